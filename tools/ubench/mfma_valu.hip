// Dev micro-benchmark (not part of the library): how many VALU / LDS instructions does a gfx950 SIMD issue in the
// shadow of v_mfma_f32_16x16x4_f32, with one and with two waves per SIMD?  Each wave runs NM MFMAs (two independent
// accumulator chains) with K independent v_fma_f32 between consecutive MFMAs, order pinned with sched_barrier.
//   hipcc --offload-arch=gfx950 -O3 tools/ubench/mfma_valu.hip -o /tmp/mfma_valu && /tmp/mfma_valu
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
using f32x4 = __attribute__((ext_vector_type(4))) float;

using bf16x8 = __attribute__((ext_vector_type(8))) __bf16;
template <int K, int MODE>   // MODE 0: fma fillers; 1: ds_read_b32 fillers (every 2nd filler); 2: no MFMA (fillers only);
                             // 3: v_mfma_f32_16x16x32_bf16 instead of the f32 MFMA, fma fillers
__global__ __launch_bounds__(256) void probe(float* out, long long* cyc, int nm) {
    __shared__ float lds[4096];
    lds[threadIdx.x] = threadIdx.x;
    __syncthreads();
    f32x4 a0 = {0, 0, 0, 0}, a1 = {0, 0, 0, 0};
    float x = threadIdx.x * 1e-3f, y = 1.0001f;
    float f[8] = {1, 2, 3, 4, 5, 6, 7, 8};
    const long long t0 = __builtin_readcyclecounter();
    for (int i = 0; i < nm; i += 2) {
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            if (MODE == 3) {
                bf16x8 p, q;
                for (int e = 0; e < 8; ++e) { p[e] = (__bf16)x; q[e] = (__bf16)y; }
                if (h == 0) a0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(p, q, a0, 0, 0, 0);
                else a1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(q, p, a1, 0, 0, 0);
            } else if (MODE != 2) {
                if (h == 0) a0 = __builtin_amdgcn_mfma_f32_16x16x4f32(x, y, a0, 0, 0, 0);
                else a1 = __builtin_amdgcn_mfma_f32_16x16x4f32(y, x, a1, 0, 0, 0);
            }
#pragma unroll
            for (int k = 0; k < K; ++k) {
                if (MODE == 1 && (k & 1)) f[k & 7] += lds[(threadIdx.x + k * 64 + i) & 4095];
                else f[k & 7] = __builtin_fmaf(f[k & 7], 1.0001f, 0.5f);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    const long long t1 = __builtin_readcyclecounter();
    float s = a0[0] + a0[1] + a0[2] + a0[3] + a1[0] + a1[1] + a1[2] + a1[3];
    for (int k = 0; k < 8; ++k) s += f[k];
    out[blockIdx.x * 256 + threadIdx.x] = s;
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

template <int K, int MODE> void run(int blocks, const char* what) {
    float* out; long long* cyc;
    hipMalloc(&out, sizeof(float) * blocks * 256);
    hipMalloc(&cyc, sizeof(long long) * blocks);
    const int nm = 2048;
    probe<K, MODE><<<blocks, 256>>>(out, cyc, nm);
    probe<K, MODE><<<blocks, 256>>>(out, cyc, nm);
    hipDeviceSynchronize();
    std::vector<long long> h(blocks);
    hipMemcpy(h.data(), cyc, sizeof(long long) * blocks, hipMemcpyDeviceToHost);
    double m = 0; for (auto v : h) m += v; m /= blocks;
    printf("%-28s K=%2d blocks=%4d : %7.1f cycles per MFMA slot (%.1f per instruction)\n", what, K, blocks, m / nm, m / nm / (K + (MODE != 2)));
    hipFree(out); hipFree(cyc);
}

int main() {
    for (int blocks : {256, 512, 1024}) {
        run<0, 0>(blocks, "mfma only");
        run<2, 0>(blocks, "mfma + K fma");
        run<4, 0>(blocks, "mfma + K fma");
        run<6, 0>(blocks, "mfma + K fma");
        run<8, 0>(blocks, "mfma + K fma");
        run<12, 0>(blocks, "mfma + K fma");
        run<6, 1>(blocks, "mfma + K (fma|ds_read)");
        run<0, 3>(blocks, "bf16 mfma only");
        run<2, 3>(blocks, "bf16 mfma + K fma");
        run<4, 3>(blocks, "bf16 mfma + K fma");
        run<8, 3>(blocks, "bf16 mfma + K fma");
        run<6, 2>(blocks, "K fma only");
        run<12, 2>(blocks, "K fma only");
    }
    return 0;
}
