// Dev probe (gfx950): issue rate of v_fma_f32 / v_pk_fma_f32 / LDS-broadcast-fed FMAs per SIMD at 1, 2, 4 waves per SIMD.
// Prints cycles per wave-instruction per SIMD (lower = faster) from the shader cycle counter of one wave.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)
constexpr int ITER = 2000;

template <int MODE>
__global__ __launch_bounds__(256) void k(float* out, long long* cyc, float s0) {
    __shared__ __attribute__((aligned(16))) float w[1024];
    for (int i = threadIdx.x; i < 1024; i += blockDim.x) w[i] = 1.0f + 1e-7f * i;
    __syncthreads();
    float a[16];
    for (int i = 0; i < 16; ++i) a[i] = threadIdx.x * 1e-3f + i;
    const float x = s0 + threadIdx.x * 1e-9f;
    const long long t0 = __builtin_readcyclecounter();
    if (MODE == 0) {            // 16 independent v_fma_f32 per iteration
        for (int it = 0; it < ITER; ++it)
#pragma unroll
            for (int i = 0; i < 16; ++i) a[i] = fmaf(a[i], x, 1.0f);
    } else if (MODE == 1) {     // 8 v_pk_fma_f32 (the same 16 FMAs)
        f32x2 p[8];
        for (int i = 0; i < 8; ++i) p[i] = f32x2{a[2 * i], a[2 * i + 1]};
        const f32x2 xx = {x, x}, one = {1.0f, 1.0f};
        for (int it = 0; it < ITER; ++it)
#pragma unroll
            for (int i = 0; i < 8; ++i) p[i] = __builtin_elementwise_fma(p[i], xx, one);
        for (int i = 0; i < 8; ++i) { a[2 * i] = p[i][0]; a[2 * i + 1] = p[i][1]; }
    } else if (MODE == 2) {     // 4 ds_read_b128 (uniform address: broadcast) + 16 v_fma_f32
        for (int it = 0; it < ITER; ++it) {
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const float4 t = *reinterpret_cast<const float4*>(w + ((it * 16 + q * 4) & 1020));
                a[4 * q] = fmaf(t.x, x, a[4 * q]); a[4 * q + 1] = fmaf(t.y, x, a[4 * q + 1]);
                a[4 * q + 2] = fmaf(t.z, x, a[4 * q + 2]); a[4 * q + 3] = fmaf(t.w, x, a[4 * q + 3]);
            }
        }
    } else if (MODE == 4 || MODE == 5) {   // 16 v_mfma_f32_4x4x1 (MODE 4) / 4 v_mfma_f32_16x16x4 (MODE 5) per iteration, 8 / 4 independent chains
        f32x4 c[8];
        for (int i = 0; i < 8; ++i) c[i] = f32x4{a[i], a[i + 1], a[i + 2], a[i + 3]};
        for (int it = 0; it < ITER; ++it) {
            if (MODE == 4) {
#pragma unroll
                for (int i = 0; i < 16; ++i) c[i & 7] = __builtin_amdgcn_mfma_f32_4x4x1f32(a[i], x, c[i & 7], 4, 3, 0);
            } else {
#pragma unroll
                for (int i = 0; i < 4; ++i) c[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[i], x, c[i], 0, 0, 0);
            }
        }
        for (int i = 0; i < 8; ++i) a[i] += c[i][0] + c[i][1] + c[i][2] + c[i][3];
    } else {                    // 4 ds_read_b128 broadcast + 8 v_pk_fma_f32
        f32x2 p[8];
        for (int i = 0; i < 8; ++i) p[i] = f32x2{a[2 * i], a[2 * i + 1]};
        const f32x2 xx = {x, x};
        for (int it = 0; it < ITER; ++it) {
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const float4 t = *reinterpret_cast<const float4*>(w + ((it * 16 + q * 4) & 1020));
                p[2 * q] = __builtin_elementwise_fma(f32x2{t.x, t.y}, xx, p[2 * q]);
                p[2 * q + 1] = __builtin_elementwise_fma(f32x2{t.z, t.w}, xx, p[2 * q + 1]);
            }
        }
        for (int i = 0; i < 8; ++i) { a[2 * i] = p[i][0]; a[2 * i + 1] = p[i][1]; }
    }
    const long long t1 = __builtin_readcyclecounter();
    float s = 0;
    for (int i = 0; i < 16; ++i) s += a[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

template <int MODE> void run(const char* name, float* out, long long* cyc) {
    for (int wps = 1; wps <= 4; wps *= 2) {           // waves per SIMD: blocks of 256 threads = 1 wave per SIMD each
        const int blocks = 256 * wps;
        k<MODE><<<blocks, 256>>>(out, cyc, 1.0000001f);
        CK(hipDeviceSynchronize());
        hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
        CK(hipEventRecord(e0));
        k<MODE><<<blocks, 256>>>(out, cyc, 1.0000001f);
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        long long c[1024]; CK(hipMemcpy(c, cyc, sizeof(long long) * blocks, hipMemcpyDeviceToHost));
        double mean = 0; for (int i = 0; i < blocks; ++i) mean += c[i]; mean /= blocks;
        // per wave: ITER * 16 FMAs; per SIMD: wps waves
        printf("%-34s waves/SIMD %d: %7.2f cycles per 16 FMAs per wave | %6.2f cycles per 64-lane FMA per SIMD | %6.1f us\n", name, wps,
               mean / ITER, mean / ITER / 16.0 / wps, ms * 1e3);
    }
}

int main() {
    float* out; long long* cyc;
    CK(hipMalloc(&out, 1024 * 256 * sizeof(float))); CK(hipMalloc(&cyc, 1024 * sizeof(long long)));
    run<0>("v_fma_f32 x16", out, cyc);
    run<1>("v_pk_fma_f32 x8", out, cyc);
    run<2>("ds_read_b128 bcast x4 + v_fma x16", out, cyc);
    run<3>("ds_read_b128 bcast x4 + v_pk_fma x8", out, cyc);
    run<4>("v_mfma_f32_4x4x1 x16 (=64 FMA/lane)", out, cyc);
    run<5>("v_mfma_f32_16x16x4 x4 (=64 FMA/lane)", out, cyc);
    return 0;
}
