// Dev probe (gfx950): what does straight-line code cost the FIRST time a CU executes it?  One kernel runs the same
// unrolled body (N independent VALU instructions, 8 bytes each, no loop inside) twice in a row and stamps both passes:
// pass 0 meets a cold instruction cache, pass 1 a warm one.  Printed per body size: cycles per instruction, cold / warm,
// for the first-dispatched wave of each CU and for a late wave (16 waves per CU share the fetches).
//   hipcc --offload-arch=gfx950 -O3 tools/ubench/icache_probe.hip -o tools/ubench/bin/icache_probe && tools/ubench/bin/icache_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

template <int N> struct Body {
    static __device__ __forceinline__ void run(float& a, float& b, float& c, float& d) {
        asm volatile("v_fma_f32 %0, %0, %0, %0\n\tv_fma_f32 %1, %1, %1, %1\n\tv_fma_f32 %2, %2, %2, %2\n\tv_fma_f32 %3, %3, %3, %3"
                     : "+v"(a), "+v"(b), "+v"(c), "+v"(d));
        Body<N - 4>::run(a, b, c, d);
    }
};
template <> struct Body<0> { static __device__ __forceinline__ void run(float&, float&, float&, float&) {} };

template <int N>
__global__ __launch_bounds__(256) void k(float* out, long long* cyc, float s) {
    float a = s, b = s + 1, c = s + 2, d = s + 3;
    long long t[3];
    t[0] = __builtin_readcyclecounter();
#pragma unroll 1
    for (int pass = 0; pass < 2; ++pass) {
        Body<N>::run(a, b, c, d);
        t[pass + 1] = __builtin_readcyclecounter();
    }
    if ((threadIdx.x & 63) == 0) {
        const int w = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
        cyc[2 * w] = t[1] - t[0]; cyc[2 * w + 1] = t[2] - t[1];
    }
    if (a + b + c + d == 12345.0f) out[0] = a;
}

template <int N> void go(int blocks) {
    float* out; long long* cyc;
    const int waves = blocks * 4;
    CK(hipMalloc(&out, 4)); CK(hipMalloc(&cyc, sizeof(long long) * 2 * waves));
    std::vector<long long> h(2 * waves);
    for (int rep = 0; rep < 3; ++rep) {
        hipLaunchKernelGGL(k<N>, dim3(blocks), dim3(256), 0, 0, out, cyc, 0.5f);
        CK(hipDeviceSynchronize());
        CK(hipMemcpy(h.data(), cyc, sizeof(long long) * 2 * waves, hipMemcpyDeviceToHost));
        std::vector<double> cold, warm;
        for (int w = 0; w < waves; ++w) { cold.push_back((double)h[2 * w] / N); warm.push_back((double)h[2 * w + 1] / N); }
        std::sort(cold.begin(), cold.end()); std::sort(warm.begin(), warm.end());
        printf("N=%5d (%3d KB) blocks=%4d launch %d: cold pass cycles/instr  min %.2f  median %.2f  max %.2f | warm pass  min %.2f  median %.2f  max %.2f\n",
               N, N * 8 / 1024, blocks, rep, cold.front(), cold[cold.size() / 2], cold.back(), warm.front(), warm[warm.size() / 2], warm.back());
    }
    CK(hipFree(out)); CK(hipFree(cyc));
}

int main() {
    for (int blocks : {256, 1024}) {
        go<256>(blocks); go<1024>(blocks); go<4096>(blocks);
    }
    return 0;
}
