// Micro-benchmark: what does a grid-wide barrier cost on MI355X (8 XCDs), against a dependent kernel launch in a replayed HIP graph?
// Decides whether merging the encoder's dependent launches into persistent kernels with in-kernel barriers can pay.
//   hipcc --offload-arch=gfx950 -O3 -o grid_barrier grid_barrier.hip && ./grid_barrier
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

// one barrier: release (make this workgroup's global writes visible device-wide), arrive, spin, acquire
__device__ __forceinline__ void grid_barrier(unsigned* counter, unsigned target) {
    __syncthreads();
    if (threadIdx.x == 0) {
        __atomic_fetch_add(counter, 1u, __ATOMIC_RELEASE);                       // agent scope by default for global atomics in HIP
        while (__atomic_load_n(counter, __ATOMIC_ACQUIRE) < target) __builtin_amdgcn_s_sleep(1);
    }
    __syncthreads();
}

// every phase: each workgroup writes a value derived from what ANOTHER workgroup wrote in the previous phase (so the barrier is load-bearing)
__global__ __launch_bounds__(256) void chain_kernel(float* buf, unsigned* counter, int phases, int payload) {
    const int wg = blockIdx.x, n = gridDim.x;
    for (int p = 0; p < phases; ++p) {
        const float* src = buf + (size_t)(p & 1) * n * payload;
        float* dst = buf + (size_t)((p + 1) & 1) * n * payload;
        const int from = (wg + 1) % n;
        for (int i = threadIdx.x; i < payload; i += 256) dst[(size_t)wg * payload + i] = src[(size_t)from * payload + i] + 1.0f;
        grid_barrier(counter, (unsigned)(p + 1) * n);
    }
}
__global__ __launch_bounds__(256) void phase_kernel(const float* src, float* dst, int payload) {
    const int wg = blockIdx.x, n = gridDim.x, from = (wg + 1) % n;
    for (int i = threadIdx.x; i < payload; i += 256) dst[(size_t)wg * payload + i] = src[(size_t)from * payload + i] + 1.0f;
}

int main() {
    const int phases = 64;
    hipStream_t s; CK(hipStreamCreate(&s));
    for (int payload : {256, 4096}) for (int n : {32, 128, 256, 512}) {
        float* buf; unsigned* counter;
        CK(hipMalloc(&buf, (size_t)2 * n * payload * 4)); CK(hipMalloc(&counter, 4));
        CK(hipMemset(buf, 0, (size_t)2 * n * payload * 4));
        hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
        float best_b = 1e9f, best_g = 1e9f;
        for (int rep = 0; rep < 5; ++rep) {
            CK(hipMemsetAsync(counter, 0, 4, s));
            CK(hipEventRecord(e0, s));
            hipLaunchKernelGGL(chain_kernel, dim3(n), dim3(256), 0, s, buf, counter, phases, payload);
            CK(hipEventRecord(e1, s)); CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1)); if (ms < best_b) best_b = ms;
        }
        std::vector<float> h((size_t)n * payload);
        CK(hipMemcpy(h.data(), buf, h.size() * 4, hipMemcpyDeviceToHost));
        const bool ok = h[0] == (float)(phases * 5) && h[h.size() - 1] == (float)(phases * 5);
        // the same chain as `phases` dependent launches in one replayed graph
        hipGraph_t g; hipGraphExec_t ge;
        CK(hipStreamBeginCapture(s, hipStreamCaptureModeGlobal));
        for (int p = 0; p < phases; ++p)
            hipLaunchKernelGGL(phase_kernel, dim3(n), dim3(256), 0, s, buf + (size_t)(p & 1) * n * payload, buf + (size_t)((p + 1) & 1) * n * payload, payload);
        CK(hipStreamEndCapture(s, &g)); CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
        for (int rep = 0; rep < 6; ++rep) {
            CK(hipEventRecord(e0, s)); CK(hipGraphLaunch(ge, s)); CK(hipEventRecord(e1, s)); CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1)); if (rep && ms < best_g) best_g = ms;
        }
        printf("payload %5d floats/wg, %3d workgroups: in-kernel barrier %.2f us/phase (%s), graph of dependent launches %.2f us/phase\n",
               payload, n, best_b * 1e3f / phases, ok ? "values ok" : "VALUES WRONG", best_g * 1e3f / phases);
        CK(hipGraphExecDestroy(ge)); CK(hipGraphDestroy(g)); CK(hipFree(buf)); CK(hipFree(counter));
    }
    return 0;
}
