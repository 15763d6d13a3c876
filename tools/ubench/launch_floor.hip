// Dev micro-benchmark: time per dependent kernel in a stream -- eager launches vs hipGraph replay -- for a trivial kernel
// and for a kernel that writes a few MB (dirty L2 lines at the boundary).
//   hipcc --offload-arch=gfx950 -O3 tools/ubench/launch_floor.hip -o /tmp/launch_floor && /tmp/launch_floor
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>

__global__ void tiny(float* p) { if (threadIdx.x == 0 && blockIdx.x == 0) p[0] += 1.0f; }
__global__ void writer(float* p, int n) {
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) p[i] = p[i] * 1.0001f + 1.0f;
}

template <typename F> double time_us(F&& f, int reps) {
    f();
    hipDeviceSynchronize();
    auto t0 = std::chrono::high_resolution_clock::now();
    for (int r = 0; r < reps; ++r) f();
    hipDeviceSynchronize();
    return std::chrono::duration<double, std::micro>(std::chrono::high_resolution_clock::now() - t0).count() / reps;
}

int main() {
    float* p;
    const int n = 1 << 20;   // 4 MB
    hipMalloc(&p, sizeof(float) * n);
    hipMemset(p, 0, sizeof(float) * n);
    hipStream_t s;
    hipStreamCreate(&s);
    const int N = 200;
    for (int mode = 0; mode < 3; ++mode) {
        auto launch = [&]() {
            if (mode == 0) tiny<<<1, 64, 0, s>>>(p);
            else if (mode == 1) tiny<<<256, 256, 0, s>>>(p);
            else writer<<<512, 256, 0, s>>>(p, n);
        };
        const char* name = mode == 0 ? "tiny 1 block" : mode == 1 ? "tiny 256 blocks" : "writer 4 MB";
        double eager = time_us([&]() { for (int i = 0; i < N; ++i) launch(); hipStreamSynchronize(s); }, 20) / N;
        hipGraph_t g; hipGraphExec_t ge;
        hipStreamBeginCapture(s, hipStreamCaptureModeGlobal);
        for (int i = 0; i < N; ++i) launch();
        hipStreamEndCapture(s, &g);
        hipGraphInstantiate(&ge, g, nullptr, nullptr, 0);
        double graph = time_us([&]() { hipGraphLaunch(ge, s); hipStreamSynchronize(s); }, 20) / N;
        printf("%-16s eager %.2f us/kernel   graph replay %.2f us/kernel\n", name, eager, graph);
        hipGraphExecDestroy(ge); hipGraphDestroy(g);
    }
    return 0;
}
