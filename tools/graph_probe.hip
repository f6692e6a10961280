// Diagnostic: dependent-launch latency of tiny kernels, stream launches against a captured hipGraph.
// Build: hipcc -O3 --offload-arch=gfx950 tools/graph_probe.hip -o /tmp/graph_probe
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

__global__ void k_tiny(float* p, int n) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) p[i] = p[i] * 1.0001f + 1.f;
}

int main() {
    const int n = 1 << 16, launches = 400;
    float* d = nullptr;
    CK(hipMalloc(&d, sizeof(float) * n));
    CK(hipMemset(d, 0, sizeof(float) * n));
    hipStream_t s;
    CK(hipStreamCreate(&s));
    hipEvent_t a, b;
    CK(hipEventCreate(&a));
    CK(hipEventCreate(&b));
    for (int grid : {16, 256, 1024}) {
        auto stream_run = [&]() {
            for (int i = 0; i < launches; ++i) k_tiny<<<grid, 256, 0, s>>>(d, n);
        };
        stream_run();
        CK(hipStreamSynchronize(s));
        float best = 1e9f;
        double best_wall = 1e9;
        for (int rep = 0; rep < 5; ++rep) {
            auto t0 = std::chrono::steady_clock::now();
            CK(hipEventRecord(a, s));
            stream_run();
            CK(hipEventRecord(b, s));
            CK(hipEventSynchronize(b));
            const double wall = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() * 1e6;
            float ms = 0.f;
            CK(hipEventElapsedTime(&ms, a, b));
            if (ms < best) best = ms;
            if (wall < best_wall) best_wall = wall;
        }
        printf("grid %5d: stream launches  %6.2f us per kernel on the GPU (%.2f us wall)\n", grid, best * 1e3 / launches, best_wall / launches);
        hipGraph_t graph;
        hipGraphExec_t exec;
        CK(hipStreamBeginCapture(s, hipStreamCaptureModeGlobal));
        stream_run();
        CK(hipStreamEndCapture(s, &graph));
        CK(hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0));
        CK(hipGraphLaunch(exec, s));
        CK(hipStreamSynchronize(s));
        best = 1e9f;
        best_wall = 1e9;
        for (int rep = 0; rep < 5; ++rep) {
            auto t0 = std::chrono::steady_clock::now();
            CK(hipEventRecord(a, s));
            CK(hipGraphLaunch(exec, s));
            CK(hipEventRecord(b, s));
            CK(hipEventSynchronize(b));
            const double wall = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() * 1e6;
            float ms = 0.f;
            CK(hipEventElapsedTime(&ms, a, b));
            if (ms < best) best = ms;
            if (wall < best_wall) best_wall = wall;
        }
        printf("grid %5d: captured graph   %6.2f us per kernel on the GPU (%.2f us wall)\n", grid, best * 1e3 / launches, best_wall / launches);
        CK(hipGraphExecDestroy(exec));
        CK(hipGraphDestroy(graph));
    }
    // short graphs launched back to back (what a loop of captured iteration pairs does)
    for (int nodes : {8, 32}) {
        hipGraph_t graph;
        hipGraphExec_t exec;
        CK(hipStreamBeginCapture(s, hipStreamCaptureModeGlobal));
        for (int i = 0; i < nodes; ++i) k_tiny<<<256, 256, 0, s>>>(d, n);
        CK(hipStreamEndCapture(s, &graph));
        auto t0 = std::chrono::steady_clock::now();
        CK(hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0));
        const double inst = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() * 1e6;
        CK(hipGraphLaunch(exec, s));
        CK(hipStreamSynchronize(s));
        const int reps = 50;
        float best = 1e9f;
        double best_wall = 1e9;
        for (int rep = 0; rep < 5; ++rep) {
            t0 = std::chrono::steady_clock::now();
            CK(hipEventRecord(a, s));
            for (int i = 0; i < reps; ++i) CK(hipGraphLaunch(exec, s));
            const double issue = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() * 1e6;
            CK(hipEventRecord(b, s));
            CK(hipEventSynchronize(b));
            float ms = 0.f;
            CK(hipEventElapsedTime(&ms, a, b));
            if (ms < best) best = ms;
            if (issue < best_wall) best_wall = issue;
        }
        printf("graph of %2d kernels x %d launches: %6.2f us per kernel on the GPU, %6.1f us of host time per hipGraphLaunch, instantiate %.0f us\n",
               nodes, reps, best * 1e3 / (nodes * reps), best_wall / reps, inst);
        CK(hipGraphExecDestroy(exec));
        CK(hipGraphDestroy(graph));
    }
    return 0;
}
