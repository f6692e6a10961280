// Diagnostic (not product): the way out of the internal id space, dst[old] = src[idx[old]] (idx < 0: the id is isolated, write 0), timed
// for index arrays read from files -- the engine's relabelling (ranks dealt to the 8 column blocks one by one) against the same ranks
// dealt in runs of 32 (tools/probe_permute_out.py writes both).  Usage: permute_probe n file [file ...]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)
constexpr int kChunk = 4096, kBlock = 256;
__global__ __launch_bounds__(kBlock) void k_out(const float* __restrict__ src, const int* __restrict__ idx, long n, float factor, float* __restrict__ dst) {
    constexpr int U = kChunk / kBlock;
    for (long base = (long)blockIdx.x * kChunk; base < n; base += (long)gridDim.x * kChunk) {
        int at[U];
        float x[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const long i = base + u * kBlock + threadIdx.x;
            at[u] = i < n ? idx[i] : -1;
        }
#pragma unroll
        for (int u = 0; u < U; ++u) x[u] = at[u] < 0 ? 0.f : src[at[u]];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const long i = base + u * kBlock + threadIdx.x;
            if (i < n) dst[i] = x[u] * factor;
        }
    }
}
int main(int argc, char** argv) {
    const long n = atol(argv[1]);
    float *src, *dst;
    int* idx;
    CK(hipMalloc(&src, 4 * (n + 65536)));
    CK(hipMalloc(&dst, 4 * n));
    CK(hipMalloc(&idx, 4 * n));
    CK(hipMemset(src, 0, 4 * (n + 65536)));
    hipEvent_t a, b;
    CK(hipEventCreate(&a));
    CK(hipEventCreate(&b));
    std::vector<int> h(n);
    for (int f = 2; f < argc; ++f) {
        FILE* in = fopen(argv[f], "rb");
        if (!in || fread(h.data(), 4, n, in) != (size_t)n) { printf("cannot read %s\n", argv[f]); return 1; }
        fclose(in);
        CK(hipMemcpy(idx, h.data(), 4 * n, hipMemcpyHostToDevice));
        const int grid = (int)((n + kChunk - 1) / kChunk) < 2048 ? (int)((n + kChunk - 1) / kChunk) : 2048;
        for (int i = 0; i < 3; ++i) k_out<<<grid, kBlock>>>(src, idx, n, 2.f, dst);
        CK(hipEventRecord(a));
        for (int i = 0; i < 20; ++i) k_out<<<grid, kBlock>>>(src, idx, n, 2.f, dst);
        CK(hipEventRecord(b));
        CK(hipEventSynchronize(b));
        float ms = 0;
        CK(hipEventElapsedTime(&ms, a, b));
        printf("%-40s %.1f us per launch\n", argv[f], ms / 20 * 1e3);
    }
    return 0;
}
