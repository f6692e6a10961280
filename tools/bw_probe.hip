// HBM ceiling probe (diagnostic, not product): read-only / write-only / copy streams with 16-byte accesses per lane at
// several launch shapes, to know what a streaming kernel of each kind can reach on this box.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)
typedef float f4 __attribute__((ext_vector_type(4)));

template <int U, bool NT>
__global__ void k_read(const f4* __restrict__ a, size_t n4, float* out) {
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    f4 acc = {0, 0, 0, 0};
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    for (; i + (U - 1) * stride < n4; i += U * stride) {
        f4 v[U];
#pragma unroll
        for (int u = 0; u < U; ++u) v[u] = NT ? __builtin_nontemporal_load(a + i + u * stride) : a[i + u * stride];
#pragma unroll
        for (int u = 0; u < U; ++u) acc += v[u];
    }
    for (; i < n4; i += stride) acc += a[i];
    if (acc.x + acc.y + acc.z + acc.w == 123.456f) *out = 1.f;
}
// contiguous share per workgroup (persistent style): every workgroup streams its own range
template <int U, bool NT>
__global__ void k_read_share(const f4* __restrict__ a, size_t n4, float* out) {
    const size_t per = (n4 + gridDim.x - 1) / gridDim.x;
    const size_t lo = per * blockIdx.x, hi = lo + per < n4 ? lo + per : n4;
    f4 acc = {0, 0, 0, 0};
    size_t i = lo + threadIdx.x;
    for (; i + (size_t)(U - 1) * blockDim.x < hi; i += (size_t)U * blockDim.x) {
        f4 v[U];
#pragma unroll
        for (int u = 0; u < U; ++u) v[u] = NT ? __builtin_nontemporal_load(a + i + (size_t)u * blockDim.x) : a[i + (size_t)u * blockDim.x];
#pragma unroll
        for (int u = 0; u < U; ++u) acc += v[u];
    }
    for (; i < hi; i += blockDim.x) acc += a[i];
    if (acc.x + acc.y + acc.z + acc.w == 123.456f) *out = 1.f;
}
template <int U, bool NT>
__global__ void k_write(f4* __restrict__ a, size_t n4) {
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    const f4 v = {1.f, 2.f, 3.f, 4.f};
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += stride) {
        if (NT) __builtin_nontemporal_store(v, a + i); else a[i] = v;
    }
}
template <int U, bool NT>
__global__ void k_copy(const f4* __restrict__ a, f4* __restrict__ b, size_t n4) {
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    for (; i + (U - 1) * stride < n4; i += U * stride) {
        f4 v[U];
#pragma unroll
        for (int u = 0; u < U; ++u) v[u] = NT ? __builtin_nontemporal_load(a + i + u * stride) : a[i + u * stride];
#pragma unroll
        for (int u = 0; u < U; ++u) { if (NT) __builtin_nontemporal_store(v[u], b + i + u * stride); else b[i + u * stride] = v[u]; }
    }
    for (; i < n4; i += stride) b[i] = a[i];
}

template <typename F>
double time_us(F f, int reps = 10) {
    hipEvent_t a, b;
    CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    f(); f();
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(a));
    for (int r = 0; r < reps; ++r) f();
    CK(hipEventRecord(b));
    CK(hipEventSynchronize(b));
    float ms = 0; CK(hipEventElapsedTime(&ms, a, b));
    return ms * 1e3 / reps;
}

int main(int argc, char** argv) {
    const size_t mb = argc > 1 ? atoi(argv[1]) : 1024;
    const size_t bytes = mb << 20, n4 = bytes / 16;
    f4 *a, *b; float* out;
    CK(hipMalloc(&a, bytes)); CK(hipMalloc(&b, bytes)); CK(hipMalloc(&out, 4));
    CK(hipMemset(a, 1, bytes)); CK(hipMemset(b, 0, bytes));
    struct Shape { int grid, block; } shapes[] = {{256, 1024}, {512, 512}, {1024, 256}, {2048, 256}, {4096, 256}, {8192, 256}, {1024, 1024}, {2048, 512}};
    for (auto s : shapes) {
        double t;
        t = time_us([&] { k_read<4, false><<<s.grid, s.block>>>(a, n4, out); });   printf("read  U4     grid %5d x %4d: %7.1f us %6.0f GB/s\n", s.grid, s.block, t, bytes / t / 1e3);
        t = time_us([&] { k_read<8, false><<<s.grid, s.block>>>(a, n4, out); });   printf("read  U8     grid %5d x %4d: %7.1f us %6.0f GB/s\n", s.grid, s.block, t, bytes / t / 1e3);
        t = time_us([&] { k_read<8, true><<<s.grid, s.block>>>(a, n4, out); });    printf("read  U8 nt  grid %5d x %4d: %7.1f us %6.0f GB/s\n", s.grid, s.block, t, bytes / t / 1e3);
        t = time_us([&] { k_read<16, true><<<s.grid, s.block>>>(a, n4, out); });   printf("read  U16 nt grid %5d x %4d: %7.1f us %6.0f GB/s\n", s.grid, s.block, t, bytes / t / 1e3);
        t = time_us([&] { k_read_share<8, true><<<s.grid, s.block>>>(a, n4, out); }); printf("share U8 nt  grid %5d x %4d: %7.1f us %6.0f GB/s\n", s.grid, s.block, t, bytes / t / 1e3);
        t = time_us([&] { k_write<1, false><<<s.grid, s.block>>>(b, n4); });       printf("write        grid %5d x %4d: %7.1f us %6.0f GB/s\n", s.grid, s.block, t, bytes / t / 1e3);
        t = time_us([&] { k_write<1, true><<<s.grid, s.block>>>(b, n4); });        printf("write nt     grid %5d x %4d: %7.1f us %6.0f GB/s\n", s.grid, s.block, t, bytes / t / 1e3);
        t = time_us([&] { k_copy<4, false><<<s.grid, s.block>>>(a, b, n4); });     printf("copy  U4     grid %5d x %4d: %7.1f us %6.0f GB/s (r+w)\n", s.grid, s.block, t, 2.0 * bytes / t / 1e3);
        t = time_us([&] { k_copy<8, true><<<s.grid, s.block>>>(a, b, n4); });      printf("copy  U8 nt  grid %5d x %4d: %7.1f us %6.0f GB/s (r+w)\n", s.grid, s.block, t, 2.0 * bytes / t / 1e3);
    }
    // small buffers (the size of one step's operands: do they come from the Infinity Cache?)
    for (size_t small_mb : {32, 64, 128, 256, 512}) {
        const size_t sn4 = (small_mb << 20) / 16;
        double t = time_us([&] { k_read<8, false><<<2048, 256>>>(a, sn4, out); }, 20);
        printf("re-read %4zu MB (default policy): %7.1f us %6.0f GB/s\n", small_mb, t, (small_mb << 20) / t / 1e3);
        t = time_us([&] { k_read<8, true><<<2048, 256>>>(a, sn4, out); }, 20);
        printf("re-read %4zu MB (nt)            : %7.1f us %6.0f GB/s\n", small_mb, t, (small_mb << 20) / t / 1e3);
    }
    return 0;
}
