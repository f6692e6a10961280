// Runtime, dense vectors, elementwise operator protocol and f64-accumulated reductions.
//
// Reference counterparts: the vector primitives of pygrank/core/backend/specification.py:5-118 as the
// numpy engine implements them (pygrank/core/backend/numpy.py:1-86) and the residual measures of
// pygrank/measures/supervised.py:93-106,133-138.  All of these are HBM-bound streaming kernels:
// 16-byte loads per lane, 64-wide wavefront shuffles for reductions, f64 accumulators.
#include "pgh_common.h"

#include <hipcub/hipcub.hpp>
#include <chrono>
#include <cstring>
#include <unordered_map>
#include <map>

#include <mutex>
#include <vector>

namespace pgh {

static thread_local std::string g_error;
static Runtime g_rt;

void set_error(const std::string& msg) { g_error = msg; }
int fail(const std::string& msg) {
    g_error = msg;
    return 1;
}
Runtime& rt() { return g_rt; }

int ensure_init() {
    if (g_rt.initialised) return 0;
    return pgh_init(0);
}

// Event pairs of the profiled launches are recorded WITHOUT waiting for them (a wait after every launch let the queue run
// dry, and every timed launch then started from an idle GPU: +4 us per kernel against rocprofv3's durations); they are read
// when the profile is read or switched off.  Events come from a pool that lives as long as the process.
namespace {
struct ProfPair {
    hipEvent_t a, b;
    int        id;
};
std::vector<ProfPair> g_prof_pending;
std::vector<std::pair<hipEvent_t, hipEvent_t>> g_prof_free;

void prof_drain() {
    if (g_prof_pending.empty()) return;
    (void)hipEventSynchronize(g_prof_pending.back().b);
    for (const ProfPair& pr : g_prof_pending) {
        float ms = 0.f;
        if (hipEventSynchronize(pr.b) == hipSuccess && hipEventElapsedTime(&ms, pr.a, pr.b) == hipSuccess) {
            g_rt.prof_count[pr.id] += 1;
            g_rt.prof_ms[pr.id] += ms;
        }
        g_prof_free.emplace_back(pr.a, pr.b);
    }
    g_prof_pending.clear();
}
}  // namespace

ProfScope::ProfScope(int kernel_id) : id(kernel_id), on(g_rt.profiling) {
    if (!on) return;
    ProfPair pr{nullptr, nullptr, id};
    if (!g_prof_free.empty()) {
        pr.a = g_prof_free.back().first;
        pr.b = g_prof_free.back().second;
        g_prof_free.pop_back();
    } else {
        if (hipEventCreate(&pr.a) != hipSuccess) {
            on = false;
            return;
        }
        if (hipEventCreate(&pr.b) != hipSuccess) {
            (void)hipEventDestroy(pr.a);
            on = false;
            return;
        }
    }
    end = pr.b;
    (void)hipEventRecord(pr.a, g_rt.stream);
    g_prof_pending.push_back(pr);
}
ProfScope::~ProfScope() {
    if (!on) return;
    (void)hipEventRecord(end, g_rt.stream);
    if (g_prof_pending.size() >= 16384) prof_drain();
}

}  // namespace pgh

using namespace pgh;

// =================================================================================================
// runtime
// =================================================================================================
extern "C" const char* pgh_last_error(void) { return g_error.c_str(); }
extern "C" const char* pgh_runtime_name(void) { return "hip:gfx950"; }

extern "C" int pgh_device_count(int* count) {
    int c = 0;
    hipError_t e = hipGetDeviceCount(&c);
    if (e != hipSuccess) {
        *count = 0;
        return fail(std::string("hipGetDeviceCount: ") + hipGetErrorString(e));
    }
    *count = c;
    return 0;
}

namespace {
std::vector<const void*>& warm_kernels() {
    static std::vector<const void*> list;      // (function-local: filled by static initialisers of other translation units)
    return list;
}
}  // namespace
void pgh::register_warm_kernel(const void* kernel) { warm_kernels().push_back(kernel); }

extern "C" int pgh_init(int device_ordinal) {
    Runtime& r = rt();
    if (r.initialised && r.device == device_ordinal) return 0;
    PGH_CHECK(!r.initialised, "pgh_init: engine already initialised on another device; call pgh_shutdown first");
    int count = 0;
    PGH_TRY(pgh_device_count(&count));
    PGH_CHECK(count > 0, "pgh_init: no HIP device visible (the MI355X engine has no CPU fallback)");
    PGH_CHECK(device_ordinal >= 0 && device_ordinal < count, "pgh_init: device ordinal out of range");
    PGH_HIP(hipSetDevice(device_ordinal));
    hipDeviceProp_t prop;
    PGH_HIP(hipGetDeviceProperties(&prop, device_ordinal));
    r.num_cus = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
    PGH_HIP(hipStreamCreateWithFlags(&r.own_stream, hipStreamNonBlocking));
    r.stream = r.own_stream;
    PGH_HIP(hipMalloc(&r.d_partials, sizeof(double) * kMaxPartials * kPartialRegions));
    PGH_HIP(hipMalloc(&r.d_scalars, sizeof(double) * kNumScalars));
    PGH_HIP(hipHostMalloc(&r.h_scalars, sizeof(double) * kNumScalars, hipHostMallocDefault));
    PGH_HIP(hipEventCreate(&r.ev_a));
    PGH_HIP(hipEventCreate(&r.ev_b));
    if (!(getenv("PGH_WARM") != nullptr && atoi(getenv("PGH_WARM")) == 0)) {
        for (const void* kernel : warm_kernels()) {
            hipFuncAttributes attr;
            (void)hipFuncGetAttributes(&attr, kernel);        // loads the code object that holds the kernel
        }
        (void)hipGetLastError();
    }
    if (!(getenv("PGH_MAILBOX") != nullptr && atoi(getenv("PGH_MAILBOX")) == 0)) {
        void* hp = nullptr;
        void* dp = nullptr;
        PGH_HIP(hipHostMalloc(&hp, 64, hipHostMallocMapped | hipHostMallocCoherent));
        memset(hp, 0, 64);
        PGH_HIP(hipHostGetDevicePointer(&dp, hp, 0));
        r.mail_host = reinterpret_cast<volatile unsigned long long*>(hp);
        r.mail_dev = reinterpret_cast<unsigned long long*>(dp);
        r.mail_tag = 0;
    }
    r.device = device_ordinal;
    r.initialised = true;
    return 0;
}

namespace {
__global__ void k_post_scalars(const double* __restrict__ src, int count, unsigned long long* __restrict__ mail, unsigned long long tag) {
    for (int i = 0; i < count; ++i) __hip_atomic_store(mail + 1 + i, (unsigned long long)__double_as_longlong(src[i]), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    __hip_atomic_store(mail, tag, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);       // the values are visible before the tag is
}
}  // namespace

namespace {
std::vector<std::pair<std::string, double>> g_build_phases;
std::chrono::steady_clock::time_point       g_build_last;
}  // namespace

void pgh::build_clock_reset() {
    g_build_phases.clear();
    g_build_last = std::chrono::steady_clock::now();
}

void pgh::build_mark(const char* label) {
    (void)hipStreamSynchronize(rt().stream);
    const auto now = std::chrono::steady_clock::now();
    const double ms = std::chrono::duration<double, std::milli>(now - g_build_last).count();
    g_build_last = now;
    for (auto& ph : g_build_phases)
        if (ph.first == label) {
            ph.second += ms;
            return;
        }
    g_build_phases.emplace_back(label, ms);
}

extern "C" int pgh_last_build_profile(char* buf, int buflen) {
    PGH_CHECK(buf != nullptr && buflen > 0, "pgh_last_build_profile: null buffer");
    std::string out;
    char num[64];
    for (const auto& ph : g_build_phases) {
        snprintf(num, sizeof(num), "=%.3f;", ph.second);
        out += ph.first + num;
    }
    snprintf(buf, (size_t)buflen, "%s", out.c_str());
    return 0;
}

int pgh::scalars_to_host(int first, int count) {
    Runtime& r = rt();
    if (r.mail_host != nullptr && count >= 1 && count <= 7) {
        const unsigned long long tag = ++r.mail_tag;
        k_post_scalars<<<1, 1, 0, r.stream>>>(r.d_scalars + first, count, r.mail_dev, tag);
        PGH_HIP(hipGetLastError());
        const auto start = std::chrono::steady_clock::now();
        long spins = 0;
        for (;;) {
            if (__atomic_load_n(const_cast<unsigned long long*>(r.mail_host), __ATOMIC_ACQUIRE) == tag) {
                for (int i = 0; i < count; ++i) {
                    const unsigned long long bits = r.mail_host[1 + i];
                    memcpy(&r.h_scalars[first + i], &bits, sizeof(double));
                }
                return 0;
            }
            if ((++spins & 0xfff) == 0) {
                // the post never came within a generous bound (a fault upstream, a stream that is not running): let the runtime say why
                if (std::chrono::duration<double>(std::chrono::steady_clock::now() - start).count() > 2.0) break;
                const hipError_t q = hipStreamQuery(r.stream);
                if (q != hipSuccess && q != hipErrorNotReady) return fail(std::string("scalars_to_host: ") + hipGetErrorString(q));
            }
        }
    }
    PGH_HIP(hipMemcpyAsync(r.h_scalars + first, r.d_scalars + first, sizeof(double) * count, hipMemcpyDeviceToHost, r.stream));
    PGH_HIP(hipStreamSynchronize(r.stream));
    return 0;
}

namespace {
// Blocks of up to half a slab are CARVED out of slabs of PGH_SLAB_MB (default 1024) MB taken from the driver in one call each: a graph
// build asks for ~150 buffers, and every driver allocation is a chance to meet the node's allocation stalls (0.2-1.0 s)
// -- with slabs the first build of a process makes ~10 driver calls instead.  A carved block goes back to the idle lists like any other and
// is never returned to the driver by itself; a slab none of whose blocks is live goes back whole when memory is short (pool_trim).
struct Slab {
    char*  base = nullptr;
    size_t size = 0, used = 0;
    int    live = 0;                            // carved blocks handed out and not yet freed
};
struct DevicePool {
    std::multimap<size_t, void*> idle;          // size -> block
    std::unordered_map<void*, size_t> live;     // block -> size
    size_t idle_bytes = 0;                      // idle blocks that came from the driver one by one (carved ones do not count against the cap)
    size_t cap_bytes = 0;
    std::vector<Slab> slabs;
    size_t slab_bytes = ~(size_t)0;             // unset; 0 = no slabs
};
DevicePool g_dev_pool;

int slab_of(const DevicePool& P, const void* p) {
    const char* c = static_cast<const char*>(p);
    for (size_t i = 0; i < P.slabs.size(); ++i)
        if (c >= P.slabs[i].base && c < P.slabs[i].base + P.slabs[i].size) return (int)i;
    return -1;
}
}  // namespace

namespace pgh {
void pool_trim() {
    DevicePool& P = g_dev_pool;
    if (P.idle.empty()) return;
    (void)hipStreamSynchronize(rt().stream);
    // blocks that came from the driver one by one go back one by one; a slab goes back when none of its blocks is in use
    for (auto it = P.idle.begin(); it != P.idle.end();) {
        const int s = slab_of(P, it->second);
        if (s >= 0 && P.slabs[(size_t)s].live > 0) {
            ++it;
            continue;
        }
        if (s < 0) (void)hipFree(it->second);
        it = P.idle.erase(it);
    }
    for (size_t i = 0; i < P.slabs.size();) {
        if (P.slabs[i].live == 0) {
            (void)hipFree(P.slabs[i].base);
            P.slabs.erase(P.slabs.begin() + (long)i);
        } else {
            ++i;
        }
    }
    P.idle_bytes = 0;
}

int pool_alloc(size_t bytes, void** out) {
    DevicePool& P = g_dev_pool;
    if (bytes == 0) bytes = 4;
    bytes = (bytes + 255) & ~(size_t)255;
    auto it = P.idle.find(bytes);
    if (it != P.idle.end()) {
        *out = it->second;
        const int s = slab_of(P, *out);
        if (s >= 0) ++P.slabs[(size_t)s].live;
        else P.idle_bytes -= bytes;
        P.idle.erase(it);
        P.live[*out] = bytes;
        return 0;
    }
    if (P.slab_bytes == ~(size_t)0) {
        const char* e = getenv("PGH_SLAB_MB");
        P.slab_bytes = (size_t)(e != nullptr ? atoll(e) : 1024) << 20;
    }
    if (P.slab_bytes > 0 && bytes <= P.slab_bytes / 2) {
        Slab* slab = nullptr;
        for (size_t i = P.slabs.size(); i-- > 0;)
            if (P.slabs[i].size - P.slabs[i].used >= bytes) {
                slab = &P.slabs[i];
                break;
            }
        if (slab == nullptr) {
            void* base = nullptr;
            hipError_t e = hipMalloc(&base, P.slab_bytes);
            if (e != hipSuccess) {
                (void)hipGetLastError();
                pool_trim();
                e = hipMalloc(&base, P.slab_bytes);
            }
            if (e == hipSuccess) {
                Slab fresh;
                fresh.base = static_cast<char*>(base);
                fresh.size = P.slab_bytes;
                P.slabs.push_back(fresh);
                slab = &P.slabs.back();
            } else {
                (void)hipGetLastError();        // no room for a slab: the block itself may still fit
            }
        }
        if (slab != nullptr) {
            void* p = slab->base + slab->used;
            slab->used += bytes;
            ++slab->live;
            P.live[p] = bytes;
            *out = p;
            return 0;
        }
    }
    void* p = nullptr;
    hipError_t e = hipMalloc(&p, bytes);
    if (e != hipSuccess) {
        (void)hipGetLastError();
        pool_trim();
        e = hipMalloc(&p, bytes);
    }
    if (e != hipSuccess) return fail(std::string("device allocation of ") + std::to_string(bytes) + " bytes failed: " + hipGetErrorString(e));
    P.live[p] = bytes;
    *out = p;
    return 0;
}

void pool_free(void* p) {
    if (p == nullptr) return;
    DevicePool& P = g_dev_pool;
    auto it = P.live.find(p);
    if (it == P.live.end()) {                  // not ours (should not happen): plain free
        (void)hipStreamSynchronize(rt().stream);
        (void)hipFree(p);
        return;
    }
    const size_t bytes = it->second;
    P.live.erase(it);
    const int s = slab_of(P, p);
    if (s >= 0) {                               // carved: back to the idle lists, whatever the cap says (its memory is the slab's)
        --P.slabs[(size_t)s].live;
        P.idle.emplace(bytes, p);
        return;
    }
    if (P.cap_bytes == 0) {                    // idle blocks kept: PGH_POOL_MB, default a quarter of the device memory
        const char* e = getenv("PGH_POOL_MB");
        if (e != nullptr) {
            P.cap_bytes = (size_t)atoll(e) << 20;
        } else {
            size_t free_b = 0, total_b = 0;
            if (hipMemGetInfo(&free_b, &total_b) != hipSuccess) total_b = (size_t)64 << 30;
            P.cap_bytes = total_b / 4;
        }
        if (P.cap_bytes == 0) P.cap_bytes = 1;
    }
    if (bytes > P.cap_bytes) {
        (void)hipStreamSynchronize(rt().stream);
        (void)hipFree(p);
        return;
    }
    if (P.idle_bytes + bytes > P.cap_bytes) {  // evict the largest idle blocks (of those that are the driver's own) until the new one fits
        (void)hipStreamSynchronize(rt().stream);
        for (auto last = P.idle.end(); last != P.idle.begin() && P.idle_bytes + bytes > P.cap_bytes;) {
            --last;
            if (slab_of(P, last->second) >= 0) continue;
            (void)hipFree(last->second);
            P.idle_bytes -= last->first;
            last = P.idle.erase(last);
        }
    }
    P.idle.emplace(bytes, p);
    P.idle_bytes += bytes;
}
}  // namespace pgh

namespace {
void staged_release();      // pinned chunks of the device -> host copies (below)
}

extern "C" int pgh_shutdown(void) {
    Runtime& r = rt();
    if (!r.initialised) return 0;
    (void)hipStreamSynchronize(r.stream);
    staged_release();
    pool_trim();
    (void)hipFree(r.d_partials);
    (void)hipFree(r.d_scalars);
    (void)hipHostFree(r.h_scalars);
    if (r.mail_host != nullptr) (void)hipHostFree(const_cast<unsigned long long*>(r.mail_host));
    (void)hipEventDestroy(r.ev_a);
    (void)hipEventDestroy(r.ev_b);
    (void)hipStreamDestroy(r.own_stream);
    r = Runtime();
    return 0;
}

extern "C" int pgh_device_name(char* buf, int buflen) {
    PGH_TRY(ensure_init());
    hipDeviceProp_t prop;
    PGH_HIP(hipGetDeviceProperties(&prop, rt().device));
    snprintf(buf, buflen, "%s (%s, %d CUs)", prop.name, prop.gcnArchName, prop.multiProcessorCount);
    return 0;
}

extern "C" int pgh_mem_info(int64_t* free_bytes, int64_t* total_bytes) {
    PGH_TRY(ensure_init());
    size_t f = 0, t = 0;
    PGH_HIP(hipMemGetInfo(&f, &t));
    *free_bytes = (int64_t)f;
    *total_bytes = (int64_t)t;
    return 0;
}

extern "C" int pgh_set_stream(void* hip_stream) {
    PGH_TRY(ensure_init());
    PGH_HIP(hipStreamSynchronize(rt().stream));
    rt().stream = hip_stream ? (hipStream_t)hip_stream : rt().own_stream;
    return 0;
}

extern "C" int pgh_sync(void) {
    PGH_TRY(ensure_init());
    PGH_HIP(hipStreamSynchronize(rt().stream));
    return 0;
}

extern "C" int pgh_timer_create(pgh_timer_t* out) {
    PGH_TRY(ensure_init());
    pgh_timer_s* t = new pgh_timer_s();
    PGH_HIP(hipEventCreate(&t->start));
    PGH_HIP(hipEventCreate(&t->stop));
    *out = t;
    return 0;
}
extern "C" int pgh_timer_destroy(pgh_timer_t t) {
    if (!t) return 0;
    (void)hipEventDestroy(t->start);
    (void)hipEventDestroy(t->stop);
    delete t;
    return 0;
}
extern "C" int pgh_timer_start(pgh_timer_t t) {
    PGH_HIP(hipEventRecord(t->start, rt().stream));
    return 0;
}
extern "C" int pgh_timer_stop(pgh_timer_t t) {
    PGH_HIP(hipEventRecord(t->stop, rt().stream));
    return 0;
}
extern "C" int pgh_timer_elapsed_ms(pgh_timer_t t, double* ms) {
    PGH_HIP(hipEventSynchronize(t->stop));
    float f = 0.f;
    PGH_HIP(hipEventElapsedTime(&f, t->start, t->stop));
    *ms = (double)f;
    return 0;
}

extern "C" int pgh_profile_enable(int on) {
    PGH_TRY(ensure_init());
    if (!on) prof_drain();
    rt().profiling = (on != 0);
    return 0;
}
extern "C" int pgh_profile_reset(void) {
    prof_drain();
    for (int i = 0; i < PGH_K_COUNT; ++i) {
        rt().prof_count[i] = 0;
        rt().prof_ms[i] = 0.0;
    }
    return 0;
}
extern "C" int pgh_profile_read(int kernel_id, int64_t* launches, double* total_ms) {
    PGH_CHECK(kernel_id >= 0 && kernel_id < PGH_K_COUNT, "pgh_profile_read: bad kernel id");
    prof_drain();
    *launches = rt().prof_count[kernel_id];
    *total_ms = rt().prof_ms[kernel_id];
    return 0;
}

// =================================================================================================
// vectors
// =================================================================================================
namespace {

constexpr int kBlock = 256;

inline int grid_for(int64_t n, int per_thread = 4) {
    int64_t blocks = (n + (int64_t)kBlock * per_thread - 1) / ((int64_t)kBlock * per_thread);
    int64_t cap = (int64_t)rt().num_cus * 8;
    if (blocks > cap) blocks = cap;
    if (blocks < 1) blocks = 1;
    return (int)blocks;
}

__global__ void k_fill(float* __restrict__ x, int64_t n, float v) {
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) x[i] = v;
}

__global__ void k_f64_to_f32(const double* __restrict__ in, float* __restrict__ out, int64_t n) {
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
        out[i] = (float)in[i];
}
__global__ void k_f32_to_f64(const float* __restrict__ in, double* __restrict__ out, int64_t n) {
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
        out[i] = (double)in[i];
}

__global__ void k_scatter_set(float* __restrict__ x, const int64_t* __restrict__ idx, const double* __restrict__ val,
                              int64_t count) {
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < count; i += (int64_t)gridDim.x * blockDim.x)
        x[idx[i]] = (float)val[i];
}

__device__ __forceinline__ float apply_bin(int op, float a, float b) {
    switch (op) {
        case PGH_ADD: return a + b;
        case PGH_SUB: return a - b;
        case PGH_MUL: return a * b;
        case PGH_DIV: return a / b;
        case PGH_POW: return powf(a, b);
        case PGH_MAXOP: return fmaxf(a, b);
        case PGH_MINOP: return fminf(a, b);
        case PGH_GT: return a > b ? 1.f : 0.f;
        case PGH_GE: return a >= b ? 1.f : 0.f;
        case PGH_LT: return a < b ? 1.f : 0.f;
        case PGH_LE: return a <= b ? 1.f : 0.f;
        case PGH_EQ: return a == b ? 1.f : 0.f;
        default: return a != b ? 1.f : 0.f;
    }
}

__device__ __forceinline__ float apply_un(int op, float a) {
    switch (op) {
        case PGH_ABS: return fabsf(a);
        case PGH_EXP: return expf(a);
        case PGH_LOG: return logf(a);
        case PGH_NEG: return -a;
        case PGH_SQRT: return sqrtf(a);
        default: return a != 0.f ? 1.f / a : 0.f;   // safe_inv, backend/__init__.py:20-23
    }
}

// 16 B per lane on the aligned body, scalar tail.  OP is a template parameter so the switch folds.
template <int OP>
__global__ void k_ewise_vv(const float* __restrict__ a, const float* __restrict__ b, float* __restrict__ out,
                           int64_t n, int vec_ok) {
    const int64_t tid = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    int64_t body = vec_ok ? (n >> 2) : 0;
    const float4* a4 = reinterpret_cast<const float4*>(a);
    const float4* b4 = reinterpret_cast<const float4*>(b);
    float4* o4 = reinterpret_cast<float4*>(out);
    for (int64_t i = tid; i < body; i += stride) {
        float4 x = a4[i], y = b4[i], r;
        r.x = apply_bin(OP, x.x, y.x);
        r.y = apply_bin(OP, x.y, y.y);
        r.z = apply_bin(OP, x.z, y.z);
        r.w = apply_bin(OP, x.w, y.w);
        o4[i] = r;
    }
    for (int64_t i = (body << 2) + tid; i < n; i += stride) out[i] = apply_bin(OP, a[i], b[i]);
}

template <int OP, int LEFT>
__global__ void k_ewise_vs(const float* __restrict__ a, float s, float* __restrict__ out, int64_t n, int vec_ok) {
    const int64_t tid = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    int64_t body = vec_ok ? (n >> 2) : 0;
    const float4* a4 = reinterpret_cast<const float4*>(a);
    float4* o4 = reinterpret_cast<float4*>(out);
    for (int64_t i = tid; i < body; i += stride) {
        float4 x = a4[i], r;
        r.x = LEFT ? apply_bin(OP, s, x.x) : apply_bin(OP, x.x, s);
        r.y = LEFT ? apply_bin(OP, s, x.y) : apply_bin(OP, x.y, s);
        r.z = LEFT ? apply_bin(OP, s, x.z) : apply_bin(OP, x.z, s);
        r.w = LEFT ? apply_bin(OP, s, x.w) : apply_bin(OP, x.w, s);
        o4[i] = r;
    }
    for (int64_t i = (body << 2) + tid; i < n; i += stride)
        out[i] = LEFT ? apply_bin(OP, s, a[i]) : apply_bin(OP, a[i], s);
}

template <int OP>
__global__ void k_ewise_un(const float* __restrict__ a, float* __restrict__ out, int64_t n, int vec_ok) {
    const int64_t tid = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    int64_t body = vec_ok ? (n >> 2) : 0;
    const float4* a4 = reinterpret_cast<const float4*>(a);
    float4* o4 = reinterpret_cast<float4*>(out);
    for (int64_t i = tid; i < body; i += stride) {
        float4 x = a4[i], r;
        r.x = apply_un(OP, x.x);
        r.y = apply_un(OP, x.y);
        r.z = apply_un(OP, x.z);
        r.w = apply_un(OP, x.w);
        o4[i] = r;
    }
    for (int64_t i = (body << 2) + tid; i < n; i += stride) out[i] = apply_un(OP, a[i]);
}

__global__ void k_axpby(float a, const float* __restrict__ x, float b, const float* __restrict__ y,
                        float* __restrict__ out, int64_t n, int vec_ok) {
    const int64_t tid = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    int64_t body = vec_ok ? (n >> 2) : 0;
    const float4* x4 = reinterpret_cast<const float4*>(x);
    const float4* y4 = reinterpret_cast<const float4*>(y);
    float4* o4 = reinterpret_cast<float4*>(out);
    for (int64_t i = tid; i < body; i += stride) {
        float4 u = x4[i], v = y4[i], r;
        r.x = a * u.x + b * v.x;
        r.y = a * u.y + b * v.y;
        r.z = a * u.z + b * v.z;
        r.w = a * u.w + b * v.w;
        o4[i] = r;
    }
    for (int64_t i = (body << 2) + tid; i < n; i += stride) out[i] = a * x[i] + b * y[i];
}

inline int aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

}  // namespace

extern "C" int pgh_vec_alloc(int64_t n, pgh_vec_t* out) {
    PGH_TRY(ensure_init());
    PGH_CHECK(n >= 0, "pgh_vec_alloc: negative length");
    pgh_vec_s* v = new pgh_vec_s();
    v->n = n;
    v->owns = true;
    size_t bytes = (size_t)(n > 0 ? n : 1) * sizeof(float);
    if (pool_alloc(bytes, (void**)&v->data) != 0) {
        delete v;
        return -1;
    }
    *out = v;
    return 0;
}

extern "C" int pgh_vec_wrap(void* device_ptr, int64_t n, pgh_vec_t* out) {
    PGH_TRY(ensure_init());
    PGH_CHECK(device_ptr != nullptr || n == 0, "pgh_vec_wrap: null pointer");
    pgh_vec_s* v = new pgh_vec_s();
    v->data = (float*)device_ptr;
    v->n = n;
    v->owns = false;
    *out = v;
    return 0;
}

extern "C" int pgh_vec_free(pgh_vec_t v) {
    if (!v) return 0;
    if (v->owns && v->data) pool_free(v->data);      // stream-ordered reuse, no device synchronisation
    delete v;
    return 0;
}

extern "C" int64_t pgh_vec_len(pgh_vec_t v) { return v ? v->n : -1; }
extern "C" void* pgh_vec_ptr(pgh_vec_t v) { return v ? (void*)v->data : nullptr; }

extern "C" int pgh_vec_h2d_f32(pgh_vec_t v, const float* host, int64_t n) {
    PGH_CHECK(v && n == v->n, "pgh_vec_h2d_f32: length mismatch");
    if (n == 0) return 0;
    PGH_HIP(hipMemcpyAsync(v->data, host, sizeof(float) * n, hipMemcpyHostToDevice, rt().stream));
    PGH_HIP(hipStreamSynchronize(rt().stream));
    return 0;
}

extern "C" int pgh_vec_h2d_f64(pgh_vec_t v, const double* host, int64_t n) {
    PGH_CHECK(v && n == v->n, "pgh_vec_h2d_f64: length mismatch");
    if (n == 0) return 0;
    double* staging = nullptr;
    PGH_TRY(pool_alloc(sizeof(double) * n, (void**)&staging));
    PGH_HIP(hipMemcpyAsync(staging, host, sizeof(double) * n, hipMemcpyHostToDevice, rt().stream));
    k_f64_to_f32<<<grid_for(n), kBlock, 0, rt().stream>>>(staging, v->data, n);
    PGH_HIP(hipGetLastError());
    PGH_HIP(hipStreamSynchronize(rt().stream));
    pool_free(staging);
    return 0;
}

// Device -> pageable host memory through two pinned chunks: the copy of chunk i + 1 is in flight while chunk i is moved into
// the caller's array (a plain hipMemcpy into pageable memory ran at ~2 GB/s: 17 ms for the 4 M doubles of a scale-22 result).
namespace {
constexpr size_t kStageBytes = 8u << 20;
void* g_stage[2] = {nullptr, nullptr};
hipEvent_t g_stage_ev[2] = {nullptr, nullptr};

void staged_release() {
    for (int k = 0; k < 2; ++k) {
        if (g_stage[k] != nullptr) (void)hipHostFree(g_stage[k]);
        if (g_stage_ev[k] != nullptr) (void)hipEventDestroy(g_stage_ev[k]);
        g_stage[k] = nullptr;
        g_stage_ev[k] = nullptr;
    }
}

int staged_d2h(void* host, const void* dev, size_t bytes) {
    Runtime& r = rt();
    if (g_stage[0] == nullptr) {
        for (int k = 0; k < 2; ++k) {
            PGH_HIP(hipHostMalloc(&g_stage[k], kStageBytes, hipHostMallocDefault));
            PGH_HIP(hipEventCreateWithFlags(&g_stage_ev[k], hipEventDisableTiming));
        }
    }
    const size_t chunks = (bytes + kStageBytes - 1) / kStageBytes;
    for (size_t c = 0; c <= chunks; ++c) {
        if (c < chunks) {
            const size_t off = c * kStageBytes, len = bytes - off < kStageBytes ? bytes - off : kStageBytes;
            PGH_HIP(hipMemcpyAsync(g_stage[c & 1], static_cast<const char*>(dev) + off, len, hipMemcpyDeviceToHost, r.stream));
            PGH_HIP(hipEventRecord(g_stage_ev[c & 1], r.stream));
        }
        if (c > 0) {
            const size_t p = c - 1, off = p * kStageBytes, len = bytes - off < kStageBytes ? bytes - off : kStageBytes;
            PGH_HIP(hipEventSynchronize(g_stage_ev[p & 1]));
            memcpy(static_cast<char*>(host) + off, g_stage[p & 1], len);
        }
    }
    return 0;
}
}  // namespace

extern "C" int pgh_vec_d2h_f32(pgh_vec_t v, float* host, int64_t n) {
    PGH_CHECK(v && n == v->n, "pgh_vec_d2h_f32: length mismatch");
    if (n == 0) return 0;
    return staged_d2h(host, v->data, sizeof(float) * (size_t)n);
}

extern "C" int pgh_vec_d2h_f64(pgh_vec_t v, double* host, int64_t n) {
    PGH_CHECK(v && n == v->n, "pgh_vec_d2h_f64: length mismatch");
    if (n == 0) return 0;
    double* staging = nullptr;
    PGH_TRY(pool_alloc(sizeof(double) * n, (void**)&staging));
    k_f32_to_f64<<<grid_for(n), kBlock, 0, rt().stream>>>(v->data, staging, n);
    PGH_HIP(hipGetLastError());
    const int rc = staged_d2h(host, staging, sizeof(double) * (size_t)n);
    PGH_HIP(hipStreamSynchronize(rt().stream));
    pool_free(staging);
    return rc;
}

extern "C" int pgh_vec_fill(pgh_vec_t v, double value) {
    PGH_CHECK(v, "pgh_vec_fill: null vector");
    if (v->n == 0) return 0;
    k_fill<<<grid_for(v->n), kBlock, 0, rt().stream>>>(v->data, v->n, (float)value);
    PGH_HIP(hipGetLastError());
    return 0;
}

extern "C" int pgh_vec_copy(pgh_vec_t dst, pgh_vec_t src) {
    PGH_CHECK(dst && src && dst->n == src->n, "pgh_vec_copy: length mismatch");
    if (dst->n == 0) return 0;
    PGH_HIP(hipMemcpyAsync(dst->data, src->data, sizeof(float) * dst->n, hipMemcpyDeviceToDevice, rt().stream));
    return 0;
}

extern "C" int pgh_vec_get(pgh_vec_t v, int64_t i, double* out) {
    PGH_CHECK(v && i >= 0 && i < v->n, "pgh_vec_get: index out of range");
    float f = 0.f;
    PGH_HIP(hipMemcpyAsync(&f, v->data + i, sizeof(float), hipMemcpyDeviceToHost, rt().stream));
    PGH_HIP(hipStreamSynchronize(rt().stream));
    *out = (double)f;
    return 0;
}

extern "C" int pgh_vec_set(pgh_vec_t v, int64_t i, double value) {
    PGH_CHECK(v && i >= 0 && i < v->n, "pgh_vec_set: index out of range");
    float f = (float)value;
    PGH_HIP(hipMemcpyAsync(v->data + i, &f, sizeof(float), hipMemcpyHostToDevice, rt().stream));
    PGH_HIP(hipStreamSynchronize(rt().stream));
    return 0;
}

extern "C" int pgh_vec_scatter_set(pgh_vec_t v, const int64_t* idx, const double* val, int64_t count) {
    PGH_CHECK(v, "pgh_vec_scatter_set: null vector");
    if (count == 0) return 0;
    for (int64_t k = 0; k < count; ++k) PGH_CHECK(idx[k] >= 0 && idx[k] < v->n, "pgh_vec_scatter_set: index out of range");
    int64_t* d_idx = nullptr;
    double* d_val = nullptr;
    PGH_HIP(hipMalloc(&d_idx, sizeof(int64_t) * count));
    PGH_HIP(hipMalloc(&d_val, sizeof(double) * count));
    PGH_HIP(hipMemcpyAsync(d_idx, idx, sizeof(int64_t) * count, hipMemcpyHostToDevice, rt().stream));
    PGH_HIP(hipMemcpyAsync(d_val, val, sizeof(double) * count, hipMemcpyHostToDevice, rt().stream));
    k_scatter_set<<<grid_for(count, 1), kBlock, 0, rt().stream>>>(v->data, d_idx, d_val, count);
    PGH_HIP(hipGetLastError());
    PGH_HIP(hipStreamSynchronize(rt().stream));
    PGH_HIP(hipFree(d_idx));
    PGH_HIP(hipFree(d_val));
    return 0;
}

// ------------------------------------------------------------------------------------------------
// elementwise dispatch
// ------------------------------------------------------------------------------------------------
#define PGH_DISPATCH_BIN(OPVAR, CALL)                      \
    switch (OPVAR) {                                       \
        case PGH_ADD: CALL(PGH_ADD); break;                \
        case PGH_SUB: CALL(PGH_SUB); break;                \
        case PGH_MUL: CALL(PGH_MUL); break;                \
        case PGH_DIV: CALL(PGH_DIV); break;                \
        case PGH_POW: CALL(PGH_POW); break;                \
        case PGH_MAXOP: CALL(PGH_MAXOP); break;            \
        case PGH_MINOP: CALL(PGH_MINOP); break;            \
        case PGH_GT: CALL(PGH_GT); break;                  \
        case PGH_GE: CALL(PGH_GE); break;                  \
        case PGH_LT: CALL(PGH_LT); break;                  \
        case PGH_LE: CALL(PGH_LE); break;                  \
        case PGH_EQ: CALL(PGH_EQ); break;                  \
        case PGH_NE: CALL(PGH_NE); break;                  \
        default: return fail("unknown binary operator");   \
    }

extern "C" int pgh_ewise_vv(int op, pgh_vec_t a, pgh_vec_t b, pgh_vec_t out) {
    PGH_CHECK(a && b && out && a->n == b->n && a->n == out->n, "pgh_ewise_vv: length mismatch");
    const int64_t n = a->n;
    if (n == 0) return 0;
    const int vec_ok = aligned16(a->data) && aligned16(b->data) && aligned16(out->data);
    const int grid = grid_for(n, 8);
#define CALL(OP) k_ewise_vv<OP><<<grid, kBlock, 0, rt().stream>>>(a->data, b->data, out->data, n, vec_ok)
    PGH_DISPATCH_BIN(op, CALL)
#undef CALL
    PGH_HIP(hipGetLastError());
    return 0;
}

extern "C" int pgh_ewise_vs(int op, pgh_vec_t a, double scalar, int scalar_on_left, pgh_vec_t out) {
    PGH_CHECK(a && out && a->n == out->n, "pgh_ewise_vs: length mismatch");
    const int64_t n = a->n;
    if (n == 0) return 0;
    const int vec_ok = aligned16(a->data) && aligned16(out->data);
    const int grid = grid_for(n, 8);
    const float s = (float)scalar;
    if (scalar_on_left) {
#define CALL(OP) k_ewise_vs<OP, 1><<<grid, kBlock, 0, rt().stream>>>(a->data, s, out->data, n, vec_ok)
        PGH_DISPATCH_BIN(op, CALL)
#undef CALL
    } else {
#define CALL(OP) k_ewise_vs<OP, 0><<<grid, kBlock, 0, rt().stream>>>(a->data, s, out->data, n, vec_ok)
        PGH_DISPATCH_BIN(op, CALL)
#undef CALL
    }
    PGH_HIP(hipGetLastError());
    return 0;
}

extern "C" int pgh_ewise_unary(int op, pgh_vec_t a, pgh_vec_t out) {
    PGH_CHECK(a && out && a->n == out->n, "pgh_ewise_unary: length mismatch");
    const int64_t n = a->n;
    if (n == 0) return 0;
    const int vec_ok = aligned16(a->data) && aligned16(out->data);
    const int grid = grid_for(n, 8);
    switch (op) {
        case PGH_ABS: k_ewise_un<PGH_ABS><<<grid, kBlock, 0, rt().stream>>>(a->data, out->data, n, vec_ok); break;
        case PGH_EXP: k_ewise_un<PGH_EXP><<<grid, kBlock, 0, rt().stream>>>(a->data, out->data, n, vec_ok); break;
        case PGH_LOG: k_ewise_un<PGH_LOG><<<grid, kBlock, 0, rt().stream>>>(a->data, out->data, n, vec_ok); break;
        case PGH_NEG: k_ewise_un<PGH_NEG><<<grid, kBlock, 0, rt().stream>>>(a->data, out->data, n, vec_ok); break;
        case PGH_SQRT: k_ewise_un<PGH_SQRT><<<grid, kBlock, 0, rt().stream>>>(a->data, out->data, n, vec_ok); break;
        case PGH_SAFE_INV: k_ewise_un<PGH_SAFE_INV><<<grid, kBlock, 0, rt().stream>>>(a->data, out->data, n, vec_ok); break;
        default: return fail("unknown unary operator");
    }
    PGH_HIP(hipGetLastError());
    return 0;
}

extern "C" int pgh_axpby(double a, pgh_vec_t x, double b, pgh_vec_t y, pgh_vec_t out) {
    PGH_CHECK(x && y && out && x->n == y->n && x->n == out->n, "pgh_axpby: length mismatch");
    const int64_t n = x->n;
    if (n == 0) return 0;
    const int vec_ok = aligned16(x->data) && aligned16(y->data) && aligned16(out->data);
    k_axpby<<<grid_for(n, 8), kBlock, 0, rt().stream>>>((float)a, x->data, (float)b, y->data, out->data, n, vec_ok);
    PGH_HIP(hipGetLastError());
    return 0;
}

// =================================================================================================
// reductions (f64 accumulation; wavefront shuffles -> LDS -> per-block partial -> single-block final)
// =================================================================================================
namespace {

// MODE: 0 sum x, 1 sum |x|, 2 max x, 3 min x, 4 dot(x, y), 5 sum |a*sa - b*sb|, 6 max |a*sa - b*sb|
template <int MODE>
__global__ void k_reduce_partials(const float* __restrict__ a, const float* __restrict__ b, double sa, double sb,
                                  int64_t n, int vec_ok, double* __restrict__ partials) {
    __shared__ double s_scratch[4];
    const int64_t tid = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    constexpr bool kMax = (MODE == 2 || MODE == 6);
    constexpr bool kMin = (MODE == 3);
    double acc = kMax ? -INFINITY : (kMin ? INFINITY : 0.0);
    if (MODE == 6) acc = 0.0;
    auto fold = [&](float u, float v) {
        if (MODE == 0) acc += (double)u;
        if (MODE == 1) acc += fabs((double)u);
        if (MODE == 2) acc = fmax(acc, (double)u);
        if (MODE == 3) acc = fmin(acc, (double)u);
        if (MODE == 4) acc += (double)u * (double)v;
        if (MODE == 5) acc += fabs((double)u * sa - (double)v * sb);
        if (MODE == 6) acc = fmax(acc, fabs((double)u * sa - (double)v * sb));
    };
    constexpr bool kTwo = (MODE >= 4);
    int64_t body = vec_ok ? (n >> 2) : 0;
    const float4* a4 = reinterpret_cast<const float4*>(a);
    const float4* b4 = reinterpret_cast<const float4*>(b);
    for (int64_t i = tid; i < body; i += stride) {
        float4 u = a4[i];
        float4 v = kTwo ? b4[i] : u;
        fold(u.x, v.x);
        fold(u.y, v.y);
        fold(u.z, v.z);
        fold(u.w, v.w);
    }
    for (int64_t i = (body << 2) + tid; i < n; i += stride) fold(a[i], kTwo ? b[i] : 0.f);
    double r = block_reduce_256<kMax ? 1 : (kMin ? 2 : 0)>(acc, s_scratch);
    if (threadIdx.x == 0) partials[blockIdx.x] = r;
}

// KIND: 0 sum, 1 max, 2 min
template <int KIND>
__global__ void k_reduce_final(const double* __restrict__ partials, int count, double* __restrict__ out) {
    __shared__ double s_scratch[4];
    double acc = KIND == 0 ? 0.0 : (KIND == 1 ? -INFINITY : INFINITY);
    for (int i = threadIdx.x; i < count; i += blockDim.x) {
        double v = partials[i];
        if (KIND == 0) acc += v;
        if (KIND == 1) acc = fmax(acc, v);
        if (KIND == 2) acc = fmin(acc, v);
    }
    double r = block_reduce_256<KIND>(acc, s_scratch);
    if (threadIdx.x == 0) out[0] = r;
}

template <int MODE>
int run_reduce(const float* a, const float* b, double sa, double sb, int64_t n, double* out) {
    Runtime& r = rt();
    int grid = grid_for(n, 16);
    if (grid > kMaxPartials) grid = kMaxPartials;
    const int vec_ok = aligned16(a) && (b == nullptr || aligned16(b));
    k_reduce_partials<MODE><<<grid, kBlock, 0, r.stream>>>(a, b ? b : a, sa, sb, n, vec_ok, r.d_partials);
    constexpr int KIND = (MODE == 2 || MODE == 6) ? 1 : (MODE == 3 ? 2 : 0);
    k_reduce_final<KIND><<<1, kBlock, 0, r.stream>>>(r.d_partials, grid, r.d_scalars);
    PGH_HIP(hipGetLastError());
    PGH_TRY(scalars_to_host(0, 1));
    *out = r.h_scalars[0];
    return 0;
}

}  // namespace

extern "C" int pgh_reduce(int kind, pgh_vec_t x, double* out) {
    PGH_CHECK(x && out, "pgh_reduce: null argument");
    if (x->n == 0) {
        PGH_CHECK(kind == PGH_SUM || kind == PGH_ABSSUM, "pgh_reduce: max/min of an empty vector");
        *out = 0.0;
        return 0;
    }
    switch (kind) {
        case PGH_SUM: return run_reduce<0>(x->data, nullptr, 1, 1, x->n, out);
        case PGH_ABSSUM: return run_reduce<1>(x->data, nullptr, 1, 1, x->n, out);
        case PGH_MAX: return run_reduce<2>(x->data, nullptr, 1, 1, x->n, out);
        case PGH_MIN: return run_reduce<3>(x->data, nullptr, 1, 1, x->n, out);
        default: return fail("pgh_reduce: unknown kind");
    }
}

extern "C" int pgh_dot(pgh_vec_t x, pgh_vec_t y, double* out) {
    PGH_CHECK(x && y && out && x->n == y->n, "pgh_dot: length mismatch");
    if (x->n == 0) {
        *out = 0.0;
        return 0;
    }
    return run_reduce<4>(x->data, y->data, 1, 1, x->n, out);
}

extern "C" int pgh_scaled_residual(int kind, pgh_vec_t y, double y_scale, pgh_vec_t x, double x_scale, double* err) {
    PGH_CHECK(y && x && err && x->n == y->n, "pgh_scaled_residual: length mismatch");
    if (x->n == 0) {
        *err = 0.0;
        return 0;
    }
    ProfScope prof(PGH_K_RESIDUAL);
    if (kind == PGH_ERR_LINF) return run_reduce<6>(y->data, x->data, y_scale, x_scale, x->n, err);
    PGH_CHECK(kind == PGH_ERR_MABS || kind == PGH_ERR_L1, "pgh_scaled_residual: unknown kind");
    PGH_TRY((run_reduce<5>(y->data, x->data, y_scale, x_scale, x->n, err)));
    if (kind == PGH_ERR_MABS) *err /= (double)x->n;      // measures/supervised.py:106
    return 0;
}

extern "C" int pgh_residual(int kind, pgh_vec_t a, pgh_vec_t b, double* out) {
    return pgh_scaled_residual(kind, a, 1.0, b, 1.0, out);
}

// =================================================================================================
// dense [n, b] slabs
// =================================================================================================
namespace {
__global__ void k_mat_set_col(float* __restrict__ m, int64_t n, int b, int col, const float* __restrict__ v) {
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
        m[i * b + col] = v[i];
}
__global__ void k_mat_get_col(const float* __restrict__ m, int64_t n, int b, int col, float* __restrict__ v) {
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
        v[i] = m[i * b + col];
}
}  // namespace

extern "C" int pgh_mat_alloc(int64_t n, int32_t b, pgh_mat_t* out) {
    PGH_TRY(ensure_init());
    PGH_CHECK(n >= 0 && b >= 1, "pgh_mat_alloc: bad shape");
    pgh_mat_s* m = new pgh_mat_s();
    m->n = n;
    m->b = b;
    size_t bytes = sizeof(float) * (size_t)(n > 0 ? n : 1) * (size_t)b;
    if (pool_alloc(bytes, (void**)&m->data) != 0) {
        delete m;
        return -1;
    }
    *out = m;
    return 0;
}
extern "C" int pgh_mat_free(pgh_mat_t m) {
    if (!m) return 0;
    pool_free(m->data);
    delete m;
    return 0;
}
extern "C" int pgh_mat_shape(pgh_mat_t m, int64_t* n, int32_t* b) {
    PGH_CHECK(m, "pgh_mat_shape: null");
    *n = m->n;
    *b = m->b;
    return 0;
}
extern "C" void* pgh_mat_ptr(pgh_mat_t m) { return m ? (void*)m->data : nullptr; }

extern "C" int pgh_mat_h2d_f64(pgh_mat_t m, const double* host) {
    PGH_CHECK(m, "pgh_mat_h2d_f64: null");
    const int64_t total = m->n * m->b;
    if (total == 0) return 0;
    double* staging = nullptr;
    PGH_TRY(pool_alloc(sizeof(double) * total, (void**)&staging));
    PGH_HIP(hipMemcpyAsync(staging, host, sizeof(double) * total, hipMemcpyHostToDevice, rt().stream));
    k_f64_to_f32<<<grid_for(total), kBlock, 0, rt().stream>>>(staging, m->data, total);
    PGH_HIP(hipGetLastError());
    PGH_HIP(hipStreamSynchronize(rt().stream));
    pool_free(staging);
    return 0;
}
extern "C" int pgh_mat_d2h_f64(pgh_mat_t m, double* host) {
    PGH_CHECK(m, "pgh_mat_d2h_f64: null");
    const int64_t total = m->n * m->b;
    if (total == 0) return 0;
    double* staging = nullptr;
    PGH_TRY(pool_alloc(sizeof(double) * total, (void**)&staging));
    k_f32_to_f64<<<grid_for(total), kBlock, 0, rt().stream>>>(m->data, staging, total);
    PGH_HIP(hipGetLastError());
    const int rc = staged_d2h(host, staging, sizeof(double) * (size_t)total);
    PGH_HIP(hipStreamSynchronize(rt().stream));
    pool_free(staging);
    return rc;
}
namespace {
// per-column sum of |.| of a row-major slab: every thread keeps ONE column (the grid stride is a multiple of b), f64;
// the threads of a workgroup that share a column are folded through LDS in thread order (deterministic)
__global__ __launch_bounds__(kBlock) void k_mat_col_abssum(const float* __restrict__ m, int64_t n, int b, double* __restrict__ partial /* [grid][b] */) {
    __shared__ double s_acc[kBlock];
    const int64_t total = n * b;
    // stride = the largest multiple of b the launch covers: the threads beyond it idle (rounding UP instead would leave
    // the elements [threads, stride) of every period to nobody)
    const int64_t stride = ((int64_t)gridDim.x * kBlock) / b * b;
    const int64_t block_first = blockIdx.x * (int64_t)kBlock;
    double acc = 0.0;
    if (block_first + threadIdx.x < stride)
        for (int64_t i = block_first + threadIdx.x; i < total; i += stride) acc += fabs((double)m[i]);
    s_acc[threadIdx.x] = acc;
    __syncthreads();
    for (int j = threadIdx.x; j < b; j += kBlock) {
        double t = 0.0;
        int first = (int)((j - block_first % b + b) % b);          // first thread of this workgroup that owns column j
        for (int k = first; k < kBlock; k += b) t += s_acc[k];
        partial[(int64_t)blockIdx.x * b + j] = t;
    }
}
__global__ void k_mat_fold_cols(const double* __restrict__ partial, int count, int b, double* __restrict__ out) {
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= b) return;
    double acc = 0.0;
    for (int i = 0; i < count; ++i) acc += partial[(int64_t)i * b + j];     // fixed order
    out[j] = acc;
}
__global__ void k_mat_div_cols(const float* __restrict__ m, int64_t total, int b, const float* __restrict__ div, float* __restrict__ out) {
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const float d = div[i % b];
        out[i] = d != 0.f ? m[i] / d : m[i];
    }
}
// dst[:, dst_first + j] = src[:, src_first + j] for j < count
__global__ void k_mat_copy_cols(const float* __restrict__ src, int ld_src, int src_first, float* __restrict__ dst, int ld_dst, int dst_first,
                                int count, int64_t n) {
    const int64_t total = n * count;
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t r = i / count;
        const int j = (int)(i - r * count);
        dst[r * ld_dst + dst_first + j] = src[r * ld_src + src_first + j];
    }
}
}  // namespace

extern "C" int pgh_mat_col_abssum(pgh_mat_t m, double* out_host) {
    PGH_CHECK(m && out_host, "pgh_mat_col_abssum: null argument");
    PGH_CHECK(m->b <= 1024, "pgh_mat_col_abssum: at most 1024 columns");
    Runtime& r = rt();
    for (int j = 0; j < m->b; ++j) out_host[j] = 0.0;
    if (m->n == 0) return 0;
    int grid = r.num_cus * 4;
    double* partial = nullptr;
    double* folded = nullptr;
    PGH_TRY(pool_alloc(sizeof(double) * (size_t)grid * m->b, (void**)&partial));
    PGH_TRY(pool_alloc(sizeof(double) * (size_t)m->b, (void**)&folded));
    k_mat_col_abssum<<<grid, kBlock, 0, r.stream>>>(m->data, m->n, m->b, partial);
    k_mat_fold_cols<<<(m->b + 63) / 64, 64, 0, r.stream>>>(partial, grid, m->b, folded);
    PGH_HIP(hipGetLastError());
    PGH_HIP(hipMemcpyAsync(out_host, folded, sizeof(double) * m->b, hipMemcpyDeviceToHost, r.stream));
    PGH_HIP(hipStreamSynchronize(r.stream));
    pool_free(partial);
    pool_free(folded);
    return 0;
}

// out[i] = sum_j m[i, j] * c[j] over the first `count` columns (f64 accumulation, one rounding to f32): a polynomial filter
// evaluated from the stored powers {(M^T)^k p} of its personalization.  A wavefront takes 64 consecutive
// rows; its lanes read the rows' leading `count` floats (the rows are contiguous, so a wavefront covers one contiguous
// range of the slab); the coefficients sit in LDS.
namespace {
__global__ __launch_bounds__(kBlock) void k_mat_gemv(const float* __restrict__ m, int64_t n, int b, const double* __restrict__ coeffs, int count,
                                                      float* __restrict__ out) {
    __shared__ double s_c[1024];
    for (int j = threadIdx.x; j < count; j += kBlock) s_c[j] = coeffs[j];
    __syncthreads();
    for (int64_t i = blockIdx.x * (int64_t)kBlock + threadIdx.x; i < n; i += (int64_t)gridDim.x * kBlock) {
        const float* __restrict__ row = m + i * b;
        double acc = 0.0;
        int j = 0;
        for (; j + 4 <= count; j += 4) {                   // four independent loads per round
            const float a0 = row[j], a1 = row[j + 1], a2 = row[j + 2], a3 = row[j + 3];
            acc += (double)a0 * s_c[j];
            acc += (double)a1 * s_c[j + 1];
            acc += (double)a2 * s_c[j + 2];
            acc += (double)a3 * s_c[j + 3];
        }
        for (; j < count; ++j) acc += (double)row[j] * s_c[j];
        out[i] = (float)acc;
    }
}
}  // namespace

extern "C" int pgh_mat_gemv(pgh_mat_t m, const double* coeffs_host, int32_t count, pgh_vec_t out) {
    PGH_CHECK(m && out && (coeffs_host || count == 0), "pgh_mat_gemv: null argument");
    PGH_CHECK(count >= 0 && count <= m->b && count <= 1024 && out->n == m->n, "pgh_mat_gemv: shape mismatch");
    if (m->n == 0) return 0;
    Runtime& r = rt();
    double* d = nullptr;
    PGH_TRY(pool_alloc(sizeof(double) * (size_t)(count > 0 ? count : 1), (void**)&d));
    if (count > 0) {
        PGH_HIP(hipMemcpyAsync(d, coeffs_host, sizeof(double) * count, hipMemcpyHostToDevice, r.stream));
        PGH_HIP(hipStreamSynchronize(r.stream));           // the caller's array may go away
    }
    k_mat_gemv<<<grid_for(m->n, 8), kBlock, 0, r.stream>>>(m->data, m->n, m->b, d, count, out->data);
    PGH_HIP(hipGetLastError());
    pool_free(d);
    return 0;
}

// [n, count] x [count, P] with P <= 64: the lanes of a row share the row's values (same address: one request) and hold four
// probes each; the coefficients sit in LDS as doubles.  Reads the slab once whatever P is.
namespace {
template <int LPR>
__global__ __launch_bounds__(kBlock) void k_mat_gemm(const float* __restrict__ m, int64_t n, int b, const double* __restrict__ coeffs, int count,
                                                      int probes, int ld_out, int accumulate, float* __restrict__ out) {
    __shared__ double s_c[64 * 64];
    for (int j = threadIdx.x; j < count * probes; j += kBlock) s_c[j] = coeffs[j];
    __syncthreads();
    constexpr int ROWS = kBlock / LPR;
    const int q0 = (threadIdx.x % LPR) * 4, r_in = threadIdx.x / LPR;
    for (int64_t i = blockIdx.x * (int64_t)ROWS + r_in; i < n; i += (int64_t)gridDim.x * ROWS) {
        const float* __restrict__ row = m + i * b;
        double acc[4] = {0.0, 0.0, 0.0, 0.0};
        for (int j = 0; j < count; ++j) {
            const double a = (double)row[j];
#pragma unroll
            for (int u = 0; u < 4; ++u)
                if (q0 + u < probes) acc[u] += a * s_c[j * probes + q0 + u];
        }
#pragma unroll
        for (int u = 0; u < 4; ++u)
            if (q0 + u < probes) {
                float* o = out + i * ld_out + q0 + u;
                *o = accumulate ? (float)((double)*o + acc[u]) : (float)acc[u];
            }
    }
}
}  // namespace

extern "C" int pgh_mat_gemm(pgh_mat_t m, const double* coeffs_host, int32_t count, int32_t probes, int32_t accumulate, pgh_mat_t out) {
    PGH_CHECK(m && out && (coeffs_host || count == 0), "pgh_mat_gemm: null argument");
    PGH_CHECK(count >= 0 && count <= m->b && count <= 64 && probes >= 1 && probes <= out->b && probes <= 64 && out->n == m->n,
              "pgh_mat_gemm: shape mismatch");
    if (m->n == 0) return 0;
    Runtime& r = rt();
    double* d = nullptr;
    const size_t elems = (size_t)(count > 0 ? count : 1) * (size_t)probes;
    PGH_TRY(pool_alloc(sizeof(double) * elems, (void**)&d));
    if (count > 0) {
        PGH_HIP(hipMemcpyAsync(d, coeffs_host, sizeof(double) * (size_t)count * probes, hipMemcpyHostToDevice, r.stream));
        PGH_HIP(hipStreamSynchronize(r.stream));           // the caller's array may go away
    }
    const int lpr = probes > 32 ? 16 : (probes > 16 ? 8 : (probes > 8 ? 4 : (probes > 4 ? 2 : 1)));
    const int grid = grid_for(m->n * lpr, 4);
#define PGH_GEMM(LPR) k_mat_gemm<LPR><<<grid, kBlock, 0, r.stream>>>(m->data, m->n, m->b, d, count, probes, out->b, accumulate, out->data)
    switch (lpr) {
        case 16: PGH_GEMM(16); break;
        case 8: PGH_GEMM(8); break;
        case 4: PGH_GEMM(4); break;
        case 2: PGH_GEMM(2); break;
        default: PGH_GEMM(1); break;
    }
#undef PGH_GEMM
    PGH_HIP(hipGetLastError());
    pool_free(d);
    return 0;
}

// Ordinals / Top (postprocess.py:163-195, 246-290): positions of the entries in descending order of value, ties in ascending
// index order (python's sorted(..., reverse=True) is stable).  One device radix sort of (value, index) pairs.
namespace {
__global__ void k_iota_i32(int32_t* p, int64_t n) {
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) p[i] = (int32_t)i;
}
__global__ void k_scatter_ordinals(const int32_t* __restrict__ order, int64_t n, float* __restrict__ out) {
    for (int64_t k = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; k < n; k += (int64_t)gridDim.x * blockDim.x)
        out[order[k]] = (float)(k + 1);
}
int sort_descending(pgh_vec_t x, float** keys_out, int32_t** order_out) {
    Runtime& r = rt();
    const int64_t n = x->n;
    PGH_CHECK(n < 2147483647LL, "sort: vector too long");
    float* keys = nullptr;
    int32_t *idx = nullptr, *order = nullptr;
    PGH_TRY(pool_alloc(sizeof(float) * (size_t)(n > 0 ? n : 1), (void**)&keys));
    PGH_TRY(pool_alloc(sizeof(int32_t) * (size_t)(n > 0 ? n : 1), (void**)&idx));
    PGH_TRY(pool_alloc(sizeof(int32_t) * (size_t)(n > 0 ? n : 1), (void**)&order));
    if (n > 0) {
        k_iota_i32<<<grid_for(n), kBlock, 0, r.stream>>>(idx, n);
        size_t temp_bytes = 0;
        PGH_HIP(hipcub::DeviceRadixSort::SortPairsDescending(nullptr, temp_bytes, x->data, keys, idx, order, (int)n, 0, 32, r.stream));
        void* temp = nullptr;
        PGH_TRY(pool_alloc(temp_bytes > 0 ? temp_bytes : 1, &temp));
        PGH_HIP(hipcub::DeviceRadixSort::SortPairsDescending(temp, temp_bytes, x->data, keys, idx, order, (int)n, 0, 32, r.stream));
        pool_free(temp);
    }
    pool_free(idx);
    *keys_out = keys;
    *order_out = order;
    return 0;
}
}  // namespace

extern "C" int pgh_vec_ordinals(pgh_vec_t x, pgh_vec_t out) {
    PGH_CHECK(x && out && x->n == out->n && x->data != out->data, "pgh_vec_ordinals: bad arguments");
    float* keys = nullptr;
    int32_t* order = nullptr;
    PGH_TRY(sort_descending(x, &keys, &order));
    if (x->n > 0) k_scatter_ordinals<<<grid_for(x->n), kBlock, 0, rt().stream>>>(order, x->n, out->data);
    PGH_HIP(hipGetLastError());
    pool_free(keys);
    pool_free(order);
    return 0;
}

extern "C" int pgh_vec_kth_largest(pgh_vec_t x, int64_t k, double* value) {
    PGH_CHECK(x && value && k >= 1 && k <= x->n, "pgh_vec_kth_largest: k outside [1, n]");
    float* keys = nullptr;
    int32_t* order = nullptr;
    PGH_TRY(sort_descending(x, &keys, &order));
    float v = 0.f;
    PGH_HIP(hipMemcpyAsync(&v, keys + (k - 1), sizeof(float), hipMemcpyDeviceToHost, rt().stream));
    PGH_HIP(hipStreamSynchronize(rt().stream));
    pool_free(keys);
    pool_free(order);
    *value = (double)v;
    return 0;
}

namespace {
// Ranks of the positives in the DESCENDING order of the scores, ties at their mid-rank (what sklearn's roc_curve / auc
// amount to, supervised.py:255-263): an element at sorted position k whose tie group spans [gs, ge] has ascending mid-rank
// n - (gs + ge) / 2.  The group is found by two binary searches in the sorted keys (positives only).
__global__ void k_auc_partials(const float* __restrict__ keys, const int32_t* __restrict__ order, const float* __restrict__ labels,
                               int64_t n, double* __restrict__ part_rank, double* __restrict__ part_pos) {
    __shared__ double s_scratch[4];
    double rank_sum = 0.0, pos = 0.0;
    for (int64_t k = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; k < n; k += (int64_t)gridDim.x * blockDim.x) {
        if (labels[order[k]] == 0.f) continue;
        const float key = keys[k];
        int64_t lo = 0, hi = k;                            // first index whose key is not above this one
        while (lo < hi) {
            const int64_t mid = (lo + hi) >> 1;
            if (keys[mid] > key) lo = mid + 1; else hi = mid;
        }
        const int64_t gs = lo;
        lo = k, hi = n;                                    // first index whose key is below this one
        while (lo < hi) {
            const int64_t mid = (lo + hi) >> 1;
            if (keys[mid] >= key) lo = mid + 1; else hi = mid;
        }
        const int64_t ge = lo - 1;
        rank_sum += (double)n - 0.5 * (double)(gs + ge);
        pos += 1.0;
    }
    const double a = block_reduce_256<0>(rank_sum, s_scratch);
    __syncthreads();
    const double b = block_reduce_256<0>(pos, s_scratch);
    if (threadIdx.x == 0) {
        part_rank[blockIdx.x] = a;
        part_pos[blockIdx.x] = b;
    }
}

// relative drop between neighbours of the descending order (postprocess.py:335-343): drop_k = (v_k - v_{k+1}) / v_k for v_k > 0
__global__ void k_gap_max(const float* __restrict__ keys, int64_t n, double* __restrict__ part_max) {
    __shared__ double s_scratch[4];
    double best = 0.0;
    for (int64_t k = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; k + 1 < n; k += (int64_t)gridDim.x * blockDim.x) {
        const double prev = (double)keys[k], cur = (double)keys[k + 1];
        if (prev > 0.0) best = fmax(best, (prev - cur) / prev);
    }
    const double r = block_reduce_256<1>(best, s_scratch);
    if (threadIdx.x == 0) part_max[blockIdx.x] = r;
}
// the FIRST position that reaches the largest drop (the reference keeps the first: strict ">")
__global__ void k_gap_first(const float* __restrict__ keys, int64_t n, const double* __restrict__ target, double* __restrict__ part_min) {
    __shared__ double s_scratch[4];
    const double want = target[0];
    double first = 1e300;
    for (int64_t k = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; k + 1 < n; k += (int64_t)gridDim.x * blockDim.x) {
        const double prev = (double)keys[k], cur = (double)keys[k + 1];
        if (prev > 0.0 && (prev - cur) / prev == want) first = fmin(first, (double)k);
    }
    const double r = block_reduce_256<2>(first, s_scratch);
    if (threadIdx.x == 0) part_min[blockIdx.x] = r;
}
}  // namespace

// AUC of `scores` against binary `labels` (non-zero = positive) with one device sort: (sum of the positives' mid-ranks -
// n_pos (n_pos + 1) / 2) / (n_pos n_neg).  Reference: measures/supervised.py:255-263 (sklearn.metrics.roc_curve + auc).
extern "C" int pgh_auc(pgh_vec_t labels, pgh_vec_t scores, double* out, int64_t* num_positive) {
    PGH_CHECK(labels && scores && out && labels->n == scores->n, "pgh_auc: bad arguments");
    Runtime& r = rt();
    const int64_t n = scores->n;
    float* keys = nullptr;
    int32_t* order = nullptr;
    PGH_TRY(sort_descending(scores, &keys, &order));
    int grid = grid_for(n, 8);
    if (grid > kMaxPartials) grid = kMaxPartials;
    if (n > 0) k_auc_partials<<<grid, kBlock, 0, r.stream>>>(keys, order, labels->data, n, r.d_partials, r.d_partials + kMaxPartials);
    else PGH_HIP(hipMemsetAsync(r.d_partials, 0, sizeof(double) * 2 * kMaxPartials, r.stream));
    k_reduce_final<0><<<1, kBlock, 0, r.stream>>>(r.d_partials, n > 0 ? grid : 1, r.d_scalars);
    k_reduce_final<0><<<1, kBlock, 0, r.stream>>>(r.d_partials + kMaxPartials, n > 0 ? grid : 1, r.d_scalars + 1);
    PGH_HIP(hipGetLastError());
    PGH_HIP(hipMemcpyAsync(r.h_scalars, r.d_scalars, sizeof(double) * 2, hipMemcpyDeviceToHost, r.stream));
    PGH_HIP(hipStreamSynchronize(r.stream));
    pool_free(keys);
    pool_free(order);
    const double rank_sum = r.h_scalars[0], pos = r.h_scalars[1], neg = (double)n - pos;
    if (num_positive) *num_positive = (int64_t)pos;
    *out = (pos > 0.0 && neg > 0.0) ? (rank_sum - pos * (pos + 1.0) / 2.0) / (pos * neg) : 0.0;   // the caller raises when all labels agree
    return 0;
}

// Threshold("gap") (postprocess.py:328-343): the score that follows the largest relative drop of the descending order (the
// first such drop), 0 when no positive score is followed by a smaller one
extern "C" int pgh_vec_gap_threshold(pgh_vec_t x, double* threshold) {
    PGH_CHECK(x && threshold, "pgh_vec_gap_threshold: null argument");
    Runtime& r = rt();
    const int64_t n = x->n;
    *threshold = 0.0;
    if (n < 2) return 0;
    float* keys = nullptr;
    int32_t* order = nullptr;
    PGH_TRY(sort_descending(x, &keys, &order));
    int grid = grid_for(n, 8);
    if (grid > kMaxPartials) grid = kMaxPartials;
    k_gap_max<<<grid, kBlock, 0, r.stream>>>(keys, n, r.d_partials);
    k_reduce_final<1><<<1, kBlock, 0, r.stream>>>(r.d_partials, grid, r.d_scalars);
    k_gap_first<<<grid, kBlock, 0, r.stream>>>(keys, n, r.d_scalars, r.d_partials + kMaxPartials);
    k_reduce_final<2><<<1, kBlock, 0, r.stream>>>(r.d_partials + kMaxPartials, grid, r.d_scalars + 1);
    PGH_HIP(hipGetLastError());
    PGH_HIP(hipMemcpyAsync(r.h_scalars, r.d_scalars, sizeof(double) * 2, hipMemcpyDeviceToHost, r.stream));
    PGH_HIP(hipStreamSynchronize(r.stream));
    const double drop = r.h_scalars[0], at = r.h_scalars[1];
    if (drop > 0.0 && at < 1e299) {
        float v = 0.f;
        PGH_HIP(hipMemcpyAsync(&v, keys + (int64_t)at + 1, sizeof(float), hipMemcpyDeviceToHost, r.stream));
        PGH_HIP(hipStreamSynchronize(r.stream));
        *threshold = (double)v;
    }
    pool_free(keys);
    pool_free(order);
    return 0;
}

extern "C" int pgh_mat_div_cols(pgh_mat_t m, const double* divisors_host, pgh_mat_t out) {
    PGH_CHECK(m && out && divisors_host && m->n == out->n && m->b == out->b, "pgh_mat_div_cols: shape mismatch");
    if (m->n == 0) return 0;
    Runtime& r = rt();
    std::vector<float> h(m->b);
    for (int j = 0; j < m->b; ++j) h[j] = (float)divisors_host[j];
    float* d = nullptr;
    PGH_TRY(pool_alloc(sizeof(float) * (size_t)m->b, (void**)&d));
    PGH_HIP(hipMemcpyAsync(d, h.data(), sizeof(float) * m->b, hipMemcpyHostToDevice, r.stream));
    PGH_HIP(hipStreamSynchronize(r.stream));                       // h goes out of scope
    const int64_t total = m->n * m->b;
    k_mat_div_cols<<<grid_for(total, 4), kBlock, 0, r.stream>>>(m->data, total, m->b, d, out->data);
    PGH_HIP(hipGetLastError());
    pool_free(d);
    return 0;
}

extern "C" int pgh_mat_get_cols(pgh_mat_t m, int32_t first, pgh_mat_t out) {
    PGH_CHECK(m && out && m->n == out->n && first >= 0 && first + out->b <= m->b, "pgh_mat_get_cols: shape mismatch");
    if (m->n == 0) return 0;
    k_mat_copy_cols<<<grid_for(m->n * out->b, 4), kBlock, 0, rt().stream>>>(m->data, m->b, first, out->data, out->b, 0, out->b, m->n);
    PGH_HIP(hipGetLastError());
    return 0;
}

extern "C" int pgh_mat_set_cols(pgh_mat_t m, int32_t first, pgh_mat_t src) {
    PGH_CHECK(m && src && m->n == src->n && first >= 0 && first + src->b <= m->b, "pgh_mat_set_cols: shape mismatch");
    if (m->n == 0) return 0;
    k_mat_copy_cols<<<grid_for(m->n * src->b, 4), kBlock, 0, rt().stream>>>(src->data, src->b, 0, m->data, m->b, first, src->b, m->n);
    PGH_HIP(hipGetLastError());
    return 0;
}

extern "C" int pgh_mat_set_col(pgh_mat_t m, int32_t col, pgh_vec_t v) {
    PGH_CHECK(m && v && v->n == m->n && col >= 0 && col < m->b, "pgh_mat_set_col: shape mismatch");
    if (m->n == 0) return 0;
    k_mat_set_col<<<grid_for(m->n, 1), kBlock, 0, rt().stream>>>(m->data, m->n, m->b, col, v->data);
    PGH_HIP(hipGetLastError());
    return 0;
}
extern "C" int pgh_mat_get_col(pgh_mat_t m, int32_t col, pgh_vec_t v) {
    PGH_CHECK(m && v && v->n == m->n && col >= 0 && col < m->b, "pgh_mat_get_col: shape mismatch");
    if (m->n == 0) return 0;
    k_mat_get_col<<<grid_for(m->n, 1), kBlock, 0, rt().stream>>>(m->data, m->n, m->b, col, v->data);
    PGH_HIP(hipGetLastError());
    return 0;
}

PGH_WARM_KERNEL(k_post_scalars)
