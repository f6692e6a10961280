// Phase A of the propagation-blocking image (pgh_pb.hip) as a device body: its own launch (k_pb_gather) and one of the two
// roles of the merged front kernel of a step (k_step_front, pgh_bsf.hip).  Internal; not part of the C-ABI.
#pragma once
#include "pgh_kernels.h"

// diagnostic builds only (tools/build_variants.sh): 1 no chunk fill, 2 fills only, 4 no stores (phase A); 8 no loads,
// 16 no atomics (phase B), 32 no epilogue
#ifndef PGH_PROBE_PB
#define PGH_PROBE_PB 0
#endif

namespace pgh {

constexpr int kPbChunk = 32768;          // sources per chunk: 128 KB of LDS in phase A
constexpr int kPbThreads = 1024;

// s_x: kPbChunk floats of LDS (the chunk's slice of the gather vector), s_amax: one more LDS word.  vblock: which share of
// the A-order stream this workgroup takes (PbFormat::task_range).
template <bool HAS_VAL>
__device__ __forceinline__ void pb_gather_body(float* __restrict__ s_x, uint32_t* __restrict__ s_amax, const PbView& f,
                                               const float* __restrict__ xg, const int vblock) {
    if (threadIdx.x == 0) *s_amax = 0u;
    uint32_t amax = 0u;                 // bit pattern of max |value| this thread wrote (NaN > inf > finite as integers)
    // this workgroup's share of the entry stream: consecutive pieces, each inside one chunk; the LDS image of the chunk
    // is refilled only when the chunk changes
    const int piece_begin = f.task_range[vblock], piece_end = f.task_range[vblock + 1];
    int loaded = -1;
    for (int piece = piece_begin; piece < piece_end; ++piece) {
        const int4 task = f.task[piece];
        if (task.x != loaded && !(PGH_PROBE_PB & 1)) {
            __syncthreads();
            // cold ids [first_id, first_id + chunk) -> positions in the gather vector, block by block: the block loop is
            // unrolled so that the layout tables are read with constant indices (scalar loads), and a thread keeps 8
            // independent loads in flight
            const int64_t first_id = (int64_t)task.x * f.chunk;
            const int64_t last_id = min(first_id + f.chunk, f.num_cold);
#pragma unroll
            for (int b = 0; b < 8; ++b) {
                if (b >= f.num_blocks) continue;
                const int64_t lo = max(first_id, f.cold_prefix[b]), hi = min(last_id, f.cold_prefix[b + 1]);
                if (lo >= hi) continue;                     // wavefront-uniform
                const float* __restrict__ src = xg + f.xg_base[b] + f.hot - f.cold_prefix[b];      // src[id] = value of cold id
                // rounds of 8 loads per thread (the whole chunk in one round of 32 was measured: 7 us SLOWER per launch; 16-byte
                // loads from the first aligned element on -- one round of 8 per chunk -- no different: 77.6 vs 78.0 us, the fills
                // of one share hide behind the streams of the others)
                constexpr int FU = 8;
                for (int64_t i0 = lo + threadIdx.x; i0 < hi; i0 += kPbThreads * FU) {
                    float v[FU];
#pragma unroll
                    for (int u = 0; u < FU; ++u) {
                        const int64_t id = i0 + (int64_t)u * kPbThreads;
                        v[u] = src[min(id, hi - 1)];
                    }
#pragma unroll
                    for (int u = 0; u < FU; ++u) {
                        const int64_t id = i0 + (int64_t)u * kPbThreads;
                        if (id < hi) s_x[id - first_id] = v[u];
                    }
                }
            }
            __syncthreads();
            loaded = task.x;
        }
        // every lane takes one group of 8 consecutive entries (pieces are whole groups): one 16-byte load of source
        // indices, one 4-byte load of the group's place in B order, two 16-byte stores of values
        const int64_t body_begin = task.y, body_end = task.z;
        // Software pipeline over rounds of P groups per lane: the loads of round i + 1 are issued BEFORE the gathers and
        // stores of round i.  vmcnt counts loads and stores in one in-order queue, so a loop that loads, gathers, stores
        // and only then loads again makes every round wait for the previous round's stores to complete (measured:
        // reads alone 40 us, with the stores 80 us -- no overlap at all).
        constexpr int P = HAS_VAL ? 2 : 4;
        if (PGH_PROBE_PB & 2) continue;
        struct Round {
            u16x8    s8[P];
            uint32_t to[P];
            f32x4    w0[HAS_VAL ? P : 1], w1[HAS_VAL ? P : 1];
        };
        const int64_t step = (int64_t)kPbThreads * 8 * P;
        auto fetch = [&](Round& r, int64_t e0) __attribute__((always_inline)) {
#pragma unroll
            for (int q = 0; q < P; ++q) {
                const int64_t e = e0 + (int64_t)q * kPbThreads * 8;
                const bool ok = e < body_end;
                r.s8[q] = ok ? __builtin_nontemporal_load(reinterpret_cast<const u16x8*>(f.sloc + e)) : u16x8{0, 0, 0, 0, 0, 0, 0, 0};
                r.to[q] = ok ? __builtin_nontemporal_load(f.dstg + (e >> 3)) : 0u;
                if (HAS_VAL) {
                    r.w0[q] = ok ? __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(f.val + e)) : f32x4{0.f, 0.f, 0.f, 0.f};
                    r.w1[q] = ok ? __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(f.val + e + 4)) : f32x4{0.f, 0.f, 0.f, 0.f};
                }
            }
        };
        auto emit = [&](const Round& r, int64_t e0) __attribute__((always_inline)) {
#pragma unroll
            for (int q = 0; q < P; ++q) {
                const int64_t e = e0 + (int64_t)q * kPbThreads * 8;
                if (e >= body_end) continue;
                f32x4 lo, hi;
                lo.x = s_x[r.s8[q][0]];
                lo.y = s_x[r.s8[q][1]];
                lo.z = s_x[r.s8[q][2]];
                lo.w = s_x[r.s8[q][3]];
                hi.x = s_x[r.s8[q][4]];
                hi.y = s_x[r.s8[q][5]];
                hi.z = s_x[r.s8[q][6]];
                hi.w = s_x[r.s8[q][7]];
                if (HAS_VAL) {
                    lo *= r.w0[q];
                    hi *= r.w1[q];
                }
                if (PGH_PROBE_PB & 4) {
                    if (lo.x + hi.w == 123.456f) f.tmp[e] = lo.y;
                    continue;
                }
#pragma unroll
                for (int k = 0; k < 4; ++k)
                    amax = max(amax, max(__float_as_uint(lo[k]) & 0x7fffffffu, __float_as_uint(hi[k]) & 0x7fffffffu));
                float* __restrict__ dst = (PGH_PROBE_PB & 64) ? f.tmp + e : f.tmp + (int64_t)r.to[q] * 8;   // 64: diagnostic, sequential stores
                *reinterpret_cast<f32x4*>(dst) = lo;
                *reinterpret_cast<f32x4*>(dst + 4) = hi;
            }
        };
        Round r0, r1;
        int64_t e0 = body_begin + (int64_t)threadIdx.x * 8;
        fetch(r0, e0);
        while (e0 < body_end) {
            fetch(r1, e0 + step);
            emit(r0, e0);
            e0 += step;
            if (e0 >= body_end) break;
            fetch(r0, e0 + step);
            emit(r1, e0);
            e0 += step;
        }
    }
    // max |value| of this launch: wavefront -> workgroup -> one global atomic
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) amax = max(amax, (uint32_t)__shfl_xor((int)amax, d, 64));
    __syncthreads();
    if ((threadIdx.x & 63) == 0 && amax != 0u) atomicMax(s_amax, amax);
    __syncthreads();
    if (threadIdx.x == 0 && *s_amax != 0u) atomicMax(f.amax, *s_amax);
}

}  // namespace pgh
