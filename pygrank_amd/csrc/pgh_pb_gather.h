// Phase A of the propagation-blocking image (pgh_pb.hip) as a device body: its own launch (k_pb_gather) and one of the two
// roles of the merged front kernel of a step (k_step_front, pgh_bsf.hip).  Internal; not part of the C-ABI.
#pragma once
#include "pgh_kernels.h"

// diagnostic builds only (tools/build_variants.sh): 1 no chunk fill, 2 fills only, 4 no stores (phase A); 8 no loads,
// 16 no atomics (phase B), 32 no epilogue, 64 sequential stores, 128 conflict-free LDS gathers (phase A), 256 conflict-free LDS atomics (phase B)
#ifndef PGH_FILL_GLDS
#define PGH_FILL_GLDS 1
#endif
#ifndef PGH_PROBE_PB
#define PGH_PROBE_PB 0
#endif
// cache policy of the values handed from phase A to phase B (diagnostic builds): streaming stores / plain loads
#ifndef PGH_PB_TMP_NTSTORE
#define PGH_PB_TMP_NTSTORE 0
#endif
#ifndef PGH_PB_TMP_PLAINLOAD
#define PGH_PB_TMP_PLAINLOAD 0
#endif

namespace pgh {

constexpr int kPbChunk = 32768;          // sources per chunk: 128 KB of LDS in phase A
constexpr int kPbThreads = 1024;

// s_x: kPbChunk floats of LDS (the chunk's slice of the gather vector), s_amax: one more LDS word.  vblock: which share of
// the A-order stream this workgroup takes (PbFormat::task_range).
// One piece of the A-order stream (whole groups of 8 entries, all inside the chunk whose slice of the gather vector sits in
// s_x): every lane takes one group per round slot -- one 16-byte load of source indices, one 4-byte load of the group's place
// in B order, 8 LDS gathers, two 16-byte stores of values.  P = round slots per lane (a round = 1024 x 8 x P entries).
template <bool HAS_VAL, int P, bool DROP = false>
__device__ __forceinline__ void pb_stream_piece(const float* __restrict__ s_x, const PbView& f, const int64_t body_begin, const int64_t body_end,
                                                uint32_t& amax, const DropView& dv = DropView{}) {
        // Software pipeline over rounds of P groups per lane: the loads of round i + 1 are issued BEFORE the gathers and
        // stores of round i.  vmcnt counts loads and stores in one in-order queue, so a loop that loads, gathers, stores
        // and only then loads again makes every round wait for the previous round's stores to complete (measured:
        // reads alone 40 us, with the stores 80 us -- no overlap at all).
        typedef int i32x4 __attribute__((ext_vector_type(4)));
        struct Round {
            u16x8    s8[P];
            uint32_t to[P];
            f32x4    w0[HAS_VAL ? P : 1], w1[HAS_VAL ? P : 1];
            i32x4    e0[DROP ? P : 1], e1[DROP ? P : 1];        // graph_dropout: the entries' indices in CSR(M^T) order
        };
        // Branch-free loads: the round base is uniform, a lane past the end of the piece repeats the piece's LAST group, so
        // every load is issued unconditionally and the compiler emits counted waits.  (Loads under `ok ? load : 0` had become
        // divergent branches with an s_waitcnt inside each -- nothing stayed in flight across the stages: a workgroup alone on
        // the chip took 63 us for its share, 31 of them waiting for its own stores.)
        constexpr int kRound = kPbThreads * 8 * P;           // entries per round
        const int span = (int)(body_end - body_begin);       // pieces are far below 2^31 entries
        const int last = span - 8;                            // first entry of the piece's last group
        const uint16_t* __restrict__ sl = f.sloc + body_begin;
        const uint32_t* __restrict__ dg = f.dstg + (body_begin >> 3);
        const float* __restrict__ vl = HAS_VAL ? f.val + body_begin : nullptr;
        const int32_t* __restrict__ el = DROP ? dv.edge + body_begin : nullptr;
        auto fetch = [&](Round& r, int rb) __attribute__((always_inline)) {
#pragma unroll
            for (int q = 0; q < P; ++q) {
                const int e = min(rb + (int)threadIdx.x * 8 + q * (kPbThreads * 8), last);
                r.s8[q] = __builtin_nontemporal_load(reinterpret_cast<const u16x8*>(sl + e));
                r.to[q] = __builtin_nontemporal_load(dg + (e >> 3));
                if (HAS_VAL) {
                    r.w0[q] = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(vl + e));
                    r.w1[q] = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(vl + e + 4));
                }
                if (DROP) {
                    r.e0[q] = __builtin_nontemporal_load(reinterpret_cast<const i32x4*>(el + e));
                    r.e1[q] = __builtin_nontemporal_load(reinterpret_cast<const i32x4*>(el + e + 4));
                }
            }
        };
        auto emit = [&](const Round& r, int rb) __attribute__((always_inline)) {
#pragma unroll
            for (int q = 0; q < P; ++q) {
                // (the LOADS of a lane past the end of the piece are clamped and unconditional; its gathers and its stores are
                // skipped: on the short pieces of a partitioned slice most lanes of a round are past the end, and thousands of
                // duplicate stores to the piece's last group queue up on one memory channel)
                if (rb + (int)threadIdx.x * 8 + q * (kPbThreads * 8) > last) continue;
                f32x4 lo, hi;
                if (PGH_PROBE_PB & 128) {                    // diagnostic: the LDS gathers without bank conflicts (wrong values)
                    const int at = (int)(threadIdx.x & 63u) + 64 * q;
                    lo.x = s_x[(r.s8[q][0] >> 15) + at];
                    lo.y = s_x[(r.s8[q][1] >> 15) + at + 512];
                    lo.z = s_x[(r.s8[q][2] >> 15) + at + 1024];
                    lo.w = s_x[(r.s8[q][3] >> 15) + at + 1536];
                    hi.x = s_x[(r.s8[q][4] >> 15) + at + 2048];
                    hi.y = s_x[(r.s8[q][5] >> 15) + at + 2560];
                    hi.z = s_x[(r.s8[q][6] >> 15) + at + 3072];
                    hi.w = s_x[(r.s8[q][7] >> 15) + at + 3584];
                } else {
                lo.x = s_x[r.s8[q][0]];
                lo.y = s_x[r.s8[q][1]];
                lo.z = s_x[r.s8[q][2]];
                lo.w = s_x[r.s8[q][3]];
                hi.x = s_x[r.s8[q][4]];
                hi.y = s_x[r.s8[q][5]];
                hi.z = s_x[r.s8[q][6]];
                hi.w = s_x[r.s8[q][7]];
                }
                if (HAS_VAL) {
                    lo *= r.w0[q];
                    hi *= r.w1[q];
                }
                if (DROP) {
#pragma unroll
                    for (int k = 0; k < 4; ++k) {
                        lo[k] *= dropout_factor(dv.seed, (uint64_t)(uint32_t)r.e0[q][k], dv.threshold, dv.keep_scale);
                        hi[k] *= dropout_factor(dv.seed, (uint64_t)(uint32_t)r.e1[q][k], dv.threshold, dv.keep_scale);
                    }
                }
                if (PGH_PROBE_PB & 4) {
                    if (lo.x + hi.w == 123.456f) f.tmp[body_begin + rb] = lo.y;
                    continue;
                }
#pragma unroll
                for (int k = 0; k < 4; ++k)
                    amax = max(amax, max(__float_as_uint(lo[k]) & 0x7fffffffu, __float_as_uint(hi[k]) & 0x7fffffffu));
                uint32_t group = r.to[q];
                if (PGH_PROBE_PB & 64)                       // diagnostic: sequential stores
                    group = (uint32_t)((body_begin + min(rb + (int)threadIdx.x * 8 + q * (kPbThreads * 8), last)) >> 3);
#if PGH_PB_TMP_NTSTORE
                __builtin_nontemporal_store(lo, reinterpret_cast<f32x4*>(f.tmp + pb_tmp_quad(group, 0, f.tmp_planes)));
                __builtin_nontemporal_store(hi, reinterpret_cast<f32x4*>(f.tmp + pb_tmp_quad(group, 1, f.tmp_planes)));
#else
                *reinterpret_cast<f32x4*>(f.tmp + pb_tmp_quad(group, 0, f.tmp_planes)) = lo;
                *reinterpret_cast<f32x4*>(f.tmp + pb_tmp_quad(group, 1, f.tmp_planes)) = hi;
#endif
            }
        };
        // (Round 5: the compiler joins the two exits of this loop with a full wait at its head -- every second round waits for the previous
        // round's stores.  Measured without it -- uniform piece descriptors, the loaded words consumed on the straight path, one exit at
        // the bottom and a dead round more per piece: phase A 58.5-58.9 against 58.0-59.0 us at scale 23, 393-436 against 388-447 at scale
        // 25, 1590-1604 against 1520-1562 at scale 27 / ef 8; with the two exits kept: 60.1 / 440-459 / 1632-1686.  The kernel is bound
        // by what a CU's memory path moves, not by this wait: left as it was.  profiles/r05/gather_shares_calibration_rejected.log)
        Round r0, r1;
        int rb = 0;
        fetch(r0, rb);
        for (;;) {
            fetch(r1, rb + kRound);                           // (clamped: the last round re-reads the last group)
            emit(r0, rb);
            rb += kRound;
            if (rb >= span) break;
            fetch(r0, rb + kRound);
            emit(r1, rb);
            rb += kRound;
            if (rb >= span) break;
        }
}

// PG: round slots per lane on long pieces.  One GPU's graph has ~190 K entries per piece at scale 23: rounds of 4.  The slices of
// a partitioned graph gather from a source space 2-16x larger -- 2-16x more chunks, and a long tail of pieces of a few
// thousand entries (88 .. 437 K on the 8-way slice of configs[4]): pieces below PbView::short_piece entries (16 K) run rounds of
// 1.  Same-box sweep of that line on the slices of the N = 2 / 4 / 8 bench, phase A us: none 113 / 153 / 194-204, 8 K 115 /
// 153 / 188, 16 K 114 / 154 / 189, 32 K 132 / 175 / 189 (profiles/r03/partition_slices.log).
template <bool HAS_VAL, int PG, bool DROP = false>
__device__ __forceinline__ void pb_gather_body(float* __restrict__ s_x, uint32_t* __restrict__ s_amax, const PbView& f,
                                               const float* __restrict__ xg, const int vblock, const DropView& dv = DropView{}) {
    if (threadIdx.x == 0) *s_amax = 0u;
    uint32_t amax = 0u;                 // bit pattern of max |value| this thread wrote (NaN > inf > finite as integers)
    // this workgroup's share of the entry stream: consecutive pieces, each inside one chunk; the LDS image of the chunk
    // is refilled only when the chunk changes
    const int piece_begin = f.task_range[vblock], piece_end = f.task_range[vblock + 1];
    const int4 first_task = f.first_task[vblock];          // (= task[piece_begin]: asked for WITH the range, not after it)
    int loaded = -1;
    for (int piece = piece_begin; piece < piece_end; ++piece) {
        const int4 task = piece == piece_begin ? first_task : f.task[piece];
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
#if PGH_FILL_GLDS
                // LDS-direct loads (global_load_lds_dword: no destination registers, so the whole slice is in flight at once -- one
                // memory round trip per fill instead of four rounds of 8 loads per thread; the destination of a wavefront's
                // instruction is its uniform base + 4 bytes per lane, which is exactly a contiguous copy)
                for (int64_t w0 = lo + (threadIdx.x & ~63); w0 < hi; w0 += kPbThreads) {
                    const int64_t id = w0 + (threadIdx.x & 63);
                    if (id < hi)
                        __builtin_amdgcn_global_load_lds(src + id, (__attribute__((address_space(3))) void*)(s_x + (w0 - first_id)), 4, 0, 0);
                }
#else
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
#endif
            }
            __syncthreads();
            loaded = task.x;
        }
        const int64_t body_begin = task.y, body_end = task.z;
#ifndef PGH_GATHER_P
#define PGH_GATHER_P 4
#endif
        if (PGH_PROBE_PB & 2) continue;
        if (body_end <= body_begin) continue;
        constexpr int PL = HAS_VAL ? (PG > 1 ? PG / 2 : 1) : PG;
        constexpr int PD = DROP ? 1 : PL;                     // the mask's hashes take the registers of the deeper rounds
        if (body_end - body_begin >= (int64_t)f.short_piece) pb_stream_piece<HAS_VAL, PD, DROP>(s_x, f, body_begin, body_end, amax, dv);
        else pb_stream_piece<HAS_VAL, 1, DROP>(s_x, f, body_begin, body_end, amax, dv);
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
