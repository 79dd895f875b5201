// q3_attn.hip -- the attention stage of one decode step (reference src/forward.c:267-291):
// per-head RMSNorm + RoPE of q and k, KV-cache append, grouped-query attention over the
// cached positions, and q8_quantize of the head outputs for the Wo GEMV.
//
// One workgroup = one kv head x one 64-position chunk slot, 4 waves.  The K and V tiles
// of a chunk are staged ONCE in LDS (64 KB at head_dim 128) and shared by the query heads
// of the group (GQA: 4 query heads per kv head on Qwen3-4B/8B read each cached byte once).
// Each wave then owns one query head end to end, so no cross-wave reduction is needed:
//   scores   lanes = 2 positions x 32 float4 slices; per step a 4-term chain + 5-level
//            butterfly inside each 32-lane half (q3_numerics.h DOT); the score of
//            position t = 2*step + half is parked in lane 32*half + step
//   softmax  wave max, q3_expf, 64-lane butterfly of the parked e_t
//   PV       two streams (even / odd positions), each lane a float4 slice of the head
// Every load that does not depend on data computed here is issued at kernel entry --
// position, raw q/k/v, norm weights, the (cos,sin) row of this position and,
// speculatively, the whole first K/V tile -- so the step costs one memory round trip.
//
// When pos >= 64 every workgroup publishes its chunk partials (m_c, l_c, O_c) and the one that
// draws the last ticket of its kv head merges them inside the same launch; below
// that the kernel normalises and quantises directly.
//
// Kernels of a decode step (one per launch shape, q3_kernels.hpp step_shape):
//   k_attn_wo     < 1024 cached positions: the attention workgroups above AND, on the CUs they leave idle, the Wo
//                 GEMV + residual (forward.c:292-298) -- its rows wait in registers and take the attention output
//                 as tagged 8-byte granules inside the launch
//   k_attn_long   >= 1024: the same attention arithmetic with the K/V tiles fetched by four loader waves per
//                 workgroup as LDS-DMA; partials to HBM
//   k_merge_wo    >= 1024: merge of the chunk partials (one wave per 64 outputs) + the Wo consumers in one launch
//   k_attn, k_attn_merge   the separate forms (Q3_FUSE=0, head_dim 64, op-level hooks, the fp16 path)
//   k_attn_block, k_kv_append   the batched prompt pass (q3_prefill)
#include <cstdio>
#include <cstdlib>
#include <unordered_map>

#include "q3_device.hpp"
#include "q3_kernels.hpp"
#include "q3_tile.hpp"

namespace q3k {

#define Q3_MAXG 8   // max query heads per kv head

#ifdef Q3_ATTN_STAMPS
// every workgroup of grid layer 0 leaves eight device-clock marks: stamps[8 * workgroup + i]
#define STAMP(i) do { if (a.stamps && blockIdx.z == 0 && threadIdx.x == 0) a.stamps[8 * (slot * a.n_kv + g) + (i)] = __builtin_amdgcn_s_memrealtime(); } while (0)
#else
#define STAMP(i) do {} while (0)
#endif

typedef __attribute__((address_space(1))) unsigned g_u32;

// A wait inside a launch gives up after this much DEVICE time (s_memrealtime, 100 MHz) unless the launch carries its own
// bound (WoView::wait_ticks): far beyond anything a healthy launch takes, short of what a watchdog calls a hung GPU.
#ifndef Q3_WAIT_TICKS
#define Q3_WAIT_TICKS 500000000ull   // 5 s
#endif

// a word that an EARLIER launch wrote (position, step counter), fetched by s_load through the scalar cache
// (volatile: the request stays where it is written, ahead of the vector loads; only the wait moves to the use)
__device__ __forceinline__ int ld_scalar(const int* p) {
    return *(const volatile __attribute__((address_space(4))) int*)(p);
}


// write-through 8-byte store / cache-bypassing loads for data handed to another workgroup
// inside the launch
__device__ __forceinline__ void st_sc1_f2(float* p, float x, float y) {
    const unsigned long long v = ((unsigned long long)__float_as_uint(y) << 32) | __float_as_uint(x);
    __hip_atomic_store((__attribute__((address_space(1))) unsigned long long*)p, v, __ATOMIC_RELAXED,
                       __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ float2 ld_sc1_f2(const float* p) {
    const unsigned long long v = __hip_atomic_load((__attribute__((address_space(1))) unsigned long long*)p,
                                                   __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    return make_float2(__uint_as_float((unsigned)v), __uint_as_float((unsigned)(v >> 32)));
}

// Final outputs of the stage (four codes per lane, one scale per 16 lanes): plain stores, or -- when workgroups of
// the SAME launch consume them (k_attn_wo) -- ONE naturally aligned 8-byte {tag, value} store each (Guideline 16, R2:
// the data is the flag; written through, sc1), which the consumers poll directly.
template <bool PUB>
__device__ __forceinline__ void out_codes(const Attn& a, unsigned tag, size_t elem, int packed) {
    if (PUB) {
        __hip_atomic_store((__attribute__((address_space(1))) unsigned long long*)(a.og + (elem >> 2)),
                           ((unsigned long long)tag << 32) | (unsigned)packed, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    } else {
        reinterpret_cast<int*>(a.oq)[elem >> 2] = packed;
    }
}
template <bool PUB>
__device__ __forceinline__ void out_scale(const Attn& a, unsigned tag, size_t elem, float scale) {
    if (PUB) {
        const size_t P4 = (size_t)a.n_heads * a.hd / 4;
        __hip_atomic_store((__attribute__((address_space(1))) unsigned long long*)(a.og + P4 + (elem >> 6)),
                           ((unsigned long long)tag << 32) | __float_as_uint(scale), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    } else {
        a.os[elem >> 6] = scale;
    }
}

// Merge of the chunk partials of one head (q3_numerics.h "attention", last three lines) +
// q8_quantize, by ONE wave.  The sums over chunks are sequential by contract, but nothing
// forces the LOADS to be: the (m_c, l_c) pairs are fetched 64 chunks at a time (one per lane)
// and the O_c rows eight chunks ahead of the accumulation.
template <int HD>
__device__ __forceinline__ float4 ld_partial_row(const float* base, int c, int lane) {
    constexpr int ST = HD + 2;
    float4 o = make_float4(0.f, 0.f, 0.f, 0.f);
    if (lane < HD / 4) {
        const float* pp = base + (size_t)c * ST + 4 * lane;
        const float2 lo = ld_sc1_f2(pp);
        const float2 hi = ld_sc1_f2(pp + 2);
        o = make_float4(lo.x, lo.y, hi.x, hi.y);
    }
    return o;
}

template <int HD, bool PUB = false>
__device__ __forceinline__ void merge_partials(const Attn& a, int h, int nchunks, int lane, unsigned tag = 0) {
    constexpr int L4 = HD / 4;
    constexpr int ST = HD + 2;
    constexpr int AH = 16;                // O_c rows per burst = every chunk the in-launch merge ever sees (Q3_ATT_LONG / 64)
    const float* base = a.part + (size_t)h * a.max_chunks * ST;
    // Everything is requested in ONE burst -- the (m_c, l_c) pairs, one per lane, and the rows O_c --
    // so the merge costs a single memory round trip before the arithmetic starts.  (Contexts beyond
    // 16 chunks take further bursts; the launch shapes in use never get there: ATT_LONG has its own kernel.)
    float2 ml0 = make_float2(-3.0e38f, 0.0f);
    if (lane < nchunks) ml0 = ld_sc1_f2(base + (size_t)lane * ST + HD);
    float4 o[AH];
#pragma unroll
    for (int k = 0; k < AH; k++) o[k] = ld_partial_row<HD>(base, k < nchunks ? k : nchunks - 1, lane);
    float M = wave_max(ml0.x);
    for (int c0 = 64; c0 < nchunks; c0 += 64) {
        const int c = c0 + lane;
        const float mc = c < nchunks ? ld_sc1_f2(base + (size_t)c * ST + HD).x : -3.0e38f;
        M = fmaxf(M, wave_max(mc));
    }
    float L = 0.0f;
    float4 A = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int c0 = 0; c0 < nchunks; c0 += 64) {
        const int c = c0 + lane;
        float2 ml = ml0;
        if (c0 > 0) ml = c < nchunks ? ld_sc1_f2(base + (size_t)c * ST + HD) : make_float2(-3.0e38f, 0.0f);
        float wc = 0.0f, wl = 0.0f;
        if (c < nchunks) {
            wc = q3_expf(ml.x - M);
            wl = wc * ml.y;
        }
        const int cnt = (nchunks - c0 < 64) ? nchunks - c0 : 64;
        for (int k0 = 0; k0 < cnt; k0 += AH) {
            if (c0 + k0 > 0) {             // a further burst (long contexts only)
#pragma unroll
                for (int k = 0; k < AH; k++) o[k] = ld_partial_row<HD>(base, c0 + k0 + k < nchunks ? c0 + k0 + k : nchunks - 1, lane);
            }
#pragma unroll
            for (int k = 0; k < AH; k++) {
                if (k0 + k < cnt) {
                    const float w = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, wc), (k0 + k) & 63));
                    const float t = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, wl), (k0 + k) & 63));
                    L = L + t;
                    A.x = A.x + w * o[k].x;
                    A.y = A.y + w * o[k].y;
                    A.z = A.z + w * o[k].z;
                    A.w = A.w + w * o[k].w;
                }
            }
        }
    }
    float4 y = make_float4(0.f, 0.f, 0.f, 0.f);
    if (lane < L4) {
        y.x = A.x / L;
        y.y = A.y / L;
        y.z = A.z / L;
        y.w = A.w / L;
    }
    float scale;
    const int packed = quantize_group16(y, scale);
    if (lane < L4) {
        out_codes<PUB>(a, tag, (size_t)h * HD + 4 * lane, packed);
        if ((lane & 15) == 0) out_scale<PUB>(a, tag, (size_t)h * HD + 4 * lane, scale);
        if (a.of) *reinterpret_cast<float4*>(a.of + (size_t)h * HD + 4 * lane) = y;
    }
}

// The same merge from partials published as {tag, value} granules (k_attn_wo, ATT_MERGE): the workgroup of chunk slot 0
// re-reads the granules of every chunk of the head until all carry this step's tag -- ONE round trip when the other
// chunks are through (they run the same program on the same amount of data), against a drain of the stores, a ticket
// round trip and a fetch round trip for the counter form.  Arithmetic: merge_partials' (q3_numerics.h "attention").
template <int HD>
__device__ __forceinline__ void merge_partials_granules(const Attn& a, int h, int nchunks, int lane, unsigned tag) {
    typedef __attribute__((address_space(1))) unsigned long long g_u64;
    constexpr int L4 = HD / 4;
    constexpr int ST = HD + 2;
    constexpr int AH = Q3_ATT_LONG / Q3_ATT_CHUNK;      // every chunk the in-launch merge ever sees
    constexpr int BR = 8;                                // rows per batch of the sweep (64 registers of granules)
    const g_u64* base = (const g_u64*)(a.pg + (size_t)h * a.max_chunks * ST);
    const int l4 = lane < L4 ? lane : L4 - 1;            // (lanes past the head re-read its last slice: unconditional loads)
    const int cm = lane < nchunks ? lane : nchunks - 1;
    const unsigned long long t_wait = __builtin_amdgcn_s_memrealtime();
    // One sweep brings the (m, l) pairs (one chunk per lane) and the first eight O rows; its loads are all in flight at
    // once, so when the other chunks are through (they run the same program on as much data) the merge costs ONE round trip.
    float2 ml0 = make_float2(-3.0e38f, 0.0f);
    float M = 0.0f, wc = 0.0f, wl = 0.0f;
    float L = 0.0f;
    float4 A = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int k0 = 0; k0 < AH && k0 < nchunks; k0 += BR) {       // (wave-uniform: the second batch only beyond eight chunks)
        unsigned long long gx[BR][4];
        for (;;) {
            unsigned long long gm = 0, gl = 0;
            if (k0 == 0) {
                gm = __hip_atomic_load(base + (size_t)cm * ST + HD, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                gl = __hip_atomic_load(base + (size_t)cm * ST + HD + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
#pragma unroll
            for (int k = 0; k < BR; k++) {
                const g_u64* row = base + (size_t)(k0 + k < nchunks ? k0 + k : nchunks - 1) * ST + 4 * l4;
#pragma unroll
                for (int e = 0; e < 4; e++) gx[k][e] = __hip_atomic_load(row + e, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
            bool hit = k0 != 0 || ((unsigned)(gm >> 32) == tag && (unsigned)(gl >> 32) == tag);
#pragma unroll
            for (int k = 0; k < BR; k++) {
#pragma unroll
                for (int e = 0; e < 4; e++) hit = hit && (unsigned)(gx[k][e] >> 32) == tag;
            }
            if (__all(hit)) {
                if (k0 == 0) {
                    if (lane < nchunks) ml0 = make_float2(__uint_as_float((unsigned)gm), __uint_as_float((unsigned)gl));
                    M = wave_max(ml0.x);
                    if (lane < nchunks) {
                        wc = q3_expf(ml0.x - M);
                        wl = wc * ml0.y;
                    }
                }
                break;
            }
            __builtin_amdgcn_s_sleep(2);
            if (__builtin_amdgcn_s_memrealtime() - t_wait > Q3_WAIT_TICKS) return;      // (the consumers give up as well and raise the flag)
        }
#pragma unroll
        for (int k = 0; k < BR; k++) {
            if (k0 + k < nchunks) {
                const float wk = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, wc), (k0 + k) & 63));
                const float t = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, wl), (k0 + k) & 63));
                L = L + t;
                A.x = A.x + wk * __uint_as_float((unsigned)gx[k][0]);
                A.y = A.y + wk * __uint_as_float((unsigned)gx[k][1]);
                A.z = A.z + wk * __uint_as_float((unsigned)gx[k][2]);
                A.w = A.w + wk * __uint_as_float((unsigned)gx[k][3]);
            }
        }
    }
    float4 y = make_float4(0.f, 0.f, 0.f, 0.f);
    if (lane < L4) {
        y.x = A.x / L;
        y.y = A.y / L;
        y.z = A.z / L;
        y.w = A.w / L;
    }
    float scale;
    const int packed = quantize_group16(y, scale);
    if (lane < L4) {
        out_codes<true>(a, tag, (size_t)h * HD + 4 * lane, packed);
        if ((lane & 15) == 0) out_scale<true>(a, tag, (size_t)h * HD + 4 * lane, scale);
    }
}

// ---- the K/V stream of the long-context launch is fetched by LOADER waves (waves 4-7 of a 512-thread workgroup).
// A wave issues in order, so a compute wave that requests a 32-KB tile sits in the issue stage until the CU's
// queue has taken it all in -- which is why k_attn holds the V requests back until the head norms are done, and the
// V tile then lands behind the scores.  A loader wave has nothing else to do: it requests K AND V of its chunk back
// to back (from the first instruction when the host vouches for the chunk: slot < sure_slots) as LDS-DMA
// (global_load_lds_dwordx4: 1 KiB = two cache rows per wave-instruction, straight into the tile, no registers), and
// meets the compute waves at the barriers the one-role kernel has (K tile ready, V tile ready) once its share of
// the tile has landed (counted vmcnt).  Row `pos` of the owning chunk is not in the cache yet: the DMA brings a stale
// row, and the compute side overwrites it behind one extra barrier that only the owning workgroup takes.
__device__ __forceinline__ void hold_back(int ticks) {
    if (ticks <= 0) return;
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    while ((long long)(__builtin_amdgcn_s_memrealtime() - t0) < (long long)ticks) __builtin_amdgcn_s_sleep(8);
}

typedef __attribute__((address_space(1))) const void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;
template <int HD>
__device__ __forceinline__ void tile_dma(const float* src, float* tile, int uwave, int lane) {
    constexpr int PIECES = Q3_ATT_CHUNK * HD * 4 / 1024;       // 1-KiB pieces of a tile: 32 at head_dim 128
#pragma unroll
    for (int j = 0; j < PIECES / 4; j++) {
        const int p = uwave + 4 * j;                           // wave-uniform: the LDS address travels in M0
        __builtin_amdgcn_global_load_lds((gptr_t)(src + p * 256 + lane * 4), (lptr_t)(tile + p * 256), 16, 0, 0);
    }
}
template <int HD>
__device__ __forceinline__ void tile_loader(const Attn& a, int g, int slot, int nslots, int sure_slots, float* Ks, float* Vs) {
    constexpr int CH = Q3_ATT_CHUNK;
    constexpr int PW = CH * HD * 4 / 1024 / 4;                 // pieces per loader wave and tile
    static_assert(PW == 8 || PW == 4, "counted waits below assume 4 or 8 pieces per wave");
    const int lane = threadIdx.x & 63;
    const int uwave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)) - 4;
    const size_t cbase = (size_t)g * a.seq_len * HD;
    const int pos = ld_scalar(&a.ctl->pos);
    __builtin_amdgcn_s_barrier();             // E: the compute waves have their few requests in the queue
    if (slot >= sure_slots && slot * CH > pos) return;
    tile_dma<HD>(a.kc + cbase + (size_t)slot * CH * HD, Ks, uwave, lane);
    // Two workgroups share a CU, and a CU returns loads in issue order: with V requested right behind K, the second
    // workgroup's K tile queues behind the first one's V tile, which nobody needs before its scores are done.  Holding the
    // V requests back (1.5 us: 548 -> 554 tok/s at 4096 positions; holding the K requests back as well, so that both
    // workgroups' small requests go first, loses 1-2 %) puts both K tiles first.
    hold_back(a.v_hold);
    tile_dma<HD>(a.vc + cbase + (size_t)slot * CH * HD, Vs, uwave, lane);
    const int T = pos + 1;
    const int nchunks = (T + CH - 1) / CH;
    if (slot >= nchunks) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");       // nothing may land in LDS after this workgroup is gone
        return;
    }
    for (int c = slot; c < nchunks; c += nslots) {
        const bool mine = pos >= c * CH && pos < (c + 1) * CH; // workgroup-uniform: this chunk holds the row of `pos`
        if (c != slot) tile_dma<HD>(a.kc + cbase + (size_t)c * CH * HD, Ks, uwave, lane);    // every wave is past the scores of the previous chunk (barrier V)
        if (c == slot) {
            if (PW == 8) asm volatile("s_waitcnt vmcnt(8)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(4)" ::: "memory");     // the K pieces are in; V still flies
        } else {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        if (mine) __builtin_amdgcn_s_barrier();                // the compute side now writes row `pos`
        __builtin_amdgcn_s_barrier();                          // barrier K
        if (c != slot) tile_dma<HD>(a.vc + cbase + (size_t)c * CH * HD, Vs, uwave, lane);    // the compute waves are past the PV of the previous chunk
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (mine) __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_s_barrier();                          // barrier V
    }
}

// The attention stage as seen by ONE workgroup: kv head g, chunk slot `slot` of `nslots`.  Returns true
// (workgroup-uniform) when THIS workgroup wrote the final outputs (codes + scales) of its kv head's query
// heads: always in ATT_SINGLE, for the drawer of the last ticket in ATT_MERGE, never in ATT_LONG.
// `rows_cap` (a multiple of 8, wave-uniform): rows of the first K/V tile that can hold cached positions --
// the host knows pos < rows_cap when it picks the launch, so the rows beyond are not even requested
// (they used to be pulled, 64 KB per workgroup whatever pos was, through one CU that takes in ~24 KB/us).
// HPW = query heads a wave may own (1 when the group has at most 4 query heads); PUB: see out_codes.
// LW = loader waves (0, or 4: the long-context launch, ATT_LONG only): the tiles are then fetched and parked by
// tile_loader() on waves 4-7, and `rows_cap` carries sure_slots instead (chunk slots the host vouches for).
// MODE >= 0: the launch shape (AttMode) at compile time -- the one-chunk kernel then carries neither the partials nor
// the merge, and its registers and scalar spills are its own (the fused kernel slowed by 10 % in its one-chunk shape
// when code was ADDED to its merge path: one kernel for all shapes pays every shape's register pressure); < 0: `multi`.
template <int HD, int HPW, bool PUB, int LW = 0, int MODE = -1>
__device__ __forceinline__ bool attn_body(const Attn& a, int multi_rt, int g, int slot, int nslots, int rows_cap,
                                          float* Ks, float* Vs, int* last_flag_p) {
    const int multi = MODE >= 0 ? MODE : multi_rt;
    // (the op-level hook's `prepared` inputs never reach the fused kernel: attn_wo_supported)
    const bool prepared = PUB ? false : a.prepared != 0;
    if (LW && threadIdx.x >= 256) {
        tile_loader<HD>(a, g, slot, nslots, rows_cap, Ks, Vs);
        return false;
    }
    constexpr int L4 = HD / 4;               // lanes holding one head as float4
    constexpr int CH = Q3_ATT_CHUNK;
    constexpr int NLD = CH * L4 / 256;       // float4 loads per thread per tile
    int& last_flag = *last_flag_p;
    const int kv_mul = a.n_heads / a.n_kv;
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int half = lane >> 5, l = lane & 31;
    const int P = a.n_heads * HD, KVD = a.n_kv * HD;
    const bool act = l < L4;
    const size_t cbase = (size_t)g * a.seq_len * HD;

    STAMP(0);
    // ---- everything that can be requested before any arithmetic ------------------
    // Request order = the order the data is needed (a CU returns loads in issue order and takes
    // in only ~40 KB of requests at once): the step's raw q/k/v + norm weights + (cos,sin) row,
    // the K tile (32 KB at head_dim 128), which lands while the head norms run, and after the
    // norms the V tile, which streams in while the scores and the softmax are being computed.
    // The position and the step counter travel on the SCALAR path (s_load: its own counter, its own
    // cache), so waiting for them never waits for a vector load -- as a vector load in front of the
    // early exit below, `pos` cost a whole memory round trip before anything else was even requested.
    const int pos = ld_scalar(&a.ctl->pos);
    const unsigned tag = PUB ? (((unsigned)ld_scalar(reinterpret_cast<const int*>(a.epoch)) << 8) | a.layer_tag) : 0u;
    float4 kt[NLD], vt[NLD];
    const int tfirst = slot * CH;    // rows beyond pos are loaded but never used
    // At head_dim 128 a head is 32 lanes of float4, so ONE norm + rope pass serves two heads:
    // k of this step in the lower half of the wave, the wave's (first) query head in the upper
    // half (bfly32 inside a half adds exactly what bfly64 adds when the other half is zero).
    constexpr bool FUSE = (L4 == 32);
    float4 kraw = make_float4(0.f, 0.f, 0.f, 0.f), vraw = kraw, qg = kraw, kg = kraw;
    float4 qraw[HPW];
#pragma unroll
    for (int hi = 0; hi < HPW; hi++) qraw[hi] = kraw;      // zeros
    float4 ca = kraw, cb = kraw;
    if constexpr (FUSE) {
        if (!half) kraw = *reinterpret_cast<const float4*>(a.qkv + P + (size_t)g * HD + 4 * l);
        if (half && wave < kv_mul) kraw = *reinterpret_cast<const float4*>(a.qkv + (size_t)(g * kv_mul + wave) * HD + 4 * l);
        vraw = *reinterpret_cast<const float4*>(a.qkv + P + KVD + (size_t)g * HD + 4 * l);     // slice l in both halves
        if (HPW > 1 && lane < L4 && wave + 4 < kv_mul)
            qraw[HPW - 1] = *reinterpret_cast<const float4*>(a.qkv + (size_t)(g * kv_mul + wave + 4) * HD + 4 * lane);
        if (!prepared) {
            kg = *reinterpret_cast<const float4*>((half ? a.qnw : a.knw) + 4 * l);
            if (HPW > 1 && lane < L4) qg = *reinterpret_cast<const float4*>(a.qnw + 4 * lane);
        }
    } else if (lane < L4) {
        kraw = *reinterpret_cast<const float4*>(a.qkv + P + (size_t)g * HD + 4 * lane);
        vraw = *reinterpret_cast<const float4*>(a.qkv + P + KVD + (size_t)g * HD + 4 * lane);
#pragma unroll
        for (int hi = 0; hi < HPW; hi++) {
            const int i = wave + 4 * hi;
            if (i < kv_mul) qraw[hi] = *reinterpret_cast<const float4*>(a.qkv + (size_t)(g * kv_mul + i) * HD + 4 * lane);
        }
        if (!prepared) {
            qg = *reinterpret_cast<const float4*>(a.qnw + 4 * lane);
            kg = *reinterpret_cast<const float4*>(a.knw + 4 * lane);
        }
    }
    if (!prepared) rope_slices<HD>(a.cs, FUSE ? l : lane, ca, cb);
    __builtin_amdgcn_sched_barrier(0);
    // Branch-free (a branch between loads makes the compiler lose count of vmcnt, and the head norms below would
    // wait for the whole tile): the tile is read through a buffer descriptor that ends after `rows_cap` rows, so
    // the requests for rows beyond it return zeros and move no bytes.
    const int tile_rows = LW ? 0 : rows_cap;
    // Slot 0 always has work (chunk 0), so it requests its tile before `pos` has even arrived;
    // the other slots first learn whether their chunk exists -- a speculative 64 KB per idle
    // workgroup would cost tens of MB of useless HBM reads per layer at short contexts.  (Their few
    // small requests above are already out; the one-chunk shape has no other slot.)
    if (MODE != ATT_SINGLE && !LW && slot != 0 && slot * CH > pos) return false;
    const __amdgpu_buffer_rsrc_t kres = __builtin_amdgcn_make_buffer_rsrc(a.kc + cbase + (size_t)tfirst * HD, 0, tile_rows * HD * 4, 0x00020000);
    const __amdgpu_buffer_rsrc_t vres = __builtin_amdgcn_make_buffer_rsrc(a.vc + cbase + (size_t)tfirst * HD, 0, tile_rows * HD * 4, 0x00020000);
    if (LW) {
        __builtin_amdgcn_s_barrier();         // E: these waves' requests are in the queue; the loader waves may flood it
        if (slot >= rows_cap && slot * CH > pos) return false;
    } else {
#pragma unroll
        for (int k = 0; k < NLD; k++) {
            const v4i r = __builtin_amdgcn_raw_buffer_load_b128(kres, (tid + k * 256) * 16, 0, 0);
            kt[k] = make_float4(__int_as_float(r.x), __int_as_float(r.y), __int_as_float(r.z), __int_as_float(r.w));
        }
    }
    __builtin_amdgcn_sched_barrier(0);
    // The head norms below start HERE, behind the tile requests: the instruction selector orders pure arithmetic
    // freely inside a basic block (sched_barrier binds only the machine scheduler), and with no branch in between
    // it had put the wait for k/q and the whole norm chain in front of the K-tile requests.
    asm volatile("" : "+v"(kraw.x), "+v"(kraw.y), "+v"(kraw.z), "+v"(kraw.w) :: "memory");

    const int T = pos + 1;
    const int nchunks = MODE == ATT_SINGLE ? 1 : (T + CH - 1) / CH;     // (the one-chunk shape: pos < 64, slot 0)
    if (slot >= nchunks) return false;
    STAMP(1);
    const bool owner = ((nchunks - 1) % nslots) == slot;

    // k of this step (every wave redundantly, no barrier) and q of this wave's head(s):
    // norm + rope; kcur_m / vraw_m / q4 end up as slice (tid % L4) in every lane
    float4 kcur_m, vraw_m = vraw, q4[HPW];
    if constexpr (FUSE) {
        float4 r = kraw;
        if (!prepared) r = headnorm_rope_halves<HD>(kraw, kg, ca, cb, l);
        const float ox = lane_xor_f<32>(r.x), oy = lane_xor_f<32>(r.y), oz = lane_xor_f<32>(r.z), ow = lane_xor_f<32>(r.w);
        const float4 other = make_float4(ox, oy, oz, ow);
        kcur_m = half ? other : r;
        q4[0] = half ? r : other;
        if (HPW > 1) {
            q4[HPW - 1] = qraw[HPW - 1];
            if (wave + 4 < kv_mul) {
                if (!prepared) q4[HPW - 1] = headnorm_rope_vals<HD>(q4[HPW - 1], qg, ca, cb, lane);
                const float ax = lane_xor_f<32>(q4[HPW - 1].x), ay = lane_xor_f<32>(q4[HPW - 1].y);
                const float az = lane_xor_f<32>(q4[HPW - 1].z), aw = lane_xor_f<32>(q4[HPW - 1].w);
                if (half) q4[HPW - 1] = make_float4(ax, ay, az, aw);
            }
        }
    } else {
        float4 kcur = kraw;
        if (!prepared) kcur = headnorm_rope_vals<HD>(kraw, kg, ca, cb, lane);
        // L4 == 16: lane's slice is lane % 16; lanes 16..63 fetch it from lane % 16
        const int src = (lane & (L4 - 1)) << 2;
        kcur_m.x = __int_as_float(__builtin_amdgcn_ds_bpermute(src, __float_as_int(kcur.x)));
        kcur_m.y = __int_as_float(__builtin_amdgcn_ds_bpermute(src, __float_as_int(kcur.y)));
        kcur_m.z = __int_as_float(__builtin_amdgcn_ds_bpermute(src, __float_as_int(kcur.z)));
        kcur_m.w = __int_as_float(__builtin_amdgcn_ds_bpermute(src, __float_as_int(kcur.w)));
        vraw_m.x = __int_as_float(__builtin_amdgcn_ds_bpermute(src, __float_as_int(vraw.x)));
        vraw_m.y = __int_as_float(__builtin_amdgcn_ds_bpermute(src, __float_as_int(vraw.y)));
        vraw_m.z = __int_as_float(__builtin_amdgcn_ds_bpermute(src, __float_as_int(vraw.z)));
        vraw_m.w = __int_as_float(__builtin_amdgcn_ds_bpermute(src, __float_as_int(vraw.w)));
#pragma unroll
        for (int hi = 0; hi < HPW; hi++) {
            q4[hi] = qraw[hi];
            if (wave + 4 * hi < kv_mul) {
                if (!prepared) q4[hi] = headnorm_rope_vals<HD>(q4[hi], qg, ca, cb, lane);
                const float ox = lane_xor_f<32>(q4[hi].x), oy = lane_xor_f<32>(q4[hi].y);
                const float oz = lane_xor_f<32>(q4[hi].z), ow = lane_xor_f<32>(q4[hi].w);
                if (half) q4[hi] = make_float4(ox, oy, oz, ow);
            }
        }
    }
    // now the V tile: the K tile has landed (loads return in order), so these requests find
    // room in the CU's queue instead of parking the wave in the issue stage
    __builtin_amdgcn_sched_barrier(0);
    if (!LW) {
#pragma unroll
        for (int k = 0; k < NLD; k++) {
            const v4i r = __builtin_amdgcn_raw_buffer_load_b128(vres, (tid + k * 256) * 16, 0, 0);
            vt[k] = make_float4(__int_as_float(r.x), __int_as_float(r.y), __int_as_float(r.z), __int_as_float(r.w));
        }
    }
    __builtin_amdgcn_sched_barrier(0);
    // wave 0 of the owning workgroup appends k and v of this step to the cache
    if (owner && wave == 0 && lane < L4) {
        *reinterpret_cast<float4*>(a.kc + cbase + (size_t)pos * HD + 4 * lane) = kcur_m;
        *reinterpret_cast<float4*>(a.vc + cbase + (size_t)pos * HD + 4 * lane) = vraw_m;
    }
    if (a.qdbg && slot == 0 && lane < L4) {
#pragma unroll
        for (int hi = 0; hi < HPW; hi++)
            if (wave + 4 * hi < kv_mul) *reinterpret_cast<float4*>(a.qdbg + (size_t)(g * kv_mul + wave + 4 * hi) * HD + 4 * lane) = q4[hi];
    }

    STAMP(2);
    const float root = sqrtf((float)HD);
    bool first = true;
    for (int c = slot; c < nchunks; c += nslots) {
        const int t0 = c * CH;
        const int Tc = (T - t0 < CH) ? T - t0 : CH;      // valid positions in this chunk
        if (!LW && !first) {      // contexts beyond nslots chunks only: the previous chunk's PV is over for this wave
            // (unconditional loads -- predicated accesses would push kt / vt out of registers -- with the
            // row clamped to the last valid one, so a short last chunk re-reads one row instead of
            // pulling 64 KB of unused cache)
#pragma unroll
            for (int k = 0; k < NLD; k++) {
                const int idx = tid + k * 256;
                const int t = idx / L4, l4 = idx - t * L4;
                const int tc = t < Tc ? t : Tc - 1;
                kt[k] = *reinterpret_cast<const float4*>(a.kc + cbase + (size_t)(t0 + tc) * HD + 4 * l4);
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int k = 0; k < NLD; k++) {
                const int idx = tid + k * 256;
                const int t = idx / L4, l4 = idx - t * L4;
                const int tc = t < Tc ? t : Tc - 1;
                vt[k] = *reinterpret_cast<const float4*>(a.vc + cbase + (size_t)(t0 + tc) * HD + 4 * l4);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        first = false;
        // ---- K tile -> LDS (every wave is past the scores of the previous chunk: barrier V below).
        // The row of this very step comes from registers (select, no branch: the thread that
        // owns slice l4 of that row holds the same slice of kcur).
        if (LW) {          // the loader waves' DMA fills the tile; row `pos` (not in the cache yet) comes from this wave's registers,
            if (pos >= t0 && pos < t0 + CH) {      // once the DMA has landed (workgroup-uniform: the loaders take the same extra barrier)
                __builtin_amdgcn_s_barrier();
                if (wave == 0 && lane < L4) *reinterpret_cast<float4*>(Ks + (pos - t0) * HD + 4 * lane) = kcur_m;
            }
        } else {
#pragma unroll
            for (int k = 0; k < NLD; k++) {
                const int idx = tid + k * 256;
                const int t = idx / L4, l4 = idx - t * L4;
                const bool cur = (t0 + t == pos);
                float4 w = kt[k];
                w.x = cur ? kcur_m.x : w.x;
                w.y = cur ? kcur_m.y : w.y;
                w.z = cur ? kcur_m.z : w.z;
                w.w = cur ? kcur_m.w : w.w;
                *reinterpret_cast<float4*>(Ks + t * HD + 4 * l4) = w;
            }
        }
        __syncthreads();                                  // barrier K
        STAMP(3);

        const int nsteps = (Tc + 1) >> 1;
        float e[HPW], m[HPW], lsum[HPW];
#pragma unroll
        for (int hi = 0; hi < HPW; hi++) {
            e[hi] = m[hi] = lsum[hi] = 0.0f;
            if (wave + 4 * hi < kv_mul) {
                // scores: step s handles position t = 2*s + half; the 4-term chains of all 32
                // steps are formed first, then ONE transposing butterfly leaves the finished
                // dot of step l in lane l of each half (= lane 32*half + l of the wave)
                float cpart[32];
#pragma unroll
                for (int blk = 0; blk < 4; blk++) {
                    if (8 * blk < nsteps) {
#pragma unroll
                        for (int k = 0; k < 8; k++) {
                            const int step = 8 * blk + k;
                            const int t = 2 * step + half;
                            float cdot = 0.0f;
                            if (act) {      // rows >= Tc hold stale bytes; their sums are masked below
                                const float4 k4 = *reinterpret_cast<const float4*>(Ks + t * HD + 4 * l);
                                cdot = q4[hi].x * k4.x;
                                cdot = cdot + q4[hi].y * k4.y;
                                cdot = cdot + q4[hi].z * k4.z;
                                cdot = cdot + q4[hi].w * k4.w;
                            }
                            cpart[step] = cdot;
                        }
                    } else {
#pragma unroll
                        for (int k = 0; k < 8; k++) cpart[8 * blk + k] = 0.0f;
                    }
                }
                const float dot = transpose_sum32(cpart, l);
                const bool valid = (2 * l + half) < Tc;
                const float mys = valid ? dot / root : -3.0e38f;
                m[hi] = wave_max(mys);
                e[hi] = valid ? q3_expf(mys - m[hi]) : 0.0f;
                lsum[hi] = bfly64(e[hi]);
            }
        }
        STAMP(4);
        // ---- V tile -> LDS (it has been landing during the scores; every wave is past the
        //      PV of the previous chunk: barrier K above)
        if (LW) {
            if (pos >= t0 && pos < t0 + CH) {
                __builtin_amdgcn_s_barrier();
                if (wave == 0 && lane < L4) *reinterpret_cast<float4*>(Vs + (pos - t0) * HD + 4 * lane) = vraw_m;
            }
        } else {
#pragma unroll
            for (int k = 0; k < NLD; k++) {
                const int idx = tid + k * 256;
                const int t = idx / L4, l4 = idx - t * L4;
                const bool cur = (t0 + t == pos);
                float4 w = vt[k];
                w.x = cur ? vraw_m.x : w.x;
                w.y = cur ? vraw_m.y : w.y;
                w.z = cur ? vraw_m.z : w.z;
                w.w = cur ? vraw_m.w : w.w;
                *reinterpret_cast<float4*>(Vs + t * HD + 4 * l4) = w;
            }
        }
        __syncthreads();                                  // barrier V
        STAMP(5);
#pragma unroll
        for (int hi = 0; hi < HPW; hi++) {
            const int i = wave + 4 * hi;
            if (i < kv_mul) {
                const int h = g * kv_mul + i;
                // weighted sum of V: stream `half` takes positions half, half+2, ...
                float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
                if (Tc == CH) {       // full chunk: no masking (same sums: every position is on)
#pragma unroll
                    for (int step = 0; step < CH / 2; step++) {
                        if ((step & 7) == 0) __builtin_amdgcn_sched_barrier(0);   // eight LDS reads in flight, not 32
                        const float e0 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, e[hi]), step));
                        const float e1 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, e[hi]), 32 + step));
                        const float et = half ? e1 : e0;
                        float4 v4 = make_float4(0.f, 0.f, 0.f, 0.f);
                        if (act) v4 = *reinterpret_cast<const float4*>(Vs + (2 * step + half) * HD + 4 * l);
                        acc.x = acc.x + et * v4.x;
                        acc.y = acc.y + et * v4.y;
                        acc.z = acc.z + et * v4.z;
                        acc.w = acc.w + et * v4.w;
                    }
                } else
#pragma unroll
                for (int blk = 0; blk < 4; blk++) {
                    if (8 * blk < nsteps) {
#pragma unroll
                        for (int k = 0; k < 8; k++) {
                            const int step = 8 * blk + k;
                            const float e0 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, e[hi]), step));
                            const float e1 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, e[hi]), 32 + step));
                            const float et = half ? e1 : e0;
                            const int t = 2 * step + half;
                            float4 v4 = make_float4(0.f, 0.f, 0.f, 0.f);
                            if (act) v4 = *reinterpret_cast<const float4*>(Vs + t * HD + 4 * l);
                            const bool on = t < Tc;
                            acc.x = on ? acc.x + et * v4.x : acc.x;
                            acc.y = on ? acc.y + et * v4.y : acc.y;
                            acc.z = on ? acc.z + et * v4.z : acc.z;
                            acc.w = on ? acc.w + et * v4.w : acc.w;
                        }
                    }
                }
                STAMP(6);
                float4 o;
                o.x = acc.x + lane_xor_f<32>(acc.x);
                o.y = acc.y + lane_xor_f<32>(acc.y);
                o.z = acc.z + lane_xor_f<32>(acc.z);
                o.w = acc.w + lane_xor_f<32>(acc.w);
                if (PUB && multi == ATT_MERGE) {
                    // the partial as {tag, value} granules: the workgroup of slot 0 polls them (merge_partials_granules)
                    typedef __attribute__((address_space(1))) unsigned long long g_u64w;
                    g_u64w* pg = (g_u64w*)(a.pg + ((size_t)h * a.max_chunks + c) * (HD + 2));
                    const unsigned long long tg = (unsigned long long)tag << 32;
                    if (lane < L4) {
                        __hip_atomic_store(pg + 4 * lane + 0, tg | __float_as_uint(o.x), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        __hip_atomic_store(pg + 4 * lane + 1, tg | __float_as_uint(o.y), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        __hip_atomic_store(pg + 4 * lane + 2, tg | __float_as_uint(o.z), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        __hip_atomic_store(pg + 4 * lane + 3, tg | __float_as_uint(o.w), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        if (lane == 0) {
                            __hip_atomic_store(pg + HD, tg | __float_as_uint(m[hi]), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                            __hip_atomic_store(pg + HD + 1, tg | __float_as_uint(lsum[hi]), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        }
                    }
                } else if (multi) {
                    if (lane < L4) {
                        // write-through (sc1) stores: another workgroup of this launch reads them
                        float* pp = a.part + ((size_t)h * a.max_chunks + c) * (HD + 2);
                        st_sc1_f2(pp + 4 * lane, o.x, o.y);
                        st_sc1_f2(pp + 4 * lane + 2, o.z, o.w);
                        if (lane == 0) st_sc1_f2(pp + HD, m[hi], lsum[hi]);
                    }
                } else {
                    // q8_quantize of the head output (forward.c:291): 64-wide groups of 16 lanes
                    float4 y = make_float4(0.f, 0.f, 0.f, 0.f);
                    if (lane < L4) {
                        y.x = o.x / lsum[hi];
                        y.y = o.y / lsum[hi];
                        y.z = o.z / lsum[hi];
                        y.w = o.w / lsum[hi];
                    }
                    float scale;
                    const int packed = quantize_group16(y, scale);
                    if (lane < L4) {
                        out_codes<PUB>(a, tag, (size_t)h * HD + 4 * lane, packed);
                        if ((lane & 15) == 0) out_scale<PUB>(a, tag, (size_t)h * HD + 4 * lane, scale);
                        if (a.of) *reinterpret_cast<float4*>(a.of + (size_t)h * HD + 4 * lane) = y;
                    }
                }
            }
        }
        if (PUB && multi == ATT_MERGE) {
            // (nslots >= nchunks in this shape: every workgroup owns one chunk, and slot 0 always has one)
            if (slot == 0) {
                for (int i = wave; i < kv_mul; i += 4) merge_partials_granules<HD>(a, g * kv_mul + i, nchunks, lane, tag);
                last_flag = 1;
            }
        } else if (multi == ATT_MERGE) {
            // Publish: every storing wave drains its write-through stores, the workgroup meets,
            // ONE lane takes a ticket (Guideline 16, counter form).  The workgroup whose ticket
            // is the last of its kv head merges the partials of the head's chunks right here.
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            if (tid == 0) {
                const unsigned t = __hip_atomic_fetch_add((g_u32*)(a.tickets + g), 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                last_flag = (t == (unsigned)(nchunks - 1)) ? 1 : 0;
                if (last_flag) __hip_atomic_store((g_u32*)(a.tickets + g), 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
            __syncthreads();
            if (last_flag) {
                for (int i = wave; i < kv_mul; i += 4) merge_partials<HD, PUB>(a, g * kv_mul + i, nchunks, lane, tag);
            }
        }
    }
    STAMP(7);
    return multi == ATT_SINGLE || (multi == ATT_MERGE && last_flag != 0);
}

// the long-context launch: four compute waves + four loader waves, two workgroups per CU
template <int HD, int HPW>
__global__ __launch_bounds__(512, 4) void k_attn_long(Attn a, int sure_slots) {
    __shared__ __attribute__((aligned(16))) float Ks[Q3_ATT_CHUNK * HD];
    __shared__ __attribute__((aligned(16))) float Vs[Q3_ATT_CHUNK * HD];
    __shared__ int last_flag;
    (void)attn_body<HD, HPW, false, 4, ATT_LONG>(a, (int)ATT_LONG, (int)blockIdx.x, (int)blockIdx.y, (int)gridDim.y, sure_slots, Ks, Vs, &last_flag);
}

template <int HD, int HPW, int MODE>
__global__ __launch_bounds__(256, 2) void k_attn(Attn a_in, int multi, int rows_cap) {
    Attn a = a_in;
    {   // batched prompt ingestion: position blockIdx.z of the launch (all strides 0 for a decode step)
        const int z = blockIdx.z;
        a.ctl += z;
        a.qkv += (size_t)z * a.zs_qkv;
        a.cs += (size_t)z * a.zs_cs;
        a.oq += (size_t)z * a.zs_oq;
        a.os += (size_t)z * a.zs_os;
        a.part += (size_t)z * a.zs_part;
        a.tickets += (size_t)z * a.zs_tickets;
        if (a.of) a.of += (size_t)z * a.zs_of;
    }
    __shared__ __attribute__((aligned(16))) float Ks[Q3_ATT_CHUNK * HD];
    __shared__ __attribute__((aligned(16))) float Vs[Q3_ATT_CHUNK * HD];
    __shared__ int last_flag;
    (void)attn_body<HD, HPW, false, 0, MODE>(a, multi, (int)blockIdx.x, (int)blockIdx.y, (int)gridDim.y, rows_cap, Ks, Vs, &last_flag);
}


// ---- attention + Wo in ONE launch (decode step, fewer than Q3_ATT_LONG cached positions) -----------
// While the few attention workgroups walk their latency-bound chain (position -> head norms -> K tile ->
// scores -> softmax -> V tile -> PV -> quantise: ~6 us during which HBM idles), the rest of the chip
//   (1) pulls the Wo matrix into REGISTERS (d/(256 - n_att) rows per workgroup: 47.9 KB per CU at Qwen3-4B), and
//   (2) waits, parked, for the attention output (P codes + P/64 scales, 4.3 KB): the finalising attention waves
//       store it as 8-byte {tag, value} granules, written through (Guideline 16 R2: the data is the flag -- no drain,
//       no counter, no second trip for the payload); every consumer wave re-reads its own granules until their tags
//       carry this step and layer, parks the values in LDS, the workgroup meets once, and ~1 us of dot products +
//       the residual add finish the stage -- instead of a kernel boundary plus a 4.7-us Wo launch.
// Arithmetic: the attention body is k_attn's, the Wo rows go through tile_dot (q3_tile.hpp) as in the GEMV
// launches, so the step stays bit-identical to the unfused path (tested both ways).
// Grid: [n_att attention workgroups][256 - n_att consumer workgroups], 256 threads each = one workgroup per CU,
// every one resident, attention first in dispatch order, and every wait is bounded: a consumer
// that gives up raises *err (host-visible) and the host stops with a message.

// A wait inside a launch gives up after this much DEVICE time (s_memrealtime, 100 MHz): far beyond anything a
// healthy launch takes -- also when another process's kernels hold the attention workgroups' CUs for a while -- and
// short of what a watchdog would call a hung GPU.  (A count of polls would expire sooner the faster the polls return.)

// The attention chain does all its memory round trips in its first ~3.5 us (position, q/k/v, K tile, V tile); weight
// traffic from the rest of the chip during that time raises their latency (profiles/r03_l2_warm_experiment.md: +1.3 us
// on the stage for 11 MB).  So the consumer workgroups sit out `ticks` x 10 ns of the device clock before they
// request their Wo rows; the rows are still in registers ~1.5 us before the attention output appears.

#ifdef Q3_ATTN_STAMPS
// consumer workgroup wb leaves its marks behind the attention workgroups': stamps[8 * (n_att + wb) + i]
#define WSTAMP(i) do { if (w.stamps && threadIdx.x == 0) w.stamps[8 * (blockIdx.x) + (i)] = __builtin_amdgcn_s_memrealtime(); } while (0)
#else
#define WSTAMP(i) do {} while (0)
#endif
// NWV = waves of the consumer workgroup (4: k_merge_wo; 8: k_attn_wo, whose attention workgroups leave four of their eight
// waves at once -- twice the waves halve a consumer's rows per wave, i.e. the dot products behind the hand-off)
template <int NJ, int RW, int NWV = 4>
__device__ __forceinline__ void wo_role(const WoView& w, int wb, int8_t* lq, float* ls, int* flag) {
    const int tid = threadIdx.x, lane = tid & 63;
    WSTAMP(0);
    const int uwave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int n = w.n;
    // the step counter is read FIRST: loads return in issue order, so a load of it placed behind the weight requests
    // (where the tag is needed) made the first poll wait for every Wo row of the wave
    const unsigned tag = ((unsigned)__builtin_amdgcn_readfirstlane((int)*w.epoch) << 8) | w.layer_tag;
    // rows of this wave: wb*rpw + uwave*RW + r while uwave*RW + r < rpw; the others are aimed past the matrix
    // (rows >= d read as zero through the descriptor and move no bytes)
    int rows[RW];
    float res[RW];
#pragma unroll
    for (int r = 0; r < RW; r++) {
        const int local = uwave * RW + r;
        const int row = wb * w.rpw + local;
        rows[r] = (local < w.rpw && row < w.d) ? row : w.d;
        res[r] = rows[r] < w.d ? w.x[rows[r]] : 0.0f;
    }
    if (tid == 0) *flag = 1;                   // cleared by a wave whose bounded wait gives up
    __syncthreads();
    const WView wv = make_wview(w.W, w.S, w.d, n);
    const TileLane tl = tile_lane<NJ>(wv, lane);
    Tile<RW, NJ> T;
    hold_back(w.delay);
    WSTAMP(1);
#pragma unroll
    for (int r = 0; r < RW; r++) {
#pragma unroll
        for (int j = 0; j < NJ; j++) tile_issue_one<RW, NJ>(T, wv, tl, rows[r] - r, r, j);
    }
    __builtin_amdgcn_sched_barrier(0);
    WSTAMP(2);
    // ---- the attention output arrives as {tag, value} granules: granule i < n/4 = dword i of the codes, n/4 + g =
    // scale g.  Thread t owns code granules t, t + 256, ... and (t < n/64) one scale granule.  A wave first re-reads
    // only its lanes' FIRST granule (a few lines per poll, so that 256 waiting workgroups do not load the fabric the
    // attention chain's own round trips go through), then sweeps the rest until every tag matches.  Bounded.
    constexpr int NT = 64 * NWV;
    constexpr int NG = (NJ * 256 + NT - 1) / NT;   // code granules per thread: (n/4) / NT, n <= 1024 NJ
    const int ncode = n >> 2, nscale = n >> 6;
    typedef __attribute__((address_space(1))) unsigned long long g_u64;
    const g_u64* gr = (const g_u64*)w.gran;
    unsigned val[NG];
    unsigned sval = 0;
    int ok = 1;
    {
        const unsigned long long t_wait = __builtin_amdgcn_s_memrealtime();
        // Every load below is unconditional (indices clamped to a granule of the same kind; the compare is masked
        // instead): behind `if (i < ncode)` each load became an exec-masked block ending in s_waitcnt vmcnt(0), and
        // a sweep cost one memory round trip PER GRANULE of a thread instead of one in all.
        // ONE wave per workgroup polls, on a 64-granule block picked by the workgroup index (the polls of the 248 waiting
        // workgroups spread over all lines of the granule array); the other waves are parked at the barrier.  Measured
        // against every wave polling its first code granules (the same 16 lines chip-wide: +0.6 us per hand-off) and
        // against every wave polling the scale granules (4 lines: +1.9 us): profiles/r04_ab_runs.txt.
        if (uwave == 0) {
            const int is = ((wb * 64) % (ncode + nscale - 63)) + lane;
            for (;;) {
                const unsigned long long x = __hip_atomic_load(gr + is, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                const bool hit = (unsigned)(x >> 32) == tag;
                if (__all(hit)) break;
                __builtin_amdgcn_s_sleep(2);
                if (__builtin_amdgcn_s_memrealtime() - t_wait > w.wait_ticks) { ok = 0; break; }
            }
        }
        __builtin_amdgcn_s_barrier();
        WSTAMP(3);
        while (ok) {
            unsigned long long x[NG];
#pragma unroll
            for (int k = 0; k < NG; k++) {
                const int i = tid + NT * k;
                x[k] = __hip_atomic_load(gr + (i < ncode ? i : ncode - 1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
            const unsigned long long xs = __hip_atomic_load(gr + ncode + (tid < nscale ? tid : nscale - 1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            bool hit = (unsigned)(xs >> 32) == tag;
            sval = (unsigned)xs;
#pragma unroll
            for (int k = 0; k < NG; k++) {
                val[k] = (unsigned)x[k];
                hit = hit && (unsigned)(x[k] >> 32) == tag;
            }
            if (__all(hit)) break;
            if (__builtin_amdgcn_s_memrealtime() - t_wait > w.wait_ticks) { ok = 0; break; }
        }
    }
#pragma unroll
    for (int k = 0; k < NG; k++) {
        const int i = tid + NT * k;
        if (i < ncode) reinterpret_cast<unsigned*>(lq)[i] = val[k];
    }
    if (tid < nscale) ls[tid] = __uint_as_float(sval);
    if (!ok && lane == 0) {
        *flag = 0;
        __hip_atomic_store(w.err, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }
    WSTAMP(4);
    __syncthreads();
    WSTAMP(5);
    ok = *flag;
    float acc[RW];
    tile_dot<RW, NJ>(T, n, lane, lq, ls, acc);
    if (lane == 0 && ok) {
#pragma unroll
        for (int r = 0; r < RW; r++) {
            if (rows[r] < w.d) w.x[rows[r]] = res[r] + acc[r];      // forward.c:295-298
        }
    }
    WSTAMP(6);
}

template <int HD, int HPW, int NJ, int RW, int MODE>
__global__ __launch_bounds__(512) void k_attn_wo(Attn a, int multi, int rows_cap, int nslots, WoView w) {
    __shared__ __attribute__((aligned(16))) float Ks[Q3_ATT_CHUNK * HD];
    __shared__ __attribute__((aligned(16))) float Vs[Q3_ATT_CHUNK * HD];
    __shared__ int last_flag;
    const int n_att = a.n_kv * nslots;
    const int b = blockIdx.x;
    if (b < n_att) {
        if (threadIdx.x >= 256) return;        // the attention body is a four-wave program: the other four waves leave at once
        const bool fin = attn_body<HD, HPW, true, 0, MODE>(a, multi, MODE == ATT_SINGLE ? b : b % a.n_kv, MODE == ATT_SINGLE ? 0 : b / a.n_kv,
                                                           MODE == ATT_SINGLE ? 1 : nslots, rows_cap, Ks, Vs, &last_flag);
        (void)fin;      // its granules are the publication: nothing to drain, no flag to raise
        return;
    }
    wo_role<NJ, RW, 8>(w, b - n_att, reinterpret_cast<int8_t*>(Ks), Vs, &last_flag);
}

// ---- batched prompt ingestion: zb consecutive positions share ONE staged K/V tile --------------
// k_attn with one grid layer per position stages the same 64 KB tile once per position: at 256 cached
// positions a 64-position pass pulls 134 MB through the CUs for 2 MB of cache.  Here a workgroup
// (kv head, chunk slot, block of zb positions) stages the tile once -- the k/v rows of the pass are
// already in the cache (kv_append) -- and walks its positions with it: wave = query head as in k_attn,
// the scores / softmax / PV code is k_attn's operation for operation (q3_numerics.h "attention"), so
// are the partials, the tickets and the merge.  Positions must ascend with z (prefill passes do).
template <int HD, int HPW>
__global__ __launch_bounds__(256, 2) void k_attn_block(Attn a, int multi, int zb) {
    constexpr int L4 = HD / 4;
    constexpr int CH = Q3_ATT_CHUNK;
    constexpr int NLD = CH * L4 / 256;
    __shared__ __attribute__((aligned(16))) float Ks[CH * HD];
    __shared__ __attribute__((aligned(16))) float Vs[CH * HD];
    __shared__ int last_flags[8];

    const int g = blockIdx.x;
    const int kv_mul = a.n_heads / a.n_kv;
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int half = lane >> 5, l = lane & 31;
    const bool act = l < L4;
    const size_t cbase = (size_t)g * a.seq_len * HD;
    const int nz = a.nz > 1 ? a.nz : 1;
    const int z0 = (int)blockIdx.z * zb;
    const int z1 = z0 + zb < nz ? z0 + zb : nz;
    const int pos_last = a.ctl[z1 - 1].pos;
    const float root = sqrtf((float)HD);

    for (int c = blockIdx.y; c * CH <= pos_last; c += gridDim.y) {
        const int t0 = c * CH;
        // ---- the tile: all 64 rows (rows beyond a position's own are masked per position below)
        {
            float4 kt[NLD], vt[NLD];
#pragma unroll
            for (int k = 0; k < NLD; k++) {
                const int idx = tid + k * 256;
                const int t = idx / L4, l4 = idx - t * L4;
                kt[k] = *reinterpret_cast<const float4*>(a.kc + cbase + (size_t)(t0 + t) * HD + 4 * l4);
            }
#pragma unroll
            for (int k = 0; k < NLD; k++) {
                const int idx = tid + k * 256;
                const int t = idx / L4, l4 = idx - t * L4;
                vt[k] = *reinterpret_cast<const float4*>(a.vc + cbase + (size_t)(t0 + t) * HD + 4 * l4);
            }
#pragma unroll
            for (int k = 0; k < NLD; k++) {
                const int idx = tid + k * 256;
                const int t = idx / L4, l4 = idx - t * L4;
                *reinterpret_cast<float4*>(Ks + t * HD + 4 * l4) = kt[k];
            }
#pragma unroll
            for (int k = 0; k < NLD; k++) {
                const int idx = tid + k * 256;
                const int t = idx / L4, l4 = idx - t * L4;
                *reinterpret_cast<float4*>(Vs + t * HD + 4 * l4) = vt[k];
            }
        }
        __syncthreads();

        // raw q of the wave's head(s), the (cos,sin) slices and the position itself are fetched one
        // position ahead of the arithmetic; the head-norm weight once
        float4 qnext[HPW], qg = make_float4(0.f, 0.f, 0.f, 0.f), ca_n = qg, cb_n = qg;
        if (!a.prepared) {
            if (lane < L4) qg = *reinterpret_cast<const float4*>(a.qnw + 4 * lane);
            rope_slices<HD>(a.cs + (size_t)z0 * a.zs_cs, lane, ca_n, cb_n);
        }
        int pos_n = a.ctl[z0].pos;
#pragma unroll
        for (int hi = 0; hi < HPW; hi++) {
            qnext[hi] = make_float4(0.f, 0.f, 0.f, 0.f);
            const int i = wave + 4 * hi;
            if (i < kv_mul && lane < L4)
                qnext[hi] = *reinterpret_cast<const float4*>(a.qkv + (size_t)z0 * a.zs_qkv + (size_t)(g * kv_mul + i) * HD + 4 * lane);
        }
        for (int z = z0; z < z1; z++) {
            float4 qraw[HPW];
            const float4 ca = ca_n, cb = cb_n;
            const int pos = pos_n;
            const int zn = z + 1 < z1 ? z + 1 : z;               // the last round re-reads its own (unused)
#pragma unroll
            for (int hi = 0; hi < HPW; hi++) {
                qraw[hi] = qnext[hi];
                const int i = wave + 4 * hi;
                if (i < kv_mul && lane < L4)
                    qnext[hi] = *reinterpret_cast<const float4*>(a.qkv + (size_t)zn * a.zs_qkv + (size_t)(g * kv_mul + i) * HD + 4 * lane);
            }
            if (!a.prepared) rope_slices<HD>(a.cs + (size_t)zn * a.zs_cs, lane, ca_n, cb_n);
            pos_n = a.ctl[zn].pos;
            const int T = pos + 1;
            const int nchunks = (T + CH - 1) / CH;
            if (c >= nchunks) continue;                          // workgroup-uniform
            const int Tc = (T - t0 < CH) ? T - t0 : CH;
            const int nsteps = (Tc + 1) >> 1;
#pragma unroll
            for (int hi = 0; hi < HPW; hi++) {
                const int i = wave + 4 * hi;
                if (i >= kv_mul) continue;                       // wave-uniform
                const int h = g * kv_mul + i;
                // q of this head: norm + rope (k_attn's general path), slice (lane % L4) in both halves
                float4 q4 = qraw[hi];
                if (!a.prepared) q4 = headnorm_rope_vals<HD>(q4, qg, ca, cb, lane);
                {
                    const float ox = lane_xor_f<32>(q4.x), oy = lane_xor_f<32>(q4.y);
                    const float oz = lane_xor_f<32>(q4.z), ow = lane_xor_f<32>(q4.w);
                    if (half) q4 = make_float4(ox, oy, oz, ow);
                }
                // scores (k_attn): step s handles position t = 2*s + half
                float cpart[32];
#pragma unroll
                for (int blk = 0; blk < 4; blk++) {
                    if (8 * blk < nsteps) {
#pragma unroll
                        for (int k = 0; k < 8; k++) {
                            const int step = 8 * blk + k;
                            const int t = 2 * step + half;
                            float cdot = 0.0f;
                            if (act) {
                                const float4 k4 = *reinterpret_cast<const float4*>(Ks + t * HD + 4 * l);
                                cdot = q4.x * k4.x;
                                cdot = cdot + q4.y * k4.y;
                                cdot = cdot + q4.z * k4.z;
                                cdot = cdot + q4.w * k4.w;
                            }
                            cpart[step] = cdot;
                        }
                    } else {
#pragma unroll
                        for (int k = 0; k < 8; k++) cpart[8 * blk + k] = 0.0f;
                    }
                }
                const float dot = transpose_sum32(cpart, l);
                const bool valid = (2 * l + half) < Tc;
                const float mys = valid ? dot / root : -3.0e38f;
                const float m = wave_max(mys);
                const float e = valid ? q3_expf(mys - m) : 0.0f;
                const float lsum = bfly64(e);
                // weighted sum of V
                float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
                if (Tc == CH) {
#pragma unroll
                    for (int step = 0; step < CH / 2; step++) {
                        if ((step & 7) == 0) __builtin_amdgcn_sched_barrier(0);
                        const float e0 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, e), step));
                        const float e1 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, e), 32 + step));
                        const float et = half ? e1 : e0;
                        float4 v4 = make_float4(0.f, 0.f, 0.f, 0.f);
                        if (act) v4 = *reinterpret_cast<const float4*>(Vs + (2 * step + half) * HD + 4 * l);
                        acc.x = acc.x + et * v4.x;
                        acc.y = acc.y + et * v4.y;
                        acc.z = acc.z + et * v4.z;
                        acc.w = acc.w + et * v4.w;
                    }
                } else
#pragma unroll
                for (int blk = 0; blk < 4; blk++) {
                    if (8 * blk < nsteps) {
#pragma unroll
                        for (int k = 0; k < 8; k++) {
                            const int step = 8 * blk + k;
                            const float e0 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, e), step));
                            const float e1 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, e), 32 + step));
                            const float et = half ? e1 : e0;
                            const int t = 2 * step + half;
                            float4 v4 = make_float4(0.f, 0.f, 0.f, 0.f);
                            if (act) v4 = *reinterpret_cast<const float4*>(Vs + t * HD + 4 * l);
                            const bool on = t < Tc;
                            acc.x = on ? acc.x + et * v4.x : acc.x;
                            acc.y = on ? acc.y + et * v4.y : acc.y;
                            acc.z = on ? acc.z + et * v4.z : acc.z;
                            acc.w = on ? acc.w + et * v4.w : acc.w;
                        }
                    }
                }
                float4 o;
                o.x = acc.x + lane_xor_f<32>(acc.x);
                o.y = acc.y + lane_xor_f<32>(acc.y);
                o.z = acc.z + lane_xor_f<32>(acc.z);
                o.w = acc.w + lane_xor_f<32>(acc.w);
                if (multi) {
                    if (lane < L4) {
                        float* pp = a.part + (size_t)z * a.zs_part + ((size_t)h * a.max_chunks + c) * (HD + 2);
                        st_sc1_f2(pp + 4 * lane, o.x, o.y);
                        st_sc1_f2(pp + 4 * lane + 2, o.z, o.w);
                        if (lane == 0) st_sc1_f2(pp + HD, m, lsum);
                    }
                } else {
                    float4 y = make_float4(0.f, 0.f, 0.f, 0.f);
                    if (lane < L4) {
                        y.x = o.x / lsum;
                        y.y = o.y / lsum;
                        y.z = o.z / lsum;
                        y.w = o.w / lsum;
                    }
                    float scale;
                    const int packed = quantize_group16(y, scale);
                    if (lane < L4) {
                        reinterpret_cast<int*>(a.oq + (size_t)z * a.zs_oq)[((size_t)h * HD + 4 * lane) >> 2] = packed;
                        if ((lane & 15) == 0) (a.os + (size_t)z * a.zs_os)[((size_t)h * HD + 4 * lane) >> 6] = scale;
                        if (a.of) *reinterpret_cast<float4*>(a.of + (size_t)z * a.zs_of + (size_t)h * HD + 4 * lane) = y;
                    }
                }
            }
        }
        if (multi == ATT_MERGE) {
            // One publication for the whole block: every storing wave drains its write-through stores,
            // the workgroup meets, lane zi takes the ticket of position z0 + zi (one round trip for all
            // of them); the positions whose last ticket this workgroup drew are merged right here.
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            if (tid < z1 - z0) {
                const int z = z0 + tid;
                const int nchunks = a.ctl[z].pos / CH + 1;
                int last = 0;
                if (c < nchunks) {
                    unsigned* tk = a.tickets + (size_t)z * a.zs_tickets + g;
                    const unsigned t = __hip_atomic_fetch_add((g_u32*)tk, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    last = (t == (unsigned)(nchunks - 1)) ? 1 : 0;
                    if (last) __hip_atomic_store((g_u32*)tk, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
                last_flags[tid] = last;
            }
            __syncthreads();
            for (int z = z0; z < z1; z++) {
                if (!last_flags[z - z0]) continue;                // workgroup-uniform
                Attn az = a;
                az.part += (size_t)z * a.zs_part;
                az.oq += (size_t)z * a.zs_oq;
                az.os += (size_t)z * a.zs_os;
                if (a.of) az.of += (size_t)z * a.zs_of;
                const int nchunks = a.ctl[z].pos / CH + 1;
                for (int i = wave; i < kv_mul; i += 4) merge_partials<HD>(az, g * kv_mul + i, nchunks, lane);
            }
        }
        __syncthreads();                                         // every wave is done with this tile
    }
}

// ---- ATT_LONG: merge of the chunk partials as its own launch ---------------------------
// One wave per 64 consecutive output values (= one quantisation group; HD/64 waves per head),
// lane = one value.  The sums over chunks are sequential by contract (q3_numerics.h), but every
// lane owns its own chain and all the loads of 32 chunks are in flight together.
__device__ __forceinline__ int q8_code1(float y, float scale, float inv) {
    const float t = fabsf(y * inv) + 0.5f;
    const float gg = fabsf(__builtin_amdgcn_fractf(t) - 0.5f);
    if (__builtin_expect(gg > 0.499f || !(scale >= 1e-30f), 0)) return q8_code_exact(y, scale);
    return (int)copysignf(fminf(floorf(t), 127.0f), y);
}
// merge of one 64-value group (head h, group grp) by ONE wave; PUB: the codes and the scale leave as tagged granules
#ifdef Q3_ATTN_STAMPS
#define MSTAMP(i) do { if (PUB && a.stamps && lane == 0) a.stamps[8 * blockIdx.x + (i)] = __builtin_amdgcn_s_memrealtime(); } while (0)
#else
#define MSTAMP(i) do {} while (0)
#endif
template <int HD, bool PUB>
__device__ __forceinline__ void merge_group(const Attn& a, int h, int grp, int lane) {
    constexpr int ST = HD + 2;
    MSTAMP(0);
    const float* base = a.part + (size_t)h * a.max_chunks * ST;
    const int d = grp * 64 + lane;
    // Everything the first 64 chunks need -- their (m_c, l_c) pairs, one per lane, and this lane's
    // value of each O_c -- is requested in ONE burst together with the position itself (rows beyond
    // the last chunk exist in the buffer; what they hold is masked below): the usual context
    // (<= 4096 positions) costs a single memory round trip.  Longer contexts walk further blocks.
    const int rmax = a.max_chunks - 1;
    float2 ml0 = *reinterpret_cast<const float2*>(base + (size_t)(lane < rmax ? lane : rmax) * ST + HD);
    float o0[64];
#pragma unroll
    for (int k = 0; k < 64; k++) o0[k] = base[(size_t)(k < rmax ? k : rmax) * ST + d];
    const int nchunks = ld_scalar(&a.ctl->pos) / Q3_ATT_CHUNK + 1;
    unsigned tag = 0;
    if (PUB) tag = ((unsigned)ld_scalar(reinterpret_cast<const int*>(a.epoch)) << 8) | a.layer_tag;
    if (lane >= nchunks) ml0 = make_float2(-3.0e38f, 0.0f);
    float M = wave_max(ml0.x);
    MSTAMP(1);
    for (int c0 = 64; c0 < nchunks; c0 += 64) {
        const int c = c0 + lane;
        const float mc = c < nchunks ? base[(size_t)c * ST + HD] : -3.0e38f;
        M = fmaxf(M, wave_max(mc));
    }
    float L = 0.0f, A = 0.0f;
    {
        float wc = 0.0f, wl = 0.0f;
        if (lane < nchunks) {
            wc = q3_expf(ml0.x - M);
            wl = wc * ml0.y;
        }
        const int cnt = nchunks < 64 ? nchunks : 64;
        // the sums over chunks are sequential by contract; blocks of eight steps run without a branch per step (a
        // branch per step cost the 64-chunk chain 1.5 us), the steps past the last chunk keep L and A by a select
#pragma unroll
        for (int k0 = 0; k0 < 64; k0 += 8) {
            if (k0 < cnt) {
#pragma unroll
                for (int k = k0; k < k0 + 8; k++) {
                    const float w = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, wc), k));
                    const float t = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, wl), k));
                    const float Ln = L + t, An = A + w * o0[k];
                    L = k < cnt ? Ln : L;
                    A = k < cnt ? An : A;
                }
            }
        }
    }
    for (int c0 = 64; c0 < nchunks; c0 += 64) {
        const int c = c0 + lane;
        float wc = 0.0f, wl = 0.0f;
        if (c < nchunks) {
            const float2 ml = *reinterpret_cast<const float2*>(base + (size_t)c * ST + HD);
            wc = q3_expf(ml.x - M);
            wl = wc * ml.y;
        }
        const int cnt = (nchunks - c0 < 64) ? nchunks - c0 : 64;
        for (int k0 = 0; k0 < cnt; k0 += 32) {
            float o[32];
#pragma unroll
            for (int k = 0; k < 32; k++) {
                const int cc = c0 + k0 + k;
                o[k] = base[(size_t)(cc < nchunks ? cc : nchunks - 1) * ST + d];
            }
#pragma unroll
            for (int k = 0; k < 32; k++) {
                if (k0 + k < cnt) {
                    const float w = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, wc), (k0 + k) & 63));
                    const float t = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, wl), (k0 + k) & 63));
                    L = L + t;
                    A = A + w * o[k];
                }
            }
        }
    }
    MSTAMP(2);
    const float y = A / L;
    const float scale = q3_q8_scale(wave_max(fabsf(y)));
    const int q = q8_code1(y, scale, __builtin_amdgcn_rcpf(scale));
    if (PUB) {
        // four lanes' codes -> one dword (little-endian, as the int8 array holds them), one granule per quad
        const int b0 = q & 255;
        const int b1 = lane_xor_i<1>(b0);                 // the other lane of the pair
        const int pair = (lane & 1) ? ((b0 << 8) | b1) : ((b1 << 8) | b0);
        const int other = lane_xor_i<2>(pair);            // the other pair of the quad
        const int packed = (lane & 2) ? ((pair << 16) | other) : ((other << 16) | pair);
        if ((lane & 3) == 0) out_codes<true>(a, tag, (size_t)h * HD + d, packed);
        if (lane == 0) out_scale<true>(a, tag, (size_t)h * HD + d, scale);
    } else {
        a.oq[(size_t)h * HD + d] = (int8_t)q;
        if (lane == 0) a.os[((size_t)h * HD + d) >> 6] = scale;
    }
    if (a.of) a.of[(size_t)h * HD + d] = y;
    MSTAMP(3);
}

template <int HD>
__global__ __launch_bounds__(64) void k_attn_merge(Attn a_in) {
    Attn a = a_in;
    {
        const int z = blockIdx.y;
        a.ctl += z;
        a.oq += (size_t)z * a.zs_oq;
        a.os += (size_t)z * a.zs_os;
        a.part += (size_t)z * a.zs_part;
        if (a.of) a.of += (size_t)z * a.zs_of;
    }
    constexpr int GPH = HD / 64;                // quantisation groups per head
    merge_group<HD, false>(a, blockIdx.x / GPH, blockIdx.x % GPH, threadIdx.x);
}

// ---- ATT_LONG: merge + Wo in ONE launch.  The K/V stream of the attention launch keeps HBM busy, so Wo cannot ride
// along there; it rides with the merge instead: n_heads*HD/64 workgroups merge the chunk partials (one wave per 64
// outputs, as k_attn_merge) while the other CUs pull their Wo rows into registers, then take the merged output as
// tagged granules (wo_role) -- instead of a merge launch, a boundary and a Wo launch.
template <int HD, int NJ, int RW>
__global__ __launch_bounds__(512) void k_merge_wo(Attn a, WoView w) {
    __shared__ __attribute__((aligned(16))) int8_t lq[4096];
    __shared__ __attribute__((aligned(16))) float ls[64];
    __shared__ int flag;
    constexpr int GPH = HD / 64;
    const int nm = a.n_heads * GPH;                     // merge workgroups: ONE wave each (a wave pulls 16.6 KB of partials at
    const int b = blockIdx.x;                           // 4096 positions; four of them behind one CU's queue took 2 us longer)
    if (b < nm) {
        if (threadIdx.x < 64) merge_group<HD, true>(a, b / GPH, b % GPH, threadIdx.x);
        return;
    }
    wo_role<NJ, RW, 8>(w, b - nm, lq, ls, &flag);
}

// ---- batched prompt ingestion: the k/v rows of a run of positions, ahead of a batched attn() ----
template <int HD>
__global__ __launch_bounds__(64) void k_kv_append(Attn a) {
    constexpr int L4 = HD / 4;
    const int g = blockIdx.x, t = blockIdx.y, lane = threadIdx.x;
    const int pos = a.ctl[t].pos;
    const int P = a.n_heads * HD, KVD = a.n_kv * HD;
    const float* row = a.qkv + (size_t)t * a.zs_qkv;
    float4 k = make_float4(0.f, 0.f, 0.f, 0.f), v = k;
    if (lane < L4) {
        k = *reinterpret_cast<const float4*>(row + P + (size_t)g * HD + 4 * lane);
        v = *reinterpret_cast<const float4*>(row + P + KVD + (size_t)g * HD + 4 * lane);
    }
    k = headnorm_rope_wave<HD>(k, a.knw, a.cs + (size_t)t * a.zs_cs, lane);
    if (lane < L4) {
        const size_t off = ((size_t)g * a.seq_len + pos) * HD + 4 * lane;
        *reinterpret_cast<float4*>(a.kc + off) = k;
        *reinterpret_cast<float4*>(a.vc + off) = v;
    }
}
void kv_append(const Attn& a, int ntok, hipStream_t st) {
    if (a.hd == 128) hipLaunchKernelGGL(k_kv_append<128>, dim3(a.n_kv, ntok), dim3(64), 0, st, a);
    else hipLaunchKernelGGL(k_kv_append<64>, dim3(a.n_kv, ntok), dim3(64), 0, st, a);
}

// Geometry of the launch: the attention workgroups and the extra ones together are ONE workgroup per CU (256) --
// an extra workgroup that shares a CU with an attention workgroup puts its tens of KB of weight requests into the
// queue the attention chain's dependent loads wait in (measured: +1.7 us on the stage).  So nwo = 256 - n_att extra
// workgroups; fused, they split the d rows of Wo: rows per workgroup, wave-loads per row, rows per wave.
// false = shape not covered (callers then launch attn() and the Wo GEMV separately).
// (cu_count(): the device's compute units, q3_kernels.hpp -- never a literal: the waits inside the launch rely on every
// workgroup of the grid being resident at once)
static int extra_workgroups(int n_att) { const int ncu = cu_count(); return n_att < ncu * 3 / 4 ? ncu - n_att : ncu / 4; }
static int merge_workgroups(const Attn& a);
static bool wo_geometry(const WoView& w, int n_att, int* rpw, int* nj, int* rw, int nwv = 4) {
    if (w.n % 64 || w.n > 4096 || w.d < 1) return false;
    const int nwo = extra_workgroups(n_att);
    *rpw = (w.d + nwo - 1) / nwo;
    *nj = (w.n + 1023) / 1024;
    *rw = (*rpw + nwv - 1) / nwv;
    // (up to 8 rows of 4 KiB per wave = 160 registers of tile: the 8B shapes' in-launch-merge launch, 32 rows per consumer)
    return (*nj == 1 || *nj == 2 || *nj == 4) && (*rw <= 5 || (*nj == 4 && *rw <= 8));
}
static int attn_slots(int chunk_slots, AttMode mode) {
    // ATT_MERGE serves positions below Q3_ATT_LONG only: never more than Q3_ATT_LONG / 64 chunks, so the
    // launch need not dispatch (and retire) workgroups for slots that can have no chunk
    const int merge_slots = chunk_slots < Q3_ATT_LONG / Q3_ATT_CHUNK ? chunk_slots : Q3_ATT_LONG / Q3_ATT_CHUNK;
    return mode == ATT_SINGLE ? 1 : (mode == ATT_MERGE ? merge_slots : chunk_slots);
}
static const void* fused_kernel(const Attn& a, const WoView& w, int chunk_slots, AttMode mode, int* grid, int* rpw_out);
static bool grid_resident(const void* kernel, int grid, int threads);
bool attn_wo_supported(const Attn& a, const WoView& w, int chunk_slots, AttMode mode) {
    int rpw, nj, rw;
    // (head_dim 128 only -- every Qwen3 size; the head_dim-64 test shapes take the separate launches)
    if (!(a.nz <= 1 && !a.of && !a.prepared && a.n_heads / a.n_kv <= Q3_MAXG && a.hd == 128)) return false;
    if (!w.W || !a.og || !a.epoch) return false;
    const int producers = mode == ATT_LONG ? merge_workgroups(a) : a.n_kv * attn_slots(chunk_slots, mode);
    if (!(w.n == a.n_heads * a.hd && w.n <= 4096 && producers < cu_count() && wo_geometry(w, producers, &rpw, &nj, &rw, 8))) return false;
    int grid = 0;
    const void* kern = fused_kernel(a, w, chunk_slots, mode, &grid, nullptr);
    if (!grid_resident(kern, grid, 512)) {
        static bool told = false;
        if (!told) fprintf(stderr, "[q3hip] the fused attention + Wo launch (%d workgroups) would not be resident at once on this device: separate launches\n", grid);
        told = true;
        return false;
    }
    return true;
}

// ---- the fused launches by function pointer: the same pointer serves the residency check and the launch ----
typedef void (*MergeWoFn)(Attn, WoView);
typedef void (*AttnWoFn)(Attn, int, int, int, WoView);
template <int HD>
static MergeWoFn pick_merge_wo(int nj, int rw) {
#define Q3_MW(NJ, RW) return k_merge_wo<HD, NJ, RW>
    switch (nj * 8 + rw) {
        case 1 * 8 + 1: Q3_MW(1, 1);
        case 1 * 8 + 2: Q3_MW(1, 2);
        case 1 * 8 + 3: Q3_MW(1, 3);
        case 1 * 8 + 4: Q3_MW(1, 4);
        case 1 * 8 + 5: Q3_MW(1, 5);
        case 2 * 8 + 1: Q3_MW(2, 1);
        case 2 * 8 + 2: Q3_MW(2, 2);
        case 2 * 8 + 3: Q3_MW(2, 3);
        case 2 * 8 + 4: Q3_MW(2, 4);
        case 2 * 8 + 5: Q3_MW(2, 5);
        case 4 * 8 + 1: Q3_MW(4, 1);
        case 4 * 8 + 2: Q3_MW(4, 2);
        case 4 * 8 + 3: Q3_MW(4, 3);
        case 4 * 8 + 4: Q3_MW(4, 4);
        case 4 * 8 + 5: Q3_MW(4, 5);
        case 4 * 8 + 6: Q3_MW(4, 6);
        case 4 * 8 + 7: Q3_MW(4, 7);
        default: Q3_MW(4, 8);
    }
#undef Q3_MW
}
template <int HD, int HPW>
static AttnWoFn pick_attn_wo(int nj, int rw, AttMode mode) {
#define Q3_AW(NJ, RW) return mode == ATT_SINGLE ? (AttnWoFn)k_attn_wo<HD, HPW, NJ, RW, ATT_SINGLE> : (AttnWoFn)k_attn_wo<HD, HPW, NJ, RW, ATT_MERGE>
    switch (nj * 8 + rw) {
        case 1 * 8 + 1: Q3_AW(1, 1);
        case 1 * 8 + 2: Q3_AW(1, 2);
        case 1 * 8 + 3: Q3_AW(1, 3);
        case 1 * 8 + 4: Q3_AW(1, 4);
        case 1 * 8 + 5: Q3_AW(1, 5);
        case 2 * 8 + 1: Q3_AW(2, 1);
        case 2 * 8 + 2: Q3_AW(2, 2);
        case 2 * 8 + 3: Q3_AW(2, 3);
        case 2 * 8 + 4: Q3_AW(2, 4);
        case 2 * 8 + 5: Q3_AW(2, 5);
        case 4 * 8 + 1: Q3_AW(4, 1);
        case 4 * 8 + 2: Q3_AW(4, 2);
        case 4 * 8 + 3: Q3_AW(4, 3);
        case 4 * 8 + 4: Q3_AW(4, 4);
        case 4 * 8 + 5: Q3_AW(4, 5);
        case 4 * 8 + 6: Q3_AW(4, 6);
        case 4 * 8 + 7: Q3_AW(4, 7);
        default: Q3_AW(4, 8);
    }
#undef Q3_AW
}
// The consumers of a fused launch spin on granules that other workgroups of the SAME launch write: that only ends if
// every workgroup of the grid holds a CU at once.  Checked per kernel against the runtime's occupancy answer (cached);
// one workgroup per CU is all these grids ask for, so a grid of at most cu_count() workgroups is resident when the
// answer is at least 1.  The bounded waits and the host's fallback (q3_shim.hip: sync_checked) stay as the backstop.
static bool grid_resident(const void* kernel, int grid, int threads) {
    static std::unordered_map<const void*, int> cache;
    auto it = cache.find(kernel);
    int per_cu;
    if (it != cache.end()) {
        per_cu = it->second;
    } else {
        per_cu = 0;
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, kernel, threads, 0) != hipSuccess) per_cu = 0;
        cache[kernel] = per_cu;
    }
    return per_cu >= 1 && grid <= per_cu * cu_count();
}
// ATT_LONG: the merge + Wo launch behind the chunk launch
static int merge_workgroups(const Attn& a) { return a.n_heads * (a.hd / 64); }
template <int HD>
static void launch_merge_wo(const Attn& a, WoView w, hipStream_t st) {
    int rpw = 1, nj = 1, rw = 1;
    const int nm = merge_workgroups(a), nwo = extra_workgroups(nm);
    wo_geometry(w, nm, &rpw, &nj, &rw, 8);
    w.rpw = rpw;
    const dim3 grid(nm + nwo), blk(512);
    const MergeWoFn fn = pick_merge_wo<HD>(nj, rw);
    hipLaunchKernelGGL(fn, grid, blk, 0, st, a, w);
}

template <int HD, int HPW>
static void launch_attn_wo(const Attn& a, AttMode mode, int rows_cap, int slots, WoView w, hipStream_t st) {
    int rpw = 1, nj = 1, rw = 1;
    const int n_att = a.n_kv * slots, nwo = extra_workgroups(n_att);
    wo_geometry(w, n_att, &rpw, &nj, &rw, 8);
    w.rpw = rpw;
    const dim3 grid(n_att + nwo), blk(512);
    const AttnWoFn fn = pick_attn_wo<HD, HPW>(nj, rw, mode);
    hipLaunchKernelGGL(fn, grid, blk, 0, st, a, (int)mode, rows_cap, slots, w);
}

static const void* fused_kernel(const Attn& a, const WoView& w, int chunk_slots, AttMode mode, int* grid, int* rpw_out) {
    int rpw = 1, nj = 1, rw = 1;
    const bool two = a.n_heads / a.n_kv > 4;
    if (mode == ATT_LONG) {
        const int nm = merge_workgroups(a);
        wo_geometry(w, nm, &rpw, &nj, &rw, 8);
        *grid = nm + extra_workgroups(nm);
        if (rpw_out) *rpw_out = rpw;
        return (const void*)pick_merge_wo<128>(nj, rw);
    }
    const int n_att = a.n_kv * attn_slots(chunk_slots, mode);
    wo_geometry(w, n_att, &rpw, &nj, &rw, 8);
    *grid = n_att + extra_workgroups(n_att);
    if (rpw_out) *rpw_out = rpw;
    return two ? (const void*)pick_attn_wo<128, 2>(nj, rw, mode) : (const void*)pick_attn_wo<128, 1>(nj, rw, mode);
}

void attn(const Attn& a, int chunk_slots, AttMode mode, hipStream_t st, int rows_cap, const WoView* wo) {
    if (a.n_heads / a.n_kv > Q3_MAXG) {
        fprintf(stderr, "[q3hip] attention: more than %d query heads per kv head\n", Q3_MAXG);
        exit(EXIT_FAILURE);
    }
    if (a.hd != 128 && a.hd != 64) {
        fprintf(stderr, "[q3hip] attention: head_dim %d not supported (64 or 128)\n", a.hd);
        exit(EXIT_FAILURE);
    }
    if (mode != ATT_LONG && (rows_cap < 8 || rows_cap > Q3_ATT_CHUNK)) rows_cap = Q3_ATT_CHUNK;
    const int nz = a.nz > 1 ? a.nz : 1;
    const bool two = a.n_heads / a.n_kv > 4;
    const int slots = attn_slots(chunk_slots, mode);
    if (wo && !attn_wo_supported(a, *wo, chunk_slots, mode)) {
        fprintf(stderr, "[q3hip] attention: fused Wo launch requested for a shape it does not cover\n");
        exit(EXIT_FAILURE);
    }
    if (wo && mode != ATT_LONG) {
        if (!two) launch_attn_wo<128, 1>(a, mode, rows_cap, slots, *wo, st);
        else launch_attn_wo<128, 2>(a, mode, rows_cap, slots, *wo, st);
        return;
    }
    if (nz > 1) {
        // a pass of the batched prompt path: blocks of positions share a staged tile; as many positions
        // per block as still leaves ~2 workgroups per CU
        int zb = a.n_kv * slots * nz / 512;
        zb = zb < 1 ? 1 : (zb > 8 ? 8 : zb);
        dim3 gridb(a.n_kv, slots, (nz + zb - 1) / zb);
        // Beyond a few chunks a pass merges in the wide second launch (one wave per 64 outputs, all
        // positions at once): in-launch, the workgroup of the last chunk would merge every position of
        // its block, one after the other, while the rest of the chip is done.  Same sums, same order.
        if (mode == ATT_MERGE && slots > 4) mode = ATT_LONG;
        if (a.hd == 128 && !two) hipLaunchKernelGGL((k_attn_block<128, 1>), gridb, dim3(256), 0, st, a, (int)mode, zb);
        else if (a.hd == 128) hipLaunchKernelGGL((k_attn_block<128, 2>), gridb, dim3(256), 0, st, a, (int)mode, zb);
        else if (!two) hipLaunchKernelGGL((k_attn_block<64, 1>), gridb, dim3(256), 0, st, a, (int)mode, zb);
        else hipLaunchKernelGGL((k_attn_block<64, 2>), gridb, dim3(256), 0, st, a, (int)mode, zb);
        if (mode == ATT_LONG) {
            if (a.hd == 128) hipLaunchKernelGGL(k_attn_merge<128>, dim3(a.n_heads * 2, nz), dim3(64), 0, st, a);
            else hipLaunchKernelGGL(k_attn_merge<64>, dim3(a.n_heads, nz), dim3(64), 0, st, a);
        }
        return;
    }
    dim3 grid(a.n_kv, slots, nz);
    static const bool long_loader = !(getenv("Q3_ATT_LOADER") && getenv("Q3_ATT_LOADER")[0] == '0');
    if (mode == ATT_LONG && long_loader && a.hd == 128) {      // (head_dim 64: the one-role kernel; its general path does not fit 128 registers)
        // chunk slots that hold cached positions for certain: what the caller vouches for (step_rows_cap: 16 per 1024
        // positions the launch shape has reached), never fewer than the 16 the mode itself implies (attn_mode() sent the
        // caller here, so at least Q3_ATT_LONG positions are cached); a caller that does not say (rows_cap 0) gets those 16
        int sure = rows_cap < Q3_ATT_LONG / Q3_ATT_CHUNK ? Q3_ATT_LONG / Q3_ATT_CHUNK : rows_cap;
        if (sure > slots) sure = slots;
        if (!two) hipLaunchKernelGGL((k_attn_long<128, 1>), grid, dim3(512), 0, st, a, sure);
        else hipLaunchKernelGGL((k_attn_long<128, 2>), grid, dim3(512), 0, st, a, sure);
    } else {
        const int rc = mode == ATT_LONG ? Q3_ATT_CHUNK : rows_cap;       // the one-role kernel takes whole tiles in ATT_LONG
#define Q3_KA(HD_, HPW_) do { if (mode == ATT_SINGLE) hipLaunchKernelGGL((k_attn<HD_, HPW_, ATT_SINGLE>), grid, dim3(256), 0, st, a, (int)mode, rc); \
                              else if (mode == ATT_MERGE) hipLaunchKernelGGL((k_attn<HD_, HPW_, ATT_MERGE>), grid, dim3(256), 0, st, a, (int)mode, rc); \
                              else hipLaunchKernelGGL((k_attn<HD_, HPW_, ATT_LONG>), grid, dim3(256), 0, st, a, (int)mode, rc); } while (0)
        if (a.hd == 128 && !two) Q3_KA(128, 1);
        else if (a.hd == 128) Q3_KA(128, 2);
        else if (!two) Q3_KA(64, 1);
        else Q3_KA(64, 2);
#undef Q3_KA
    }
    if (mode == ATT_LONG && wo) {
        launch_merge_wo<128>(a, *wo, st);
    } else if (mode == ATT_LONG) {
        if (a.hd == 128) hipLaunchKernelGGL(k_attn_merge<128>, dim3(a.n_heads * 2, nz), dim3(64), 0, st, a);
        else hipLaunchKernelGGL(k_attn_merge<64>, dim3(a.n_heads, nz), dim3(64), 0, st, a);
    }
}

// start of a step on one device: embedding row (when this stage owns it) and the
// (cos,sin) row of `pos` copied to a fixed address so that no later load depends on pos
__global__ __launch_bounds__(256) void k_begin(const Ctl* ctl, const int8_t* __restrict__ eq,
                                               const float* __restrict__ es, int dim, float* __restrict__ x,
                                               const float* __restrict__ rope, int hd, float* __restrict__ cs,
                                               unsigned* __restrict__ epoch, int vocab) {
    const int pos = ctl->pos;
    if (blockIdx.x == 0) {
        for (int i = threadIdx.x; i < hd; i += 256) cs[i] = rope[(size_t)pos * hd + i];
        // the step counter that tags this step's in-launch hand-offs (k_attn_wo): a granule left by an earlier step,
        // or by one that was cut short, can never carry it
        if (epoch && threadIdx.x == 0) epoch[0] = epoch[0] + 1u;
    }
    if (eq) {
        // x = q*s of one embedding row (reference model.c:201-206 dequantises the whole table
        // on the host and forward.c:237 copies a row; the product q*s is the same single rounding)
        // the token may come from device memory (the previous step's argmax or sampler) or, on a pipeline's first stage,
        // from the slot a neighbour sent: whatever it holds, the row fetched is a row of the table
        int token = ctl->token;
        if ((unsigned)token >= (unsigned)vocab) token = 0;
        const size_t base = (size_t)token * dim;
        for (int i = blockIdx.x * 256 + threadIdx.x; i < dim; i += gridDim.x * 256) {
            x[i] = (float)eq[base + i] * es[(base + i) >> 6];
        }
    }
}
void begin_step(const Ctl* ctl, const int8_t* eq, const float* es, int dim, float* x, const float* rope,
                int hd, float* cs, hipStream_t st, unsigned* epoch, int vocab) {
    const int blocks = eq ? (dim + 255) / 256 : 1;
    hipLaunchKernelGGL(k_begin, dim3(blocks), dim3(256), 0, st, ctl, eq, es, dim, x, rope, hd, cs, epoch, vocab);
}


// ctl = {token, pos} for the step about to run: the token comes from device memory when
// the previous step's argmax (or a pipeline recv) produced it, else from the argument
__global__ void k_set_ctl(Ctl* ctl, const int* tok_src, int tok_imm, int pos) {
    ctl->token = tok_src ? *tok_src : tok_imm;
    ctl->pos = pos;
}
void set_ctl(Ctl* ctl, const int* tok_src, int tok_imm, int pos, hipStream_t st) {
    hipLaunchKernelGGL(k_set_ctl, dim3(1), dim3(1), 0, st, ctl, tok_src, tok_imm, pos);
}

}  // namespace q3k
