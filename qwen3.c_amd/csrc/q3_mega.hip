// q3_mega.hip -- the whole decode step of one pipeline stage as ONE persistent launch.
//
// Why: at batch 1 a layer is five dependent GEMV/attention stages of 2-9 us of HBM
// transfer each.  As separate launches every stage pays launch + kernarg + first-byte
// latency and its own tail (~3-4 us) plus ~1.9 us between kernels, and HBM idles through
// all of it.  Here one workgroup per CU stays resident for the whole
// step and the weight stream is decoupled from the dependency chain:
//
//   * 6 WORKER waves per workgroup do nothing but stream weights and multiply: their
//     vector-memory queue holds only weight tiles (q3_tile.hpp), always two tiles ahead of
//     the one being consumed, across stage and layer boundaries.  vmcnt retires in order,
//     so a worker never issues any other global access -- activations reach it through LDS.
//   * 2 AUX waves per workgroup own everything small and latency-bound: the hand-offs
//     between stages (results out with write-through stores, grid-wide arrival counters,
//     the all-gather of the next activation vector into LDS), the attention stage, and the
//     embedding row.  While they wait, the workers' prefetch keeps HBM busy.
//
// Hand-off protocol (cdna_hip_programming.md Guideline 16, counter form): producers store
// with sc1 (write-through), drain with s_waitcnt vmcnt(0), then ONE lane adds to the
// workgroup's shard of a monotonic 8-shard arrival counter (agent scope); consumers poll
// all 8 shards with relaxed agent-scope loads and then read the payload with sc1 loads.
// Every workgroup arrives at every stage, so the counters only ever count up (no reset,
// 64 bit); the phase number they have reached is carried from launch to launch in
// MegaSync::epoch.  Every spin is bounded: on timeout the kernel raises MegaSync::error and
// every wave leaves, so a protocol bug cannot hang the GPU.
//
// Arithmetic is the same device code the stand-alone kernels use (tile_dot, quantize_group16,
// headnorm_rope_vals, the attention trees of q3_numerics.h), so the logits are bit-identical
// to the multi-kernel path and to the oracle's tree order.
#include <cstdio>
#include <cstdlib>

#include "q3_device.hpp"
#include "q3_kernels.hpp"
#include "q3_tile.hpp"

namespace q3k {

#define Q3M_NWG 256          // one workgroup per CU of an MI355X
#define Q3M_WORKERS 6
#define Q3M_AUX 2
#define Q3M_WAVES (Q3M_WORKERS + Q3M_AUX)
#define Q3M_SPIN_LIMIT (1 << 21)
// keep the issue order of the weight pipeline exactly as written: hipcc would otherwise
// hoist later tiles' loads above earlier tiles' arithmetic and blow the register budget
#define PIN() do { asm volatile("" ::: "memory"); __builtin_amdgcn_sched_barrier(0); } while (0)

// ---- intra-workgroup signalling through LDS (no s_barrier: the two wave classes run
//      different programs and wait for different things) ---------------------------------
struct LdsCtl {
    unsigned raw;      // aux -> all : fp32 activation of the coming stage is in LDS
    unsigned codes;    // all -> all : quantised activation complete (8 per stage)
    unsigned out;      // workers -> aux : results of the stage are in out_stage (6 per stage)
    unsigned aux;      // aux1 -> aux0 : "my global stores are drained"
    unsigned pair;     // aux <-> aux : attention tile staging
    unsigned grid;     // aux0 -> aux1 : number of grid stages known to be complete
    unsigned abort;    // anybody -> all : leave the kernel
};

// control words are touched only through address-space-qualified pointers, so that they
// can never become flat accesses (a flat access ties up vmcnt AND lgkmcnt)
typedef __attribute__((address_space(3))) unsigned lds_u32;
typedef __attribute__((address_space(1))) unsigned long long g_u64;
typedef __attribute__((address_space(1))) unsigned g_u32;
#define LDSW(field) ((lds_u32*)(&lc->field))

__device__ __forceinline__ void lds_fence() { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); }
__device__ __forceinline__ void lds_add(lds_u32* p, unsigned v, int lane) {
    lds_fence();
    if (lane == 0) __hip_atomic_fetch_add(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    asm volatile("" ::: "memory");
}
// returns false when the kernel is being aborted
__device__ __forceinline__ bool lds_wait(lds_u32* p, unsigned target, lds_u32* abort) {
    for (int spin = 0; spin < Q3M_SPIN_LIMIT; spin++) {
        const unsigned v = __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        if ((int)(v - target) >= 0) {
            asm volatile("" ::: "memory");
            return true;
        }
        if (__hip_atomic_load(abort, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP)) return false;
        __builtin_amdgcn_s_sleep(1);
    }
    __hip_atomic_store(abort, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    return false;
}

// ---- grid-wide arrival counters -------------------------------------------------------
__device__ __forceinline__ void grid_arrive(MegaSync* s, int lane) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // this wave's write-through stores have left
    if (lane == 0) {
        __hip_atomic_fetch_add((g_u64*)&s->shard[blockIdx.x & 7][0], 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}
// every shard must have counted `per_shard_target` arrivals; one wave polls (lane i -> shard i)
__device__ __forceinline__ bool grid_wait(MegaSync* s, unsigned long long per_shard_target, int lane,
                                          lds_u32* abort) {
    for (int spin = 0; spin < Q3M_SPIN_LIMIT; spin++) {
        const unsigned long long v =
            __hip_atomic_load((g_u64*)&s->shard[lane & 7][0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (__all(v >= per_shard_target)) {
            asm volatile("" ::: "memory");
            return true;
        }
        if (__hip_atomic_load(abort, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP)) return false;
        __builtin_amdgcn_s_sleep(1);
    }
    if (lane == 0) {
        __hip_atomic_store((g_u32*)&s->error, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(abort, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    }
    return false;
}

// ---- write-through / cache-bypassing accesses to in-kernel hand-off data ----------------
__device__ __forceinline__ void st_sc1(float* p, float v) {
    __hip_atomic_store((__attribute__((address_space(1))) float*)p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void st_sc1_i(int* p, int v) {
    __hip_atomic_store((__attribute__((address_space(1))) int*)p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ v4i ld_sc1_16(__amdgpu_buffer_rsrc_t r, int byte_off) {
    return __builtin_amdgcn_raw_buffer_load_b128(r, byte_off, 0, 16 /* sc1 */);
}
__device__ __forceinline__ __amdgpu_buffer_rsrc_t rsrc_of(const void* p, size_t bytes) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p), 0, (int)bytes, 0x00020000);
}

// all-gather helper: `bytes` (multiple of 16) from global (sc1 loads, the counter-form hand-off
// of Guideline 16) into LDS, split over the aux waves; each workgroup starts at a different
// 1-KiB piece so that 256 readers do not march over the memory channels in lock step.
// (Measured alternative: ONE agent-scope acquire + plain loads -- the XCD's L2 then serves 31 of
// 32 readers -- was slower here, 4-6 us per gather instead of 1-3.)
__device__ __forceinline__ void gather_to_lds(const void* src, int bytes, char* dst, int asub, int lane, int rot) {
    const __amdgpu_buffer_rsrc_t r = rsrc_of(src, (size_t)bytes);
    const int pieces = (bytes + 1023) >> 10;
    const int start = rot % pieces;
    for (int p0 = asub; p0 < pieces; p0 += Q3M_AUX * 8) {
        v4i v[8];
#pragma unroll
        for (int k = 0; k < 8; k++) {
            int p = p0 + k * Q3M_AUX;
            const bool on = p < pieces;
            p = p + start;
            if (p >= pieces) p -= pieces;
            const int off = p * 1024 + lane * 16;
            v[k] = v4i{0, 0, 0, 0};
            if (on && off < bytes) v[k] = ld_sc1_16(r, off);
        }
#pragma unroll
        for (int k = 0; k < 8; k++) {
            int p = p0 + k * Q3M_AUX;
            const bool on = p < pieces;
            p = p + start;
            if (p >= pieces) p -= pieces;
            const int off = p * 1024 + lane * 16;
            if (on && off < bytes) *reinterpret_cast<v4i*>(dst + off) = v[k];
        }
    }
}

// quantise the 256-element blocks b = wave, wave+8, ... of y (fp32 in LDS; optionally
// y = nw*(s*x)) into codes/scales in LDS -- the activation prologue of the GEMV stages.
// Up to 6 blocks per wave, processed together so that their cross-lane max chains overlap.
__device__ __forceinline__ void quantize_blocks(const float* src, const float* nw, float sc, int n, int8_t* lq,
                                                float* ls, int wave, int lane) {
    asm volatile("" : "+v"(lane));
    constexpr int MAXB = 6;
    float4 y[MAXB];
#pragma unroll
    for (int k = 0; k < MAXB; k++) {
        const int i = (wave + k * Q3M_WAVES) * 256 + 4 * lane;
        y[k] = make_float4(0.f, 0.f, 0.f, 0.f);
        if (i < n) {
            const float4 v = *reinterpret_cast<const float4*>(src + i);
            if (nw) {
                const float4 g = *reinterpret_cast<const float4*>(nw + i);
                y[k].x = g.x * (sc * v.x);
                y[k].y = g.y * (sc * v.y);
                y[k].z = g.z * (sc * v.z);
                y[k].w = g.w * (sc * v.w);
            } else {
                y[k] = v;
            }
        }
    }
#pragma unroll
    for (int k = 0; k < MAXB; k++) {
        const int i = (wave + k * Q3M_WAVES) * 256 + 4 * lane;
        float scale;
        const int packed = quantize_group16(y[k], scale);
        if (i < n) {
            reinterpret_cast<int*>(lq)[i >> 2] = packed;
            if ((lane & 15) == 0) ls[i >> 6] = scale;
        }
    }
}
__device__ __forceinline__ float lds_sum256_sq(const float* x, int n, int lane) {
    asm volatile("" : "+v"(lane));
    float c0 = 0.f, c1 = 0.f, c2 = 0.f, c3 = 0.f;
    for (int i = 4 * lane; i < n; i += 256) {
        const float4 v = *reinterpret_cast<const float4*>(x + i);
        c0 = c0 + v.x * v.x;
        c1 = c1 + v.y * v.y;
        c2 = c2 + v.z * v.z;
        c3 = c3 + v.w * v.w;
    }
    return bfly64((c0 + c1) + (c2 + c3));
}

// a pointer field of the LDS layer table as a wave-uniform (SGPR) value
template <typename T>
__device__ __forceinline__ T* lds_ptr(const void* field) {
    const unsigned long long v = *reinterpret_cast<const unsigned long long*>(field);
    const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)v);
    const unsigned hi = __builtin_amdgcn_readfirstlane((unsigned)(v >> 32));
    return reinterpret_cast<T*>(((unsigned long long)hi << 32) | lo);
}

// Shape constants of one model family: wave-loads per row of the three GEMV widths and the
// tiling of each stage over the 6 workers of a workgroup (RW rows per workgroup, tiles of R
// rows, T tiles per worker; tile ti of a stage = rows [RW*b + R*ti, +R)).
struct Cfg4B {   // Qwen3-4B: dim 2560, P 4096, hidden 9728
    static constexpr int NJD = 3, NJP = 4, NJH = 10;
    static constexpr int RW_Q = 24, R_Q = 4, T_Q = 1;
    static constexpr int RW_W = 10, R_W = 2, T_W = 1;
    static constexpr int RW_G = 76, R_G = 4, T_G = 4;
    static constexpr int RW_D = 10, R_D = 1, T_D = 2;
    static constexpr int R_C = 4;
};

#define Q3M_ATT_HD 128

// LDS carve-up, as byte offsets from the start of the dynamic segment (all multiples of 16);
// each wave class rebuilds address-space-3 pointers from them, so that every LDS access is
// a ds_* instruction (never flat) no matter how the code is split into functions.
struct LdsMap {
    unsigned xl, hl, nwl, lq, ls, out_stage, Ks, Vs, lc, ltab;
};
#define LDS_PTR(T, off) (reinterpret_cast<T*>(smem + (off)))

__device__ __forceinline__ LdsMap make_ldsmap(unsigned base, int dim, int hid, int P) {
    LdsMap m;
    const int nmax = hid > P ? hid : P;
    unsigned o = base;
    m.xl = o; o += dim * 4;
    m.hl = o; o += hid * 4;
    m.nwl = o; o += dim * 4;
    m.lq = o; o += nmax;
    m.ls = o; o += ((nmax / 64 + 3) & ~3) * 4;
    m.out_stage = o; o += 128 * 4;
    m.Ks = o; o += 64 * Q3M_ATT_HD * 4;
    m.Vs = o; o += 64 * Q3M_ATT_HD * 4;
    m.lc = o; o += 64;
    m.ltab = o;
    return m;
}

template <class CFG>
__device__ __forceinline__ void worker_main(const Mega* __restrict__ mp, char* smem, int wave, int lane, int b) {
    const Mega& a = *mp;
    const int dim = a.dim, hid = a.hid, P = a.P, KVD = a.KVD;
    const LdsMap lm = make_ldsmap(0, dim, hid, P);
    float* xl = LDS_PTR(float, lm.xl);
    float* hl = LDS_PTR(float, lm.hl);
    float* nwl = LDS_PTR(float, lm.nwl);
    int8_t* lq = LDS_PTR(int8_t, lm.lq);
    float* ls = LDS_PTR(float, lm.ls);
    float* out_stage = LDS_PTR(float, lm.out_stage);
    LdsCtl* lc = LDS_PTR(LdsCtl, lm.lc);
    MegaLayer* ltab = LDS_PTR(MegaLayer, lm.ltab);
    lds_u32* abortp = LDSW(abort);
    const int L0 = a.l0, L1 = a.l1;
    unsigned t_raw = Q3M_AUX, t_codes = 0;
    {
        const int w = wave;
        // register sets: the tile being consumed plus the two behind it are live at any time
        Tile<CFG::R_Q, CFG::NJD> tq;
        Tile<CFG::R_W, CFG::NJP> tw;
        Tile<CFG::R_G, CFG::NJD> ta, tb;          // gate/up tiles alternate between these; so do the classifier's
        Tile<CFG::R_D, CFG::NJH> td0, td1;
        static_assert(CFG::R_C == CFG::R_G, "classifier tiles reuse the gate/up registers");

        const int dq = P + 2 * KVD;
#define ROW_Q (CFG::RW_Q * b + CFG::R_Q * w)
#define ROW_W (CFG::RW_W * b + CFG::R_W * w)
#define ROW_G(t_) (CFG::RW_G * b + CFG::R_G * ((t_) * Q3M_WORKERS + w))
#define ROW_D(t_) (CFG::RW_D * b + CFG::R_D * ((t_) * Q3M_WORKERS + w))
        // A tile whose rows lie beyond its workgroup's share (or the matrix) reads zeros: the
        // descriptor ends with the share.  Pointers come from the LDS copy of the layer table.
        auto share = [&](int d, int rw) {
            const int end = rw * (b + 1);
            return end > d ? d : end;
        };
        auto view_qkv = [&](int l, bool on) {
            const MegaLayer* ly = ltab + ((on ? l : L0) - L0);
            return make_wview(lds_ptr<const int8_t>(&ly->qkv_q), lds_ptr<const float>(&ly->qkv_s), on ? share(dq, CFG::RW_Q) : 0, dim);
        };
        auto view_wo = [&](int l, bool on) {
            const MegaLayer* ly = ltab + ((on ? l : L0) - L0);
            return make_wview(lds_ptr<const int8_t>(&ly->wo_q), lds_ptr<const float>(&ly->wo_s), on ? share(dim, CFG::RW_W) : 0, P);
        };
        auto view_gu = [&](int l) {
            const MegaLayer* ly = ltab + (l - L0);
            return make_wview(lds_ptr<const int8_t>(&ly->gu_q), lds_ptr<const float>(&ly->gu_s), share(2 * hid, CFG::RW_G), dim);
        };
        auto view_dn = [&](int l) {
            const MegaLayer* ly = ltab + (l - L0);
            return make_wview(lds_ptr<const int8_t>(&ly->dn_q), lds_ptr<const float>(&ly->dn_s), share(dim, CFG::RW_D), hid);
        };
        const int ncls_rows = a.cls_q ? (a.V + Q3M_NWG - 1) / Q3M_NWG : 0;
        const int cls_row0 = ncls_rows * b;
        const int cls_end = (cls_row0 + ncls_rows < a.V) ? cls_row0 + ncls_rows : a.V;
        const int cls_rounds = (ncls_rows + CFG::R_C * Q3M_WORKERS - 1) / (CFG::R_C * Q3M_WORKERS);
        // `on` false gives an empty descriptor: every load of the tile returns zero without
        // touching memory.  Tiles are always (re)issued unconditionally -- a conditional issue
        // would keep the previous contents of the registers alive across the whole layer.
        auto view_cls = [&](bool on) { return make_wview(a.cls_q, a.cls_s, (on && cls_end > 0) ? cls_end : 0, dim); };
#define ROW_C(r_) (cls_row0 + CFG::R_C * ((r_) * Q3M_WORKERS + w))

        // wait for the quantised activation of the coming stage; in quantising stages every
        // wave converts its own blocks first
        auto stage_in_norm = [&](int n) -> bool {     // rmsnorm + quantise of xl with nwl
            if (!lds_wait(LDSW(raw), t_raw, abortp)) return false;
            t_raw += Q3M_AUX;
            const float ss = lds_sum256_sq(xl, n, lane);
            const float sc = 1.0f / sqrtf(ss / (float)n + 1e-6f);
            quantize_blocks(xl, nwl, sc, n, lq, ls, wave, lane);
            lds_add(LDSW(codes), 1, lane);
            t_codes += Q3M_WAVES;
            return lds_wait(LDSW(codes), t_codes, abortp);
        };
        auto stage_in_f32 = [&](const float* src, int n) -> bool {
            if (!lds_wait(LDSW(raw), t_raw, abortp)) return false;
            t_raw += Q3M_AUX;
            quantize_blocks(src, nullptr, 0.0f, n, lq, ls, wave, lane);
            lds_add(LDSW(codes), 1, lane);
            t_codes += Q3M_WAVES;
            return lds_wait(LDSW(codes), t_codes, abortp);
        };
        auto stage_in_q8 = [&]() -> bool {            // codes written by the aux waves
            t_codes += Q3M_WAVES;
            return lds_wait(LDSW(codes), t_codes, abortp);
        };

        static_assert(CFG::R_C == CFG::R_Q, "classifier round 0 is prefetched into the QKV tile registers");
        {
            // the first tile of whatever comes first: QKV of layer L0, or classifier round 0
            const bool any = L1 > L0;
            tile_issue(tq, any ? view_qkv(L0, true) : view_cls(true), any ? ROW_Q : ROW_C(0), lane);
        }

#define GU_TILE(T_, IDX_)                                                                     \
        {                                                                                     \
            float acc[CFG::R_G];                                                              \
            tile_dot(T_, dim, lane, lq, ls, acc);                                             \
            if (lane == 0) {                                                                  \
                _Pragma("unroll") for (int r = 0; r < CFG::R_G; r += 2)                       \
                    out_stage[(CFG::R_G / 2) * ((IDX_) * Q3M_WORKERS + w) + (r >> 1)] =       \
                        swiglu_pair(acc[r], acc[r + 1]);                                      \
            }                                                                                 \
        }
        // One tile beyond the one being consumed is in flight per worker (6 x ~10 KB per CU):
        // enough to keep this CU's share of HBM busy, while a deeper queue would only sit in
        // the CU's memory pipeline in front of the aux waves' latency-critical hand-off loads.
        for (int l = L0; l < L1; l++) {
            // ---- QKV (activation: rmsnorm(x)) ----
            tile_issue(tw, view_wo(l, true), ROW_W, lane);
            PIN();
            if (!stage_in_norm(dim)) return;
            {
                float acc[CFG::R_Q];
                tile_dot(tq, dim, lane, lq, ls, acc);
                if (lane == 0) {
#pragma unroll
                    for (int r = 0; r < CFG::R_Q; r++) out_stage[CFG::R_Q * w + r] = acc[r];
                }
            }
            lds_add(LDSW(out), 1, lane);
            PIN();
            // ---- Wo (activation: attention output codes, put into LDS by the aux waves) ----
            tile_issue(ta, view_gu(l), ROW_G(0), lane);
            PIN();
            if (!stage_in_q8()) return;
            {
                float acc[CFG::R_W];
                tile_dot(tw, P, lane, lq, ls, acc);
                if (lane == 0) {
#pragma unroll
                    for (int r = 0; r < CFG::R_W; r++) out_stage[CFG::R_W * w + r] = acc[r];
                }
            }
            lds_add(LDSW(out), 1, lane);
            PIN();
            // ---- gate/up (activation: rmsnorm(x)) + SwiGLU ----
            tile_issue(tb, view_gu(l), ROW_G(1), lane);
            PIN();
            if (!stage_in_norm(dim)) return;
            GU_TILE(ta, 0)
            PIN();
            tile_issue(ta, view_gu(l), ROW_G(2), lane);
            PIN();
            GU_TILE(tb, 1)
            PIN();
            tile_issue(tb, view_gu(l), ROW_G(3), lane);
            PIN();
            GU_TILE(ta, 2)
            PIN();
            tile_issue(td0, view_dn(l), ROW_D(0), lane);
            PIN();
            GU_TILE(tb, 3)
            PIN();
            lds_add(LDSW(out), 1, lane);
            PIN();
            // ---- down (activation: quantised h) ----
            const bool more = l + 1 < L1;
            tile_issue(td1, view_dn(l), ROW_D(1), lane);
            PIN();
            if (!stage_in_f32(hl, hid)) return;
            {
                float acc[CFG::R_D];
                tile_dot(td0, hid, lane, lq, ls, acc);
                if (lane == 0) {
#pragma unroll
                    for (int r = 0; r < CFG::R_D; r++) out_stage[CFG::R_D * w + r] = acc[r];
                }
            }
            PIN();
            // next layer's QKV tile, or (last layer) the classifier's first tile: same registers
            tile_issue(tq, more ? view_qkv(l + 1, true) : view_cls(true), more ? ROW_Q : ROW_C(0), lane);
            PIN();
            {
                float acc[CFG::R_D];
                tile_dot(td1, hid, lane, lq, ls, acc);
                if (lane == 0) {
#pragma unroll
                    for (int r = 0; r < CFG::R_D; r++) out_stage[CFG::R_D * (Q3M_WORKERS + w) + r] = acc[r];
                }
            }
            lds_add(LDSW(out), 1, lane);
            PIN();
        }
#undef GU_TILE

        // ---- classifier: final rmsnorm + quantise + the lm_head rows of this workgroup ----
        if (a.cls_q) {
            if (!stage_in_norm(dim)) return;
#define CLS_TILE(T_, RND_)                                                                    \
            {                                                                                 \
                float acc[CFG::R_C];                                                          \
                tile_dot(T_, dim, lane, lq, ls, acc);                                         \
                if (lane == 0) {                                                              \
                    _Pragma("unroll") for (int r = 0; r < CFG::R_C; r++) {                    \
                        const int row = ROW_C(RND_) + r;                                      \
                        if (row < cls_end) a.logits[row] = acc[r];                            \
                    }                                                                         \
                }                                                                             \
            }
            // round 0 is already in flight in tq; rounds 1.. alternate between ta and tb
            tile_issue(ta, view_cls(1 < cls_rounds), ROW_C(1), lane);
            PIN();
            CLS_TILE(tq, 0)
            PIN();
            tile_issue(tb, view_cls(2 < cls_rounds), ROW_C(2), lane);
            PIN();
            for (int rnd = 1; rnd < cls_rounds; rnd += 2) {
                CLS_TILE(ta, rnd)
                PIN();
                tile_issue(ta, view_cls(rnd + 2 < cls_rounds), ROW_C(rnd + 2), lane);
                PIN();
                if (rnd + 1 < cls_rounds) {
                    CLS_TILE(tb, rnd + 1)
                    PIN();
                    tile_issue(tb, view_cls(rnd + 3 < cls_rounds), ROW_C(rnd + 3), lane);
                    PIN();
                }
            }
#undef CLS_TILE
        }
        return;
    }
}

template <class CFG>
__device__ __forceinline__ void aux_main(const Mega* __restrict__ mp, char* smem, int wave, int lane, int b) {
    const Mega& a = *mp;
    const int dim = a.dim, hid = a.hid, P = a.P, KVD = a.KVD;
    const LdsMap lm = make_ldsmap(0, dim, hid, P);
    float* xl = LDS_PTR(float, lm.xl);
    float* hl = LDS_PTR(float, lm.hl);
    float* nwl = LDS_PTR(float, lm.nwl);
    int8_t* lq = LDS_PTR(int8_t, lm.lq);
    float* ls = LDS_PTR(float, lm.ls);
    float* out_stage = LDS_PTR(float, lm.out_stage);
    float* Ks = LDS_PTR(float, lm.Ks);
    float* Vs = LDS_PTR(float, lm.Vs);
    LdsCtl* lc = LDS_PTR(LdsCtl, lm.lc);
    lds_u32* abortp = LDSW(abort);
    const int pos = a.ctl->pos;
    const int token = a.ctl->token;
    const unsigned long long epoch0 = a.sync->epoch;   // stages completed before this launch
    const int L0 = a.l0, L1 = a.l1;
    const int asub = wave - Q3M_WORKERS;               // 0 / 1
    unsigned t_raw = Q3M_AUX, t_codes = 0, t_out = 0;
    (void)t_codes;
    unsigned long long stage = 0;                       // grid stages completed in this launch
    int stamp_i = 0;
#define MSTAMP() do { if (a.stamps && b == 0 && asub == 0 && lane == 0 && l == L0 + 1 && stamp_i < 60) a.stamps[stamp_i++] = __builtin_amdgcn_s_memrealtime(); } while (0)

    // ---- stage 0: the residual entering the first layer ----------------------------------
        if (a.emb_q) {
            // x = q*s of the embedding row (reference forward.c:237 via model.c:201-206)
            const size_t base = (size_t)token * dim;
            for (int i = asub * 64 + lane; i < dim; i += Q3M_AUX * 64) {
                xl[i] = (float)a.emb_q[base + i] * a.emb_s[(base + i) >> 6];
            }
        } else {
            // later pipeline stage: the residual was delivered to a.x before this launch
            for (int i = asub * 64 + lane; i < dim / 4; i += Q3M_AUX * 64) {
                reinterpret_cast<float4*>(xl)[i] = reinterpret_cast<const float4*>(a.x)[i];
            }
        }
        {
            const float* nw = (L1 > L0) ? a.layers[L0].att_nw : a.out_nw;
            if (nw) {
                for (int i = asub * 64 + lane; i < dim / 4; i += Q3M_AUX * 64) {
                    reinterpret_cast<float4*>(nwl)[i] = reinterpret_cast<const float4*>(nw)[i];
                }
            }
        }
        lds_add(LDSW(raw), 1, lane);

    {
        const int dq = P + 2 * KVD;
        const int kv_mul = a.H / a.KV;
        const int T = pos + 1;
        const int nchunks = (T + Q3_ATT_CHUNK - 1) / Q3_ATT_CHUNK;
        unsigned t_aux = 0, t_pair = 0;

        // complete one grid stage: aux1 drains and tells aux0, aux0 drains, arrives, waits
        auto finish_stage = [&]() -> bool {
            if (asub == 1) {
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                lds_add(LDSW(aux), 1, lane);
            }
            t_aux += 1;
            stage += 1;
            if (asub == 0) {
                if (!lds_wait(LDSW(aux), t_aux, abortp)) return false;
                grid_arrive(a.sync, lane);
                // ONE wave per workgroup polls the counters; its partner waits on an LDS word
                if (!grid_wait(a.sync, (epoch0 + stage) * (Q3M_NWG / 8), lane, abortp)) return false;
                lds_add(LDSW(grid), 1, lane);
            } else {
                if (!lds_wait(LDSW(grid), (unsigned)stage, abortp)) return false;
            }
            return true;
        };
        // quantise this wave's blocks together with the workers
        auto join_norm = [&](int n) -> bool {
            const float ss = lds_sum256_sq(xl, n, lane);
            const float sc = 1.0f / sqrtf(ss / (float)n + 1e-6f);
            quantize_blocks(xl, nwl, sc, n, lq, ls, wave, lane);
            lds_add(LDSW(codes), 1, lane);
            t_codes += Q3M_WAVES;
            return true;
        };

        for (int l = L0; l < L1; l++) {
            const MegaLayer& ly = a.layers[l];
            // ================= stage A: QKV =================
            if (!lds_wait(LDSW(raw), t_raw, abortp)) return;     // both halves of x (and of the norm weight)
            t_raw += Q3M_AUX;
            MSTAMP();   /* 0: x ready */
            join_norm(dim);
            MSTAMP();   /* 1: my blocks quantised */
            t_out += Q3M_WORKERS;
            if (!lds_wait(LDSW(out), t_out, abortp)) return;
            MSTAMP();   /* 2: workers done QKV */
            if (asub == 0 && lane < CFG::RW_Q) {
                const int row = CFG::RW_Q * b + lane;
                if (row < dq) st_sc1(a.qkv + row, out_stage[lane]);
            }
            if (!finish_stage()) return;
            MSTAMP();   /* 3: grid A done */

            // ================= stage B: attention =================
            // single chunk: item = (kv group, half of its query heads), one head per aux wave;
            // several chunks: item = (kv group, chunk), the group's heads split over the aux waves
            {
                const int hpi = nchunks == 1 ? (kv_mul >= 2 ? kv_mul / 2 : 1) : kv_mul;   // heads per item
                const int ipg = nchunks == 1 ? kv_mul / hpi : nchunks;                     // items per group
                const int nitems = a.KV * ipg;
                for (int item = b; item < nitems; item += Q3M_NWG) {
                    const int g = item / ipg, sub = item - g * ipg;
                    const int c = nchunks == 1 ? 0 : sub;
                    const int h0 = nchunks == 1 ? sub * hpi : 0;
                    const bool owner = (c == nchunks - 1) && (nchunks > 1 || sub == 0);
                    constexpr int HD = Q3M_ATT_HD, L4 = HD / 4, CH = Q3_ATT_CHUNK;
                    const size_t cbase = ((size_t)g * a.seq_pad) * HD;
                    const int half = lane >> 5, li = lane & 31;
                    const int t0 = c * CH;
                    const int Tc = (T - t0 < CH) ? T - t0 : CH;
                    // K/V tile: 128 threads stage 64 x 32 float4 per tile
                    const int atid = asub * 64 + lane;
                    float4 kraw = make_float4(0.f, 0.f, 0.f, 0.f), vraw = kraw, kg = kraw, qg = kraw, ca = kraw, cb = kraw;
                    if (lane < L4) {
                        kraw = __builtin_bit_cast(float4, ld_sc1_16(rsrc_of(a.qkv, (size_t)dq * 4), (P + g * HD + 4 * lane) * 4));
                        vraw = __builtin_bit_cast(float4, ld_sc1_16(rsrc_of(a.qkv, (size_t)dq * 4), (P + KVD + g * HD + 4 * lane) * 4));
                        kg = *reinterpret_cast<const float4*>(ly.knw + 4 * lane);
                        qg = *reinterpret_cast<const float4*>(ly.qnw + 4 * lane);
                    }
                    rope_slices<HD>(a.rope + (size_t)pos * HD, lane, ca, cb);
                    for (int idx = atid; idx < CH * L4; idx += Q3M_AUX * 64) {
                        const int t = idx / L4, l4 = idx - t * L4;
                        if (t < Tc && t0 + t != pos) {
                            *reinterpret_cast<float4*>(Ks + t * HD + 4 * l4) =
                                *reinterpret_cast<const float4*>(ly.kc + cbase + (size_t)(t0 + t) * HD + 4 * l4);
                            *reinterpret_cast<float4*>(Vs + t * HD + 4 * l4) =
                                *reinterpret_cast<const float4*>(ly.vc + cbase + (size_t)(t0 + t) * HD + 4 * l4);
                        }
                    }
                    const float4 kcur = headnorm_rope_vals<HD>(kraw, kg, ca, cb, lane);
                    if (asub == 0 && lane < L4 && pos >= t0 && pos < t0 + CH) {
                        *reinterpret_cast<float4*>(Ks + (pos - t0) * HD + 4 * lane) = kcur;
                        *reinterpret_cast<float4*>(Vs + (pos - t0) * HD + 4 * lane) = vraw;
                        if (owner) {
                            *reinterpret_cast<float4*>(ly.kc + cbase + (size_t)pos * HD + 4 * lane) = kcur;
                            *reinterpret_cast<float4*>(ly.vc + cbase + (size_t)pos * HD + 4 * lane) = vraw;
                        }
                    }
                    lds_add(LDSW(pair), 1, lane);
                    t_pair += Q3M_AUX;
                    if (!lds_wait(LDSW(pair), t_pair, abortp)) return;

                    const float root = sqrtf((float)HD);
                    const int nsteps = (Tc + 1) >> 1;
                    const bool act = li < L4;
                    for (int i = h0 + asub; i < h0 + hpi; i += Q3M_AUX) {
                        const int h = g * kv_mul + i;
                        float4 q4 = make_float4(0.f, 0.f, 0.f, 0.f);
                        if (lane < L4) q4 = __builtin_bit_cast(float4, ld_sc1_16(rsrc_of(a.qkv, (size_t)dq * 4), (h * HD + 4 * lane) * 4));
                        q4 = headnorm_rope_vals<HD>(q4, qg, ca, cb, lane);
                        {
                            const float ox = lane_xor_f<32>(q4.x), oy = lane_xor_f<32>(q4.y);
                            const float oz = lane_xor_f<32>(q4.z), ow = lane_xor_f<32>(q4.w);
                            if (half) q4 = make_float4(ox, oy, oz, ow);
                        }
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
                                        const float4 k4 = *reinterpret_cast<const float4*>(Ks + t * HD + 4 * li);
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
                        const float dot = transpose_sum32(cpart, li);
                        const bool valid = (2 * li + half) < Tc;
                        const float mys = valid ? dot / root : -3.0e38f;
                        const float m = wave_max(mys);
                        const float e = valid ? q3_expf(mys - m) : 0.0f;
                        const float lsum = bfly64(e);
                        float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
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
                                    if (act) v4 = *reinterpret_cast<const float4*>(Vs + t * HD + 4 * li);
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
                        if (nchunks > 1) {
                            if (lane < L4) {
                                float* pp = a.part + ((size_t)h * a.max_chunks + c) * (HD + 2);
                                st_sc1(pp + 4 * lane, o.x);
                                st_sc1(pp + 4 * lane + 1, o.y);
                                st_sc1(pp + 4 * lane + 2, o.z);
                                st_sc1(pp + 4 * lane + 3, o.w);
                                if (lane == 0) {
                                    st_sc1(pp + HD, m);
                                    st_sc1(pp + HD + 1, lsum);
                                }
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
                                st_sc1_i(reinterpret_cast<int*>(a.att_q) + ((h * HD + 4 * lane) >> 2), packed);
                                if ((lane & 15) == 0) st_sc1(a.att_s + ((h * HD + 4 * lane) >> 6), scale);
                            }
                        }
                    }
                    // both waves are done with the tiles before the next item restages them
                    lds_add(LDSW(pair), 1, lane);
                    t_pair += Q3M_AUX;
                    if (!lds_wait(LDSW(pair), t_pair, abortp)) return;
                }
            }
            MSTAMP();   /* 4: attention items done */
            if (!finish_stage()) return;
            MSTAMP();   /* 5: grid B done */
            if (nchunks > 1) {
                // ---- stage B2: merge the chunk partials of head h = b, b+256, ... (aux0) ----
                constexpr int HD = Q3M_ATT_HD, L4 = HD / 4;
                if (asub == 0) {
                    for (int h = b; h < a.H; h += Q3M_NWG) {
                        const float* base = a.part + (size_t)h * a.max_chunks * (HD + 2);
                        const __amdgpu_buffer_rsrc_t rp = rsrc_of(base, (size_t)a.max_chunks * (HD + 2) * 4);
                        auto ldf = [&](int idx) {
                            return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rp, idx * 4, 0, 16));
                        };
                        float M = ldf(HD);
                        for (int c = 1; c < nchunks; c++) M = fmaxf(M, ldf(c * (HD + 2) + HD));
                        float Lsum = 0.0f;
                        float4 A = make_float4(0.f, 0.f, 0.f, 0.f);
                        for (int c = 0; c < nchunks; c++) {
                            const int o = c * (HD + 2);
                            const float wgt = q3_expf(ldf(o + HD) - M);
                            Lsum = Lsum + wgt * ldf(o + HD + 1);
                            if (lane < L4) {
                                A.x = A.x + wgt * ldf(o + 4 * lane);
                                A.y = A.y + wgt * ldf(o + 4 * lane + 1);
                                A.z = A.z + wgt * ldf(o + 4 * lane + 2);
                                A.w = A.w + wgt * ldf(o + 4 * lane + 3);
                            }
                        }
                        float4 y = make_float4(0.f, 0.f, 0.f, 0.f);
                        if (lane < L4) {
                            y.x = A.x / Lsum;
                            y.y = A.y / Lsum;
                            y.z = A.z / Lsum;
                            y.w = A.w / Lsum;
                        }
                        float scale;
                        const int packed = quantize_group16(y, scale);
                        if (lane < L4) {
                            st_sc1_i(reinterpret_cast<int*>(a.att_q) + ((h * HD + 4 * lane) >> 2), packed);
                            if ((lane & 15) == 0) st_sc1(a.att_s + ((h * HD + 4 * lane) >> 6), scale);
                        }
                    }
                }
                if (!finish_stage()) return;
            }

            // ================= stage C: Wo + residual =================
            // attention output codes + scales straight into the activation buffers
            gather_to_lds(a.att_q, P, reinterpret_cast<char*>(lq), asub, lane, b);
            gather_to_lds(a.att_s, (P / 64) * 4, reinterpret_cast<char*>(ls), asub, lane, 0);
            // norm weight of the gate/up stage, needed only after the next gather
            for (int i = asub * 64 + lane; i < dim / 4; i += Q3M_AUX * 64) {
                reinterpret_cast<float4*>(nwl)[i] = reinterpret_cast<const float4*>(ly.ffn_nw)[i];
            }
            lds_add(LDSW(codes), Q3M_WAVES / Q3M_AUX, lane);
            MSTAMP();   /* 6: att codes gathered */
            t_codes += Q3M_WAVES;
            t_out += Q3M_WORKERS;
            if (!lds_wait(LDSW(out), t_out, abortp)) return;
            MSTAMP();   /* 7: workers done Wo */
            if (asub == 0 && lane < CFG::RW_W) {
                const int row = CFG::RW_W * b + lane;
                if (row < dim) st_sc1(a.x + row, xl[row] + out_stage[lane]);
            }
            if (!finish_stage()) return;
            MSTAMP();   /* 8: grid C done */
            gather_to_lds(a.x, dim * 4, reinterpret_cast<char*>(xl), asub, lane, b);
            lds_add(LDSW(raw), 1, lane);
            MSTAMP();   /* 9: x gathered */

            // ================= stage D: gate/up + SwiGLU =================
            if (!lds_wait(LDSW(raw), t_raw, abortp)) return;     // the other aux wave's half of x
            t_raw += Q3M_AUX;
            join_norm(dim);
            MSTAMP();   /* 10: quantised */
            t_out += Q3M_WORKERS;
            if (!lds_wait(LDSW(out), t_out, abortp)) return;
            MSTAMP();   /* 11: workers done gate/up */
            if (asub == 0 && lane < CFG::RW_G / 2) {
                const int i = (CFG::RW_G / 2) * b + lane;
                if (i < hid) st_sc1(a.h + i, out_stage[lane]);
            }
            if (!finish_stage()) return;
            MSTAMP();   /* 12: grid D done */
            gather_to_lds(a.h, hid * 4, reinterpret_cast<char*>(hl), asub, lane, b);
            lds_add(LDSW(raw), 1, lane);
            MSTAMP();   /* 13: h gathered */

            // ================= stage E: down + residual =================
            if (!lds_wait(LDSW(raw), t_raw, abortp)) return;
            t_raw += Q3M_AUX;
            quantize_blocks(hl, nullptr, 0.0f, hid, lq, ls, wave, lane);
            lds_add(LDSW(codes), 1, lane);
            t_codes += Q3M_WAVES;
            MSTAMP();   /* 14: quantised h */
            t_out += Q3M_WORKERS;
            if (!lds_wait(LDSW(out), t_out, abortp)) return;
            MSTAMP();   /* 15: workers done down */
            if (asub == 0 && lane < CFG::RW_D) {
                const int row = CFG::RW_D * b + lane;
                if (row < dim) st_sc1(a.x + row, xl[row] + out_stage[lane]);
            }
            if (!finish_stage()) return;
            MSTAMP();   /* 16: grid E done */
            // norm weight of the next consumer of x, then x itself
            {
                const float* nw = (l + 1 < L1) ? a.layers[l + 1].att_nw : a.out_nw;
                if (nw) {
                    for (int i = asub * 64 + lane; i < dim / 4; i += Q3M_AUX * 64) {
                        reinterpret_cast<float4*>(nwl)[i] = reinterpret_cast<const float4*>(nw)[i];
                    }
                }
            }
            gather_to_lds(a.x, dim * 4, reinterpret_cast<char*>(xl), asub, lane, b);
            lds_add(LDSW(raw), 1, lane);
        }

        // ================= classifier =================
        if (a.cls_q) {
            if (!lds_wait(LDSW(raw), t_raw, abortp)) return;
            t_raw += Q3M_AUX;
            join_norm(dim);
        }
        // the launch is over for the grid counters: remember how far they have counted
        if (b == 0 && asub == 0 && lane == 0) a.sync->epoch = epoch0 + stage;
    }
}

template <class CFG>
__global__ __launch_bounds__(Q3M_WAVES * 64) void k_step(const Mega* __restrict__ mp) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const LdsMap lm = make_ldsmap(0, mp->dim, mp->hid, mp->P);       // offsets relative to smem here
    LdsCtl* lc = reinterpret_cast<LdsCtl*>(smem + lm.lc);
    if (tid == 0) {
        lc->raw = 0; lc->codes = 0; lc->out = 0; lc->aux = 0; lc->pair = 0; lc->grid = 0; lc->abort = 0;
    }
    {
        // the layer table of this launch, copied to LDS once: workers must not touch global
        // memory for anything but weight tiles, not even for pointers
        const int nwords = (mp->l1 - mp->l0) * (int)(sizeof(MegaLayer) / 8);
        const unsigned long long* src = reinterpret_cast<const unsigned long long*>(mp->layers + mp->l0);
        unsigned long long* dst = reinterpret_cast<unsigned long long*>(smem + lm.ltab);
        for (int i = tid; i < nwords; i += Q3M_WAVES * 64) dst[i] = src[i];
    }
    __syncthreads();       // the only s_barrier of the kernel: before the wave classes diverge
    if (wave < Q3M_WORKERS) worker_main<CFG>(mp, smem, wave, lane, blockIdx.x);
    else aux_main<CFG>(mp, smem, wave, lane, blockIdx.x);
}


size_t step_lds_bytes(const Mega& m) {
    const int nmax = m.hid > m.P ? m.hid : m.P;
    return (size_t)(m.dim + m.hid + m.dim) * 4 + (size_t)nmax + (size_t)((nmax / 64 + 3) & ~3) * 4 + 128 * 4
           + 2 * 64 * Q3M_ATT_HD * 4 + 64 + (size_t)(m.l1 - m.l0) * sizeof(MegaLayer) + 64;
}

// `dev` = device copy of `host` (the kernel reads its parameters from memory on demand)
void step(const Mega* dev, const Mega& host, hipStream_t st) {
    static bool attr_set = false;
    if (!attr_set) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k_step<Cfg4B>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        attr_set = true;
    }
    hipLaunchKernelGGL(k_step<Cfg4B>, dim3(Q3M_NWG), dim3(Q3M_WAVES * 64), step_lds_bytes(host), st, dev);
}

// Which model shapes the persistent kernel is compiled for.
bool mega_supported(int dim, int hid, int H, int KV, int hd, int seq_pad) {
    (void)seq_pad;
    return dim == 2560 && hid == 9728 && H == 32 && KV == 8 && hd == 128;
}

}  // namespace q3k
