// q3_engine.hip -- the weight-streaming engine: the GEMV stages between two attention
// stages (Wo + residual, gate/up + SwiGLU, down + residual, and the next layer's
// Wq|Wk|Wv: reference src/forward.c:292-338 and 254-262) as ONE launch per layer.
//
// At batch 1 a stage is ~2-8 us of HBM transfer behind a grid-wide dependency (every output
// row needs the whole activation vector).  As separate launches each of them pays the kernel
// boundary, the ramp of its weight stream and its own tail, and HBM idles through all of it.
// Here one workgroup per CU stays resident across the four stages and the weight stream is
// decoupled from the dependency chain:
//
//   * wave 0 is the LOADER: it walks the static list of 16-KiB slots this workgroup will
//     need -- a few rows of codes plus their scales -- and moves them from HBM into an
//     8-slot ring in LDS with LDS-DMA (buffer_load_dwordx4 ... lds, nt), two slots in
//     flight, never looking at the dependency chain: while the consumers wait for a
//     hand-off, the next stage's weights keep arriving (up to 128 KiB ahead).
//   * waves 1..15 are CONSUMERS: per stage they fetch the activation vector the other
//     workgroups are publishing, run the stage's prologue (rmsnorm / q8_quantize, the
//     arithmetic of q3_gemv.hip), keep the quantised activation in registers, and turn
//     the ring's rows into dot products (same int32 group dots, same fp32 order as
//     tile_dot); a row's result goes straight to global memory.
//
// Hand-off between the workgroups: the data is the flag.  Every vector that crosses
// workgroups (x after Wo, h, x after down) lives in device memory that holds a SENTINEL (a
// NaN no arithmetic here produces) until its producer stores the value, write-through (sc1,
// one 4-byte store per value, which is single-copy atomic).  A consumer wave polls its
// 256-value block with sc1 loads until no lane sees the sentinel -- one memory round trip
// after the last producer's store, no counter, no second trip for the payload
// (cdna_hip_programming.md Guideline 16, R2 "the data IS the flag", with the value's own
// bits as the tag).  The vectors are double-buffered by launch parity; a launch poisons the
// set the NEXT launch will use, so nobody ever polls a buffer that still holds last time's
// values.  Every poll is bounded by the 100 MHz clock; a give-up raises *error (pinned host
// word), sets a sticky device flag and lets the launch drain.
//
// Wave classes never meet at an s_barrier after the first one (they run different
// programs): the loader publishes `landed` (slots complete, in stream order), every consumer
// publishes the first slot it may still read, both as single-writer LDS words; the
// consumers meet each other at an LDS counter.
#include <cstdio>
#include <cstdlib>

#include "q3_device.hpp"
#include "q3_kernels.hpp"
#include "q3_tile.hpp"

namespace q3k {

#define ENG_NWG 256
#define ENG_WAVES 16
#define ENG_CW (ENG_WAVES - 1)          // consumer waves
#define ENG_INFLIGHT 2                  // slots in flight beyond the one being published
#define ENG_TIMEOUT_TICKS 200000000ull  // 2 s of the 100 MHz clock
#define ENG_LDS_TOTAL 163840

typedef __attribute__((address_space(3))) unsigned lds_u32;
typedef __attribute__((address_space(3))) char lds_char;
typedef __attribute__((address_space(1))) unsigned ge_u32;
typedef __attribute__((address_space(1))) unsigned long long ge_u64;

#ifdef Q3_ENG_STAMPS
// diagnostic build: stamps[64 * workgroup + slot] <- s_memrealtime.  [0,16) consumer 0, [16,32) consumer 14, [32,64) loader (issue of stream slot s)
#define ESTAMP(a_, base_, i_) do { if ((a_).stamps && lane == 0) (a_).stamps[64 * blockIdx.x + (base_) + (i_)] = __builtin_amdgcn_s_memrealtime(); } while (0)
#else
#define ESTAMP(a_, base_, i_) do {} while (0)
#endif
#define CSTAMP(i_) do { if (cw == 0) ESTAMP(a, 0, i_); else if (cw == ENG_CW - 1) ESTAMP(a, 16, i_); } while (0)

// control words in LDS
struct EngCtl {
    unsigned landed;          // loader -> consumers: slots complete (stream order)
    unsigned csync;           // consumer <-> consumer: arrival counter (monotonic)
    unsigned abort;           // anybody -> all: leave
    unsigned pad0;
    unsigned prog[16];        // consumer cw -> loader: first slot this wave may still read
    unsigned hold[4];         // consumers -> loader: hold[t] = waves polling for the activation of stage t right now
    unsigned pad[8];
};

// ---- shape of the streamed stages (Qwen3-4B) ------------------------------------------
// A SLOT is RS rows of one matrix: their codes (rows*n bytes, in 1-KiB DMA pieces) followed, at the
// next 1-KiB boundary, by their scales.  The ring is a byte ring: slot s sits at TAB.pos[s]; it
// may be (re)filled once every consumer is past slot TAB.need[s], the youngest earlier slot that
// overlaps its bytes.  Both tables are compile-time constants of the launch's slot list.
constexpr int eng_pieces(int rows, int n) { return (rows * n + 1023) / 1024; }
struct Cfg4B {
    static constexpr int DIM = 2560, HID = 9728, P = 4096, QKV = 6144;
    static constexpr int NJ_X = 3, NJ_P = 4, NJ_H = 10;                 // wave-loads per row
    static constexpr int RW_W = 10, RS_W = 3, NS_W = 4;                 // Wo
    static constexpr int RW_G = 76, RS_G = 6, NS_G = 13;                // gate/up (rows interleaved)
    static constexpr int RW_D = 10, RS_D = 1, NS_D = 10;                // down
    static constexpr int RW_Q = 24, RS_Q = 6, NS_Q = 4;                 // Wq|Wk|Wv of the next layer
    static constexpr int S0_W = 0, S0_G = NS_W, S0_D = S0_G + NS_G, S0_Q = S0_D + NS_D, S_END = S0_Q + NS_Q;
    static constexpr int SOFF_W = eng_pieces(RS_W, P) * 1024, SOFF_G = eng_pieces(RS_G, DIM) * 1024;
    static constexpr int SOFF_D = eng_pieces(RS_D, HID) * 1024, SOFF_Q = eng_pieces(RS_Q, DIM) * 1024;
    static constexpr int SZ_W = SOFF_W + 1024, SZ_G = SOFF_G + 1024, SZ_D = SOFF_D + 1024, SZ_Q = SOFF_Q + 1024;
    // LDS: [ring][R1: fp32 x of the norm stages / codes of the Wo and down stages][R2: codes of the norm
    //       stages][scales][1 KiB dump for the DMA pieces that move nothing][control]
    static constexpr int R1 = (DIM * 4 > HID ? DIM * 4 : HID), R2 = DIM, R3 = ((HID / 64 + 3) & ~3) * 4;
    static constexpr int RING = (ENG_LDS_TOTAL - R1 - R2 - R3 - 1024 - (int)sizeof(EngCtl)) / 1024 * 1024;
    static constexpr int OFF_R1 = RING, OFF_R2 = OFF_R1 + R1, OFF_R3 = OFF_R2 + R2, OFF_DUMP = OFF_R3 + R3, OFF_CTL = OFF_DUMP + 1024;
    static constexpr int LDS_BYTES = OFF_CTL + (int)sizeof(EngCtl);
    static constexpr int stage_of(int s) { return s < S0_G ? 0 : (s < S0_D ? 1 : (s < S0_Q ? 2 : 3)); }
    static constexpr int stage_end(int s) { return s < S0_G ? S0_G : (s < S0_D ? S0_D : (s < S0_Q ? S0_Q : S_END)); }
    static constexpr int slot_size(int s) { return s < S0_G ? SZ_W : (s < S0_D ? SZ_G : (s < S0_Q ? SZ_D : SZ_Q)); }
};
template <class CFG>
struct SlotTab {
    int pos[CFG::S_END];
    int need[CFG::S_END];
};
template <class CFG>
constexpr SlotTab<CFG> make_slot_tab() {
    SlotTab<CFG> t{};
    int pos = 0;
    for (int s = 0; s < CFG::S_END; s++) {
        const int sz = CFG::slot_size(s);
        if (pos + sz > CFG::RING) pos = 0;
        t.pos[s] = pos;
        t.need[s] = -1;
        for (int j = s - 1; j >= 0; j--) {
            if (t.pos[j] < pos + sz && pos < t.pos[j] + CFG::slot_size(j)) { t.need[s] = j; break; }
        }
        pos += sz;
    }
    return t;
}
template <class CFG>
struct Tabs { static constexpr SlotTab<CFG> T = make_slot_tab<CFG>(); };
static_assert(Cfg4B::LDS_BYTES <= ENG_LDS_TOTAL, "LDS");

// The loader's own LDS traffic goes through inline asm: hipcc treats an LDS-DMA in flight as a
// pending LDS write that any ds_read / ds_write of the wave may alias and would put
// s_waitcnt vmcnt(0) in front of each of them -- draining the stream the loader exists to keep full.
__device__ __forceinline__ void ldsa_store(const void* p, unsigned v) {
    const unsigned addr = (unsigned)(unsigned long long)(const lds_char*)p;
    asm volatile("ds_write_b32 %0, %1" ::"v"(addr), "v"(v) : "memory");
}
__device__ __forceinline__ unsigned ldsa_load(const void* p) {
    const unsigned addr = (unsigned)(unsigned long long)(const lds_char*)p;
    unsigned v;
    asm volatile("ds_read_b32 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(addr) : "memory");
    return v;
}
// min over each row of 16 lanes, by DPP only
__device__ __forceinline__ unsigned row_min_u32(unsigned v) {
    v = min(v, (unsigned)__builtin_amdgcn_mov_dpp((int)v, 0xB1, 0xF, 0xF, true));    // quad_perm [1,0,3,2]
    v = min(v, (unsigned)__builtin_amdgcn_mov_dpp((int)v, 0x4E, 0xF, 0xF, true));    // quad_perm [2,3,0,1]
    v = min(v, (unsigned)__builtin_amdgcn_mov_dpp((int)v, 0x141, 0xF, 0xF, true));   // row_half_mirror
    v = min(v, (unsigned)__builtin_amdgcn_mov_dpp((int)v, 0x140, 0xF, 0xF, true));   // row_mirror
    return v;
}

__device__ __forceinline__ bool eng_dead(lds_u32* abortp) {
    return __hip_atomic_load(abortp, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) != 0;
}
__device__ __forceinline__ void eng_give_up(const Engine& a, lds_u32* abortp) {
    __hip_atomic_store(abortp, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    __hip_atomic_store((ge_u64*)&a.sync->aborted, 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __hip_atomic_store((ge_u32*)a.error, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}

// ======================================================================================
// LOADER
// ======================================================================================
// One slot = rows [row0, row0 + rows) of a [d][n] matrix: `pieces` 1-KiB DMA pieces of codes to the
// slot's start, the scales to soff.  Always 16 DMA instructions (vmcnt counts instructions, and
// its immediate must be a constant): the surplus ones go to the dump area with an empty descriptor.
__device__ __forceinline__ void load_slot(const int8_t* W, const float* S, int n, int row0, int rows, int pieces, int soff,
                                          lds_char* dst, lds_char* dump, int lane) {
    const int cbytes = (row0 + rows) * n;                 // < 2^31 for every matrix of these models
    const int sbytes = (row0 + rows) * (n >> 6) * 4;
    const __amdgpu_buffer_rsrc_t rc = __builtin_amdgcn_make_buffer_rsrc(const_cast<int8_t*>(W), 0, cbytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(S), 0, sbytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rz = __builtin_amdgcn_make_buffer_rsrc(const_cast<int8_t*>(W), 0, 0, 0x00020000);
    const int voff = row0 * n + lane * 16;
#pragma unroll
    for (int k = 0; k < 15; k++) {
        if (k < pieces)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rc, (__attribute__((address_space(3))) void*)(dst + k * 1024), 16, voff, k * 1024, 0, 2 /* nt */);
        else
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rz, (__attribute__((address_space(3))) void*)dump, 16, lane * 16, 0, 0, 2);
        // (without this hipcc folds the identical surplus pieces into one, and the vmcnt arithmetic
        // of the loader -- 16 instructions per slot -- publishes slots that have not landed)
        asm volatile("" ::: "memory");
    }
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (__attribute__((address_space(3))) void*)(dst + soff), 16,
                                             row0 * (n >> 6) * 4 + lane * 16, 0, 0, 2);
}

template <class CFG>
__device__ __forceinline__ void loader_main(const Engine& a, lds_char* ring, EngCtl* lc, int lane, int b) {
    lds_u32* abortp = (lds_u32*)&lc->abort;
    const int total = a.qkv_q ? CFG::S_END : CFG::S0_Q;
    lds_char* dump = ring + CFG::OFF_DUMP;
    const bool use_hold = (a.flags & 1) != 0;
    unsigned issued = 0, landed = 0;
    for (int s = 0; s < total; s++) {
        const int need = Tabs<CFG>::T.need[s];
        const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
        for (;;) {
            // the slot's bytes are free once no consumer can still read slot `need`; while the consumers
            // are polling global memory for a hand-off the loader keeps out of their way (flags & 1)
            bool ok = true;
            if (need >= 0 || use_hold) {
                // min over the 15 consumers' words (lane 15 repeats lane 0's)
                const unsigned p = row_min_u32(ldsa_load(&lc->prog[(lane & 15) < ENG_CW ? (lane & 15) : 0]));
                const int mp = __builtin_amdgcn_readfirstlane((int)p);
                ok = mp > need;
                // a poll for the activation of stage t holds back the slots of stages >= t (never those of
                // the stage that produces it: the vector could not complete without them)
                if (ok && use_hold) {
                    const int t = CFG::stage_of(s);
                    for (int u = 1; u <= t; u++) ok = ok && ldsa_load(&lc->hold[u]) == 0;
                }
                (void)mp;
            }
            if (ok) break;
            // blocked: nothing more will be requested for a while, so publish what is in flight
            if (issued - landed == 2) {
                asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
                ldsa_store(&lc->landed, ++landed);
            } else if (issued - landed == 1) {
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                ldsa_store(&lc->landed, ++landed);
            } else {
                __builtin_amdgcn_s_sleep(1);
            }
            if (ldsa_load(&lc->abort)) return;
            if (__builtin_amdgcn_s_memrealtime() - t0 > ENG_TIMEOUT_TICKS) { if (lane == 0) eng_give_up(a, abortp); return; }
        }
        lds_char* dst = ring + Tabs<CFG>::T.pos[s];
        ESTAMP(a, 32, s);
        if (s < CFG::S0_G) {
            const int k = s - CFG::S0_W, r0 = CFG::RS_W * k, rows = min(CFG::RS_W, CFG::RW_W - r0);
            load_slot(a.wo_q, a.wo_s, CFG::P, CFG::RW_W * b + r0, rows, eng_pieces(rows, CFG::P), CFG::SOFF_W, dst, dump, lane);
        } else if (s < CFG::S0_D) {
            const int k = s - CFG::S0_G, r0 = CFG::RS_G * k, rows = min(CFG::RS_G, CFG::RW_G - r0);
            load_slot(a.gu_q, a.gu_s, CFG::DIM, CFG::RW_G * b + r0, rows, eng_pieces(rows, CFG::DIM), CFG::SOFF_G, dst, dump, lane);
        } else if (s < CFG::S0_Q) {
            const int k = s - CFG::S0_D, r0 = CFG::RS_D * k, rows = min(CFG::RS_D, CFG::RW_D - r0);
            load_slot(a.dn_q, a.dn_s, CFG::HID, CFG::RW_D * b + r0, rows, eng_pieces(rows, CFG::HID), CFG::SOFF_D, dst, dump, lane);
        } else {
            const int k = s - CFG::S0_Q, r0 = CFG::RS_Q * k, rows = min(CFG::RS_Q, CFG::RW_Q - r0);
            load_slot(a.qkv_q, a.qkv_s, CFG::DIM, CFG::RW_Q * b + r0, rows, eng_pieces(rows, CFG::DIM), CFG::SOFF_Q, dst, dump, lane);
        }
        issued++;
        if (issued - landed > ENG_INFLIGHT) {
            // all but the youngest INFLIGHT slots (16 DMA instructions each) have landed
            asm volatile("s_waitcnt vmcnt(%0)" ::"n"(ENG_INFLIGHT * 16) : "memory");
            landed = issued - ENG_INFLIGHT;
            ldsa_store(&lc->landed, landed);
        }
    }
    while (landed < issued) {
        if (issued - landed == 2) asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        ldsa_store(&lc->landed, ++landed);
    }
}

// ======================================================================================
// CONSUMERS
// ======================================================================================
struct ConsCtx {
    const Engine* a;
    EngCtl* lc;
    lds_u32* abortp;
    unsigned sync_target;     // next value of lc->csync that means "all consumers arrived"
    int cw, lane;
    bool dead;
};

// consumers meet (LDS counter; the loader is not part of it)
__device__ __forceinline__ void cons_sync(ConsCtx& c) {
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    c.sync_target += ENG_CW;
    if (c.lane == 0) __hip_atomic_fetch_add((lds_u32*)&c.lc->csync, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    for (;;) {
        const unsigned v = __hip_atomic_load((lds_u32*)&c.lc->csync, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        if ((int)(v - c.sync_target) >= 0) break;
        if (eng_dead(c.abortp)) { c.dead = true; break; }
        if (__builtin_amdgcn_s_memrealtime() - t0 > ENG_TIMEOUT_TICKS) { if (c.lane == 0) eng_give_up(*c.a, c.abortp); c.dead = true; break; }
        __builtin_amdgcn_s_sleep(1);
    }
    asm volatile("" ::: "memory");
}
// wait until stream slot s has landed in the ring
__device__ __forceinline__ void cons_wait_slot(ConsCtx& c, int s) {
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    for (;;) {
        const unsigned v = __hip_atomic_load((lds_u32*)&c.lc->landed, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        if (v > (unsigned)s) break;
        if (eng_dead(c.abortp)) { c.dead = true; break; }
        if (__builtin_amdgcn_s_memrealtime() - t0 > ENG_TIMEOUT_TICKS) { if (c.lane == 0) eng_give_up(*c.a, c.abortp); c.dead = true; break; }
        __builtin_amdgcn_s_sleep(1);
    }
    asm volatile("" ::: "memory");
}
__device__ __forceinline__ void cons_progress(ConsCtx& c, int next_slot) {
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");    // this wave's reads of earlier slots have returned
    if (c.lane == 0) __hip_atomic_store((lds_u32*)&c.lc->prog[c.cw], (unsigned)next_slot, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}

// While a wave polls global memory it holds the loader off (flags & 1): the poll's round trip is
// then one trip through an idle memory pipeline instead of a wait behind 32-48 KB of weight pieces.
__device__ __forceinline__ void hold_on(ConsCtx& c, int t) {
    if (c.lane == 0) __hip_atomic_fetch_add((lds_u32*)&c.lc->hold[t], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}
__device__ __forceinline__ void hold_off(ConsCtx& c, int t) {
    if (c.lane == 0) __hip_atomic_fetch_sub((lds_u32*)&c.lc->hold[t], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}
__device__ __forceinline__ bool no_sentinel(v4i v) {
    return (unsigned)v.x != Q3_ENG_SENTINEL && (unsigned)v.y != Q3_ENG_SENTINEL && (unsigned)v.z != Q3_ENG_SENTINEL &&
           (unsigned)v.w != Q3_ENG_SENTINEL;
}
// K 256-value blocks (blk0, blk0 + 15, ...) of a handed-off fp32 vector: poll, all K loads in flight together,
// until no lane sees the sentinel in any of them
template <int K>
__device__ __forceinline__ void poll_blocks(ConsCtx& c, __amdgpu_buffer_rsrc_t r, int blk0, int nblk, int stage, float4 (&out)[K]) {
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    v4i v[K];
    hold_on(c, stage);
    for (;;) {
        bool ok = true;
#pragma unroll
        for (int k = 0; k < K; k++) {
            const int blk = blk0 + ENG_CW * k;
            // (a block past the vector: offset outside the descriptor, reads zero)
            const int off = blk < nblk ? (blk * 256 + 4 * c.lane) * 4 : 0x7ffffff0;
            v[k] = __builtin_amdgcn_raw_buffer_load_b128(r, off, 0, 16 /* sc1 */);
        }
#pragma unroll
        for (int k = 0; k < K; k++) ok = ok && no_sentinel(v[k]);
        if (__all(ok)) break;
        if (eng_dead(c.abortp)) { c.dead = true; break; }
        if (__builtin_amdgcn_s_memrealtime() - t0 > ENG_TIMEOUT_TICKS) { if (c.lane == 0) eng_give_up(*c.a, c.abortp); c.dead = true; break; }
        __builtin_amdgcn_s_sleep(1);
    }
    hold_off(c, stage);
#pragma unroll
    for (int k = 0; k < K; k++) out[k] = __builtin_bit_cast(float4, v[k]);
}
// publish one value of a handed-off vector (write-through; a value that happens to carry the
// sentinel's bits -- a NaN with that payload -- goes out as the canonical NaN instead)
__device__ __forceinline__ void publish(float* p, float v) {
    unsigned u = __float_as_uint(v);
    if (u == Q3_ENG_SENTINEL) u = 0x7fc00000u;
    __hip_atomic_store((ge_u32*)p, u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// row r of a landed slot . quantised activation held in registers; tile_dot's arithmetic
template <int NJ>
__device__ __forceinline__ float row_dot(const lds_char* slot, int soff, int r, int n, const v4i (&xv)[NJ], const float (&sx)[NJ], int lane) {
    const lds_char* codes = slot + r * n;
    const __attribute__((address_space(3))) float* sc =
        (const __attribute__((address_space(3))) float*)(slot + soff) + r * (n >> 6);
    const int quad = lane >> 2;
    float acc = 0.0f;
#pragma unroll
    for (int j = 0; j < NJ; j++) {
        const int off = j * 1024 + lane * 16;
        const bool act = off < n;
        // (a lane past the row reads the next row's bytes or the slot's tail: unused)
        const v4i w = *reinterpret_cast<const __attribute__((address_space(3))) v4i*>(codes + off);
        const float ws = sc[act ? j * 16 + quad : 0];
        const int dsum = quad_sum(dot16(w, xv[j]));
        const float pp = ((float)dsum * ws) * sx[j];
        acc = act ? acc + pp : acc;
    }
    return bfly_quads(acc);
}
// the quantised activation of a stage, from LDS into registers (lane l of wave-load j holds codes [1024 j + 16 l, +16))
template <int NJ>
__device__ __forceinline__ void load_act(const int8_t* lq, const float* ls, int n, int lane, v4i (&xv)[NJ], float (&sx)[NJ]) {
    const int quad = lane >> 2;
#pragma unroll
    for (int j = 0; j < NJ; j++) {
        const int off = j * 1024 + lane * 16;
        xv[j] = v4i{0, 0, 0, 0};
        sx[j] = 0.0f;
        if (off < n) {
            xv[j] = *reinterpret_cast<const v4i*>(lq + off);
            sx[j] = ls[j * 16 + quad];
        }
    }
}

// gather a handed-off fp32 vector of n values into LDS (waves take blocks cw, cw+15, ..), then
// rmsnorm with weight nw + q8_quantize (prepare_activation's arithmetic): codes/scales in LDS
__device__ __forceinline__ void stage_in_norm(ConsCtx& c, const float* vec, const float* nw, int n, int stage, float* xl,
                                              int8_t* lq, float* ls) {
    const __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(vec), 0, n * 4, 0x00020000);
    const int nb = (n + 255) >> 8;
    float4 gw = make_float4(0.f, 0.f, 0.f, 0.f);
    const int myblk = c.cw;                               // n <= 15 * 256 for the norm stages
    if (myblk < nb && myblk * 256 + 4 * c.lane < n) gw = *reinterpret_cast<const float4*>(nw + myblk * 256 + 4 * c.lane);
    float4 own = make_float4(0.f, 0.f, 0.f, 0.f);
    if (myblk < nb) {
        float4 got[1];
        poll_blocks<1>(c, r, myblk, nb, stage, got);
        own = got[0];
        if (myblk * 256 + 4 * c.lane < n) *reinterpret_cast<float4*>(xl + myblk * 256 + 4 * c.lane) = own;
    }
    cons_sync(c);
    // SUM256 over all of x, every wave redundantly (q3_numerics.h "rmsnorm")
    float c0 = 0.f, c1 = 0.f, c2 = 0.f, c3 = 0.f;
    for (int i = 4 * c.lane; i < n; i += 256) {
        const float4 v = *reinterpret_cast<const float4*>(xl + i);
        c0 = c0 + v.x * v.x;
        c1 = c1 + v.y * v.y;
        c2 = c2 + v.z * v.z;
        c3 = c3 + v.w * v.w;
    }
    const float ss = bfly64((c0 + c1) + (c2 + c3));
    const float sc = 1.0f / sqrtf(ss / (float)n + 1e-6f);
    if (myblk < nb) {
        const int i = myblk * 256 + 4 * c.lane;
        float4 y;
        y.x = gw.x * (sc * own.x);
        y.y = gw.y * (sc * own.y);
        y.z = gw.z * (sc * own.z);
        y.w = gw.w * (sc * own.w);
        float scale;
        const int packed = quantize_group16(y, scale);
        if (i < n) {
            reinterpret_cast<int*>(lq)[i >> 2] = packed;
            if ((c.lane & 15) == 0) ls[i >> 6] = scale;
        }
    }
    cons_sync(c);
}

template <class CFG>
__device__ __forceinline__ void consumer_main(const Engine& a, char* smem, lds_char* ring, EngCtl* lc, int cw, int lane, int b) {
    ConsCtx c;
    c.a = &a; c.lc = lc; c.abortp = (lds_u32*)&lc->abort; c.sync_target = 0; c.cw = cw; c.lane = lane; c.dead = false;
    float* xl = reinterpret_cast<float*>(smem + CFG::OFF_R1);        // fp32 x of the norm stages ...
    int8_t* lq1 = reinterpret_cast<int8_t*>(smem + CFG::OFF_R1);     // ... or the codes of the Wo / down stages
    int8_t* lq2 = reinterpret_cast<int8_t*>(smem + CFG::OFF_R2);     // codes of the norm stages
    float* ls = reinterpret_cast<float*>(smem + CFG::OFF_R3);
    constexpr auto& TAB = Tabs<CFG>::T;
    const bool has_q = a.qkv_q != nullptr;

    // which set of hand-off vectors this launch uses (the other one is poisoned for the next launch)
    const unsigned long long epoch = __hip_atomic_load((ge_u64*)&a.sync->epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    const int set = (int)(epoch & 1ull);
    float* xw = a.xw + set * CFG::DIM;
    float* hv = a.hv + set * CFG::HID;
    float* xd = a.xd + set * CFG::DIM;
    {
        float* oxw = a.xw + (set ^ 1) * CFG::DIM;
        float* ohv = a.hv + (set ^ 1) * CFG::HID;
        float* oxd = a.xd + (set ^ 1) * CFG::DIM;
        const float sent = __uint_as_float(Q3_ENG_SENTINEL);
        if (cw == 12 && lane < CFG::RW_W) __hip_atomic_store((ge_u32*)(oxw + CFG::RW_W * b + lane), Q3_ENG_SENTINEL, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (cw == 13 && lane < CFG::RW_G / 2) __hip_atomic_store((ge_u32*)(ohv + (CFG::RW_G / 2) * b + lane), Q3_ENG_SENTINEL, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (cw == 14 && lane < CFG::RW_D) __hip_atomic_store((ge_u32*)(oxd + CFG::RW_D * b + lane), Q3_ENG_SENTINEL, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        (void)sent;
    }

    CSTAMP(0);
    // ================= Wo + residual (activation: attention output codes, plain memory) =================
    float xres = 0.0f;                                    // this wave's residual row (waves 0..9), carried Wo -> down
    if (cw < CFG::RW_W) xres = a.x[CFG::RW_W * b + cw];
    {
        if (cw < CFG::P / 1024) {
            const v4i v = reinterpret_cast<const v4i*>(a.att_q)[cw * 64 + lane];
            reinterpret_cast<v4i*>(lq1)[cw * 64 + lane] = v;
        } else if (cw == CFG::P / 1024 && lane < CFG::P / 64) {
            ls[lane] = a.att_s[lane];
        }
        cons_sync(c);
        v4i xv[CFG::NJ_P];
        float sx[CFG::NJ_P];
        load_act<CFG::NJ_P>(lq1, ls, CFG::P, lane, xv, sx);
        cons_sync(c);                                     // every wave has its copy: lq/ls may be rewritten
        CSTAMP(1);
        if (cw < CFG::RW_W) {
            const int s = CFG::S0_W + cw / CFG::RS_W;
            cons_wait_slot(c, s);
            const float acc = row_dot<CFG::NJ_P>(ring + TAB.pos[s], CFG::SOFF_W, cw % CFG::RS_W, CFG::P, xv, sx, lane);
            xres = xres + acc;                            // forward.c:295-298
            if (lane == 0) publish(xw + CFG::RW_W * b + cw, xres);
        }
        CSTAMP(2);
        // next: the first gate/up slot holding one of this wave's row pairs (pair p = rows 2p, 2p+1)
        cons_progress(c, CFG::S0_G + (2 * cw) / CFG::RS_G);
    }
    if (c.dead) return;

    // ================= gate/up + SwiGLU (activation: rmsnorm(x)) =================
    {
        stage_in_norm(c, xw, a.ffn_nw, CFG::DIM, 1, xl, lq2, ls);
        CSTAMP(3);
        v4i xv[CFG::NJ_X];
        float sx[CFG::NJ_X];
        load_act<CFG::NJ_X>(lq2, ls, CFG::DIM, lane, xv, sx);
        cons_sync(c);
        CSTAMP(4);
        constexpr int NPAIR = CFG::RW_G / 2;
#pragma unroll 1
        for (int p = cw; p < NPAIR; p += ENG_CW) {
            const int s = CFG::S0_G + (2 * p) / CFG::RS_G, r = (2 * p) % CFG::RS_G;
            cons_wait_slot(c, s);
            const lds_char* slot = ring + TAB.pos[s];
            const float g = row_dot<CFG::NJ_X>(slot, CFG::SOFF_G, r, CFG::DIM, xv, sx, lane);
            const float u = row_dot<CFG::NJ_X>(slot, CFG::SOFF_G, r + 1, CFG::DIM, xv, sx, lane);
            const float h = swiglu_pair(g, u);
            if (lane == 0) publish(hv + NPAIR * b + p, h);
            const int pn = p + ENG_CW;
            cons_progress(c, pn < NPAIR ? CFG::S0_G + (2 * pn) / CFG::RS_G
                                        : (cw < CFG::RW_D ? CFG::S0_D + cw : (has_q ? CFG::S0_Q + cw / CFG::RS_Q : 0x7fffffff)));
        }
    }
    if (c.dead) return;
    CSTAMP(5);

    // ================= down + residual (activation: q8_quantize(h)) =================
    {
        const __amdgpu_buffer_rsrc_t rh = __builtin_amdgcn_make_buffer_rsrc(hv, 0, CFG::HID * 4, 0x00020000);
        constexpr int NB = (CFG::HID + 255) / 256;
        constexpr int KB = (NB + ENG_CW - 1) / ENG_CW;        // blocks a wave may have to quantise
        float4 hb[KB];
        poll_blocks<KB>(c, rh, cw, NB, 2, hb);
#pragma unroll
        for (int k = 0; k < KB; k++) {
            const int blk = cw + ENG_CW * k;
            if (blk < NB) {                                   // wave-uniform
                const int i = blk * 256 + 4 * lane;
                float scale;
                const int packed = quantize_group16(hb[k], scale);
                if (i < CFG::HID) {
                    reinterpret_cast<int*>(lq1)[i >> 2] = packed;
                    if ((lane & 15) == 0) ls[i >> 6] = scale;
                }
            }
        }
        CSTAMP(6);
        cons_sync(c);
        CSTAMP(7);
        if (cw < CFG::RW_D) {
            v4i xv[CFG::NJ_H];
            float sx[CFG::NJ_H];
            load_act<CFG::NJ_H>(lq1, ls, CFG::HID, lane, xv, sx);
            const int s = CFG::S0_D + cw;
            cons_wait_slot(c, s);
            const float acc = row_dot<CFG::NJ_H>(ring + TAB.pos[s], CFG::SOFF_D, 0, CFG::HID, xv, sx, lane);
            xres = xres + acc;                            // forward.c:335-338
            if (lane == 0) {
                a.x[CFG::RW_D * b + cw] = xres;           // the residual the next launch starts from
                if (has_q) publish(xd + CFG::RW_D * b + cw, xres);
            }
        }
        CSTAMP(8);
        cons_progress(c, has_q ? CFG::S0_Q + cw / CFG::RS_Q : 0x7fffffff);
        cons_sync(c);                                     // lq/ls are rewritten by the next stage
        CSTAMP(9);
    }
    if (c.dead || !has_q) return;

    // ================= next layer's Wq|Wk|Wv (activation: rmsnorm(x)) =================
    {
        stage_in_norm(c, xd, a.att_nw_next, CFG::DIM, 3, xl, lq2, ls);
        CSTAMP(10);
        v4i xv[CFG::NJ_X];
        float sx[CFG::NJ_X];
        load_act<CFG::NJ_X>(lq2, ls, CFG::DIM, lane, xv, sx);
#pragma unroll 1
        for (int u = cw; u < CFG::RW_Q; u += ENG_CW) {
            const int s = CFG::S0_Q + u / CFG::RS_Q;
            cons_wait_slot(c, s);
            const float acc = row_dot<CFG::NJ_X>(ring + TAB.pos[s], CFG::SOFF_Q, u % CFG::RS_Q, CFG::DIM, xv, sx, lane);
            if (lane == 0) a.qkv[CFG::RW_Q * b + u] = acc;    // read by the attention launch that follows
        }
        CSTAMP(11);
    }
}

template <class CFG>
__global__ __launch_bounds__(ENG_WAVES * 64) void k_engine(Engine a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    if (a.clk && tid == 0) atomicMin(a.clk, (unsigned long long)__builtin_amdgcn_s_memrealtime());
    EngCtl* lc = reinterpret_cast<EngCtl*>(smem + CFG::OFF_CTL);
    if (tid < 32) reinterpret_cast<unsigned*>(lc)[tid] = 0;     // landed, csync, abort, hold, prog[] = 0
    __syncthreads();       // the only s_barrier: before the wave classes diverge
    lds_char* ring = (lds_char*)smem;
    if (wave == 0) loader_main<CFG>(a, ring, lc, lane, blockIdx.x);
    else consumer_main<CFG>(a, smem, ring, lc, wave - 1, lane, blockIdx.x);
    if (a.clk) {
        // (profiling launches only: a barrier here is fine, both classes are done)
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (tid == 64) atomicMax(a.clk + 1, (unsigned long long)__builtin_amdgcn_s_memrealtime());
    }
    if (blockIdx.x == 0 && tid == 64) {
        // launches alternate between the two sets of hand-off vectors
        const unsigned long long e = __hip_atomic_load((ge_u64*)&a.sync->epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        a.sync->epoch = e + 1ull;
    }
}

size_t engine_lds_bytes() {
    return (size_t)Cfg4B::LDS_BYTES;
}

bool engine_supported(int dim, int hid, int H, int KV, int hd, int n_cus) {
    return dim == 2560 && hid == 9728 && H == 32 && KV == 8 && hd == 128 && n_cus >= ENG_NWG;
}

void engine_layer(const Engine& e, hipStream_t st) {
    static bool attr_set = false;
    if (!attr_set) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k_engine<Cfg4B>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        attr_set = true;
    }
    hipLaunchKernelGGL(k_engine<Cfg4B>, dim3(ENG_NWG), dim3(ENG_WAVES * 64), engine_lds_bytes(), st, e);
}

// fill `n` words with the sentinel (attach: both sets of hand-off vectors start poisoned)
__global__ void k_fill_u32(unsigned* p, size_t n, unsigned v) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) p[i] = v;
}
void engine_poison(float* p, size_t n, hipStream_t st) {
    hipLaunchKernelGGL(k_fill_u32, dim3(64), dim3(256), 0, st, reinterpret_cast<unsigned*>(p), n, Q3_ENG_SENTINEL);
}

}  // namespace q3k
