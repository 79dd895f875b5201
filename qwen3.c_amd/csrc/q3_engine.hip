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
#define ENG_NS 8                        // ring slots
#define ENG_SLOT 16384
#define ENG_SCALES_OFF 15360            // scales of a slot's rows sit behind 15 KiB of codes
#define ENG_INFLIGHT 2                  // slots in flight beyond the one being published
#define ENG_TIMEOUT_TICKS 200000000ull  // 2 s of the 100 MHz clock

typedef __attribute__((address_space(3))) unsigned lds_u32;
typedef __attribute__((address_space(3))) char lds_char;
typedef __attribute__((address_space(1))) unsigned ge_u32;
typedef __attribute__((address_space(1))) unsigned long long ge_u64;

// control words in LDS
struct EngCtl {
    unsigned landed;          // loader -> consumers: slots complete (stream order)
    unsigned csync;           // consumer <-> consumer: arrival counter (monotonic)
    unsigned abort;           // anybody -> all: leave
    unsigned pad0;
    unsigned prog[16];        // consumer cw -> loader: first slot this wave may still read
};

// ---- shape of the streamed stages (Qwen3-4B) ------------------------------------------
// Stage s of a launch: rows of width n, RW rows per workgroup, RS rows per slot.
struct Cfg4B {
    static constexpr int DIM = 2560, HID = 9728, P = 4096, QKV = 6144;
    static constexpr int NJ_X = 3, NJ_P = 4, NJ_H = 10;                 // wave-loads per row
    static constexpr int RW_W = 10, RS_W = 3, NS_W = 4;                 // Wo
    static constexpr int RW_G = 76, RS_G = 6, NS_G = 13;                // gate/up (rows interleaved)
    static constexpr int RW_D = 10, RS_D = 1, NS_D = 10;                // down
    static constexpr int RW_Q = 24, RS_Q = 6, NS_Q = 4;                 // Wq|Wk|Wv of the next layer
    static constexpr int S0_W = 0, S0_G = NS_W, S0_D = S0_G + NS_G, S0_Q = S0_D + NS_D, S_END = S0_Q + NS_Q;
};

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
// One slot = rows [row0, row0 + rows) of a [d][n] matrix: rows*n bytes of codes to the slot's
// start, rows*(n/64) scales to ENG_SCALES_OFF.  Always 16 DMA instructions (vmcnt counts
// instructions): the descriptors end with the slot's data, so pieces past it move nothing.
__device__ __forceinline__ void load_slot(const int8_t* W, const float* S, int n, int row0, int rows,
                                          lds_char* dst, int lane) {
    const int cbytes = (row0 + rows) * n;                 // < 2^31 for every matrix of these models
    const int sbytes = (row0 + rows) * (n >> 6) * 4;
    const __amdgpu_buffer_rsrc_t rc = __builtin_amdgcn_make_buffer_rsrc(const_cast<int8_t*>(W), 0, cbytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(S), 0, sbytes, 0x00020000);
    const int voff = row0 * n + lane * 16;
#pragma unroll
    for (int k = 0; k < 15; k++)
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rc, (__attribute__((address_space(3))) void*)(dst + k * 1024), 16, voff, k * 1024, 0, 2 /* nt */);
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (__attribute__((address_space(3))) void*)(dst + ENG_SCALES_OFF), 16,
                                             row0 * (n >> 6) * 4 + lane * 16, 0, 0, 2);
}

template <class CFG>
__device__ __forceinline__ void loader_main(const Engine& a, lds_char* ring, EngCtl* lc, int lane, int b) {
    lds_u32* abortp = (lds_u32*)&lc->abort;
    const int total = a.qkv_q ? CFG::S_END : CFG::S0_Q;
    unsigned issued = 0, landed = 0;
    for (int s = 0; s < total; s++) {
        if (s >= ENG_NS) {
            // ring position s % NS is free once no consumer can still read slot s - NS
            const unsigned need = (unsigned)(s - ENG_NS);
            const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
            for (;;) {
                // min over the 15 consumers' words (lane 15 repeats lane 0's)
                const unsigned p = row_min_u32(ldsa_load(&lc->prog[(lane & 15) < ENG_CW ? (lane & 15) : 0]));
                const unsigned m = (unsigned)__builtin_amdgcn_readfirstlane((int)p);
                if (m > need) break;
                if (ldsa_load(&lc->abort)) return;
                if (__builtin_amdgcn_s_memrealtime() - t0 > ENG_TIMEOUT_TICKS) { if (lane == 0) eng_give_up(a, abortp); return; }
                __builtin_amdgcn_s_sleep(1);
            }
        }
        lds_char* dst = ring + (s % ENG_NS) * ENG_SLOT;
        if (s < CFG::S0_G) {
            const int k = s - CFG::S0_W, r0 = CFG::RS_W * k;
            load_slot(a.wo_q, a.wo_s, CFG::P, CFG::RW_W * b + r0, min(CFG::RS_W, CFG::RW_W - r0), dst, lane);
        } else if (s < CFG::S0_D) {
            const int k = s - CFG::S0_G, r0 = CFG::RS_G * k;
            load_slot(a.gu_q, a.gu_s, CFG::DIM, CFG::RW_G * b + r0, min(CFG::RS_G, CFG::RW_G - r0), dst, lane);
        } else if (s < CFG::S0_Q) {
            const int k = s - CFG::S0_D, r0 = CFG::RS_D * k;
            load_slot(a.dn_q, a.dn_s, CFG::HID, CFG::RW_D * b + r0, min(CFG::RS_D, CFG::RW_D - r0), dst, lane);
        } else {
            const int k = s - CFG::S0_Q, r0 = CFG::RS_Q * k;
            load_slot(a.qkv_q, a.qkv_s, CFG::DIM, CFG::RW_Q * b + r0, min(CFG::RS_Q, CFG::RW_Q - r0), dst, lane);
        }
        issued++;
        if (issued - landed > ENG_INFLIGHT) {
            // all but the youngest INFLIGHT slots (16 DMA instructions each) have landed
            asm volatile("s_waitcnt vmcnt(%0)" ::"n"(ENG_INFLIGHT * 16) : "memory");
            landed = issued - ENG_INFLIGHT;
            ldsa_store(&lc->landed, landed);
        }
    }
    asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
    ldsa_store(&lc->landed, issued - 1);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    ldsa_store(&lc->landed, issued);
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

// one 256-value block of a handed-off fp32 vector: poll until no lane sees the sentinel
__device__ __forceinline__ float4 poll_block(ConsCtx& c, __amdgpu_buffer_rsrc_t r, int blk) {
    const int off = (blk * 256 + 4 * c.lane) * 4;         // past the vector: outside the descriptor, reads zero
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    v4i v;
    for (;;) {
        v = __builtin_amdgcn_raw_buffer_load_b128(r, off, 0, 16 /* sc1 */);
        const bool ok = (unsigned)v.x != Q3_ENG_SENTINEL && (unsigned)v.y != Q3_ENG_SENTINEL &&
                        (unsigned)v.z != Q3_ENG_SENTINEL && (unsigned)v.w != Q3_ENG_SENTINEL;
        if (__all(ok)) break;
        if (eng_dead(c.abortp)) { c.dead = true; break; }
        if (__builtin_amdgcn_s_memrealtime() - t0 > ENG_TIMEOUT_TICKS) { if (c.lane == 0) eng_give_up(*c.a, c.abortp); c.dead = true; break; }
        __builtin_amdgcn_s_sleep(1);
    }
    return __builtin_bit_cast(float4, v);
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
__device__ __forceinline__ float row_dot(const lds_char* slot, int r, int n, const v4i (&xv)[NJ], const float (&sx)[NJ], int lane) {
    const lds_char* codes = slot + r * n;
    const __attribute__((address_space(3))) float* sc =
        (const __attribute__((address_space(3))) float*)(slot + ENG_SCALES_OFF) + r * (n >> 6);
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
__device__ __forceinline__ void stage_in_norm(ConsCtx& c, const float* vec, const float* nw, int n, float* xl,
                                              int8_t* lq, float* ls) {
    const __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(vec), 0, n * 4, 0x00020000);
    const int nb = (n + 255) >> 8;
    float4 gw = make_float4(0.f, 0.f, 0.f, 0.f);
    const int myblk = c.cw;                               // n <= 15 * 256 for the norm stages
    if (myblk < nb && myblk * 256 + 4 * c.lane < n) gw = *reinterpret_cast<const float4*>(nw + myblk * 256 + 4 * c.lane);
    float4 own = make_float4(0.f, 0.f, 0.f, 0.f);
    if (myblk < nb) {
        own = poll_block(c, r, myblk);
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
    float* xl = reinterpret_cast<float*>(smem + ENG_NS * ENG_SLOT);
    int8_t* lq = reinterpret_cast<int8_t*>(xl + CFG::DIM);
    float* ls = reinterpret_cast<float*>(lq + CFG::HID);
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

    // ================= Wo + residual (activation: attention output codes, plain memory) =================
    float xres = 0.0f;                                    // this wave's residual row (waves 0..9), carried Wo -> down
    if (cw < CFG::RW_W) xres = a.x[CFG::RW_W * b + cw];
    {
        if (cw < CFG::P / 1024) {
            const v4i v = reinterpret_cast<const v4i*>(a.att_q)[cw * 64 + lane];
            reinterpret_cast<v4i*>(lq)[cw * 64 + lane] = v;
        } else if (cw == CFG::P / 1024 && lane < CFG::P / 64) {
            ls[lane] = a.att_s[lane];
        }
        cons_sync(c);
        v4i xv[CFG::NJ_P];
        float sx[CFG::NJ_P];
        load_act<CFG::NJ_P>(lq, ls, CFG::P, lane, xv, sx);
        cons_sync(c);                                     // every wave has its copy: lq/ls may be rewritten
        if (cw < CFG::RW_W) {
            const int s = CFG::S0_W + cw / CFG::RS_W;
            cons_wait_slot(c, s);
            const float acc = row_dot<CFG::NJ_P>(ring + (s % ENG_NS) * ENG_SLOT, cw % CFG::RS_W, CFG::P, xv, sx, lane);
            xres = xres + acc;                            // forward.c:295-298
            if (lane == 0) publish(xw + CFG::RW_W * b + cw, xres);
        }
        // next: the first gate/up slot holding one of this wave's row pairs (pair p = rows 2p, 2p+1)
        cons_progress(c, CFG::S0_G + (2 * cw) / CFG::RS_G);
    }
    if (c.dead) return;

    // ================= gate/up + SwiGLU (activation: rmsnorm(x)) =================
    {
        stage_in_norm(c, xw, a.ffn_nw, CFG::DIM, xl, lq, ls);
        v4i xv[CFG::NJ_X];
        float sx[CFG::NJ_X];
        load_act<CFG::NJ_X>(lq, ls, CFG::DIM, lane, xv, sx);
        cons_sync(c);
        constexpr int NPAIR = CFG::RW_G / 2;
#pragma unroll 1
        for (int p = cw; p < NPAIR; p += ENG_CW) {
            const int s = CFG::S0_G + (2 * p) / CFG::RS_G, r = (2 * p) % CFG::RS_G;
            cons_wait_slot(c, s);
            const lds_char* slot = ring + (s % ENG_NS) * ENG_SLOT;
            const float g = row_dot<CFG::NJ_X>(slot, r, CFG::DIM, xv, sx, lane);
            const float u = row_dot<CFG::NJ_X>(slot, r + 1, CFG::DIM, xv, sx, lane);
            const float h = swiglu_pair(g, u);
            if (lane == 0) publish(hv + NPAIR * b + p, h);
            const int pn = p + ENG_CW;
            cons_progress(c, pn < NPAIR ? CFG::S0_G + (2 * pn) / CFG::RS_G
                                        : (cw < CFG::RW_D ? CFG::S0_D + cw : (has_q ? CFG::S0_Q + cw / CFG::RS_Q : 0x7fffffff)));
        }
    }
    if (c.dead) return;

    // ================= down + residual (activation: q8_quantize(h)) =================
    {
        const __amdgpu_buffer_rsrc_t rh = __builtin_amdgcn_make_buffer_rsrc(hv, 0, CFG::HID * 4, 0x00020000);
        constexpr int NB = (CFG::HID + 255) / 256;
#pragma unroll 1
        for (int blk = cw; blk < NB; blk += ENG_CW) {
            const float4 v = poll_block(c, rh, blk);
            const int i = blk * 256 + 4 * lane;
            float scale;
            const int packed = quantize_group16(v, scale);
            if (i < CFG::HID) {
                reinterpret_cast<int*>(lq)[i >> 2] = packed;
                if ((lane & 15) == 0) ls[i >> 6] = scale;
            }
        }
        cons_sync(c);
        if (cw < CFG::RW_D) {
            v4i xv[CFG::NJ_H];
            float sx[CFG::NJ_H];
            load_act<CFG::NJ_H>(lq, ls, CFG::HID, lane, xv, sx);
            const int s = CFG::S0_D + cw;
            cons_wait_slot(c, s);
            const float acc = row_dot<CFG::NJ_H>(ring + (s % ENG_NS) * ENG_SLOT, 0, CFG::HID, xv, sx, lane);
            xres = xres + acc;                            // forward.c:335-338
            if (lane == 0) {
                a.x[CFG::RW_D * b + cw] = xres;           // the residual the next launch starts from
                if (has_q) publish(xd + CFG::RW_D * b + cw, xres);
            }
        }
        cons_progress(c, has_q ? CFG::S0_Q + cw / CFG::RS_Q : 0x7fffffff);
        cons_sync(c);                                     // lq/ls are rewritten by the next stage
    }
    if (c.dead || !has_q) return;

    // ================= next layer's Wq|Wk|Wv (activation: rmsnorm(x)) =================
    {
        stage_in_norm(c, xd, a.att_nw_next, CFG::DIM, xl, lq, ls);
        v4i xv[CFG::NJ_X];
        float sx[CFG::NJ_X];
        load_act<CFG::NJ_X>(lq, ls, CFG::DIM, lane, xv, sx);
#pragma unroll 1
        for (int u = cw; u < CFG::RW_Q; u += ENG_CW) {
            const int s = CFG::S0_Q + u / CFG::RS_Q;
            cons_wait_slot(c, s);
            const float acc = row_dot<CFG::NJ_X>(ring + (s % ENG_NS) * ENG_SLOT, u % CFG::RS_Q, CFG::DIM, xv, sx, lane);
            if (lane == 0) a.qkv[CFG::RW_Q * b + u] = acc;    // read by the attention launch that follows
        }
    }
}

template <class CFG>
__global__ __launch_bounds__(ENG_WAVES * 64) void k_engine(Engine a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    if (a.clk && tid == 0) atomicMin(a.clk, (unsigned long long)__builtin_amdgcn_s_memrealtime());
    EngCtl* lc = reinterpret_cast<EngCtl*>(smem + ENG_NS * ENG_SLOT + CFG::DIM * 4 + CFG::HID + ((CFG::HID / 64 + 3) & ~3) * 4);
    if (tid < 32) reinterpret_cast<unsigned*>(lc)[tid] = 0;     // landed, csync, abort, prog[] = 0
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
    using CFG = Cfg4B;
    return (size_t)ENG_NS * ENG_SLOT + CFG::DIM * 4 + CFG::HID + ((CFG::HID / 64 + 3) & ~3) * 4 + sizeof(EngCtl);
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
