// q3_gemv.hip -- the Q8_0 GEMV of the decode step (reference matmul(), src/forward.c:79-101)
// with its activation producer fused in front and its consumer fused behind.
//
//   PRO_NORM : rmsnorm (forward.c:12-28) + q8_quantize (q8.c:5-30) of the fp32 residual
//   PRO_F32  : q8_quantize of an fp32 vector (the SwiGLU output)
//   PRO_Q8   : activation already quantised (attention output)
//   EPI_STORE: out = W x     EPI_RESID: x += W x (forward.c:295-298, 335-338)
//   EPI_SWIGLU: rows are interleaved (gate_i, up_i); out_i = silu(gate_i)*up_i (forward.c:122-139)
//
// HBM is the only resource that matters here (1 MAC per weight byte), so the kernel is
// built around getting every weight byte of the launch in flight as early as possible:
//
//   * ONE-SHOT: the launch is sized so that each wave owns exactly one task of R rows
//     (R*NJ 1-KiB wave-loads); a wave issues ALL its weight loads before anything else
//     depends on them, so the whole matrix (<= 64 MB at these shapes) is requested from
//     HBM within the first microsecond and the kernel lasts one memory round trip plus
//     the transfer.  Only the classifier (hundreds of MB) takes the looping variant,
//     which keeps two tasks per wave in flight.
//   * the activation prologue (norm, quantise -> LDS) reads a few KB that the previous
//     kernel left in L2; those loads are issued BEFORE the weight loads (vmcnt retires
//     in order) and the arithmetic runs while the weights are in flight.
//   * lane l of wave-load j reads bytes [1024 j + 16 l, +16) of a row: 1 KiB contiguous
//     per instruction; a quad of lanes = one 64-wide quantisation group, so the int32
//     group dot is 4 x v_dot4_i32_i8 + a DPP quad sum, then scaled in fp32 exactly as
//     the reference does (((float)dot * ws) * xs) and summed in the SUM16 tree.
//
// int8 MFMA (v_mfma_i32_16x16x64_i8) was considered and rejected for batch 1: the B
// operand would carry one useful column of 16, the A fragment wants 16 rows x 64 B
// (64-B segments, half a cache line each) instead of 1 KiB rows, and the dot products
// above already cost < 10 % of the VALU issue slots of a CU that is waiting on HBM.
#include <cstdio>
#include <cstdlib>

#include "q3_device.hpp"
#include "q3_kernels.hpp"
#include "q3_tile.hpp"

namespace q3k {

// wave-loads requested before the activation prologue (the rest follow one per consumed load)
#ifndef Q3_GEMV_PRE
#define Q3_GEMV_PRE 8
#endif
// experiments that did not pay on Qwen3-4B shapes (-2 % / -0.2 %): tiles of at most Q3_GEMV_SMALL wave-loads requested
// whole before the barrier; Q3_GEMV_BALANCE: 256 workgroups with the tasks dealt evenly instead of ceil(ntasks/tw)
#ifndef Q3_GEMV_SMALL
#define Q3_GEMV_SMALL 0
#endif
#ifndef Q3_GEMV_BALANCE
#define Q3_GEMV_BALANCE 0
#endif

#ifdef Q3_GEMV_STAMPS
// [0,48): eight phase marks of the first / last wave of three workgroups;
// [64 + 4*wg + k): entry (k=0, first wave), barrier open (k=1) and end (k=2) of the last wave of EVERY workgroup
#define GSTAMP(i) do { if (a.stamps && (threadIdx.x == 0 || threadIdx.x == blockDim.x - 64)) { \
    const unsigned long long t_ = __builtin_amdgcn_s_memrealtime(); \
    const int b_ = blockIdx.x == 0 ? 0 : (blockIdx.x == gridDim.x / 2 ? 1 : (blockIdx.x == gridDim.x - 1 ? 2 : -1)); \
    if (b_ >= 0) a.stamps[(b_ * 2 + (threadIdx.x ? 1 : 0)) * 8 + (i)] = t_; \
    if (blockIdx.x < 256 && ((i) == 0 || (i) == 4 || (i) == 7) && (threadIdx.x != 0) == ((i) != 0)) \
        a.stamps[64 + 4 * blockIdx.x + ((i) == 0 ? 0 : ((i) == 4 ? 1 : 2))] = t_; } } while (0)
#else
#define GSTAMP(i) do {} while (0)
#endif

__device__ __forceinline__ WView make_wview(const Gemv& a) {
    return make_wview(a.W, a.S, a.d, a.n);
}

// one task = R consecutive rows starting at task*R; `task` must be wave-uniform
template <int R, int NJ>
__device__ __forceinline__ void tile_load(Tile<R, NJ>& t, const Gemv& a, const WView& wv, int task, int lane) {
    (void)a;
    tile_issue<R, NJ>(t, wv, task * R, lane);
}

template <int EPI, int R, int NJ>
__device__ __forceinline__ void tile_compute(const Tile<R, NJ>& t, const Gemv& a, int task, int lane,
                                             const int8_t* lq, const float* ls) {
    const int row0 = task * R;
    if (row0 >= a.d) return;
    float acc[R];
    tile_dot<R, NJ>(t, a.n, lane, lq, ls, acc);
    if (lane == 0) {
#pragma unroll
        for (int r = 0; r < R; r++) {
            if (row0 + r < a.d) {
                if (EPI == EPI_STORE) {
                    a.out[row0 + r] = acc[r];
                } else if (EPI == EPI_RESID) {
                    a.out[row0 + r] = a.out[row0 + r] + acc[r];
                } else if ((r & 1) == 0) {
                    a.out[(row0 + r) >> 1] = swiglu_pair(acc[r], acc[(r + 1) % R]);
                }
            }
        }
    }
}

// ---- activation prologue, shared by both kernels ---------------------------------
// LDS image: [fp32 x : PRO_NORM only][int8 codes n][scales n/64]
constexpr int QB = 4;             // 256-element blocks a wave may have to quantise

struct ProRegs {
    float4 xo[QB], go[QB];
    v4i cq[2];
    float cs;
};

// the loads of the prologue; issued before any weight load so that they retire first
template <int PRO>
__device__ __forceinline__ void pro_loads(ProRegs& p, const Gemv& a, int tid, int NT, int uwave, int NW, int lane) {
    const int n = a.n;
    if (PRO == PRO_NORM) {
#pragma unroll
        for (int k = 0; k < QB; k++) {       // fp32 x -> LDS, one float4 per thread per round
            const int i = 4 * (tid + k * NT);
            p.xo[k] = make_float4(0.f, 0.f, 0.f, 0.f);
            if (i < n) p.xo[k] = *reinterpret_cast<const float4*>(a.xf + i);
        }
#pragma unroll
        for (int k = 0; k < QB; k++) {       // norm weights of the blocks this wave quantises
            const int blk = uwave + k * NW;  // wave-uniform
            p.go[k] = make_float4(0.f, 0.f, 0.f, 0.f);
            if (blk * 256 + 4 * lane < n) p.go[k] = *reinterpret_cast<const float4*>(a.nw + blk * 256 + 4 * lane);
        }
    } else if (PRO == PRO_F32) {
#pragma unroll
        for (int k = 0; k < QB; k++) {
            const int blk = uwave + k * NW;
            p.xo[k] = make_float4(0.f, 0.f, 0.f, 0.f);
            if (blk * 256 + 4 * lane < n) p.xo[k] = *reinterpret_cast<const float4*>(a.xf + blk * 256 + 4 * lane);
        }
    } else {
#pragma unroll
        for (int k = 0; k < 2; k++) {
            const int c = tid + k * NT;
            p.cq[k] = v4i{0, 0, 0, 0};
            if (c < (n >> 4)) p.cq[k] = reinterpret_cast<const v4i*>(a.xq)[c];
        }
        p.cs = 0.0f;
        if (tid < (n >> 6)) p.cs = a.xs[tid];
    }
}

// norm / quantise into LDS; ends with the barrier after which lq / ls are complete.  Blocks
// are skipped with wave-uniform branches, so a wave pays only for the blocks it owns.
template <int PRO>
__device__ __forceinline__ void pro_compute(const ProRegs& p, const Gemv& a, int tid, int NT, int uwave, int NW,
                                            int lane, float* lx, int8_t* lq, float* ls) {
    const int n = a.n;
    if (PRO == PRO_NORM) {
#pragma unroll
        for (int k = 0; k < QB; k++) {
            const int i = 4 * (tid + k * NT);
            if (i < n) *reinterpret_cast<float4*>(lx + i) = p.xo[k];
        }
        GSTAMP(2);
        __syncthreads();
        GSTAMP(5);
        float c0 = 0.f, c1 = 0.f, c2 = 0.f, c3 = 0.f;     // SUM256, every wave redundantly
        for (int i0 = 4 * lane; i0 < n; i0 += 1024) {     // four LDS reads in flight per round
            float4 v[4];
#pragma unroll
            for (int u = 0; u < 4; u++) {
                v[u] = make_float4(0.f, 0.f, 0.f, 0.f);
                if (i0 + 256 * u < n) v[u] = *reinterpret_cast<const float4*>(lx + i0 + 256 * u);
            }
#pragma unroll
            for (int u = 0; u < 4; u++) {
                if (i0 + 256 * u < n) {
                    c0 = c0 + v[u].x * v[u].x;
                    c1 = c1 + v[u].y * v[u].y;
                    c2 = c2 + v[u].z * v[u].z;
                    c3 = c3 + v[u].w * v[u].w;
                }
            }
        }
        const float ss = bfly64((c0 + c1) + (c2 + c3));
        const float sc = 1.0f / sqrtf(ss / (float)n + 1e-6f);
#pragma unroll
        for (int k = 0; k < QB; k++) {
            const int blk = uwave + k * NW;
            if (blk * 256 < n) {             // wave-uniform; the last block may be partial (n % 64 == 0)
                const int i = blk * 256 + 4 * lane;
                const bool act = i < n;
                float4 y = make_float4(0.f, 0.f, 0.f, 0.f);
                if (act) {
                    const float4 v = *reinterpret_cast<const float4*>(lx + i);
                    y.x = p.go[k].x * (sc * v.x);
                    y.y = p.go[k].y * (sc * v.y);
                    y.z = p.go[k].z * (sc * v.z);
                    y.w = p.go[k].w * (sc * v.w);
                }
                float scale;
                const int packed = quantize_group16(y, scale);
                if (act) {
                    reinterpret_cast<int*>(lq)[i >> 2] = packed;
                    if ((lane & 15) == 0) ls[i >> 6] = scale;
                }
            }
        }
    } else if (PRO == PRO_F32) {
#pragma unroll
        for (int k = 0; k < QB; k++) {
            const int blk = uwave + k * NW;
            if (blk * 256 < n) {
                const int i = blk * 256 + 4 * lane;
                float scale;
                const int packed = quantize_group16(p.xo[k], scale);
                if (i < n) {
                    reinterpret_cast<int*>(lq)[i >> 2] = packed;
                    if ((lane & 15) == 0) ls[i >> 6] = scale;
                }
            }
        }
    } else {
#pragma unroll
        for (int k = 0; k < 2; k++) {
            const int c = tid + k * NT;
            if (c < (n >> 4)) reinterpret_cast<v4i*>(lq)[c] = p.cq[k];
        }
        if (tid < (n >> 6)) ls[tid] = p.cs;
    }
    GSTAMP(3);
    __syncthreads();
}

// ---- ONE-SHOT kernel: one task per wave, dots in the order the loads land ------------
// The last tw waves STREAM: each owns one task of R rows (R*NJ wave-loads):
//   1. request the first PRE wave-loads, 2. wait at the barrier for the quantised activation,
//   3. for every wave-load q in issue order: request load q+PRE, consume load q.
// A CU holds only a few tens of KB of requests in flight, so a wave that requests a whole
// 24-KiB tile sits in the issue stage for microseconds; with step 3 the dot products ride
// inside those stalls and the kernel ends right after its last byte lands.
// The first NW-tw waves PREPARE the activation (norm, quantise -> LDS) and exit.  They request no
// weights, so nothing of the prologue queues behind the weight stream: the barrier opens
// ~2 us after launch instead of ~5 us when every wave did both jobs.
constexpr int XR = 12;            // PRO_NORM: blocks of x a preparing wave sums per round
constexpr int QN = 5;             // PRO_NORM: blocks a preparing wave may have to quantise
constexpr int QF = 8;             // PRO_F32 / PRO_Q8: rounds of a preparing wave

// First of the two workgroup barriers of k_gemv3: the streaming waves hold their weight
// requests until the preparing waves have ISSUED their (first round of) loads.  The vector
// memory pipeline of a CU returns data in issue order, so a load queued behind the weight
// burst would come back only after ~2.5 us.
__device__ __forceinline__ void release_stream_waves() {
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
}

// 16 bytes at byte offset `off` of a buffer; past the end of the descriptor: zeros, no memory traffic
__device__ __forceinline__ float4 ld_f4(__amdgpu_buffer_rsrc_t r, int off) {
    const v4i t = __builtin_amdgcn_raw_buffer_load_b128(r, off, 0, 0);
    return make_float4(__int_as_float(t.x), __int_as_float(t.y), __int_as_float(t.z), __int_as_float(t.w));
}

template <int PRO>
__device__ __forceinline__ void prepare_activation(const Gemv& a, int nid, int NN, int lane, int8_t* lq, float* ls) {
    const int n = a.n;
    if (PRO == PRO_NORM) {
        // Branch-free loads through buffer descriptors that end at n floats: a lane past the end reads zeros and
        // moves no bytes.  (Written as `if (i < n) v = load`, every load became its own exec-masked block that
        // ENDS WITH s_waitcnt vmcnt(0): twelve dependent memory round trips in front of the barrier.)
        const __amdgpu_buffer_rsrc_t xr = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.xf), 0, n * 4, 0x00020000);
        const __amdgpu_buffer_rsrc_t gr = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.nw), 0, n * 4, 0x00020000);
        float4 own[QN], gw[QN];
#pragma unroll
        for (int k = 0; k < QN; k++) {                    // the blocks this wave quantises, and their norm weights
            const int i = (nid + k * NN) * 256 + 4 * lane;
            own[k] = ld_f4(xr, i * 4);
            gw[k] = ld_f4(gr, i * 4);
        }
        float c0 = 0.f, c1 = 0.f, c2 = 0.f, c3 = 0.f;     // SUM256 over all of x, every preparing wave redundantly
        for (int b0 = 0; b0 * 256 < n; b0 += XR) {
            float4 v[XR];
#pragma unroll
            for (int u = 0; u < XR; u++) v[u] = ld_f4(xr, ((b0 + u) * 256 + 4 * lane) * 4);
            if (b0 == 0) { GSTAMP(1); release_stream_waves(); GSTAMP(4); }
#pragma unroll
            for (int u = 0; u < XR; u++) {                // c + 0*0 == c exactly: slots past n change nothing
                c0 = c0 + v[u].x * v[u].x;
                c1 = c1 + v[u].y * v[u].y;
                c2 = c2 + v[u].z * v[u].z;
                c3 = c3 + v[u].w * v[u].w;
            }
        }
#ifdef Q3_GEMV_STAMPS
        asm volatile("" : "+v"(c0));
        GSTAMP(2);
#endif
        const float ss = bfly64((c0 + c1) + (c2 + c3));
        const float sc = 1.0f / sqrtf(ss / (float)n + 1e-6f);
#ifdef Q3_GEMV_STAMPS
        { float t_ = sc; asm volatile("" : "+v"(t_)); }
        GSTAMP(5);
#endif
#pragma unroll
        for (int k = 0; k < QN; k++) {
            const int blk = nid + k * NN;                  // wave-uniform
            if (blk * 256 < n) {
                const int i = blk * 256 + 4 * lane;
                float4 y;                                  // lanes past n hold zeros
                y.x = gw[k].x * (sc * own[k].x);
                y.y = gw[k].y * (sc * own[k].y);
                y.z = gw[k].z * (sc * own[k].z);
                y.w = gw[k].w * (sc * own[k].w);
                float scale;
                const int packed = quantize_group16(y, scale);
                if (i < n) {
                    reinterpret_cast<int*>(lq)[i >> 2] = packed;
                    if ((lane & 15) == 0) ls[i >> 6] = scale;
                }
            }
        }
    } else if (PRO == PRO_F32) {
        const __amdgpu_buffer_rsrc_t xr = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.xf), 0, n * 4, 0x00020000);
        float4 xo[QF];
#pragma unroll
        for (int k = 0; k < QF; k++) xo[k] = ld_f4(xr, ((nid + k * NN) * 256 + 4 * lane) * 4);
        release_stream_waves();
#pragma unroll
        for (int k = 0; k < QF; k++) {
            const int blk = nid + k * NN;
            if (blk * 256 < n) {
                const int i = blk * 256 + 4 * lane;
                float scale;
                const int packed = quantize_group16(xo[k], scale);
                if (i < n) {
                    reinterpret_cast<int*>(lq)[i >> 2] = packed;
                    if ((lane & 15) == 0) ls[i >> 6] = scale;
                }
            }
        }
    } else {
        // plain copy; addresses are clamped instead of predicated so that the loads stay branch-free
        const int t = nid * 64 + lane, T = NN * 64, nc = n >> 4, ns = n >> 6;
        const v4i* xq4 = reinterpret_cast<const v4i*>(a.xq);
        int c0 = t;
        do {                                   // at least one round: every wave must reach the barrier
            v4i v[4];
            float sv[4];
#pragma unroll
            for (int u = 0; u < 4; u++) {
                const int c = c0 + u * T;
                v[u] = xq4[c < nc ? c : nc - 1];
                sv[u] = a.xs[c < ns ? c : ns - 1];
            }
            if (c0 == t) release_stream_waves();
#pragma unroll
            for (int u = 0; u < 4; u++) {
                const int c = c0 + u * T;
                if (c < nc) reinterpret_cast<v4i*>(lq)[c] = v[u];
                if (c < ns) ls[c] = sv[u];
            }
            c0 += 4 * T;
        } while (c0 < nc);
    }
}

// How many preparing waves a launch can use (0 = shape not covered by prepare_activation):
// `lo` is the fewest that cover the activation, `hi` the most that still have work.
static void prep_waves(Pro pro, int n, int& lo, int& hi) {
    const int NB = (n + 255) / 256;
    if (pro == PRO_NORM) {
        lo = (NB + QN - 1) / QN;
        hi = NB;
    } else if (pro == PRO_F32) {
        lo = (NB + QF - 1) / QF;
        hi = NB;
    } else {
        lo = hi = 2;
    }
}

template <int PRO, int EPI, int NJ, int R>
__global__ __launch_bounds__(1024) void k_gemv3(Gemv a, int ntasks, int tw, int early8, int tq, int tr, int nn) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int NL = R * NJ;
    // look-ahead of the streaming loop: the whole tile when it is small (the launch is then bound by
    // its prologue, and a load requested after the barrier would add its latency to the tail)
    constexpr int PRE = NL <= Q3_GEMV_SMALL ? NL : (NL < Q3_GEMV_PRE ? NL : Q3_GEMV_PRE);
    constexpr int E4 = PRE < 4 ? PRE : 4, E8 = Q3_GEMV_SMALL ? PRE : (PRE < 8 ? PRE : 8);   // requested before the barrier: 4 or all of the look-ahead
    const int n = a.n;
    const int tid = threadIdx.x, wave = tid >> 6;
    int lane = tid & 63;
    if (a.clk && tid == 0) atomicMin(a.clk, (unsigned long long)__builtin_amdgcn_s_memrealtime());
    GSTAMP(0);
    int8_t* lq = reinterpret_cast<int8_t*>(smem);
    float* ls = reinterpret_cast<float*>(lq + n);
    const int uwave = __builtin_amdgcn_readfirstlane(wave);

    // the preparing waves are the FIRST waves of the workgroup: they start first, and their few
    // loads enter the CU's memory pipeline ahead of the weight requests (a load issued behind
    // the weight burst waits ~2.5 us in that queue)
    // nn = preparing waves = waves of the workgroup - tw (from the host: blockDim is one more scalar load away)
    if (uwave < nn) {
        prepare_activation<PRO>(a, uwave, nn, lane, lq, ls);
        GSTAMP(3);
        __syncthreads();
        return;
    }

    // tasks are dealt evenly: with tq = ntasks / G and tr = ntasks % G (from the host: two 64-bit divisions here were
    // ~260 instructions on every streaming wave's way to its first request), the first tr workgroups own tq + 1
    // consecutive tasks, the others tq -- at most tw of them either way
    const int bx = (int)blockIdx.x;
    const int tfirst = bx * tq + (bx < tr ? bx : tr);
    const int tcount = tq + (bx < tr ? 1 : 0);
    int task = (uwave - nn < tcount) ? tfirst + (uwave - nn) : ntasks;   // rows >= d read as zero through the descriptor
    const int row0 = task * R;
    float res[R];
    if (EPI == EPI_RESID) {
#pragma unroll
        for (int r = 0; r < R; r++) res[r] = (row0 + r < a.d) ? a.out[row0 + r] : 0.0f;
    }
    const WView wd = make_wview(a);
    Tile<R, NJ> T;
    const TileLane tl = tile_lane<NJ>(wd, lane);
    GSTAMP(2);
    release_stream_waves();
    GSTAMP(3);
    // A CU takes in ~40 KB of requests at once and ~27 KB/us after the first bytes are back
    // (~2 us); a wave that tries to request more just sits in the issue stage, and the barrier
    // below would wait for it.  So only as much as the CU accepts by the time the activation is
    // ready (4 wave-loads per wave, or the whole look-ahead when there are few streaming waves) goes out before the
    // barrier, the rest of the look-ahead right after it.
#pragma unroll
    for (int q = 0; q < E4; q++) tile_issue_one<R, NJ>(T, wd, tl, row0, q / NJ, q % NJ);
    if (early8) {
#pragma unroll
        for (int q = E4; q < E8; q++) tile_issue_one<R, NJ>(T, wd, tl, row0, q / NJ, q % NJ);
    }
    __builtin_amdgcn_sched_barrier(0);
    GSTAMP(1);
    __syncthreads();
    __builtin_amdgcn_sched_barrier(0);
    GSTAMP(4);
    if (!early8) {
#pragma unroll
        for (int q = E4; q < E8; q++) tile_issue_one<R, NJ>(T, wd, tl, row0, q / NJ, q % NJ);
    }
#pragma unroll
    for (int q = E8; q < PRE; q++) tile_issue_one<R, NJ>(T, wd, tl, row0, q / NJ, q % NJ);
    __builtin_amdgcn_sched_barrier(0);

    asm volatile("" : "+v"(lane));
    const int quad = lane >> 2;
    float acc[R];
#pragma unroll
    for (int q = 0; q < NL; q++) {
        const int r = q / NJ, j = q % NJ;
        if (q + PRE < NL) tile_issue_one<R, NJ>(T, wd, tl, row0, (q + PRE) / NJ, (q + PRE) % NJ);
        const int off = j * 1024 + lane * 16;
        const bool act = off < n;
        v4i xv = {0, 0, 0, 0};
        float sx = 0.0f;
        if (act) {
            xv = *reinterpret_cast<const v4i*>(lq + off);
            sx = ls[j * 16 + quad];
        }
        const int dsum = quad_sum(dot16(T.w[r][j], xv));
        const float pp = ((float)dsum * T.s[r][j]) * sx;
        if (j == 0) acc[r] = 0.0f;
        acc[r] = act ? acc[r] + pp : acc[r];
        if (j == NJ - 1) {
            acc[r] = bfly_quads(acc[r]);
            if (EPI == EPI_SWIGLU) {
                if (r & 1) {
                    const float h = swiglu_pair(acc[r - 1], acc[r]);
                    if (lane == 0 && row0 + r < a.d) a.out[(row0 + r) >> 1] = h;
                }
            } else if (lane == 0 && row0 + r < a.d) {
                a.out[row0 + r] = (EPI == EPI_RESID) ? res[r] + acc[r] : acc[r];
            }
        }
        __builtin_amdgcn_sched_barrier(0);
    }
    GSTAMP(6);
#ifdef Q3_GEMV_STAMPS
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    GSTAMP(7);
#endif
    if (a.clk) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (tid == (int)blockDim.x - 64) atomicMax(a.clk + 1, (unsigned long long)__builtin_amdgcn_s_memrealtime());
    }
}

// ---- LOOP kernel (the classifier): waves walk over tasks, two tiles in flight ----------
template <int PRO, int EPI, int NJ, int R, int MAXT>
__global__ __launch_bounds__(MAXT) void k_gemv2(Gemv a, int ntasks, int tw) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int n = a.n;
    const int NT = blockDim.x, NW = NT >> 6;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    if (a.clk && tid == 0) atomicMin(a.clk, (unsigned long long)__builtin_amdgcn_s_memrealtime());
    float* lx = reinterpret_cast<float*>(smem);
    int8_t* lq = reinterpret_cast<int8_t*>(smem + (PRO == PRO_NORM ? (size_t)n * 4 : 0));
    float* ls = reinterpret_cast<float*>(lq + n);
    const int uwave = __builtin_amdgcn_readfirstlane(wave);

    ProRegs pr;
    pro_loads<PRO>(pr, a, tid, NT, uwave, NW, lane);

    const int stride = gridDim.x * tw;
    int task = (uwave < tw) ? (int)blockIdx.x * tw + uwave : ntasks;
    if (task > ntasks) task = ntasks;
    const WView wd = make_wview(a);
    Tile<R, NJ> A;
    tile_load<R, NJ>(A, a, wd, task, lane);

    pro_compute<PRO>(pr, a, tid, NT, uwave, NW, lane, lx, lq, ls);

    Tile<R, NJ> B;
    while (task < ntasks) {
        int nxt = task + stride;
        tile_load<R, NJ>(B, a, wd, nxt < ntasks ? nxt : ntasks, lane);
        tile_compute<EPI, R, NJ>(A, a, task, lane, lq, ls);
        task = nxt;
        if (task >= ntasks) break;
        nxt = task + stride;
        tile_load<R, NJ>(A, a, wd, nxt < ntasks ? nxt : ntasks, lane);
        tile_compute<EPI, R, NJ>(B, a, task, lane, lq, ls);
        task = nxt;
    }
    if (a.clk) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (tid == 0) atomicMax(a.clk + 1, (unsigned long long)__builtin_amdgcn_s_memrealtime());
    }
}

// ---------------------------------------------------------------- launch plan -----

struct Plan {
    int R;      // rows per task
    int NW;     // waves per workgroup
    int TW;     // waves per workgroup that own a task (one-shot: the others prepare the activation)
    int grid;
    bool loop;
};

static int maxt_for(int loads) { return loads <= 12 ? 1024 : (loads <= 24 ? 768 : 512); }

static Plan make_plan(int d, int n, bool pairs, Pro pro) {
    const int NJ = (n + 1023) / 1024, NB = (n + 255) / 256;
    const int ncu = cu_count();
    const int rows_per_cu = (d + ncu - 1) / ncu;
    const int cand[4] = {1, 2, 4, 8};
    Plan p;
    // one pass: every streaming wave owns one task of R rows.  Fewer, fatter streaming waves
    // leave room for more preparing waves, i.e. fewer quantisation rounds before the barrier
    // opens; 5 streaming waves x 8 wave-loads already fill the CU's request queue.
    int lo, hi, best_rounds = 1 << 30;
    prep_waves(pro, n, lo, hi);
    p.R = 0;
    for (int ci = pairs ? 1 : 0; ci < 4; ci++) {
        const int R = cand[ci];
        if (R * NJ > 32) break;
        const int tw = (rows_per_cu + R - 1) / R;
        if (tw + lo > 16 || (tw < 5 && p.R)) continue;
        int nn = 16 - tw;
        if (nn > hi) nn = hi;
        const int rounds = (pro == PRO_Q8) ? 1 : (NB + nn - 1) / nn;
        if (rounds < best_rounds) {
            const int ntasks = (d + R - 1) / R;
            best_rounds = rounds;
            p.R = R;
            p.TW = tw;
            p.NW = tw + nn;
            p.grid = (Q3_GEMV_BALANCE && ntasks >= ncu) ? ncu : (ntasks + tw - 1) / tw;     // tasks are dealt evenly over the grid
            p.loop = false;
        }
    }
    if (p.R) return p;
    // too big for one pass: stream, two tasks per wave in flight; every wave also quantises
    int nwmin = (NB + 3) / 4;                 // ... at most 4 blocks of 256
    if (nwmin < 4) nwmin = 4;
    p.R = (NJ <= 3) ? 4 : (NJ <= 6 ? 2 : (pairs ? 2 : 1));
    const int cap = maxt_for(2 * p.R * NJ) / 64;
    p.NW = cap < 12 ? cap : 12;
    if (p.NW < nwmin) p.NW = nwmin < cap ? nwmin : cap;
    p.TW = p.NW;
    p.grid = ncu;
    p.loop = true;
    return p;
}

// experiments: Q3_GEMV_EARLY8=0/1 forces how many wave-loads go out before the barrier
static int early8_for(int tw) {
    static const char* e = getenv("Q3_GEMV_EARLY8");
    return e ? atoi(e) : (tw <= 8 ? 1 : 0);
}

template <int PRO, int EPI, int NJ, int R, bool LOOP>
static void launch_one(const Gemv& g, const Plan& p, hipStream_t st) {
    constexpr int LOADS = (LOOP ? 2 : 1) * R * NJ;
    constexpr int MAXT = LOADS <= 12 ? 1024 : (LOADS <= 24 ? 768 : 512);
    const int ntasks = (g.d + R - 1) / R;
    if constexpr (LOOP) {
        const size_t lds = (PRO == PRO_NORM ? (size_t)g.n * 4 : 0) + (size_t)g.n + (size_t)(g.n / 64) * 4;
        hipLaunchKernelGGL((k_gemv2<PRO, EPI, NJ, R, MAXT>), dim3(p.grid), dim3(p.NW * 64), lds, st, g, ntasks, p.TW);
    } else {
        const size_t lds = (size_t)g.n + (size_t)(g.n / 64) * 4;
        hipLaunchKernelGGL((k_gemv3<PRO, EPI, NJ, R>), dim3(p.grid), dim3(p.NW * 64), lds, st, g, ntasks, p.TW,
                           early8_for(p.TW), ntasks / p.grid, ntasks % p.grid, p.NW - p.TW);
    }
}

template <int PRO, int EPI, int NJ>
static bool launch_nj(const Gemv& g, const Plan& p, hipStream_t st) {
    if (p.loop) {
        switch (p.R) {
            case 1: if constexpr (EPI != EPI_SWIGLU) { launch_one<PRO, EPI, NJ, 1, true>(g, p, st); return true; } return false;
            case 2: launch_one<PRO, EPI, NJ, 2, true>(g, p, st); return true;
            case 4: if constexpr (NJ <= 4) { launch_one<PRO, EPI, NJ, 4, true>(g, p, st); return true; } return false;
            default: return false;
        }
    }
    switch (p.R) {
        case 1: if constexpr (EPI != EPI_SWIGLU) { launch_one<PRO, EPI, NJ, 1, false>(g, p, st); return true; } return false;
        case 2: launch_one<PRO, EPI, NJ, 2, false>(g, p, st); return true;
        case 4: if constexpr (NJ <= 8) { launch_one<PRO, EPI, NJ, 4, false>(g, p, st); return true; } return false;
        case 8: if constexpr (NJ <= 4) { launch_one<PRO, EPI, NJ, 8, false>(g, p, st); return true; } return false;
        default: return false;
    }
}

template <int PRO, int EPI>
static bool launch_pe(const Gemv& g, hipStream_t st) {
    const int NJ = (g.n + 1023) / 1024;
    if (g.n > 16384 || (PRO == PRO_NORM && g.n > 4 * 4 * 1024)) return false;
    const Plan p = make_plan(g.d, g.n, EPI == EPI_SWIGLU, (Pro)PRO);
    switch (NJ) {
        case 1: return launch_nj<PRO, EPI, 1>(g, p, st);
        case 2: return launch_nj<PRO, EPI, 2>(g, p, st);
        case 3: return launch_nj<PRO, EPI, 3>(g, p, st);
        case 4: return launch_nj<PRO, EPI, 4>(g, p, st);
        case 6: return launch_nj<PRO, EPI, 6>(g, p, st);
        case 10: return launch_nj<PRO, EPI, 10>(g, p, st);
        case 12: return launch_nj<PRO, EPI, 12>(g, p, st);
        default: return false;
    }
}

void gemv_generic(const Gemv& g, Pro pro, Epi epi, hipStream_t st);   // q3_kernels.hip

void gemv(const Gemv& g, Pro pro, Epi epi, hipStream_t st) {
    if (g.n % 64 || g.d % 2) {
        fprintf(stderr, "[q3hip] gemv: n must be a multiple of 64 and d even (n=%d d=%d)\n", g.n, g.d);
        exit(EXIT_FAILURE);
    }
    bool done = false;
    if (pro == PRO_Q8 && epi == EPI_STORE) done = launch_pe<PRO_Q8, EPI_STORE>(g, st);
    else if (pro == PRO_Q8 && epi == EPI_RESID) done = launch_pe<PRO_Q8, EPI_RESID>(g, st);
    else if (pro == PRO_NORM && epi == EPI_STORE) done = launch_pe<PRO_NORM, EPI_STORE>(g, st);
    else if (pro == PRO_NORM && epi == EPI_SWIGLU) done = launch_pe<PRO_NORM, EPI_SWIGLU>(g, st);
    else if (pro == PRO_F32 && epi == EPI_RESID) done = launch_pe<PRO_F32, EPI_RESID>(g, st);
    if (!done) gemv_generic(g, pro, epi, st);   // unusual shapes: the plain grid-stride kernel
}

}  // namespace q3k
