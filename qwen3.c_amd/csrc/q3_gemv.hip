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

// LDS image: [fp32 x : PRO_NORM only][int8 codes n][scales n/64]
template <int PRO, int EPI, int NJ, int R, bool LOOP, int MAXT>
__global__ __launch_bounds__(MAXT) void k_gemv2(Gemv a, int ntasks, int tw) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int n = a.n;
    const int NT = blockDim.x, NW = NT >> 6;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    if (a.clk && tid == 0) atomicMin(a.clk, (unsigned long long)__builtin_amdgcn_s_memrealtime());
    float* lx = reinterpret_cast<float*>(smem);
    int8_t* lq = reinterpret_cast<int8_t*>(smem + (PRO == PRO_NORM ? (size_t)n * 4 : 0));
    float* ls = reinterpret_cast<float*>(lq + n);

    // ---- activation loads first (they retire first) -------------------------------
    constexpr int QB = 4;             // 256-element blocks a wave may have to quantise
    float4 xo[QB], go[QB];
    v4i cq[2];
    float cs = 0.0f;
    if (PRO == PRO_NORM) {
#pragma unroll
        for (int k = 0; k < QB; k++) {       // fp32 x -> LDS, one float4 per thread per round
            const int i = 4 * (tid + k * NT);
            xo[k] = make_float4(0.f, 0.f, 0.f, 0.f);
            if (i < n) xo[k] = *reinterpret_cast<const float4*>(a.xf + i);
        }
#pragma unroll
        for (int k = 0; k < QB; k++) {       // norm weights of the blocks this wave quantises
            const int i = (wave + k * NW) * 256 + 4 * lane;
            go[k] = make_float4(0.f, 0.f, 0.f, 0.f);
            if (i < n) go[k] = *reinterpret_cast<const float4*>(a.nw + i);
        }
    } else if (PRO == PRO_F32) {
#pragma unroll
        for (int k = 0; k < QB; k++) {
            const int i = (wave + k * NW) * 256 + 4 * lane;
            xo[k] = make_float4(0.f, 0.f, 0.f, 0.f);
            if (i < n) xo[k] = *reinterpret_cast<const float4*>(a.xf + i);
        }
    } else {
#pragma unroll
        for (int k = 0; k < 2; k++) {
            const int c = tid + k * NT;
            cq[k] = v4i{0, 0, 0, 0};
            if (c < (n >> 4)) cq[k] = reinterpret_cast<const v4i*>(a.xq)[c];
        }
        if (tid < (n >> 6)) cs = a.xs[tid];
    }

    // ---- then every weight byte this wave will need (first task) -------------------
    // waves [0, tw) of a workgroup own tasks; the others only help with the prologue
    const int stride = LOOP ? gridDim.x * tw : 0;
    const int uwave = __builtin_amdgcn_readfirstlane(wave);
    int task = (uwave < tw) ? (int)blockIdx.x * tw + uwave : ntasks;
    if (task > ntasks) task = ntasks;            // rows >= d read as zero through the descriptor
    const WView wd = make_wview(a);
    Tile<R, NJ> A;
    tile_load<R, NJ>(A, a, wd, task, lane);

    // ---- prologue arithmetic while the weights are in flight -----------------------
    if (PRO == PRO_NORM) {
#pragma unroll
        for (int k = 0; k < QB; k++) {
            const int i = 4 * (tid + k * NT);
            if (i < n) *reinterpret_cast<float4*>(lx + i) = xo[k];
        }
        __syncthreads();
        float c0 = 0.f, c1 = 0.f, c2 = 0.f, c3 = 0.f;     // SUM256, every wave redundantly
        for (int i = 4 * lane; i < n; i += 256) {
            const float4 v = *reinterpret_cast<const float4*>(lx + i);
            c0 = c0 + v.x * v.x;
            c1 = c1 + v.y * v.y;
            c2 = c2 + v.z * v.z;
            c3 = c3 + v.w * v.w;
        }
        const float ss = bfly64((c0 + c1) + (c2 + c3));
        const float sc = 1.0f / sqrtf(ss / (float)n + 1e-6f);
#pragma unroll
        for (int k = 0; k < QB; k++) {
            const int i = (wave + k * NW) * 256 + 4 * lane;
            const bool act = i < n;
            float4 y = make_float4(0.f, 0.f, 0.f, 0.f);
            if (act) {
                const float4 v = *reinterpret_cast<const float4*>(lx + i);
                y.x = go[k].x * (sc * v.x);
                y.y = go[k].y * (sc * v.y);
                y.z = go[k].z * (sc * v.z);
                y.w = go[k].w * (sc * v.w);
            }
            float scale;
            const int packed = quantize_group16(y, scale);
            if (act) {
                reinterpret_cast<int*>(lq)[i >> 2] = packed;
                if ((lane & 15) == 0) ls[i >> 6] = scale;
            }
        }
    } else if (PRO == PRO_F32) {
#pragma unroll
        for (int k = 0; k < QB; k++) {
            const int i = (wave + k * NW) * 256 + 4 * lane;
            const bool act = i < n;
            float scale;
            const int packed = quantize_group16(xo[k], scale);
            if (act) {
                reinterpret_cast<int*>(lq)[i >> 2] = packed;
                if ((lane & 15) == 0) ls[i >> 6] = scale;
            }
        }
    } else {
#pragma unroll
        for (int k = 0; k < 2; k++) {
            const int c = tid + k * NT;
            if (c < (n >> 4)) reinterpret_cast<v4i*>(lq)[c] = cq[k];
        }
        if (tid < (n >> 6)) ls[tid] = cs;
    }
    __syncthreads();

    // ---- dot products ------------------------------------------------------------
    if (!LOOP) {
        if (task < ntasks) tile_compute<EPI, R, NJ>(A, a, task, lane, lq, ls);
    } else {
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
    int NW;     // waves per workgroup (all of them run the prologue)
    int TW;     // waves per workgroup that own a task (<= NW)
    int grid;
    bool loop;
};

static int maxt_for(int loads) { return loads <= 12 ? 1024 : (loads <= 24 ? 768 : 512); }

static Plan make_plan(int d, int n, bool pairs) {
    const int NJ = (n + 1023) / 1024, NB = (n + 255) / 256;
    const int ncu = 256;
    const int rows_per_cu = (d + ncu - 1) / ncu;
    int nwmin = (NB + 3) / 4;                 // each wave quantises at most 4 blocks of 256
    if (nwmin < 4) nwmin = 4;
    const int cand[4] = {1, 2, 4, 8};
    Plan p;
    for (int ci = pairs ? 1 : 0; ci < 4; ci++) {
        const int R = cand[ci];
        if (R * NJ > 32) break;
        const int cap = maxt_for(R * NJ) / 64;
        const int tw = (rows_per_cu + R - 1) / R;
        if (tw <= cap && nwmin <= cap) {
            const int ntasks = (d + R - 1) / R;
            p.R = R;
            p.TW = tw;
            p.NW = tw > nwmin ? tw : nwmin;
            p.grid = (ntasks + tw - 1) / tw;
            p.loop = false;
            return p;
        }
    }
    // too big for one pass: stream, two tasks per wave in flight
    p.R = (NJ <= 3) ? 4 : (NJ <= 6 ? 2 : (pairs ? 2 : 1));
    const int cap = maxt_for(2 * p.R * NJ) / 64;
    p.NW = cap < 12 ? cap : 12;
    if (p.NW < nwmin) p.NW = nwmin < cap ? nwmin : cap;
    p.TW = p.NW;
    p.grid = ncu;
    p.loop = true;
    return p;
}

template <int PRO, int EPI, int NJ, int R, bool LOOP>
static void launch_one(const Gemv& g, const Plan& p, hipStream_t st) {
    constexpr int LOADS = (LOOP ? 2 : 1) * R * NJ;
    constexpr int MAXT = LOADS <= 12 ? 1024 : (LOADS <= 24 ? 768 : 512);
    const int ntasks = (g.d + R - 1) / R;
    const size_t lds = (PRO == PRO_NORM ? (size_t)g.n * 4 : 0) + (size_t)g.n + (size_t)(g.n / 64) * 4;
    hipLaunchKernelGGL((k_gemv2<PRO, EPI, NJ, R, LOOP, MAXT>), dim3(p.grid), dim3(p.NW * 64), lds, st, g,
                       ntasks, p.TW);
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
    const Plan p = make_plan(g.d, g.n, EPI == EPI_SWIGLU);
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
