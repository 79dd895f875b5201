// q3_kernels.hip -- hand-written gfx950 (CDNA4, wave64) kernels of the Qwen3 Q8_0
// decode step.  Arithmetic contract: q3_numerics.h (fixed reduction trees, no
// implicit FMA; this file is compiled with -ffp-contract=off).
//
// What the reference does in src/forward.c with one OpenMP region per loop is
// regrouped here around the HBM stream of the weights:
//
//   gemv<PRO,EPI>   Q8_0 int8 x int8 GEMV (reference matmul(), forward.c:79-101) with
//                   the producer of its activation fused in front
//                     PRO_NORM : rmsnorm (forward.c:12-28) + q8_quantize (q8.c:5-30)
//                     PRO_F32  : q8_quantize of an fp32 vector
//                     PRO_Q8   : activation already quantised
//                   and the consumer fused behind
//                     EPI_RESID : x += out (forward.c:295-298, 335-338)
//                     EPI_SWIGLU: silu(gate)*up on interleaved gate/up rows (forward.c:122-139)
//   attn            per-head RMSNorm + RoPE of q and k (forward.c:267-280), KV-cache
//                   append, GQA attention over LDS-staged K/V tiles (forward.c:141-195)
//                   and q8_quantize of the head outputs for the Wo GEMV
//
// Wave mapping of the GEMV: one wave owns two output rows at a time; lane l of load
// j reads the 16 weight bytes [1024 j + 16 l, +16) of the row (1 KiB fully coalesced
// per wave-load), so a QUAD of lanes holds one 64-wide quantisation group: the int32
// group dot is a 4 x v_dot4_i32_i8 + quad reduction, scaled in fp32 per group
// exactly as the reference does, and summed in the SUM16 tree.
#include "q3_kernels.hpp"

#include <cstdio>
#include <cstdlib>

#include "q3_device.hpp"
#include "q3_numerics.h"

namespace q3k {

// ---------------------------------------------------------------- GEMV -----

// Activation prologues: leave int8 codes in lq[n] and scales in ls[n/64] (LDS).
template <int NT>
__device__ __forceinline__ void stage_q8(const int8_t* __restrict__ xq, const float* __restrict__ xs,
                                         int n, int8_t* lq, float* ls) {
    const v4i* src = reinterpret_cast<const v4i*>(xq);
    v4i* dst = reinterpret_cast<v4i*>(lq);
    for (int c = threadIdx.x; c < (n >> 4); c += NT) dst[c] = src[c];
    for (int g = threadIdx.x; g < (n >> 6); g += NT) ls[g] = xs[g];
}

// Quantise y = (nw ? nw*(s*x) : x) block-wise: wave w takes the 256-element blocks
// b = w, w+NW, ...; 16 lanes per 64-group.  `normed` (optional) receives y.
template <int NT>
__device__ __forceinline__ void stage_quantize(const float* __restrict__ x, const float* __restrict__ nw,
                                               float s, int n, int8_t* lq, float* ls,
                                               float* __restrict__ normed) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    constexpr int NW = NT / 64;
    int* lq32 = reinterpret_cast<int*>(lq);
    for (int base = wave * 256; base < n; base += NW * 256) {
        const int i = base + 4 * lane;
        const bool act = i < n;
        float4 y = make_float4(0.f, 0.f, 0.f, 0.f);
        if (act) {
            const float4 v = *reinterpret_cast<const float4*>(x + i);
            if (nw) {
                const float4 g = *reinterpret_cast<const float4*>(nw + i);
                y.x = g.x * (s * v.x);
                y.y = g.y * (s * v.y);
                y.z = g.z * (s * v.z);
                y.w = g.w * (s * v.w);
            } else {
                y = v;
            }
        }
        float scale;
        const int packed = quantize_group16(y, scale);
        if (act) {
            lq32[i >> 2] = packed;
            if ((lane & 15) == 0) ls[i >> 6] = scale;
            if (normed) *reinterpret_cast<float4*>(normed + i) = y;
        }
    }
}

// NJ = number of 1 KiB wave-loads per row (compile-time when > 0, else runtime).
template <int PRO, int EPI, int NJ, int NT>
__global__ __launch_bounds__(NT) void k_gemv(Gemv a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int n = a.n, d = a.d;
    const int ngroups = n >> 6;
    int8_t* lq = reinterpret_cast<int8_t*>(smem);
    float* ls = reinterpret_cast<float*>(smem + n);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;

    if (PRO == PRO_Q8) {
        stage_q8<NT>(a.xq, a.xs, n, lq, ls);
    } else if (PRO == PRO_NORM) {
        const float ss = sum256_sq(a.xf, n, lane);   // every wave, redundantly: no barrier needed
        const float s = 1.0f / sqrtf(ss / (float)n + 1e-6f);
        stage_quantize<NT>(a.xf, a.nw, s, n, lq, ls, nullptr);
    } else {
        stage_quantize<NT>(a.xf, nullptr, 0.0f, n, lq, ls, nullptr);
    }
    __syncthreads();

    const int nj = (NJ > 0) ? NJ : ((n + 1023) >> 10);
    const int quad = lane >> 2;
    const int npairs = d >> 1;
    const int wave_stride = gridDim.x * (NT / 64);
    for (int rp = blockIdx.x * (NT / 64) + wave; rp < npairs; rp += wave_stride) {
        const int row0 = rp * 2;
        const int8_t* w0 = a.W + (size_t)row0 * n;
        const int8_t* w1 = w0 + n;
        const float* s0 = a.S + (size_t)row0 * ngroups;
        const float* s1 = s0 + ngroups;
        float acc0 = 0.0f, acc1 = 0.0f;
#pragma unroll
        for (int j = 0; j < nj; j++) {
            const int off = j * 1024 + lane * 16;
            const bool act = off < n;
            v4i wa = {0, 0, 0, 0}, wb = {0, 0, 0, 0}, xv = {0, 0, 0, 0};
            float sa = 0.0f, sb = 0.0f, sx = 0.0f;
            if (act) {
                wa = *reinterpret_cast<const v4i*>(w0 + off);
                wb = *reinterpret_cast<const v4i*>(w1 + off);
                sa = s0[j * 16 + quad];
                sb = s1[j * 16 + quad];
                xv = *reinterpret_cast<const v4i*>(lq + off);
                sx = ls[j * 16 + quad];
            }
            int da = dot16(wa, xv);
            int db = dot16(wb, xv);
            da += lane_xor_i<1>(da);
            db += lane_xor_i<1>(db);
            da += lane_xor_i<2>(da);
            db += lane_xor_i<2>(db);
            const float pa = ((float)da * sa) * sx;
            const float pb = ((float)db * sb) * sx;
            acc0 = act ? acc0 + pa : acc0;
            acc1 = act ? acc1 + pb : acc1;
        }
        // SUM16 butterfly over the 16 quads (column c = lane/4): xor 8,4,2,1 on c
        acc0 = acc0 + lane_xor_f<32>(acc0);
        acc1 = acc1 + lane_xor_f<32>(acc1);
        acc0 = acc0 + lane_xor_f<16>(acc0);
        acc1 = acc1 + lane_xor_f<16>(acc1);
        acc0 = acc0 + lane_xor_f<8>(acc0);
        acc1 = acc1 + lane_xor_f<8>(acc1);
        acc0 = acc0 + lane_xor_f<4>(acc0);
        acc1 = acc1 + lane_xor_f<4>(acc1);
        if (lane == 0) {
            if (EPI == EPI_STORE) {
                a.out[row0] = acc0;
                a.out[row0 + 1] = acc1;
            } else if (EPI == EPI_RESID) {
                a.out[row0] = a.out[row0] + acc0;
                a.out[row0 + 1] = a.out[row0 + 1] + acc1;
            } else {
                // rows (2i, 2i+1) = (gate_i, up_i): reference swiglu(), forward.c:134-139
                const float sig = 1.0f / (1.0f + q3_expf(-acc0));
                a.out[rp] = (acc0 * sig) * acc1;
            }
        }
    }
}

template <int PRO, int EPI>
static void gemv_launch(const Gemv& g, hipStream_t st) {
    constexpr int NT = 256;
    const int npairs = g.d / 2;
    int blocks = (npairs + (NT / 64) - 1) / (NT / 64);
    if (blocks > 1024) blocks = 1024;
    const size_t lds = (size_t)g.n + (size_t)(g.n / 64) * 4;
    const int nj = (g.n + 1023) / 1024;
#define Q3_GEMV_CASE(J) case J: hipLaunchKernelGGL((k_gemv<PRO, EPI, J, NT>), dim3(blocks), dim3(NT), lds, st, g); break;
    switch (nj) {
        Q3_GEMV_CASE(1)
        Q3_GEMV_CASE(2)
        Q3_GEMV_CASE(3)
        Q3_GEMV_CASE(4)
        Q3_GEMV_CASE(6)
        Q3_GEMV_CASE(10)
        Q3_GEMV_CASE(12)
        default: hipLaunchKernelGGL((k_gemv<PRO, EPI, 0, NT>), dim3(blocks), dim3(NT), lds, st, g); break;
    }
#undef Q3_GEMV_CASE
}

void gemv_generic(const Gemv& g, Pro pro, Epi epi, hipStream_t st) {
    if (pro == PRO_Q8 && epi == EPI_STORE) gemv_launch<PRO_Q8, EPI_STORE>(g, st);
    else if (pro == PRO_Q8 && epi == EPI_RESID) gemv_launch<PRO_Q8, EPI_RESID>(g, st);
    else if (pro == PRO_NORM && epi == EPI_STORE) gemv_launch<PRO_NORM, EPI_STORE>(g, st);
    else if (pro == PRO_NORM && epi == EPI_SWIGLU) gemv_launch<PRO_NORM, EPI_SWIGLU>(g, st);
    else if (pro == PRO_F32 && epi == EPI_RESID) gemv_launch<PRO_F32, EPI_RESID>(g, st);
    else if (pro == PRO_F32 && epi == EPI_STORE) gemv_launch<PRO_F32, EPI_STORE>(g, st);
    else {
        fprintf(stderr, "[q3hip] gemv: unsupported prologue/epilogue pair %d/%d\n", (int)pro, (int)epi);
        exit(EXIT_FAILURE);
    }
}

// ------------------------------------------------------------ attention ----

#define Q3_MAXG 8   // max query heads per kv head

template <int HD>
__global__ __launch_bounds__(256) void k_attn(Attn a, int multi) {
    constexpr int L4 = HD / 4;
    constexpr int CH = Q3_ATT_CHUNK;
    __shared__ __attribute__((aligned(16))) float Ks[CH * HD];
    __shared__ __attribute__((aligned(16))) float Vs[CH * HD];
    __shared__ __attribute__((aligned(16))) float qs[Q3_MAXG * HD];
    __shared__ __attribute__((aligned(16))) float kcur[HD];
    __shared__ __attribute__((aligned(16))) float vcur[HD];
    __shared__ float sc[Q3_MAXG * CH];
    __shared__ float es[Q3_MAXG * CH];
    __shared__ float mc[Q3_MAXG];
    __shared__ float lc[Q3_MAXG];
    float* red = Ks;   // [4][kv_mul][HD], reused after the score phase
    float* ofin = Vs;  // [kv_mul][HD], reused after the PV phase

    const int g = blockIdx.x;
    const int kv_mul = a.n_heads / a.n_kv;
    const int pos = a.ctl->pos;
    const int T = pos + 1;
    const int nchunks = (T + CH - 1) / CH;
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int P = a.n_heads * HD, KVD = a.n_kv * HD;
    const float* cs = a.rope + (size_t)pos * HD;   // [HD/2][2]
    const bool owner = ((nchunks - 1) % (int)gridDim.y) == (int)blockIdx.y;
    if ((int)blockIdx.y >= nchunks) return;

    // q heads of this kv group: norm + rope -> qs
    for (int i = wave; i < kv_mul; i += 4) {
        const int h = g * kv_mul + i;
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (lane < L4) v = *reinterpret_cast<const float4*>(a.qkv + (size_t)h * HD + 4 * lane);
        const float4 r = a.prepared ? v : headnorm_rope_wave<HD>(v, a.qnw, cs, lane);
        if (lane < L4) {
            *reinterpret_cast<float4*>(qs + i * HD + 4 * lane) = r;
            if (a.qdbg && blockIdx.y == 0) *reinterpret_cast<float4*>(a.qdbg + (size_t)h * HD + 4 * lane) = r;
        }
    }
    // the workgroup that owns the last chunk appends k (norm + rope) and v at `pos`
    if (owner) {
        if (wave == 0) {
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (lane < L4) v = *reinterpret_cast<const float4*>(a.qkv + P + (size_t)g * HD + 4 * lane);
            const float4 r = a.prepared ? v : headnorm_rope_wave<HD>(v, a.knw, cs, lane);
            if (lane < L4) {
                *reinterpret_cast<float4*>(kcur + 4 * lane) = r;
                *reinterpret_cast<float4*>(a.kc + ((size_t)g * a.seq_len + pos) * HD + 4 * lane) = r;
            }
        } else if (wave == 1) {
            if (lane < L4) {
                const float4 v = *reinterpret_cast<const float4*>(a.qkv + P + KVD + (size_t)g * HD + 4 * lane);
                *reinterpret_cast<float4*>(vcur + 4 * lane) = v;
                *reinterpret_cast<float4*>(a.vc + ((size_t)g * a.seq_len + pos) * HD + 4 * lane) = v;
            }
        }
    }
    __syncthreads();

    const float root = sqrtf((float)HD);
    for (int c = blockIdx.y; c < nchunks; c += gridDim.y) {
        const int t0 = c * CH;
        // stage the K and V tiles of this chunk in LDS
        for (int idx = tid; idx < CH * L4; idx += 256) {
            const int t = idx / L4, l4 = idx - t * L4;
            const int tt = t0 + t;
            if (tt < T) {
                float4 kk, vv;
                if (tt == pos) {
                    kk = *reinterpret_cast<const float4*>(kcur + 4 * l4);
                    vv = *reinterpret_cast<const float4*>(vcur + 4 * l4);
                } else {
                    kk = *reinterpret_cast<const float4*>(a.kc + ((size_t)g * a.seq_len + tt) * HD + 4 * l4);
                    vv = *reinterpret_cast<const float4*>(a.vc + ((size_t)g * a.seq_len + tt) * HD + 4 * l4);
                }
                *reinterpret_cast<float4*>(Ks + t * HD + 4 * l4) = kk;
                *reinterpret_cast<float4*>(Vs + t * HD + 4 * l4) = vv;
            }
        }
        __syncthreads();

        // scores: wave w owns positions [16w, 16w+16), two per step (one per 32-lane half)
        const int half = lane >> 5, l = lane & 31;
        for (int i = 0; i < kv_mul; i++) {
            float4 q4 = make_float4(0.f, 0.f, 0.f, 0.f);
            if (l < L4) q4 = *reinterpret_cast<const float4*>(qs + i * HD + 4 * l);
#pragma unroll
            for (int it = 0; it < 8; it++) {
                const int t = wave * 16 + it * 2 + half;
                float cdot = 0.0f;
                if (l < L4 && t0 + t < T) {
                    const float4 k4 = *reinterpret_cast<const float4*>(Ks + t * HD + 4 * l);
                    cdot = q4.x * k4.x;
                    cdot = cdot + q4.y * k4.y;
                    cdot = cdot + q4.z * k4.z;
                    cdot = cdot + q4.w * k4.w;
                }
                cdot = bfly32(cdot);
                if (l == 0) sc[i * CH + t] = cdot / root;
            }
        }
        __syncthreads();

        // chunk softmax statistics: one wave per head, lane = position
        for (int i = wave; i < kv_mul; i += 4) {
            const bool valid = t0 + lane < T;
            const float s = valid ? sc[i * CH + lane] : -3.0e38f;
            const float m = wave_max(s);
            const float e = valid ? q3_expf(s - m) : 0.0f;
            const float lsum = bfly64(e);
            es[i * CH + lane] = e;
            if (lane == 0) {
                mc[i] = m;
                lc[i] = lsum;
            }
        }
        __syncthreads();

        // weighted sum of V: stream s = t % 8 -> (wave = s/2, half = s%2)
        const int strm = 2 * wave + half;
        for (int i = 0; i < kv_mul; i++) {
            float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
            if (l < L4) {
#pragma unroll
                for (int mm = 0; mm < 8; mm++) {
                    const int t = 8 * mm + strm;
                    if (t0 + t < T) {
                        const float e = es[i * CH + t];
                        const float4 v4 = *reinterpret_cast<const float4*>(Vs + t * HD + 4 * l);
                        acc.x = acc.x + e * v4.x;
                        acc.y = acc.y + e * v4.y;
                        acc.z = acc.z + e * v4.z;
                        acc.w = acc.w + e * v4.w;
                    }
                }
            }
            float4 oth;
            oth.x = lane_xor_f<32>(acc.x);
            oth.y = lane_xor_f<32>(acc.y);
            oth.z = lane_xor_f<32>(acc.z);
            oth.w = lane_xor_f<32>(acc.w);
            if (half == 0 && l < L4) {
                float4 sum;
                sum.x = acc.x + oth.x;
                sum.y = acc.y + oth.y;
                sum.z = acc.z + oth.z;
                sum.w = acc.w + oth.w;
                *reinterpret_cast<float4*>(red + (wave * kv_mul + i) * HD + 4 * l) = sum;
            }
        }
        __syncthreads();

        for (int idx = tid; idx < kv_mul * HD; idx += 256) {
            const int i = idx / HD, j = idx - i * HD;
            const float o = (red[(0 * kv_mul + i) * HD + j] + red[(1 * kv_mul + i) * HD + j])
                            + (red[(2 * kv_mul + i) * HD + j] + red[(3 * kv_mul + i) * HD + j]);
            const int h = g * kv_mul + i;
            if (multi) {
                float* pp = a.part + ((size_t)h * a.max_chunks + c) * (HD + 2);
                pp[j] = o;
                if (j == 0) {
                    pp[HD] = mc[i];
                    pp[HD + 1] = lc[i];
                }
            } else {
                ofin[i * HD + j] = o / lc[i];
            }
        }
        __syncthreads();
        if (!multi) {
            // q8_quantize of the head outputs (forward.c:291): 2 groups per 128-wide head
            for (int i = wave; i < kv_mul; i += 4) {
                const int h = g * kv_mul + i;
                float4 y = make_float4(0.f, 0.f, 0.f, 0.f);
                if (lane < L4) y = *reinterpret_cast<const float4*>(ofin + i * HD + 4 * lane);
                float scale;
                const int packed = quantize_group16(y, scale);
                if (lane < L4) {
                    reinterpret_cast<int*>(a.oq)[((size_t)h * HD + 4 * lane) >> 2] = packed;
                    if ((lane & 15) == 0) a.os[((size_t)h * HD + 4 * lane) >> 6] = scale;
                    if (a.of) *reinterpret_cast<float4*>(a.of + (size_t)h * HD + 4 * lane) = y;
                }
            }
            __syncthreads();
        }
    }
}

// merge of the chunk partials (q3_numerics.h "attention", last three lines) + quantise
template <int HD>
__global__ __launch_bounds__(64) void k_attn_combine(Attn a) {
    constexpr int L4 = HD / 4;
    const int h = blockIdx.x, lane = threadIdx.x;
    const int T = a.ctl->pos + 1;
    const int nchunks = (T + Q3_ATT_CHUNK - 1) / Q3_ATT_CHUNK;
    const float* base = a.part + (size_t)h * a.max_chunks * (HD + 2);
    float M = base[HD];
    for (int c = 1; c < nchunks; c++) M = fmaxf(M, base[(size_t)c * (HD + 2) + HD]);
    float L = 0.0f;
    float4 A = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int c = 0; c < nchunks; c++) {
        const float* pp = base + (size_t)c * (HD + 2);
        const float w = q3_expf(pp[HD] - M);
        L = L + w * pp[HD + 1];
        if (lane < L4) {
            const float2 o01 = *reinterpret_cast<const float2*>(pp + 4 * lane);
            const float2 o23 = *reinterpret_cast<const float2*>(pp + 4 * lane + 2);
            A.x = A.x + w * o01.x;
            A.y = A.y + w * o01.y;
            A.z = A.z + w * o23.x;
            A.w = A.w + w * o23.y;
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
        reinterpret_cast<int*>(a.oq)[((size_t)h * HD + 4 * lane) >> 2] = packed;
        if ((lane & 15) == 0) a.os[((size_t)h * HD + 4 * lane) >> 6] = scale;
        if (a.of) *reinterpret_cast<float4*>(a.of + (size_t)h * HD + 4 * lane) = y;
    }
}

void attn(const Attn& a, int chunk_slots, bool multi, hipStream_t st) {
    if (a.n_heads / a.n_kv > Q3_MAXG) {
        fprintf(stderr, "[q3hip] attention: more than %d query heads per kv head\n", Q3_MAXG);
        exit(EXIT_FAILURE);
    }
    dim3 grid(a.n_kv, chunk_slots);
    if (a.hd == 128) hipLaunchKernelGGL(k_attn<128>, grid, dim3(256), 0, st, a, multi ? 1 : 0);
    else if (a.hd == 64) hipLaunchKernelGGL(k_attn<64>, grid, dim3(256), 0, st, a, multi ? 1 : 0);
    else {
        fprintf(stderr, "[q3hip] attention: head_dim %d not supported (64 or 128)\n", a.hd);
        exit(EXIT_FAILURE);
    }
}

void attn_combine(const Attn& a, hipStream_t st) {
    if (a.hd == 128) hipLaunchKernelGGL(k_attn_combine<128>, dim3(a.n_heads), dim3(64), 0, st, a);
    else hipLaunchKernelGGL(k_attn_combine<64>, dim3(a.n_heads), dim3(64), 0, st, a);
}

// ------------------------------------------------------------- small ops ---

// x = q*s of one embedding row (reference model.c:201-206 dequantises the whole table
// on the host and forward.c:237 copies a row; the product q*s is the same single rounding)
__global__ void k_embed(const Ctl* ctl, const int8_t* __restrict__ eq, const float* __restrict__ es,
                        int dim, float* __restrict__ x) {
    const size_t base = (size_t)ctl->token * dim;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < dim; i += gridDim.x * blockDim.x) {
        x[i] = (float)eq[base + i] * es[(base + i) >> 6];
    }
}
void embed(const Ctl* ctl, const int8_t* eq, const float* es, int dim, float* x, hipStream_t st) {
    hipLaunchKernelGGL(k_embed, dim3((dim + 255) / 256), dim3(256), 0, st, ctl, eq, es, dim, x);
}

// first index of the maximum (what a strict `>` scan returns), one workgroup
__global__ __launch_bounds__(1024) void k_argmax(const float* __restrict__ logits, int n, int* out,
                                                 Ctl* next) {
    __shared__ float bv[16];
    __shared__ int bi[16];
    float v = -3.4e38f;
    int idx = 0x7fffffff;
    for (int i = threadIdx.x; i < n; i += 1024) {
        const float x = logits[i];
        if (x > v) {
            v = x;
            idx = i;
        }
    }
    for (int m = 32; m >= 1; m >>= 1) {
        const float ov = __shfl_xor(v, m, 64);
        const int oi = __shfl_xor(idx, m, 64);
        if (ov > v || (ov == v && oi < idx)) {
            v = ov;
            idx = oi;
        }
    }
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    if (lane == 0) {
        bv[wave] = v;
        bi[wave] = idx;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int w = 1; w < 16; w++) {
            if (bv[w] > v || (bv[w] == v && bi[w] < idx)) {
                v = bv[w];
                idx = bi[w];
            }
        }
        *out = idx;
        if (next) {
            next->token = idx;
            next->pos = next->pos + 1;
        }
    }
}
void argmax(const float* logits, int n, int* out, Ctl* ctl_next, hipStream_t st) {
    hipLaunchKernelGGL(k_argmax, dim3(1), dim3(1024), 0, st, logits, n, out, ctl_next);
}

__global__ __launch_bounds__(64) void k_rmsnorm(float* out, const float* x, const float* w, int n) {
    const int lane = threadIdx.x;
    const float ss = sum256_sq(x, n, lane);
    const float s = 1.0f / sqrtf(ss / (float)n + 1e-6f);
    for (int i = lane; i < n; i += 64) out[i] = w[i] * (s * x[i]);
}
void rmsnorm(float* out, const float* x, const float* w, int n, hipStream_t st) {
    hipLaunchKernelGGL(k_rmsnorm, dim3(1), dim3(64), 0, st, out, x, w, n);
}

// softmax over an arbitrary length (reference forward.c:34-77): max, q3_expf, SUM256, divide
__global__ __launch_bounds__(1024) void k_softmax(float* x, int n) {
    __shared__ float red[16];
    __shared__ float bc;
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    float m = -3.4e38f;
    for (int i = tid; i < n; i += 1024) m = fmaxf(m, x[i]);
    m = wave_max(m);
    if (lane == 0) red[wave] = m;
    __syncthreads();
    m = red[0];
    for (int w = 1; w < 16; w++) m = fmaxf(m, red[w]);
    for (int i = tid; i < n; i += 1024) x[i] = q3_expf(x[i] - m);
    __syncthreads();
    if (wave == 0) {
        float c0 = 0.f, c1 = 0.f, c2 = 0.f, c3 = 0.f;
        for (int i = 4 * lane; i < n; i += 256) {
            c0 = c0 + x[i];
            if (i + 1 < n) c1 = c1 + x[i + 1];
            if (i + 2 < n) c2 = c2 + x[i + 2];
            if (i + 3 < n) c3 = c3 + x[i + 3];
        }
        const float sum = bfly64((c0 + c1) + (c2 + c3));
        if (lane == 0) bc = sum;
    }
    __syncthreads();
    const float sum = bc;
    for (int i = tid; i < n; i += 1024) x[i] = x[i] / sum;
}
void softmax(float* x, int n, hipStream_t st) {
    hipLaunchKernelGGL(k_softmax, dim3(1), dim3(1024), 0, st, x, n);
}

// stand-alone activation quantisers: run the SAME staging code as the GEMV prologue,
// then spill the LDS image
template <bool NORM>
__global__ __launch_bounds__(256) void k_quantize(const float* x, const float* w, int n, float* normed,
                                                  int8_t* q, float* s) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    int8_t* lq = reinterpret_cast<int8_t*>(smem);
    float* ls = reinterpret_cast<float*>(smem + n);
    float sc = 0.0f;
    if (NORM) {
        const float ss = sum256_sq(x, n, threadIdx.x & 63);
        sc = 1.0f / sqrtf(ss / (float)n + 1e-6f);
    }
    stage_quantize<256>(x, NORM ? w : nullptr, sc, n, lq, ls, normed);
    __syncthreads();
    for (int i = threadIdx.x; i < n; i += 256) q[i] = lq[i];
    for (int g = threadIdx.x; g < (n >> 6); g += 256) s[g] = ls[g];
}
void quantize(const float* x, int n, int8_t* q, float* s, hipStream_t st) {
    hipLaunchKernelGGL(k_quantize<false>, dim3(1), dim3(256), (size_t)n + (n / 64) * 4, st, x, nullptr, n,
                       nullptr, q, s);
}
void rmsnorm_quantize(const float* x, const float* w, int n, float* normed, int8_t* q, float* s,
                      hipStream_t st) {
    hipLaunchKernelGGL(k_quantize<true>, dim3(1), dim3(256), (size_t)n + (n / 64) * 4, st, x, w, n, normed,
                       q, s);
}

__global__ void k_dequantize(const int8_t* q, const float* s, int n, float* x) {
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        x[i] = (float)q[i] * s[i >> 6];
    }
}
void dequantize(const int8_t* q, const float* s, int n, float* x, hipStream_t st) {
    int blocks = (n + 255) / 256;
    if (blocks > 2048) blocks = 2048;
    hipLaunchKernelGGL(k_dequantize, dim3(blocks), dim3(256), 0, st, q, s, n, x);
}

// reference rotary() (forward.c:104-118) on n_heads consecutive heads, (cos,sin) given
__global__ void k_rope_pairs(float* x, int n_heads, int hd, const float* cs) {
    const int half = hd / 2;
    for (int idx = blockIdx.x * blockDim.x + threadIdx.x; idx < n_heads * half; idx += gridDim.x * blockDim.x) {
        const int h = idx / half, i = idx - h * half;
        float* p = x + (size_t)h * hd;
        const float c = cs[2 * i], s = cs[2 * i + 1];
        const float real = p[i], imag = p[i + half];
        p[i] = real * c - imag * s;
        p[i + half] = real * s + imag * c;
    }
}
void rope_pairs(float* x, int n_heads, int hd, const float* cs, hipStream_t st) {
    hipLaunchKernelGGL(k_rope_pairs, dim3(1), dim3(256), 0, st, x, n_heads, hd, cs);
}

template <int HD>
__global__ __launch_bounds__(64) void k_headnorm_rope(float* heads, const float* w, const float* cs) {
    const int lane = threadIdx.x;
    float* p = heads + (size_t)blockIdx.x * HD;
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (lane < HD / 4) v = *reinterpret_cast<const float4*>(p + 4 * lane);
    const float4 r = headnorm_rope_wave<HD>(v, w, cs, lane);
    if (lane < HD / 4) *reinterpret_cast<float4*>(p + 4 * lane) = r;
}
void headnorm_rope(float* heads, int n_heads, int hd, const float* w, const float* cs, hipStream_t st) {
    if (hd == 128) hipLaunchKernelGGL(k_headnorm_rope<128>, dim3(n_heads), dim3(64), 0, st, heads, w, cs);
    else if (hd == 64) hipLaunchKernelGGL(k_headnorm_rope<64>, dim3(n_heads), dim3(64), 0, st, heads, w, cs);
    else {
        fprintf(stderr, "[q3hip] head_dim %d not supported (64 or 128)\n", hd);
        exit(EXIT_FAILURE);
    }
}

__global__ void k_swiglu(const float* g, const float* u, int n, float* out) {
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        const float a = g[i];
        const float sig = 1.0f / (1.0f + q3_expf(-a));
        out[i] = (a * sig) * u[i];
    }
}
void swiglu(const float* g, const float* u, int n, float* out, hipStream_t st) {
    hipLaunchKernelGGL(k_swiglu, dim3((n + 255) / 256), dim3(256), 0, st, g, u, n, out);
}

__global__ void k_expf(const float* x, int n, float* out) {
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) out[i] = q3_expf(x[i]);
}
void expf_map(const float* x, int n, float* out, hipStream_t st) {
    hipLaunchKernelGGL(k_expf, dim3((n + 255) / 256), dim3(256), 0, st, x, n, out);
}

__global__ void k_fill_random(float* p, size_t n, uint64_t seed) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        uint64_t z = seed + i * 0x9E3779B97F4A7C15ull;
        z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
        z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
        z ^= z >> 31;
        p[i] = ((float)(z >> 40) * (1.0f / 16777216.0f) - 0.5f) * 2.0f;
    }
}
void fill_random(float* p, size_t n, uint64_t seed, hipStream_t st) {
    hipLaunchKernelGGL(k_fill_random, dim3(2048), dim3(256), 0, st, p, n, seed);
}

}  // namespace q3k
