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

static int g_cu_count = 0;
void set_cu_count(int n) { if (n > 0) g_cu_count = n; }
int cu_count() {
    if (!g_cu_count) {           // (ops called before any Model was attached)
        int dev = 0;
        hipDeviceProp_t p;
        if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&p, dev) != hipSuccess) {
            fprintf(stderr, "[q3hip] no device properties\n");
            exit(EXIT_FAILURE);
        }
        g_cu_count = p.multiProcessorCount;
    }
    return g_cu_count;
}

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

// first index of the maximum (what a strict `>` scan returns), in two small launches:
// 128 workgroups reduce slices to (value, index) pairs, one wave picks among those
__device__ __forceinline__ void argmax_merge(float& v, int& idx, float ov, int oi) {
    if (ov > v || (ov == v && oi < idx)) {
        v = ov;
        idx = oi;
    }
}
#define Q3_ARGMAX_WGS 128
__global__ __launch_bounds__(256) void k_argmax_part(const float* __restrict__ logits, int n, float* pv, int* pi) {
    __shared__ float bv[4];
    __shared__ int bi[4];
    float v = -3.4e38f;
    int idx = 0x7fffffff;
    for (int i = blockIdx.x * 256 + threadIdx.x; i < n; i += Q3_ARGMAX_WGS * 256) {
        const float x = logits[i];
        if (x > v) {        // ascending i per thread: strict > keeps the first
            v = x;
            idx = i;
        }
    }
    for (int m = 32; m >= 1; m >>= 1) argmax_merge(v, idx, __shfl_xor(v, m, 64), __shfl_xor(idx, m, 64));
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    if (lane == 0) {
        bv[wave] = v;
        bi[wave] = idx;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int w = 1; w < 4; w++) argmax_merge(v, idx, bv[w], bi[w]);
        pv[blockIdx.x] = v;
        pi[blockIdx.x] = idx;
    }
}
__global__ __launch_bounds__(64) void k_argmax_final(const float* pv, const int* pi, int n, int* out, int* out2) {
    const int lane = threadIdx.x;
    float v = pv[lane];
    int idx = pi[lane];
    argmax_merge(v, idx, pv[lane + 64], pi[lane + 64]);
    for (int m = 32; m >= 1; m >>= 1) argmax_merge(v, idx, __shfl_xor(v, m, 64), __shfl_xor(idx, m, 64));
    if (lane == 0) {
        // no logit compared greater than anything (all NaN: a broken checkpoint): the pick feeds the next step's embedding
        // fetch on the device, so it must stay a valid token id
        if ((unsigned)idx >= (unsigned)n) idx = 0;
        *out = idx;
        if (out2) *out2 = idx;
    }
}
void argmax(const float* logits, int n, float* scratch /* 2*128 words */, int* out, int* out2, hipStream_t st) {
    float* pv = scratch;
    int* pi = reinterpret_cast<int*>(scratch + Q3_ARGMAX_WGS);
    hipLaunchKernelGGL(k_argmax_part, dim3(Q3_ARGMAX_WGS), dim3(256), 0, st, logits, n, pv, pi);
    hipLaunchKernelGGL(k_argmax_final, dim3(1), dim3(64), 0, st, pv, pi, n, out, out2);
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

// SUM256 of n floats (any n >= 1) by ONE wave: lane l, component k walks x[256 b + 4 l + k]
__device__ __forceinline__ float sum256_wave(const float* x, int n, int lane) {
    float c0 = 0.f, c1 = 0.f, c2 = 0.f, c3 = 0.f;
    for (int i = 4 * lane; i < n; i += 256) {
        c0 = c0 + x[i];
        if (i + 1 < n) c1 = c1 + x[i + 1];
        if (i + 2 < n) c2 = c2 + x[i + 2];
        if (i + 3 < n) c3 = c3 + x[i + 3];
    }
    return bfly64((c0 + c1) + (c2 + c3));
}

// softmax over an arbitrary length (reference forward.c:34-77; tree: q3_numerics.h "softmax"):
// max, q3_expf, chunk sums by the 16 waves, sequential sum of the chunk sums, divide.  One workgroup
// (the exported softmax() is a host-pointer convenience; the sampler has its own multi-workgroup form).
__global__ __launch_bounds__(1024) void k_softmax(float* x, int n) {
    __shared__ float red[16];
    __shared__ float ps[1024];          // chunk sums: n <= 4 Mi elements
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    float m = -3.4e38f;
    for (int i = tid; i < n; i += 1024) m = fmaxf(m, x[i]);
    m = wave_max(m);
    if (lane == 0) red[wave] = m;
    __syncthreads();
    m = red[0];
    for (int w = 1; w < 16; w++) m = fmaxf(m, red[w]);
    for (int i = tid; i < n; i += 1024) x[i] = q3_expf(x[i] - m);
    __threadfence_block();
    __syncthreads();
    const int nchunks = (n + Q3_SM_CHUNK - 1) / Q3_SM_CHUNK;
    for (int c = wave; c < nchunks; c += 16) {
        const int c0 = c * Q3_SM_CHUNK;
        const float s = sum256_wave(x + c0, n - c0 < Q3_SM_CHUNK ? n - c0 : Q3_SM_CHUNK, lane);
        if (lane == 0) ps[c] = s;
    }
    __syncthreads();
    float sum = ps[0];
    if (nchunks > 1) {
        sum = 0.0f;
        for (int c = 0; c < nchunks; c++) sum = sum + ps[c];
    }
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
