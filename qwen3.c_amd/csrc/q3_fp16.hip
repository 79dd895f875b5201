// q3_fp16.hip -- the fp16 CONTRAST path (BASELINE config 5; no counterpart in the reference): every
// Q8_0 matrix is dequantised (q*s, reference q8_dequantize, src/q8.c:32-40) and rounded to binary16
// once at attach, activations stay fp32 and are never quantised.  It exists to put a number next to
// the Q8_0 path: twice the weight bytes per token through the same memory system.
//
// At batch 1 this is still a GEMV -- 1 MAC per 2 weight bytes -- so it is HBM-bound like the Q8_0
// path and f16 MFMA would idle exactly as int8 MFMA would (q3_gemv.hip header); the products are
// formed with v_dot2-free fp32 arithmetic after converting the halves, summed per lane and reduced
// with the wave butterfly.  The kernels are deliberately the simple grid-stride form: the tuned
// wave-role structure of q3_gemv.hip is what the headline path gets.
#include <hip/hip_fp16.h>

#include <cstdio>
#include <cstdlib>

#include "q3_device.hpp"
#include "q3_kernels.hpp"
#include "q3_tile.hpp"

namespace q3k {

typedef _Float16 h8 __attribute__((ext_vector_type(8)));

__global__ void k_to_half(const int8_t* __restrict__ q, const float* __restrict__ s, size_t n, __half* __restrict__ out) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
        out[i] = __float2half_rn((float)q[i] * s[i >> 6]);
}
void to_half(const int8_t* q, const float* s, size_t n, void* out, hipStream_t st) {
    hipLaunchKernelGGL(k_to_half, dim3(4096), dim3(256), 0, st, q, s, n, reinterpret_cast<__half*>(out));
}

__global__ void k_embed_half(const Ctl* ctl, const __half* __restrict__ e, int dim, float* __restrict__ x, int vocab) {
    int token = ctl->token;                       // may come from device memory: keep it a row of the table (k_begin)
    if ((unsigned)token >= (unsigned)vocab) token = 0;
    const size_t base = (size_t)token * dim;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < dim; i += gridDim.x * blockDim.x) x[i] = __half2float(e[base + i]);
}
void embed_half(const Ctl* ctl, const void* e, int dim, float* x, hipStream_t st, int vocab) {
    hipLaunchKernelGGL(k_embed_half, dim3((dim + 255) / 256), dim3(256), 0, st, ctl, reinterpret_cast<const __half*>(e), dim, x, vocab);
}

// out = W x with the activation (rmsnorm'ed when NORM) staged in LDS as fp32.  One wave per row
// (row pair for SWIGLU): lane l takes the 8 halves at k = 512 j + 8 l of every wave-load j.
// (Measured and dropped: requesting the first row's weights before the activation prologue, with one
// wave computing the sum of squares for the others -- 390 against 402 tok/s on 4B shapes.)
template <bool NORM, int EPI>
__global__ __launch_bounds__(512) void k_gemv_f16(const __half* __restrict__ W, int n, int d, const float* __restrict__ x,
                                                  const float* __restrict__ nw, float* __restrict__ out) {
    extern __shared__ __attribute__((aligned(16))) float lx[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, NW = blockDim.x >> 6;
    float sc = 1.0f;
    if (NORM) {
        const float ss = sum256_sq(x, n, lane);          // every wave redundantly
        sc = 1.0f / sqrtf(ss / (float)n + 1e-6f);
    }
    for (int i = 4 * tid; i < n; i += 4 * blockDim.x) {
        float4 v = *reinterpret_cast<const float4*>(x + i);
        if (NORM) {
            const float4 g = *reinterpret_cast<const float4*>(nw + i);
            v.x = g.x * (sc * v.x);
            v.y = g.y * (sc * v.y);
            v.z = g.z * (sc * v.z);
            v.w = g.w * (sc * v.w);
        }
        *reinterpret_cast<float4*>(lx + i) = v;
    }
    __syncthreads();
    constexpr int RP = (EPI == EPI_SWIGLU) ? 2 : 1;       // rows per wave step
    const int nsteps = d / RP;
    for (int rs = blockIdx.x * NW + wave; rs < nsteps; rs += gridDim.x * NW) {
        float acc[RP];
#pragma unroll
        for (int r = 0; r < RP; r++) acc[r] = 0.0f;
        constexpr int U = 8 / RP;                         // eight wave-loads (nt: read once) in flight per wave
        for (int k0 = 8 * lane; k0 < n; k0 += 512 * U) {
            h8 w[RP][U];
#pragma unroll
            for (int u = 0; u < U; u++) {
                const int k = k0 + 512 * u;
#pragma unroll
                for (int r = 0; r < RP; r++) {
                    if (k < n) w[r][u] = __builtin_nontemporal_load(reinterpret_cast<const h8*>(W + (size_t)(rs * RP + r) * n + k));
                }
            }
#pragma unroll
            for (int u = 0; u < U; u++) {
                const int k = k0 + 512 * u;
                if (k < n) {
                    const float4 xa = *reinterpret_cast<const float4*>(lx + k);
                    const float4 xb = *reinterpret_cast<const float4*>(lx + k + 4);
#pragma unroll
                    for (int r = 0; r < RP; r++) {
                        float a = acc[r];
                        a = a + (float)w[r][u][0] * xa.x;
                        a = a + (float)w[r][u][1] * xa.y;
                        a = a + (float)w[r][u][2] * xa.z;
                        a = a + (float)w[r][u][3] * xa.w;
                        a = a + (float)w[r][u][4] * xb.x;
                        a = a + (float)w[r][u][5] * xb.y;
                        a = a + (float)w[r][u][6] * xb.z;
                        a = a + (float)w[r][u][7] * xb.w;
                        acc[r] = a;
                    }
                }
            }
        }
#pragma unroll
        for (int r = 0; r < RP; r++) acc[r] = bfly64(acc[r]);
        if (lane == 0) {
            if (EPI == EPI_SWIGLU) out[rs] = swiglu_pair(acc[0], acc[RP - 1]);
            else if (EPI == EPI_RESID) out[rs] = out[rs] + acc[0];
            else out[rs] = acc[0];
        }
    }
}

// ---- batched form on the matrix cores (prompt ingestion of an fp16-attached model) ---------------
// BASELINE config 5 asks for the fp16 path "on CDNA4 bf16/fp16 MFMA": at batch 1 a GEMV cannot use it, the
// batched prompt pass can.  out[t][r] = W[r][:] . x_t with v_mfma_f32_16x16x32_f16: binary16 weights as at
// batch 1, the activations of the 16..64 positions rounded to binary16 as well (the instruction takes both
// operands in one type; at batch 1 they stay fp32), fp32 accumulation inside the MFMA.  One wave = 16 rows x
// 16 tokens x all of K, four token tiles per workgroup sharing the weight lines through L1 -- the shape of
// k_gemm_q8 without the per-group scales.  Parity: unpinned, like the whole contrast path; checked against
// orc_forward_f16 (double accumulation, fp32 activations) with the activation rounding in the tolerance.
//   A: lane l holds A[row l&15][k = 8*(l>>4) .. +7];  B the same with token l&15
//   D: lane l, register i holds D[row 4*(l>>4) + i][token l&15]
typedef float f4 __attribute__((ext_vector_type(4)));

// (rmsnorm with weight w when given, then) fp32 -> binary16 of `rows` activation rows of n floats
template <bool NORM>
__global__ __launch_bounds__(256) void k_rows_half(const float* __restrict__ x, int ldx, const float* __restrict__ w, int n,
                                                   __half* __restrict__ out) {
    const float* xr = x + (size_t)blockIdx.x * ldx;
    __half* o = out + (size_t)blockIdx.x * n;
    const int lane = threadIdx.x & 63;
    float sc = 1.0f;
    if (NORM) {
        const float ss = sum256_sq(xr, n, lane);          // every wave redundantly
        sc = 1.0f / sqrtf(ss / (float)n + 1e-6f);
    }
    for (int i = threadIdx.x; i < n; i += 256) {
        float v = xr[i];
        if (NORM) v = w[i] * (sc * v);
        o[i] = __float2half_rn(v);
    }
}
void rows_half(const float* x, int ldx, const float* w, int n, int rows, void* out, hipStream_t st) {
    if (w) hipLaunchKernelGGL(k_rows_half<true>, dim3(rows), dim3(256), 0, st, x, ldx, w, n, reinterpret_cast<__half*>(out));
    else hipLaunchKernelGGL(k_rows_half<false>, dim3(rows), dim3(256), 0, st, x, ldx, w, n, reinterpret_cast<__half*>(out));
}

template <int EPI>
__global__ __launch_bounds__(256) void k_gemm_f16(const __half* __restrict__ W, int n, int d, const __half* __restrict__ X, int ntok,
                                                  float* __restrict__ out, int ldo) {
    const int lane = threadIdx.x & 63;
    const int tt = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int r0 = (int)blockIdx.x * 16;
    const int li = lane & 15, kb = lane >> 4;
    const int arow = r0 + li < d ? r0 + li : d - 1;
    const int tok = tt * 16 + li < ntok ? tt * 16 + li : ntok - 1;
    const __amdgpu_buffer_rsrc_t rW = __builtin_amdgcn_make_buffer_rsrc(const_cast<__half*>(W), 0, d * n * 2, 0x00020000);
    const __amdgpu_buffer_rsrc_t rX = __builtin_amdgcn_make_buffer_rsrc(const_cast<__half*>(X), 0, ntok * n * 2, 0x00020000);
    const int va = arow * n * 2 + 16 * kb, vb = tok * n * 2 + 16 * kb;
    const int steps = n >> 5;                               // K = 32 per MFMA
    f4 acc = {0.f, 0.f, 0.f, 0.f};
    v4i pa[4], pb[4], qa[4], qb[4];                          // ping / pong: four K steps computing, the next four in flight
    auto load4 = [&](v4i (&a)[4], v4i (&b)[4], int s0) {
#pragma unroll
        for (int u = 0; u < 4; u++) {
            const int sidx = s0 + u < steps ? s0 + u : steps - 1;
            a[u] = __builtin_amdgcn_raw_buffer_load_b128(rW, va, sidx * 64, 0);
            b[u] = __builtin_amdgcn_raw_buffer_load_b128(rX, vb, sidx * 64, 0);
        }
    };
    auto mac4 = [&](const v4i (&a)[4], const v4i (&b)[4], int s0) {
#pragma unroll
        for (int u = 0; u < 4; u++) {
            if (s0 + u < steps) acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(h8, a[u]), __builtin_bit_cast(h8, b[u]), acc, 0, 0, 0);
        }
    };
    load4(pa, pb, 0);
    for (int s0 = 0; s0 < steps; s0 += 8) {
        if (s0 + 4 < steps) load4(qa, qb, s0 + 4);
        __builtin_amdgcn_sched_barrier(0);
        mac4(pa, pb, s0);
        __builtin_amdgcn_sched_barrier(0);
        if (s0 + 4 >= steps) break;
        if (s0 + 8 < steps) load4(pa, pb, s0 + 8);
        __builtin_amdgcn_sched_barrier(0);
        mac4(qa, qb, s0 + 4);
        __builtin_amdgcn_sched_barrier(0);
    }
    const int tokj = tt * 16 + li;
    if (tokj >= ntok) return;
    const int row = r0 + 4 * kb;
    if (EPI == EPI_SWIGLU) {
        float* o = out + (size_t)tokj * ldo + (row >> 1);
        if (row < d) o[0] = swiglu_pair(acc[0], acc[1]);
        if (row + 2 < d) o[1] = swiglu_pair(acc[2], acc[3]);
    } else {
        float* o = out + (size_t)tokj * ldo + row;
#pragma unroll
        for (int i = 0; i < 4; i++) {
            if (row + i < d) o[i] = (EPI == EPI_RESID) ? o[i] + acc[i] : acc[i];
        }
    }
}
// ---- the same product with both operands staged through LDS (the structure of k_gemm_q8_lds) ----
// A workgroup owns 16*RT rows x 64 tokens and walks K in slabs of 256 halves (512 bytes per row):
// waves 4..7 fetch (whole 128-byte lines, three slabs ahead in three register sets, parked in one of
// two swizzled LDS buffers), waves 0..3 multiply (wave t = token tile t, all RT row tiles: per k-step
// one activation read, RT weight reads, RT MFMAs accumulating in fp32 in the matrix core).  There is
// no reduction-order contract to keep here, so a 16 x 16 tile costs 4 accumulator registers and the
// row tile can be wide.  Persistent: the slabs of a workgroup's row tiles form one stream.
template <int RT> struct HalfSlab {
    v4i a[2 * RT];
    v4i b[8];
};

template <int EPI, int RT>
__global__ __launch_bounds__(512) void k_gemm_f16_lds(const __half* __restrict__ W, int n, int d, const __half* __restrict__ X,
                                                      int ntok, float* __restrict__ out, int ldo, int ntiles) {
    constexpr int R = 16 * RT;
    constexpr int A_BYTES = R * 512, B_BYTES = 64 * 512;
    constexpr int OFF_B = A_BYTES, STAGE = A_BYTES + B_BYTES;
    __shared__ __attribute__((aligned(16))) unsigned char smem[2 * STAGE];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int nslab = n >> 8;                                  // 256 halves per slab
    const int ntok16 = (ntok + 15) & ~15;
    const int mytiles = (ntiles - (int)blockIdx.x + (int)gridDim.x - 1) / (int)gridDim.x;
    const int nq = mytiles * nslab;
    const int rowb = n * 2;                                    // bytes per row

    if (wave >= 4) {
        // ---------------------------------------------------------------- FETCH
        const __amdgpu_buffer_rsrc_t rW = __builtin_amdgcn_make_buffer_rsrc(const_cast<__half*>(W), 0, d * rowb, 0x00020000);
        const __amdgpu_buffer_rsrc_t rX = __builtin_amdgcn_make_buffer_rsrc(const_cast<__half*>(X), 0, ntok * rowb, 0x00020000);
        const int t = tid - 256;
        const int prow = t >> 5, piece = t & 31;
        const int va = prow * rowb + piece * 16;
        int f_tile = (int)blockIdx.x, f_sl = 0;
        auto fetch = [&](HalfSlab<RT>& r) {
            const int sl = f_sl, r0 = f_tile * R;
            if (++f_sl == nslab) { f_sl = 0; f_tile += (int)gridDim.x; }
#pragma unroll
            for (int k = 0; k < 2 * RT; k++) r.a[k] = __builtin_amdgcn_raw_buffer_load_b128(rW, va + k * 8 * rowb, r0 * rowb + sl * 512, 0);
#pragma unroll
            for (int k = 0; k < 8; k++) {
                if (prow + 8 * k < ntok16) r.b[k] = __builtin_amdgcn_raw_buffer_load_b128(rX, va + k * 8 * rowb, sl * 512, 0);
            }
        };
        auto park = [&](const HalfSlab<RT>& r, int buf) {
            unsigned char* base = smem + buf * STAGE;
#pragma unroll
            for (int k = 0; k < 2 * RT; k++) *reinterpret_cast<v4i*>(base + slab_off(prow + 8 * k, piece)) = r.a[k];
#pragma unroll
            for (int k = 0; k < 8; k++) {
                if (prow + 8 * k < ntok16) *reinterpret_cast<v4i*>(base + OFF_B + slab_off(prow + 8 * k, piece)) = r.b[k];
            }
        };
        HalfSlab<RT> S0, S1, S2;
        fetch(S0);
        if (nq > 1) fetch(S1);
        if (nq > 2) fetch(S2);
        for (int q = 0; q < nq; q += 3) {
            park(S0, q & 1);
            if (q + 3 < nq) fetch(S0);
            __syncthreads();
            if (q + 1 >= nq) break;
            park(S1, (q + 1) & 1);
            if (q + 4 < nq) fetch(S1);
            __syncthreads();
            if (q + 2 >= nq) break;
            park(S2, q & 1);
            if (q + 5 < nq) fetch(S2);
            __syncthreads();
        }
        return;
    }

    // -------------------------------------------------------------------- MULTIPLY
    const int tt = wave;
    const int li = lane & 15, kb = lane >> 4;
    const int brow = tt * 16 + li;
    const bool live = tt * 16 < ntok;
    const int tokj = tt * 16 + li;
    f4 acc[RT];
    int q = 0;
    for (int tile = (int)blockIdx.x; tile < ntiles; tile += (int)gridDim.x) {
#pragma unroll
        for (int r = 0; r < RT; r++) acc[r] = f4{0.f, 0.f, 0.f, 0.f};
        for (int sl = 0; sl < nslab; sl++, q++) {
            __syncthreads();                                   // stage q is in LDS buffer q & 1
            if (!live) continue;
            const unsigned char* base = smem + (q & 1) * STAGE;
#pragma unroll
            for (int ks = 0; ks < 8; ks++) {                   // 32 halves = 64 bytes = pieces 4*ks .. 4*ks+3
                const v4i b = *reinterpret_cast<const v4i*>(base + OFF_B + slab_off(brow, 4 * ks + kb));
#pragma unroll
                for (int r = 0; r < RT; r++) {
                    const v4i a = *reinterpret_cast<const v4i*>(base + slab_off(r * 16 + li, 4 * ks + kb));
                    acc[r] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(h8, a), __builtin_bit_cast(h8, b), acc[r], 0, 0, 0);
                }
            }
        }
        if (tokj < ntok) {
#pragma unroll
            for (int r = 0; r < RT; r++) {
                const int row = tile * R + r * 16 + 4 * kb;
                if (EPI == EPI_SWIGLU) {
                    float* o = out + (size_t)tokj * ldo + (row >> 1);
                    if (row < d) o[0] = swiglu_pair(acc[r][0], acc[r][1]);
                    if (row + 2 < d) o[1] = swiglu_pair(acc[r][2], acc[r][3]);
                } else {
                    float* o = out + (size_t)tokj * ldo + row;
#pragma unroll
                    for (int i = 0; i < 4; i++) {
                        if (row + i < d) o[i] = (EPI == EPI_RESID) ? o[i] + acc[r][i] : acc[r][i];
                    }
                }
            }
        }
    }
}

template <int RT>
static void launch_gemm_f16_lds(const __half* w, int n, int d, const __half* x, int ntok, float* out, int ldo, Epi epi, hipStream_t st) {
    static int ncu = 0;
    if (!ncu) {
        int dev = 0;
        hipDeviceProp_t p;
        if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&p, dev) != hipSuccess) {
            fprintf(stderr, "[q3hip] gemm_f16: no device properties\n");
            exit(EXIT_FAILURE);
        }
        ncu = p.multiProcessorCount;
    }
    const int ntiles = (d + 16 * RT - 1) / (16 * RT);
    const dim3 grid(ntiles < ncu ? ntiles : ncu), block(512);
    if (epi == EPI_STORE) hipLaunchKernelGGL((k_gemm_f16_lds<EPI_STORE, RT>), grid, block, 0, st, w, n, d, x, ntok, out, ldo, ntiles);
    else if (epi == EPI_RESID) hipLaunchKernelGGL((k_gemm_f16_lds<EPI_RESID, RT>), grid, block, 0, st, w, n, d, x, ntok, out, ldo, ntiles);
    else hipLaunchKernelGGL((k_gemm_f16_lds<EPI_SWIGLU, RT>), grid, block, 0, st, w, n, d, x, ntok, out, ldo, ntiles);
}

void gemm_f16(const void* W, int n, int d, const void* X, int ntok, float* out, int ldo, Epi epi, hipStream_t st) {
    if (n % 32 || d % 2 || ntok < 1 || ntok > 64 || (long long)d * n * 2 >= (1ll << 31)) {
        fprintf(stderr, "[q3hip] gemm_f16: bad shape (n=%d d=%d tokens=%d)\n", n, d, ntok);
        exit(EXIT_FAILURE);
    }
    const __half* w = reinterpret_cast<const __half*>(W);
    const __half* x = reinterpret_cast<const __half*>(X);
    if (n % 256 == 0) {
        if (d >= 16384) launch_gemm_f16_lds<4>(w, n, d, x, ntok, out, ldo, epi, st);
        else if (d >= 6144) launch_gemm_f16_lds<2>(w, n, d, x, ntok, out, ldo, epi, st);
        else launch_gemm_f16_lds<1>(w, n, d, x, ntok, out, ldo, epi, st);
        return;
    }
    const dim3 grid((d + 15) / 16), block(64 * ((ntok + 15) / 16));
    if (epi == EPI_STORE) hipLaunchKernelGGL(k_gemm_f16<EPI_STORE>, grid, block, 0, st, w, n, d, x, ntok, out, ldo);
    else if (epi == EPI_RESID) hipLaunchKernelGGL(k_gemm_f16<EPI_RESID>, grid, block, 0, st, w, n, d, x, ntok, out, ldo);
    else hipLaunchKernelGGL(k_gemm_f16<EPI_SWIGLU>, grid, block, 0, st, w, n, d, x, ntok, out, ldo);
}

// x rows of a prompt chunk from the binary16 embedding table
__global__ void k_embed_rows_half(const int* __restrict__ tokens, const __half* __restrict__ e, int dim, float* __restrict__ x, int ldx) {
    const size_t base = (size_t)tokens[blockIdx.x] * dim;
    for (int i = threadIdx.x; i < dim; i += blockDim.x) x[(size_t)blockIdx.x * ldx + i] = __half2float(e[base + i]);
}
void embed_rows_half(const int* tokens, int ntok, const void* e, int dim, float* x, int ldx, hipStream_t st) {
    hipLaunchKernelGGL(k_embed_rows_half, dim3(ntok), dim3(256), 0, st, tokens, reinterpret_cast<const __half*>(e), dim, x, ldx);
}

void gemv_f16(const void* W, int n, int d, const float* x, const float* nw, float* out, Epi epi, hipStream_t st) {
    if (n % 8 || (epi == EPI_SWIGLU && d % 2)) {
        fprintf(stderr, "[q3hip] gemv_f16: bad shape (n=%d d=%d)\n", n, d);
        exit(EXIT_FAILURE);
    }
    const __half* w = reinterpret_cast<const __half*>(W);
    const int rows = epi == EPI_SWIGLU ? d / 2 : d;
    int grid = (rows + 7) / 8;
    if (grid > 1024) grid = 1024;
    const size_t lds = (size_t)n * 4;
    const dim3 g(grid), b(512);
    if (nw) {
        if (epi == EPI_SWIGLU) hipLaunchKernelGGL((k_gemv_f16<true, EPI_SWIGLU>), g, b, lds, st, w, n, d, x, nw, out);
        else hipLaunchKernelGGL((k_gemv_f16<true, EPI_STORE>), g, b, lds, st, w, n, d, x, nw, out);
    } else {
        hipLaunchKernelGGL((k_gemv_f16<false, EPI_RESID>), g, b, lds, st, w, n, d, x, nw, out);
    }
}

}  // namespace q3k
