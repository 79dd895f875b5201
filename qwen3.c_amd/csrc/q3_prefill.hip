// q3_prefill.hip -- batched prompt ingestion: up to 64 prompt positions go
// through a layer together, so every weight byte is read once per 64 tokens instead of once per
// token, and the Q8_0 product runs on the matrix cores.
//
//   k_rows_quantize   the GEMV prologue (rmsnorm + q8_quantize, or q8_quantize alone) for B
//                     activation rows at once: one workgroup per row
//   k_gemm_q8         out[t][r] = W[r][:] . x_t for 16 rows x 16 tokens per wave with
//                     v_mfma_i32_16x16x64_i8: K = 64 is exactly one Q8_0 group, so ONE MFMA
//                     yields the exact int32 group dots of 256 (row, token) pairs; each is then
//                     scaled ((float)dot * ws) * xs and accumulated in the SUM16 tree of
//                     q3_numerics.h -- the same operations in the same order as the decode GEMV,
//                     which is why a prefilled prompt leaves bit-identical logits and KV cache
//                     (tests/test_gpu_prefill.py)
//
// Operand maps of v_mfma_i32_16x16x64_i8 (checked with exact integer data by the op test):
//   A: lane l holds A[row l&15][k = 16*(l>>4) .. +15]  (16 bytes);  B the same with col l&15
//   D: lane l, register i holds D[row 4*(l>>4) + i][col l&15]
// W rows are the A operand, the quantised activations of the tokens the B operand, so a lane ends
// up with 4 consecutive output rows of ONE token: one 16-byte store.
#include <cstdio>
#include <cstdlib>

#include "q3_device.hpp"
#include "q3_kernels.hpp"
#include "q3_tile.hpp"

namespace q3k {

// ---- B rows: (rmsnorm +) q8_quantize ------------------------------------------------------
template <bool NORM>
__global__ __launch_bounds__(256) void k_rows_quantize(const float* __restrict__ x, int ldx, const float* __restrict__ w,
                                                       int n, int8_t* __restrict__ q, float* __restrict__ s) {
    const float* xr = x + (size_t)blockIdx.x * ldx;
    int8_t* qr = q + (size_t)blockIdx.x * n;
    float* sr = s + (size_t)blockIdx.x * (n >> 6);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    float sc = 0.0f;
    if (NORM) {
        const float ss = sum256_sq(xr, n, lane);          // every wave redundantly
        sc = 1.0f / sqrtf(ss / (float)n + 1e-6f);
    }
    for (int base = wave * 256; base < n; base += 4 * 256) {
        const int i = base + 4 * lane;
        const bool act = i < n;
        float4 y = make_float4(0.f, 0.f, 0.f, 0.f);
        if (act) {
            const float4 v = *reinterpret_cast<const float4*>(xr + i);
            if (NORM) {
                const float4 g = *reinterpret_cast<const float4*>(w + i);
                y.x = g.x * (sc * v.x);
                y.y = g.y * (sc * v.y);
                y.z = g.z * (sc * v.z);
                y.w = g.w * (sc * v.w);
            } else {
                y = v;
            }
        }
        float scale;
        const int packed = quantize_group16(y, scale);
        if (act) {
            reinterpret_cast<int*>(qr)[i >> 2] = packed;
            if ((lane & 15) == 0) sr[i >> 6] = scale;
        }
    }
}

void rows_quantize(const float* x, int ldx, const float* w, int n, int rows, int8_t* q, float* s, hipStream_t st) {
    if (w) hipLaunchKernelGGL(k_rows_quantize<true>, dim3(rows), dim3(256), 0, st, x, ldx, w, n, q, s);
    else hipLaunchKernelGGL(k_rows_quantize<false>, dim3(rows), dim3(256), 0, st, x, ldx, w, n, q, s);
}

// ---- Q8_0 GEMM on int8 MFMA ----------------------------------------------------------------
typedef int v4i32 __attribute__((ext_vector_type(4)));

// One workgroup = one tile of 16 rows x 16 tokens; its 4 waves split the groups by SUM16 column:
// wave w owns the columns 4w..4w+3, i.e. groups 16b + 4w .. 16b + 4w + 3 of every block b of sixteen
// groups -- four consecutive groups, so the group scales come as one float4 per row.  The sixteen
// column sums then meet in LDS and wave 0 runs the butterfly: the same additions in the same order
// as one wave doing it all, with four times the waves streaming the matrix.
template <int EPI, int NT>            // NT = token tiles of 16 per workgroup: the weights are loaded once for all of them
__global__ __launch_bounds__(256, NT == 1 ? 4 : (NT == 2 ? 2 : 1)) void k_gemm_q8(const int8_t* __restrict__ W, const float* __restrict__ S, int n,
                                                                  int d, const int8_t* __restrict__ xq,
                                                                  const float* __restrict__ xs, int ntok,
                                                                  float* __restrict__ out, int ldo) {
    __shared__ float cols[NT][16][4][64];       // [token tile][column][output register][lane]
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int r0 = (int)blockIdx.x * 16;
    const int ng = n >> 6;
    const int li = lane & 15, kb = lane >> 4;
    // A: row r0 + li (clamped: rows >= d are computed and dropped), 16 bytes at k-block kb of each group
    const int arow = r0 + li < d ? r0 + li : d - 1;
    const int8_t* ap = W + (size_t)arow * n + 16 * kb;
    // B: token 16j + li of tile j (tokens >= ntok: clamped, dropped at the store)
    const int8_t* bp[NT];
    const float* xsp[NT];
#pragma unroll
    for (int j = 0; j < NT; j++) {
        const int tok = 16 * j + li < ntok ? 16 * j + li : ntok - 1;
        bp[j] = xq + (size_t)tok * n + 16 * kb;
        xsp[j] = xs + (size_t)tok * ng;
    }
    const float* wsp[4];                         // scales of the 4 output rows of this lane
#pragma unroll
    for (int i = 0; i < 4; i++) {
        const int r = r0 + 4 * kb + i < d ? r0 + 4 * kb + i : d - 1;
        wsp[i] = S + (size_t)r * ng;
    }
    float col[NT][4][4];                         // [token tile][column 4w + c][output register]
#pragma unroll
    for (int j = 0; j < NT; j++)
#pragma unroll
        for (int c = 0; c < 4; c++)
#pragma unroll
            for (int i = 0; i < 4; i++) col[j][c][i] = 0.0f;

    for (int g0 = 4 * wave; g0 < ng; g0 += 16) {             // groups g0 .. g0+3
        v4i a[4], b[NT][4];
        float4 sw[4], sx[NT];
#pragma unroll
        for (int c = 0; c < 4; c++) {
            const int g = g0 + c < ng ? g0 + c : ng - 1;
            a[c] = *reinterpret_cast<const v4i*>(ap + (size_t)g * 64);
#pragma unroll
            for (int j = 0; j < NT; j++) b[j][c] = *reinterpret_cast<const v4i*>(bp[j] + (size_t)g * 64);
        }
        if (g0 + 3 < ng && (ng & 3) == 0) {                   // 16-byte aligned scale quads
#pragma unroll
            for (int i = 0; i < 4; i++) sw[i] = *reinterpret_cast<const float4*>(wsp[i] + g0);
#pragma unroll
            for (int j = 0; j < NT; j++) sx[j] = *reinterpret_cast<const float4*>(xsp[j] + g0);
        } else {
            const int g1 = g0 < ng ? g0 : ng - 1, g2 = g0 + 1 < ng ? g0 + 1 : ng - 1;
            const int g3 = g0 + 2 < ng ? g0 + 2 : ng - 1, g4 = g0 + 3 < ng ? g0 + 3 : ng - 1;
#pragma unroll
            for (int i = 0; i < 4; i++) sw[i] = make_float4(wsp[i][g1], wsp[i][g2], wsp[i][g3], wsp[i][g4]);
#pragma unroll
            for (int j = 0; j < NT; j++) sx[j] = make_float4(xsp[j][g1], xsp[j][g2], xsp[j][g3], xsp[j][g4]);
        }
#pragma unroll
        for (int c = 0; c < 4; c++) {
            if (g0 + c < ng) {                                // wave-uniform
#pragma unroll
                for (int j = 0; j < NT; j++) {
                    const v4i32 zero = {0, 0, 0, 0};
                    const v4i32 dsum = __builtin_amdgcn_mfma_i32_16x16x64_i8(a[c], b[j][c], zero, 0, 0, 0);
                    const float sxc = c == 0 ? sx[j].x : (c == 1 ? sx[j].y : (c == 2 ? sx[j].z : sx[j].w));
#pragma unroll
                    for (int i = 0; i < 4; i++) {
                        const float swc = c == 0 ? sw[i].x : (c == 1 ? sw[i].y : (c == 2 ? sw[i].z : sw[i].w));
                        const float p = ((float)dsum[i] * swc) * sxc;
                        col[j][c][i] = col[j][c][i] + p;
                    }
                }
            }
        }
    }
#pragma unroll
    for (int j = 0; j < NT; j++)
#pragma unroll
        for (int c = 0; c < 4; c++)
#pragma unroll
            for (int i = 0; i < 4; i++) cols[j][4 * wave + c][i][lane] = col[j][c][i];
    __syncthreads();
    if (wave >= NT) return;
    // wave j finishes token tile j: butterfly col[c] += col[c^8], ^4, ^2, ^1
    const int j = wave;
    float res[4];
#pragma unroll
    for (int i = 0; i < 4; i++) {
        float t8[8], t4[4], t2[2];
#pragma unroll
        for (int c = 0; c < 8; c++) t8[c] = cols[j][c][i][lane] + cols[j][c + 8][i][lane];
#pragma unroll
        for (int c = 0; c < 4; c++) t4[c] = t8[c] + t8[c + 4];
#pragma unroll
        for (int c = 0; c < 2; c++) t2[c] = t4[c] + t4[c + 2];
        res[i] = t2[0] + t2[1];
    }
    const int tokj = 16 * j + li;
    if (tokj >= ntok) return;
    const int row = r0 + 4 * kb;
    if (EPI == EPI_SWIGLU) {
        // rows are interleaved (gate_i, up_i): two outputs per lane
        float* o = out + (size_t)tokj * ldo + (row >> 1);
        if (row < d) o[0] = swiglu_pair(res[0], res[1]);
        if (row + 2 < d) o[1] = swiglu_pair(res[2], res[3]);
    } else {
        float* o = out + (size_t)tokj * ldo + row;
#pragma unroll
        for (int i = 0; i < 4; i++) {
            if (row + i < d) o[i] = (EPI == EPI_RESID) ? o[i] + res[i] : res[i];
        }
    }
}

template <int NT>
static void launch_gemm(const int8_t* W, const float* S, int n, int d, const int8_t* xq, const float* xs, int ntok,
                        float* out, int ldo, Epi epi, hipStream_t st) {
    const dim3 grid((d + 15) / 16), block(256);
    if (epi == EPI_STORE) hipLaunchKernelGGL((k_gemm_q8<EPI_STORE, NT>), grid, block, 0, st, W, S, n, d, xq, xs, ntok, out, ldo);
    else if (epi == EPI_RESID) hipLaunchKernelGGL((k_gemm_q8<EPI_RESID, NT>), grid, block, 0, st, W, S, n, d, xq, xs, ntok, out, ldo);
    else hipLaunchKernelGGL((k_gemm_q8<EPI_SWIGLU, NT>), grid, block, 0, st, W, S, n, d, xq, xs, ntok, out, ldo);
}
void gemm_q8(const int8_t* W, const float* S, int n, int d, const int8_t* xq, const float* xs, int ntok, float* out,
             int ldo, Epi epi, hipStream_t st) {
    if (n % 64 || d % 2 || ntok < 1 || ntok > 64) {
        fprintf(stderr, "[q3hip] gemm_q8: bad shape (n=%d d=%d tokens=%d)\n", n, d, ntok);
        exit(EXIT_FAILURE);
    }
    if (ntok <= 16) launch_gemm<1>(W, S, n, d, xq, xs, ntok, out, ldo, epi, st);
    else if (ntok <= 32) launch_gemm<2>(W, S, n, d, xq, xs, ntok, out, ldo, epi, st);
    else launch_gemm<4>(W, S, n, d, xq, xs, ntok, out, ldo, epi, st);
}

// ---- per-token bookkeeping of a prefill chunk ----------------------------------------------
// x row of every token (embedding dequant, reference model.c:201-206 / forward.c:237), its Ctl and
// its (cos,sin) row
__global__ __launch_bounds__(256) void k_prefill_begin(const int* __restrict__ tokens, int pos0, const int8_t* __restrict__ eq,
                                                       const float* __restrict__ es, int dim, float* __restrict__ x, int ldx,
                                                       const float* __restrict__ rope, int hd, float* __restrict__ cs, Ctl* ctl) {
    const int t = blockIdx.x;
    const int tok = tokens[t];
    if (threadIdx.x == 0) {
        ctl[t].token = tok;
        ctl[t].pos = pos0 + t;
    }
    for (int i = threadIdx.x; i < hd; i += 256) cs[(size_t)t * hd + i] = rope[(size_t)(pos0 + t) * hd + i];
    if (eq) {
        const size_t base = (size_t)tok * dim;
        for (int i = threadIdx.x; i < dim; i += 256) x[(size_t)t * ldx + i] = (float)eq[base + i] * es[(base + i) >> 6];
    }
}
void prefill_begin(const int* tokens, int ntok, int pos0, const int8_t* eq, const float* es, int dim, float* x, int ldx,
                   const float* rope, int hd, float* cs, Ctl* ctl, hipStream_t st) {
    hipLaunchKernelGGL(k_prefill_begin, dim3(ntok), dim3(256), 0, st, tokens, pos0, eq, es, dim, x, ldx, rope, hd, cs, ctl);
}

}  // namespace q3k
