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

__global__ void k_to_half(const int8_t* __restrict__ q, const float* __restrict__ s, size_t n, __half* __restrict__ out) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
        out[i] = __float2half_rn((float)q[i] * s[i >> 6]);
}
void to_half(const int8_t* q, const float* s, size_t n, void* out, hipStream_t st) {
    hipLaunchKernelGGL(k_to_half, dim3(4096), dim3(256), 0, st, q, s, n, reinterpret_cast<__half*>(out));
}

__global__ void k_embed_half(const Ctl* ctl, const __half* __restrict__ e, int dim, float* __restrict__ x) {
    const size_t base = (size_t)ctl->token * dim;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < dim; i += gridDim.x * blockDim.x) x[i] = __half2float(e[base + i]);
}
void embed_half(const Ctl* ctl, const void* e, int dim, float* x, hipStream_t st) {
    hipLaunchKernelGGL(k_embed_half, dim3((dim + 255) / 256), dim3(256), 0, st, ctl, reinterpret_cast<const __half*>(e), dim, x);
}

typedef _Float16 h8 __attribute__((ext_vector_type(8)));

// out = W x with the activation (rmsnorm'ed when NORM) staged in LDS as fp32.  One wave per row
// (row pair for SWIGLU): lane l takes the 8 halves at k = 512 j + 8 l of every wave-load j.
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
        for (int k0 = 8 * lane; k0 < n; k0 += 2048) {     // four wave-loads in flight per row
            h8 w[RP][4];
#pragma unroll
            for (int u = 0; u < 4; u++) {
                const int k = k0 + 512 * u;
#pragma unroll
                for (int r = 0; r < RP; r++) {
                    if (k < n) w[r][u] = *reinterpret_cast<const h8*>(W + (size_t)(rs * RP + r) * n + k);
                }
            }
#pragma unroll
            for (int u = 0; u < 4; u++) {
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
