/*
 * q3_numerics.h -- the arithmetic contract of the MI355X forward pass.
 *
 * Every floating-point reduction in the device code uses a FIXED tree that
 * does not depend on grid size, tile shape or scheduling, so that
 *   (1) two runs give bit-identical logits (the reference does not manage
 *       that at >1 OpenMP thread: SURVEY.md section 0.5), and
 *   (2) the CPU port in oracle/q3_oracle.c (mode ORC_TREE) can restate the
 *       same trees and be compared bit-for-bit with the GPU.
 * The reference's own arithmetic (src/forward.c, src/q8.c) differs from this
 * contract ONLY in the order of fp32 additions and in expf (glibc vs q3_expf,
 * both < 1 ulp); every individually rounded operation is the same:
 *
 *   q8 quantise   wmax = max|x|; scale = wmax==0 ? 1e-6f : wmax/127.0f;
 *                 q = (int8) clamp(roundf(x/scale), -127, 127)      [src/q8.c:5-30]
 *                 -> no sum inside: bit-exact with the reference.
 *   matmul        dot_g = exact int32 over one 64-group;
 *                 p_g = ((float)dot_g * ws_g) * xs_g                [src/forward.c:94-96]
 *                 SUM16: col[c] = sum of p_g, g = c, c+16, c+32.. (ascending),
 *                 then butterfly col[c] += col[c^8], ^4, ^2, ^1.
 *   rmsnorm       ss = SUM256(x_i*x_i): P[j] = sum over i = j, j+256, ..;
 *                 lane l = (P[4l]+P[4l+1]) + (P[4l+2]+P[4l+3]);
 *                 butterfly over 64 lanes, xor 32,16,8,4,2,1;
 *                 s = 1.0f/sqrtf(ss/size + 1e-6f); out = w*(s*x)    [src/forward.c:12-28]
 *   rotary        (cos,sin) of pos*powf(1e6,-i/half) from host libm, table
 *                 built once per Model; x' = x*c - y*s, y' = x*s + y*c
 *                 with four separately rounded products              [src/forward.c:104-118]
 *   attention     per 64-position chunk c of the cache:
 *                   s_t = DOT(q,k_t)/sqrtf(hd), DOT: lane l<hd/4 holds
 *                         ((q0k0+q1k1)+q2k2)+q3k3 of elements 4l..4l+3,
 *                         butterfly over 32 lanes xor 16,8,4,2,1
 *                   m_c = max s_t ; e_t = q3_expf(s_t - m_c)
 *                   l_c = butterfly-64 sum of e_t, e_t sitting in lane
 *                         32*(t%2) + (t%64)/2   (how the kernel parks the scores)
 *                   O_c[j] = two streams: a_h[j] += e_t*v_t[j] for t%2==h,
 *                            ascending t; O_c = a_0 + a_1
 *                 M = max m_c ; w_c = q3_expf(m_c - M)
 *                 L = sum_c w_c*l_c ; A[j] = sum_c w_c*O_c[j] (ascending c)
 *                 out[j] = A[j]/L                                    [src/forward.c:141-195]
 *   swiglu        (x1 * (1.0f/(1.0f+q3_expf(-x1)))) * x3             [src/forward.c:122-139]
 *   softmax       m = max x_i ; e_i = q3_expf(x_i - m) ;
 *                 sum = SUM256(e) when size <= Q3_SM_CHUNK, else the sums SUM256(e[c]) of the
 *                 consecutive Q3_SM_CHUNK-element chunks added in ascending order of c (one
 *                 workgroup per chunk on the device) ; out_i = e_i/sum   [src/forward.c:34-77]
 *   sample        logits/temperature (true division), softmax as above, then the reference's own
 *                 sequential arithmetic on the stably sorted distribution (src/sampler.c:88-136)
 *   residual add, embedding dequant (q*s): single rounded ops, exact.
 *
 * No fused multiply-add is ever formed implicitly: device code is built with
 * -ffp-contract=off (the x86-64 golden build of the reference has no FMA
 * either).  q3_expf uses explicit fmaf, which is exact on both sides.
 */
#ifndef Q3_NUMERICS_H
#define Q3_NUMERICS_H

#include <stdint.h>

#if defined(__HIPCC__) || defined(__HIP__)
#define Q3_HD __host__ __device__ static inline
#else
#define Q3_HD static inline
#endif

#define Q3_ATT_CHUNK 64      /* cache positions per attention chunk          */
#define Q3_ATT_STREAMS 2     /* interleaved accumulation streams per chunk   */
#define Q3_MM_COLS 16        /* column partials of the matmul group sum      */
#define Q3_SM_CHUNK 4096     /* elements per partial sum of a long softmax   */

Q3_HD float q3_bits_to_float(uint32_t u) {
    union { uint32_t u; float f; } c; c.u = u; return c.f;
}

/*
 * expf with one code path for host and device: round-to-nearest range
 * reduction by the 1.5*2^23 trick, two-constant ln2, degree-6 polynomial
 * (Cephes coefficients), exact power-of-two scaling in two steps.
 * Max error < 1 ulp on [-86, 88.7]; below -86 the result is 0 (no denormals
 * are produced on either side), above 88.72283 it is +inf.
 * Restates what the reference gets from glibc expf at src/forward.c:57,66,125.
 */
Q3_HD float q3_expf(float x) {
    if (!(x <= 88.72283f)) {
        return (x != x) ? x : q3_bits_to_float(0x7f800000u);
    }
    if (x < -86.0f) {
        return 0.0f;
    }
    const float t = x * 1.44269504088896341f;
    const float n = (t + 12582912.0f) - 12582912.0f;
    float r = __builtin_fmaf(n, -0.693359375f, x);
    r = __builtin_fmaf(n, 2.12194440e-4f, r);
    float p = 1.9875691500e-4f;
    p = __builtin_fmaf(p, r, 1.3981999507e-3f);
    p = __builtin_fmaf(p, r, 8.3334519073e-3f);
    p = __builtin_fmaf(p, r, 4.1665795894e-2f);
    p = __builtin_fmaf(p, r, 1.6666665459e-1f);
    p = __builtin_fmaf(p, r, 5.0000001201e-1f);
    const float r2 = r * r;
    float y = __builtin_fmaf(p, r2, r);
    y = y + 1.0f;
    const int ni = (int)n;
    const int n1 = ni >> 1;          /* arithmetic shift: floor(ni/2) */
    const int n2 = ni - n1;
    const float s1 = q3_bits_to_float((uint32_t)(n1 + 127) << 23);
    const float s2 = q3_bits_to_float((uint32_t)(n2 + 127) << 23);
    return (y * s1) * s2;
}

/* Q8_0 activation quantisation of one value (reference src/q8.c:26-27). */
Q3_HD float q3_q8_scale(float wmax) {
    return (wmax == 0.0f) ? 1e-6f : (wmax / 127.0f);
}

#endif /* Q3_NUMERICS_H */
