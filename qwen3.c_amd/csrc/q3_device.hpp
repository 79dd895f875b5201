// q3_device.hpp -- wave64 building blocks shared by the gfx950 kernels.
// Arithmetic contract: q3_numerics.h.  All cross-lane traffic uses the cheapest
// gfx950 instruction that realises the contract's pairing:
//   xor 1, 2      DPP quad_perm (no LDS hardware involved)
//   xor 4         two DPP row rotations under bank masks
//   xor 8         DPP row_ror:8
//   xor 16, 32    v_permlane16_swap / v_permlane32_swap
// (until round 4 xor 4, 8, 16 went through ds_swizzle_b32: the LDS crossbar, ~100 cycles per dependent level)
// (hipcc lowers __shfl_xor to ds_bpermute_b32 with a computed address for all of them.)
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "q3_numerics.h"

namespace q3k {

typedef int v4i __attribute__((ext_vector_type(4)));

template <int M>
__device__ __forceinline__ int lane_xor_i(int v) {
    static_assert(M == 1 || M == 2 || M == 4 || M == 8 || M == 16 || M == 32, "xor mask");
    if constexpr (M == 1) {
        return __builtin_amdgcn_mov_dpp(v, 0xB1, 0xF, 0xF, true);   // quad_perm [1,0,3,2]
    } else if constexpr (M == 2) {
        return __builtin_amdgcn_mov_dpp(v, 0x4E, 0xF, 0xF, true);   // quad_perm [2,3,0,1]
    } else if constexpr (M == 32) {
        auto r = __builtin_amdgcn_permlane32_swap((unsigned)v, (unsigned)v, false, false);
        return (int)((threadIdx.x & 32) ? r[0] : r[1]);
#ifndef Q3_XOR_SWIZZLE          // (round 4: no trip through the LDS crossbar for xor 16 / 8 / 4 either -- ~100 cycles of
                                //  latency per dependent butterfly level against ~10; -DQ3_XOR_SWIZZLE restores ds_swizzle)
    } else if constexpr (M == 16) {
        auto r = __builtin_amdgcn_permlane16_swap((unsigned)v, (unsigned)v, false, false);     // r[0] = rows 0,0,2,2; r[1] = rows 1,1,3,3
        return (int)((threadIdx.x & 16) ? r[0] : r[1]);
    } else if constexpr (M == 8) {
        return __builtin_amdgcn_update_dpp(v, v, 0x128, 0xF, 0xF, false);      // row_ror:8 (lane i of a row of 16 gets lane i ^ 8)
    } else {
        // xor 4 = rotate a row of 16 by 12 in the banks whose lanes have bit 2 clear (0, 2), by 4 in the others (1, 3)
        int t = __builtin_amdgcn_update_dpp(v, v, 0x12C, 0xF, 0x5, false);
        return __builtin_amdgcn_update_dpp(t, v, 0x124, 0xF, 0xA, false);
    }
#else
    } else {
        return __builtin_amdgcn_ds_swizzle(v, (M << 10) | 0x1F);
    }
#endif
}
template <int M>
__device__ __forceinline__ float lane_xor_f(float v) {
    return __int_as_float(lane_xor_i<M>(__float_as_int(v)));
}

// butterfly over the 64 lanes of a wave: xor 32,16,8,4,2,1
__device__ __forceinline__ float bfly64(float v) {
    v = v + lane_xor_f<32>(v);
    v = v + lane_xor_f<16>(v);
    v = v + lane_xor_f<8>(v);
    v = v + lane_xor_f<4>(v);
    v = v + lane_xor_f<2>(v);
    v = v + lane_xor_f<1>(v);
    return v;
}
// butterfly inside each 32-lane half: xor 16,8,4,2,1
__device__ __forceinline__ float bfly32(float v) {
    v = v + lane_xor_f<16>(v);
    v = v + lane_xor_f<8>(v);
    v = v + lane_xor_f<4>(v);
    v = v + lane_xor_f<2>(v);
    v = v + lane_xor_f<1>(v);
    return v;
}
// SUM16 butterfly over the 16 quads of a wave (all 4 lanes of a quad hold the same
// value): xor 8,4,2,1 on the quad index = xor 32,16,8,4 on the lane
__device__ __forceinline__ float bfly_quads(float v) {
    v = v + lane_xor_f<32>(v);
    v = v + lane_xor_f<16>(v);
    v = v + lane_xor_f<8>(v);
    v = v + lane_xor_f<4>(v);
    return v;
}
// max over the 64 lanes, in every lane.  max is exact in any order, so the cheapest pairing
// will do: DPP inside each row of 16, then two row broadcasts and a readlane -- no LDS crossbar.
__device__ __forceinline__ float wave_max(float v) {
    v = fmaxf(v, lane_xor_f<1>(v));
    v = fmaxf(v, lane_xor_f<2>(v));
    v = fmaxf(v, __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(v), 0x141, 0xF, 0xF, true)));   // row_half_mirror
    v = fmaxf(v, __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(v), 0x140, 0xF, 0xF, true)));   // row_mirror
    // every lane now holds the max of its row of 16; rows 0..3 -> four readlanes
    const float r0 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 0));
    const float r1 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 16));
    const float r2 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 32));
    const float r3 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 48));
    return fmaxf(fmaxf(r0, r1), fmaxf(r2, r3));
}

// SUM256 of x_i^2 over n floats (n % 4 == 0), computed by ONE wave.
__device__ __forceinline__ float sum256_sq(const float* __restrict__ x, int n, int lane) {
    float c0 = 0.0f, c1 = 0.0f, c2 = 0.0f, c3 = 0.0f;
    const float4* x4 = reinterpret_cast<const float4*>(x);
    // eight loads in flight per round (a load per iteration costs a memory round trip each); slots
    // past the end hold zeros, and c + 0*0 == c exactly
    for (int i0 = 4 * lane; i0 < n; i0 += 8 * 256) {
        float4 v[8];
#pragma unroll
        for (int k = 0; k < 8; k++) {
            const int i = i0 + 256 * k;
            v[k] = i < n ? x4[i >> 2] : make_float4(0.f, 0.f, 0.f, 0.f);
        }
#pragma unroll
        for (int k = 0; k < 8; k++) {
            c0 = c0 + v[k].x * v[k].x;
            c1 = c1 + v[k].y * v[k].y;
            c2 = c2 + v[k].z * v[k].z;
            c3 = c3 + v[k].w * v[k].w;
        }
    }
    return bfly64((c0 + c1) + (c2 + c3));
}

// clamp(roundf(x / scale), +-127) (reference src/q8.c:26-27), the slow exact way
__device__ __forceinline__ int q8_code_exact(float x, float scale) {
    return (int)fminf(fmaxf(roundf(x / scale), -127.0f), 127.0f);
}

// The same for four values, packed little-endian, without paying for IEEE divisions:
// r = x * rcp(scale) is within ~2 ulp (< 4e-5 absolute, |r| <= 127) of the correctly rounded
// quotient, so unless |r| lies within 1e-3 of a half-integer both round to the same integer,
// which away from those points is floor(|r| + 0.5) with the sign of x.  Only a lane that holds
// such a near-tie (fract(|r| + 0.5) within 1e-3 of 0 or 1) carries out the true divisions.
// The result is the reference's in every case.
__device__ __forceinline__ int q8_pack4(float4 y, float scale, float inv) {
    const float t0 = fabsf(y.x * inv) + 0.5f, t1 = fabsf(y.y * inv) + 0.5f;
    const float t2 = fabsf(y.z * inv) + 0.5f, t3 = fabsf(y.w * inv) + 0.5f;
    const float g0 = fabsf(__builtin_amdgcn_fractf(t0) - 0.5f), g1 = fabsf(__builtin_amdgcn_fractf(t1) - 0.5f);
    const float g2 = fabsf(__builtin_amdgcn_fractf(t2) - 0.5f), g3 = fabsf(__builtin_amdgcn_fractf(t3) - 0.5f);
    int q0, q1, q2, q3;
    // (a denormal scale makes rcp overflow: those groups take the exact path as well)
    if (__builtin_expect(fmaxf(fmaxf(g0, g1), fmaxf(g2, g3)) > 0.499f || !(scale >= 1e-30f), 0)) {
        q0 = q8_code_exact(y.x, scale);
        q1 = q8_code_exact(y.y, scale);
        q2 = q8_code_exact(y.z, scale);
        q3 = q8_code_exact(y.w, scale);
    } else {
        q0 = (int)copysignf(fminf(floorf(t0), 127.0f), y.x);
        q1 = (int)copysignf(fminf(floorf(t1), 127.0f), y.y);
        q2 = (int)copysignf(fminf(floorf(t2), 127.0f), y.z);
        q3 = (int)copysignf(fminf(floorf(t3), 127.0f), y.w);
    }
    const unsigned lo = __builtin_amdgcn_perm((unsigned)q1, (unsigned)q0, 0x0c0c0400u);
    const unsigned hi = __builtin_amdgcn_perm((unsigned)q3, (unsigned)q2, 0x0c0c0400u);
    return (int)__builtin_amdgcn_perm(hi, lo, 0x05040100u);
}

// q8_quantize (reference src/q8.c:5-30) of one 64-wide group held by 16 consecutive
// lanes, four consecutive values each.  Returns the 4 packed codes; `scale` gets the
// group scale in every lane of the group.
__device__ __forceinline__ int quantize_group16(float4 y, float& scale) {
    float amax = fmaxf(fmaxf(fabsf(y.x), fabsf(y.y)), fmaxf(fabsf(y.z), fabsf(y.w)));
    // max is exact in any order, so the 16-lane reduction may pair lanes however is cheapest:
    // DPP quad_perm, then row_half_mirror / row_mirror (no trip through the LDS crossbar)
    amax = fmaxf(amax, lane_xor_f<1>(amax));
    amax = fmaxf(amax, lane_xor_f<2>(amax));
    amax = fmaxf(amax, __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(amax), 0x141, 0xF, 0xF, true)));
    amax = fmaxf(amax, __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(amax), 0x140, 0xF, 0xF, true)));
    scale = q3_q8_scale(amax);
    return q8_pack4(y, scale, __builtin_amdgcn_rcpf(scale));
}

__device__ __forceinline__ int dot16(v4i w, v4i x) {
    int d = __builtin_amdgcn_sdot4(w.x, x.x, 0, false);
    d = __builtin_amdgcn_sdot4(w.y, x.y, d, false);
    d = __builtin_amdgcn_sdot4(w.z, x.z, d, false);
    d = __builtin_amdgcn_sdot4(w.w, x.w, d, false);
    return d;
}
// exact int32 sum over the 4 lanes of a quad
__device__ __forceinline__ int quad_sum(int d) {
    d += lane_xor_i<1>(d);
    d += lane_xor_i<2>(d);
    return d;
}

// RMSNorm (size HD, weight w) + half-split RoPE of one head held as float4 per lane in
// lanes [0, HD/4) of a wave (reference forward.c:270-280, 104-118).
// cs = (cos,sin) pairs of this position, [HD/2][2].
// (cos,sin) slices a lane needs for RoPE of its float4: elements i0..i0+3 of the pair index
template <int HD>
__device__ __forceinline__ void rope_slices(const float* __restrict__ cs, int lane, float4& ca, float4& cb) {
    constexpr int L4 = HD / 4;
    ca = make_float4(0.f, 0.f, 0.f, 0.f);
    cb = ca;
    if (lane < L4) {
        const int i0 = 4 * (lane < L4 / 2 ? lane : lane - L4 / 2);
        ca = *reinterpret_cast<const float4*>(cs + 2 * i0);      // c0 s0 c1 s1
        cb = *reinterpret_cast<const float4*>(cs + 2 * i0 + 4);  // c2 s2 c3 s3
    }
}

// RMSNorm (size HD, weight slice g) + half-split RoPE of one head held as float4 per lane
// in lanes [0, HD/4) of a wave (reference forward.c:270-280, 104-118); all operands in
// registers, so callers can issue the loads long before.
template <int HD>
__device__ __forceinline__ float4 headnorm_rope_vals(float4 v, float4 g, float4 ca, float4 cb, int lane) {
    constexpr int L4 = HD / 4;
    const bool act = lane < L4;
    float c0 = 0.f, c1 = 0.f, c2 = 0.f, c3 = 0.f;
    if (act) {
        c0 = c0 + v.x * v.x;
        c1 = c1 + v.y * v.y;
        c2 = c2 + v.z * v.z;
        c3 = c3 + v.w * v.w;
    }
    const float ss = bfly64((c0 + c1) + (c2 + c3));
    const float s = 1.0f / sqrtf(ss / (float)HD + 1e-6f);
    float4 y = make_float4(0.f, 0.f, 0.f, 0.f);
    if (act) {
        y.x = g.x * (s * v.x);
        y.y = g.y * (s * v.y);
        y.z = g.z * (s * v.z);
        y.w = g.w * (s * v.w);
    }
    // element i < HD/2 pairs with i + HD/2: partner lane = lane ^ (L4/2)
    float4 o;
    o.x = lane_xor_f<L4 / 2>(y.x);
    o.y = lane_xor_f<L4 / 2>(y.y);
    o.z = lane_xor_f<L4 / 2>(y.z);
    o.w = lane_xor_f<L4 / 2>(y.w);
    float4 r = y;
    if (act) {
        if (lane < L4 / 2) {   // own = real, other = imag: real*cos - imag*sin
            r.x = y.x * ca.x - o.x * ca.y;
            r.y = y.y * ca.z - o.y * ca.w;
            r.z = y.z * cb.x - o.z * cb.y;
            r.w = y.w * cb.z - o.w * cb.w;
        } else {               // own = imag, other = real: real*sin + imag*cos
            r.x = o.x * ca.y + y.x * ca.x;
            r.y = o.y * ca.w + y.y * ca.z;
            r.z = o.z * cb.y + y.z * cb.x;
            r.w = o.w * cb.w + y.w * cb.z;
        }
    }
    return r;
}

// Two heads of head_dim 128 in one pass: lanes [0,32) hold one head, lanes [32,64) another
// (l = lane & 31 is the float4 slice in both).  bfly32 inside a half adds exactly what bfly64
// adds for a head that sits in the lower half with zeros above it, so each half gets the
// result headnorm_rope_vals would give it.
template <int HD>
__device__ __forceinline__ float4 headnorm_rope_halves(float4 v, float4 g, float4 ca, float4 cb, int l) {
    static_assert(HD == 128, "two heads per wave need 32 lanes per head");
    float c0 = 0.f, c1 = 0.f, c2 = 0.f, c3 = 0.f;
    c0 = c0 + v.x * v.x;
    c1 = c1 + v.y * v.y;
    c2 = c2 + v.z * v.z;
    c3 = c3 + v.w * v.w;
    const float ss = bfly32((c0 + c1) + (c2 + c3));
    const float s = 1.0f / sqrtf(ss / (float)HD + 1e-6f);
    float4 y;
    y.x = g.x * (s * v.x);
    y.y = g.y * (s * v.y);
    y.z = g.z * (s * v.z);
    y.w = g.w * (s * v.w);
    float4 o;
    o.x = lane_xor_f<16>(y.x);
    o.y = lane_xor_f<16>(y.y);
    o.z = lane_xor_f<16>(y.z);
    o.w = lane_xor_f<16>(y.w);
    float4 r;
    if (l < 16) {          // own = real, other = imag: real*cos - imag*sin
        r.x = y.x * ca.x - o.x * ca.y;
        r.y = y.y * ca.z - o.y * ca.w;
        r.z = y.z * cb.x - o.z * cb.y;
        r.w = y.w * cb.z - o.w * cb.w;
    } else {               // own = imag, other = real: real*sin + imag*cos
        r.x = o.x * ca.y + y.x * ca.x;
        r.y = o.y * ca.w + y.y * ca.z;
        r.z = o.z * cb.y + y.z * cb.x;
        r.w = o.w * cb.w + y.w * cb.z;
    }
    return r;
}

template <int HD>
__device__ __forceinline__ float4 headnorm_rope_wave(float4 v, const float* __restrict__ w,
                                                     const float* __restrict__ cs, int lane) {
    float4 g = make_float4(0.f, 0.f, 0.f, 0.f), ca, cb;
    if (lane < HD / 4) g = *reinterpret_cast<const float4*>(w + 4 * lane);
    rope_slices<HD>(cs, lane, ca, cb);
    return headnorm_rope_vals<HD>(v, g, ca, cb, lane);
}

// Transposing butterfly: every lane l of a 32-lane half holds 32 partial values
// c[s] (s = 0..31); returns, in lane l, the sum over the half's lanes of c[l],
// added in exactly the tree of bfly32 (pairs xor 16, 8, 4, 2, 1).  31 exchanges
// instead of 32 x 5, and each level's exchanges are independent of one another.
__device__ __forceinline__ float transpose_sum32(float (&c)[32], int l) {
#define Q3_TS_LEVEL(M, CNT)                                              \
    {                                                                    \
        const bool up = (l & M) != 0;                                    \
        _Pragma("unroll") for (int k = 0; k < CNT; k++) {                \
            const float keep = up ? c[k + CNT] : c[k];                   \
            const float send = up ? c[k] : c[k + CNT];                   \
            c[k] = keep + lane_xor_f<M>(send);                           \
        }                                                                \
    }
    Q3_TS_LEVEL(16, 16)
    Q3_TS_LEVEL(8, 8)
    Q3_TS_LEVEL(4, 4)
    Q3_TS_LEVEL(2, 2)
    Q3_TS_LEVEL(1, 1)
#undef Q3_TS_LEVEL
    return c[0];
}

// LDS slab of the batched GEMMs (q3_prefill.hip, q3_fp16.hip): rows of 512 bytes = 32 pieces of 16 bytes;
// the low four bits of the piece index are XORed with the row, so the 16 rows (or tokens) of an MFMA
// k-block land in 16 different bank groups.  Byte offset of (row, piece).
__device__ __forceinline__ int slab_off(int row, int piece) {
    return row * 512 + ((((piece ^ row) & 15) | (piece & 16)) << 4);
}

}  // namespace q3k
