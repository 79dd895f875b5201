// q3_tile.hpp -- the unit of weight streaming of the GEMV launches (q3_gemv.hip; also used
// by the fp16 and prompt-pass kernels): a TILE is R consecutive rows of a Q8_0
// matrix as seen by one wave -- R*NJ wave-loads of 1 KiB codes (lane l holds bytes
// [1024 j + 16 l, +16) of each row) plus the group scales each quad needs.
//
//   tile_issue  requests the tile from HBM: `buffer_load_dwordx4 ... nt` with all row
//               arithmetic in SGPRs and one per-lane offset VGPR; nothing waits, so a
//               wave can keep several tiles in flight in its registers
//   tile_dot    consumes it against the quantised activation in LDS: per 64-group an exact
//               int32 dot (4 x v_dot4_i32_i8 + DPP quad sum), scaled ((float)dot*ws)*xs as
//               reference matmul() does (src/forward.c:94-96), summed in the SUM16 tree
#pragma once
#include "q3_device.hpp"

// cache policy of the weight stream: 2 = nt (read once, do not keep)
#ifndef Q3_W_AUX
#define Q3_W_AUX 2
#endif

namespace q3k {

template <int R, int NJ>
struct Tile {
    v4i w[R][NJ];
    float s[R][NJ];
};

// Buffer descriptors of one matrix (wave-uniform, in SGPRs).  Rows >= d fall outside
// num_records and read as zero, so callers never predicate on the row index.
struct WView {
    __amdgpu_buffer_rsrc_t w, s;
    int n;
};
__device__ __forceinline__ WView make_wview(const int8_t* W, const float* S, int d, int n) {
    WView v;
    const size_t wbytes = (size_t)d * n;
    v.w = __builtin_amdgcn_make_buffer_rsrc(const_cast<int8_t*>(W), 0, (int)wbytes, 0x00020000);
    v.s = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(S), 0, (int)(wbytes >> 4), 0x00020000);
    v.n = n;
    return v;
}

// `row0` must be wave-uniform (an SGPR value).
template <int R, int NJ>
__device__ __forceinline__ void tile_issue(Tile<R, NJ>& t, const WView& wv, int row0, int lane) {
    const int n = wv.n, ngroups = wv.n >> 6;
    // The per-lane offsets are recomputed at every issue (two VALU ops) from an opaque copy of
    // the lane id: otherwise hipcc hoists voff + j*1024, vsoff + j*64 ... for every j out of
    // the caller's loops and parks dozens of loop-invariant registers next to the tiles.
    asm volatile("" : "+v"(lane));
    asm volatile("" : "+s"(row0));
    const int voff = lane * 16, vsoff = (lane >> 2) * 4;
    const int tail = n - (NJ - 1) * 1024;        // bytes of the last wave-load of a row (<= 1024)
#pragma unroll
    for (int r = 0; r < R; r++) {
        const int wbase = (row0 + r) * n;        // < 2^31 for every tensor of these models
        const int sbase = (row0 + r) * ngroups * 4;
#pragma unroll
        for (int j = 0; j < NJ; j++) {
            v4i w = {0, 0, 0, 0};
            float s = 0.0f;
            if (j < NJ - 1 || voff < tail) {
                // the constant part below 4 KiB rides in the instruction's immediate offset, so a
                // row needs one scalar offset per 4 KiB of codes and one for its scales
                w = __builtin_amdgcn_raw_buffer_load_b128(wv.w, voff + (j & 3) * 1024, wbase + (j >> 2) * 4096, Q3_W_AUX);
                s = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(wv.s, vsoff + j * 64, sbase, Q3_W_AUX));
            }
            t.w[r][j] = w;
            t.s[r][j] = s;
        }
    }
}

// The same loads one at a time, for kernels that interleave issue and use.  Branch-free: lanes
// past the end of a row get an offset beyond num_records (the load returns zero and moves no
// bytes), so the compiler sees straight-line code and counts vmcnt exactly.
struct TileLane {
    int voff, voff_last, vsoff, vsoff_last;
};
template <int NJ>
__device__ __forceinline__ TileLane tile_lane(const WView& wv, int lane) {
    asm volatile("" : "+v"(lane));
    constexpr int OOB = 0x7ffff000;
    const int tail = wv.n - (NJ - 1) * 1024;
    TileLane t;
    t.voff = lane * 16;
    t.vsoff = (lane >> 2) * 4;
    t.voff_last = t.voff < tail ? t.voff : OOB;
    t.vsoff_last = t.voff < tail ? t.vsoff : OOB;
    return t;
}
template <int R, int NJ>
__device__ __forceinline__ void tile_issue_one(Tile<R, NJ>& t, const WView& wv, const TileLane& tl, int row0, int r, int j) {
    const int n = wv.n, ngroups = wv.n >> 6;
    const int wbase = (row0 + r) * n + (j >> 2) * 4096;
    const int sbase = (row0 + r) * ngroups * 4;
    const bool last = j == NJ - 1;
    t.w[r][j] = __builtin_amdgcn_raw_buffer_load_b128(wv.w, (last ? tl.voff_last : tl.voff) + (j & 3) * 1024, wbase, Q3_W_AUX);
    t.s[r][j] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(wv.s, (last ? tl.vsoff_last : tl.vsoff) + j * 64, sbase, Q3_W_AUX));
}

// acc[r] = row r of the tile . activation; every lane ends with the full sums.
template <int R, int NJ>
__device__ __forceinline__ void tile_dot(const Tile<R, NJ>& t, int n, int lane, const int8_t* lq,
                                         const float* ls, float (&acc)[R]) {
    asm volatile("" : "+v"(lane));          // see tile_issue: keep the LDS offsets out of outer loops
    const int quad = lane >> 2;
#pragma unroll
    for (int r = 0; r < R; r++) acc[r] = 0.0f;
#pragma unroll
    for (int j = 0; j < NJ; j++) {
        const int off = j * 1024 + lane * 16;
        const bool act = off < n;
        v4i xv = {0, 0, 0, 0};
        float sx = 0.0f;
        if (act) {
            xv = *reinterpret_cast<const v4i*>(lq + off);
            sx = ls[j * 16 + quad];
        }
#pragma unroll
        for (int r = 0; r < R; r++) {
            const int dsum = quad_sum(dot16(t.w[r][j], xv));
            const float p = ((float)dsum * t.s[r][j]) * sx;
            acc[r] = act ? acc[r] + p : acc[r];
        }
    }
#pragma unroll
    for (int r = 0; r < R; r++) acc[r] = bfly_quads(acc[r]);
}

// reference swiglu() on one (gate, up) pair (src/forward.c:122-139)
__device__ __forceinline__ float swiglu_pair(float g, float u) {
    const float sig = 1.0f / (1.0f + q3_expf(-g));
    return (g * sig) * u;
}

}  // namespace q3k
