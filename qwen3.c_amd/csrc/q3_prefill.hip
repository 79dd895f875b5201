// q3_prefill.hip -- batched prompt ingestion: up to 64 prompt positions go
// through a layer together, so every weight byte is read once per 64 tokens instead of once per
// token, and the Q8_0 product runs on the matrix cores.
//
//   k_rows_quantize   the GEMV prologue (rmsnorm + q8_quantize, or q8_quantize alone) for B
//                     activation rows at once: a workgroup per row and 1024-element span
//   k_gemm_q8_lds /   out[t][r] = W[r][:] . x_t with v_mfma_i32_16x16x64_i8, operands staged through LDS in
//   k_gemm_q8         whole lines (or gathered per lane, for narrow rows): K = 64 is exactly one Q8_0 group, so ONE MFMA
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
#include <type_traits>

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
    for (int base = ((int)blockIdx.y * 4 + wave) * 256; base < n; base += (int)gridDim.y * 4 * 256) {
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
    // a row is split over up to 8 workgroups of 1024 elements (each recomputes the row's sum of squares)
    const int split = (n + 1023) / 1024 < 8 ? (n + 1023) / 1024 : 8;
    if (w) hipLaunchKernelGGL(k_rows_quantize<true>, dim3(rows, split), dim3(256), 0, st, x, ldx, w, n, q, s);
    else hipLaunchKernelGGL(k_rows_quantize<false>, dim3(rows, split), dim3(256), 0, st, x, ldx, w, n, q, s);
}

// ---- Q8_0 GEMM on int8 MFMA ----------------------------------------------------------------
typedef int v4i32 __attribute__((ext_vector_type(4)));

// One wave = 16 rows x 16 tokens, all of K.  A workgroup = one row tile x up to four token tiles
// (the waves read the same weight lines: one fetch from L2, the rest from the CU's L1).
//
// The decode GEMV (q3_gemv.hip) defines the result: per row, sixteen column sums col[c] = the
// products p_g = ((float)dot_g * ws_g) * xs_g of the groups g = c, c+16, c+32, .. added in ascending
// order, then the butterfly col[c] += col[c^8], ^4, ^2, ^1.  K = 64 is one Q8_0 group, so ONE
// v_mfma_i32_16x16x64_i8 gives the exact int32 group dots of 16 rows x 16 tokens; a lane holds 4 of
// them (4 rows of one token) and keeps the sixteen column sums of each in registers (64 VGPRs), walks
// the groups in memory order -- every lane streams through its row 64 bytes at a time, four groups per
// round with the next round's loads already in flight, every load a buffer load whose per-lane offset
// is computed once -- and runs the butterfly once at the end: the GEMV's additions with the GEMV's
// operands, so a prefilled prompt leaves bit-identical logits and KV cache (tests/test_gpu_prefill.py).
//
// Operand maps of v_mfma_i32_16x16x64_i8 (checked with asymmetric integer data by the op test):
//   A: lane l holds A[row l&15][k = 16*(l>>4) .. +15];  B the same with token l&15
//   D: lane l, register i holds D[row 4*(l>>4) + i][token l&15]
//
// This kernel lets every lane gather its own 16-byte operand slices (32 half lines per MFMA through the CU's
// vector-memory path) and its time follows that count: 63 us for the gate/up GEMM of 64 tokens
// (profiles/r02_prefill_gemm_variants.json lists the register-only variants measured, none faster).  It now
// serves the shapes k_gemm_q8_lds below does not take (rows narrower than 512 bytes: the small fixtures).
struct GemmSub {          // four consecutive groups: this lane's k-slices and the scales it needs
    v4i a[4], b[4];
    float sw[4][4];       // [output row i][group]
    float sx[4];
};

template <int EPI, bool AL4>          // AL4: n/64 is a multiple of 4 (scale quads are 16-byte aligned, no partial round)
__global__ __launch_bounds__(256) void k_gemm_q8(const int8_t* __restrict__ W, const float* __restrict__ S, int n,
                                                 int d, const int8_t* __restrict__ xq,
                                                 const float* __restrict__ xs, int ntok,
                                                 float* __restrict__ out, int ldo) {
    const int lane = threadIdx.x & 63;
    const int tt = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));        // token tile of this wave
    const int r0 = (int)blockIdx.x * 16;
    const int ng = n >> 6;
    const int li = lane & 15, kb = lane >> 4;
    const int arow = r0 + li < d ? r0 + li : d - 1;          // rows >= d: computed and dropped
    const int tok = tt * 16 + li < ntok ? tt * 16 + li : ntok - 1;     // tokens >= ntok: clamped, dropped at the store
    // buffer loads: per-lane offsets computed once, the group's offset is a scalar
    const __amdgpu_buffer_rsrc_t rW = __builtin_amdgcn_make_buffer_rsrc(const_cast<int8_t*>(W), 0, d * n, 0x00020000);
    const __amdgpu_buffer_rsrc_t rS = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(S), 0, d * ng * 4, 0x00020000);
    const __amdgpu_buffer_rsrc_t rX = __builtin_amdgcn_make_buffer_rsrc(const_cast<int8_t*>(xq), 0, ntok * n, 0x00020000);
    const __amdgpu_buffer_rsrc_t rXS = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(xs), 0, ntok * ng * 4, 0x00020000);
    const int va = arow * n + 16 * kb, vb = tok * n + 16 * kb, vxs = tok * ng * 4;
    int vs[4];
#pragma unroll
    for (int i = 0; i < 4; i++) vs[i] = (r0 + 4 * kb + i < d ? r0 + 4 * kb + i : d - 1) * ng * 4;

    auto load_sub = [&](GemmSub& r, int g) {                 // groups g .. g+3 (g wave-uniform); past ng: clamped, never used
#pragma unroll
        for (int c = 0; c < 4; c++) {
            const int gc = AL4 ? g + c : (g + c < ng ? g + c : ng - 1);
            r.a[c] = __builtin_amdgcn_raw_buffer_load_b128(rW, va, gc * 64, 0);
            r.b[c] = __builtin_amdgcn_raw_buffer_load_b128(rX, vb, gc * 64, 0);
        }
        if (AL4) {
#pragma unroll
            for (int i = 0; i < 4; i++) {
                const v4i q = __builtin_amdgcn_raw_buffer_load_b128(rS, vs[i], g * 4, 0);
                r.sw[i][0] = __int_as_float(q.x); r.sw[i][1] = __int_as_float(q.y);
                r.sw[i][2] = __int_as_float(q.z); r.sw[i][3] = __int_as_float(q.w);
            }
            const v4i q = __builtin_amdgcn_raw_buffer_load_b128(rXS, vxs, g * 4, 0);
            r.sx[0] = __int_as_float(q.x); r.sx[1] = __int_as_float(q.y); r.sx[2] = __int_as_float(q.z); r.sx[3] = __int_as_float(q.w);
        } else {
#pragma unroll
            for (int c = 0; c < 4; c++) {
                const int gc = g + c < ng ? g + c : ng - 1;
#pragma unroll
                for (int i = 0; i < 4; i++) r.sw[i][c] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rS, vs[i], gc * 4, 0));
                r.sx[c] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rXS, vxs, gc * 4, 0));
            }
        }
    };

    float col[16][4];                                        // [SUM16 column][output register]
#pragma unroll
    for (int c = 0; c < 16; c++)
#pragma unroll
        for (int i = 0; i < 4; i++) col[c][i] = 0.0f;

    GemmSub P, Q;                                            // ping / pong: one round computing, the next in flight
    load_sub(P, 0);
    for (int g0 = 0; g0 < ng; g0 += 16) {                    // 16 groups = one pass over the columns
#pragma unroll
        for (int q4 = 0; q4 < 4; q4++) {                     // rounds of 4 groups, alternating register sets
            const int g = g0 + 4 * q4;
            if (g >= ng) break;                              // wave-uniform
            GemmSub& cur = (q4 & 1) ? Q : P;
            GemmSub& nxt = (q4 & 1) ? P : Q;
            if (g + 4 < ng) load_sub(nxt, g + 4);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int c = 0; c < 4; c++) {
                if (AL4 || g + c < ng) {
                    const v4i32 zero = {0, 0, 0, 0};
                    const v4i32 dd = __builtin_amdgcn_mfma_i32_16x16x64_i8(cur.a[c], cur.b[c], zero, 0, 0, 0);
#pragma unroll
                    for (int i = 0; i < 4; i++) {
                        const float p = ((float)dd[i] * cur.sw[i][c]) * cur.sx[c];
                        col[4 * q4 + c][i] = col[4 * q4 + c][i] + p;
                    }
                }
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    // butterfly col[c] += col[c^8], ^4, ^2, ^1
    float res[4];
#pragma unroll
    for (int i = 0; i < 4; i++) {
        float t8[8], t4[4], t2[2];
#pragma unroll
        for (int c = 0; c < 8; c++) t8[c] = col[c][i] + col[c + 8][i];
#pragma unroll
        for (int c = 0; c < 4; c++) t4[c] = t8[c] + t8[c + 4];
#pragma unroll
        for (int c = 0; c < 2; c++) t2[c] = t4[c] + t4[c + 2];
        res[i] = t2[0] + t2[1];
    }
    const int tokj = tt * 16 + li;
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

// ---- the same GEMM with both operands staged through LDS in whole 128-byte lines ---------------
// k_gemm_q8 above makes every lane gather its own 16-byte operand slices: per MFMA the CU's vector
// memory path touches 32 half lines, and that, not HBM or the VALU, is what its time follows.  Here a
// workgroup owns 16*RT rows x 64 tokens and walks K in slabs of 512 bytes (8 groups) with its eight
// waves in two roles:
//   waves 4..7  FETCH.  Every thread requests 16-byte pieces that are CONSECUTIVE across the lanes (a
//               wave-load = two rows x 512 B = eight whole lines), three slabs ahead in three register
//               sets, and parks a landed slab in one of two LDS buffers whose 16-byte slots are
//               XOR-swizzled by the row, so the operand reads of the 16 rows (or tokens) of a k-block
//               hit 16 different bank groups.  A CU takes in only ~40 KB of requests at once, so these
//               waves spend their life in the issue stage -- which is why they do nothing else.
//   waves 0..3  MULTIPLY.  Wave t owns token tile t and all RT row tiles: per group one activation
//               read, RT weight reads, RT MFMAs, and the scale-accumulate into its 16 column sums per
//               output (64*RT registers).  They never issue a global load, so they never stall in it.
// One barrier per slab: it says "slab s is in LDS and slab s-1 has been consumed" (slab s+1 goes into
// the buffer slab s-1 occupied).  The weight slab is shared by the four token tiles, the activation
// slab by the RT row tiles; the per-group scales ride along ([group][row] in LDS: a lane's four row
// scales are one 16-byte read).  Arithmetic per (row, token): exactly k_gemm_q8's -- the group dots
// of one MFMA, ((float)dot * ws) * xs, added to column (g & 15) in ascending g; the butterfly at the
// end.  Needs n % 512 == 0 (all Qwen3 shapes; fixtures with narrower rows take k_gemm_q8).
template <int RT> struct GemmSlab {       // what one fetching thread holds of a slab
    v4i a[2 * RT];
    v4i b[8];
    float ws;
    float xs[2];
};

template <int EPI, int RT>
__global__ __launch_bounds__(512) void k_gemm_q8_lds(const int8_t* __restrict__ W, const float* __restrict__ S, int n,
                                                     int d, const int8_t* __restrict__ xq,
                                                     const float* __restrict__ xs, int ntok,
                                                     float* __restrict__ out, int ldo, int ntiles) {
    constexpr int R = 16 * RT;
    // scales in LDS: [group][row] and [group][token] with a group stride of 72 floats (= 8 mod 64): the fetch
    // waves write them 8 groups x 8 rows per wave, which then lands in 64 different banks
    constexpr int SS = 72;
    constexpr int A_BYTES = R * 512, B_BYTES = 64 * 512, WS_BYTES = 8 * SS * 4, XS_BYTES = 8 * SS * 4;
    constexpr int OFF_B = A_BYTES, OFF_WS = A_BYTES + B_BYTES, OFF_XS = OFF_WS + WS_BYTES;
    constexpr int STAGE = OFF_XS + XS_BYTES;
    __shared__ __attribute__((aligned(16))) unsigned char smem[2 * STAGE];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int ng = n >> 6, nslab = n >> 9;
    const int ntok16 = (ntok + 15) & ~15;
    // Persistent: this workgroup takes the row tiles blockIdx.x, blockIdx.x + gridDim.x, ..; the slabs of
    // all of them form ONE stream (stage q = slab q % nslab of the workgroup's tile q / nslab), so the
    // fetch waves are already deep in the next tile while the last slab of this one is multiplied.
    const int mytiles = (ntiles - (int)blockIdx.x + (int)gridDim.x - 1) / (int)gridDim.x;
    const int nq = mytiles * nslab;

    if (wave >= 4) {
        // ---------------------------------------------------------------- FETCH
        // rows >= d and tokens >= ntok lie beyond the buffer ranges: they read as zero and are dropped at the store
        const __amdgpu_buffer_rsrc_t rW = __builtin_amdgcn_make_buffer_rsrc(const_cast<int8_t*>(W), 0, d * n, 0x00020000);
        const __amdgpu_buffer_rsrc_t rS = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(S), 0, d * ng * 4, 0x00020000);
        const __amdgpu_buffer_rsrc_t rX = __builtin_amdgcn_make_buffer_rsrc(const_cast<int8_t*>(xq), 0, ntok * n, 0x00020000);
        const __amdgpu_buffer_rsrc_t rXS = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(xs), 0, ntok * ng * 4, 0x00020000);
        const int t = tid - 256;
        const int prow = t >> 5, piece = t & 31;                 // this thread's (row, piece) of a slab; +8 rows per further load
        const int va = prow * n + piece * 16;
        const int vb = prow * n + piece * 16;
        const int vws = ((t >> 3) * ng + (t & 7)) * 4;           // threads < 8*R
        const int vxs = ((t >> 3) * ng + (t & 7)) * 4;
        int f_tile = (int)blockIdx.x, f_sl = 0;                  // the (tile, slab) the next fetch takes
        auto fetch = [&](GemmSlab<RT>& r) {
            const int sl = f_sl, r0 = f_tile * R;                // wave-uniform
            if (++f_sl == nslab) { f_sl = 0; f_tile += (int)gridDim.x; }
#pragma unroll
            for (int k = 0; k < 2 * RT; k++) r.a[k] = __builtin_amdgcn_raw_buffer_load_b128(rW, va + k * 8 * n, r0 * n + sl * 512, 0);
#pragma unroll
            for (int k = 0; k < 8; k++) {
                if (prow + 8 * k < ntok16)                        // wave-uniform: a wave covers two rows, ntok16 is a multiple of 16
                    r.b[k] = __builtin_amdgcn_raw_buffer_load_b128(rX, vb + k * 8 * n, sl * 512, 0);
            }
            if (t < 8 * R) r.ws = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rS, vws, r0 * ng * 4 + sl * 32, 0));
#pragma unroll
            for (int k = 0; k < 2; k++)
                r.xs[k] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rXS, vxs + k * 32 * ng * 4, sl * 32, 0));
        };
        auto park = [&](const GemmSlab<RT>& r, int buf) {        // registers -> LDS buffer `buf`
            unsigned char* base = smem + buf * STAGE;
#pragma unroll
            for (int k = 0; k < 2 * RT; k++) *reinterpret_cast<v4i*>(base + slab_off(prow + 8 * k, piece)) = r.a[k];
#pragma unroll
            for (int k = 0; k < 8; k++) {
                if (prow + 8 * k < ntok16) *reinterpret_cast<v4i*>(base + OFF_B + slab_off(prow + 8 * k, piece)) = r.b[k];
            }
            if (t < 8 * R) reinterpret_cast<float*>(base + OFF_WS)[(t & 7) * SS + (t >> 3)] = r.ws;
#pragma unroll
            for (int k = 0; k < 2; k++) reinterpret_cast<float*>(base + OFF_XS)[(t & 7) * SS + (t >> 3) + 32 * k] = r.xs[k];
        };
        GemmSlab<RT> S0, S1, S2;                                 // three slabs in flight
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
    const bool live = tt * 16 < ntok;                        // wave-uniform: this wave's token tile exists
    float col[RT][16][4];                                    // [row tile][SUM16 column][output register]
#pragma unroll
    for (int r = 0; r < RT; r++)
#pragma unroll
        for (int c = 0; c < 16; c++)
#pragma unroll
            for (int i = 0; i < 4; i++) col[r][c][i] = 0.0f;

    auto compute = [&](int buf, auto par) {                  // slab in LDS buffer `buf`; PAR = slab index & 1: columns 8*PAR ..
        constexpr int PAR = decltype(par)::value;
        const unsigned char* base = smem + buf * STAGE;
#pragma unroll
        for (int g = 0; g < 8; g++) {
            const v4i b = *reinterpret_cast<const v4i*>(base + OFF_B + slab_off(brow, 4 * g + kb));
            const float sx = *reinterpret_cast<const float*>(base + OFF_XS + (g * SS + brow) * 4);
#pragma unroll
            for (int r = 0; r < RT; r++) {
                const v4i a = *reinterpret_cast<const v4i*>(base + slab_off(r * 16 + li, 4 * g + kb));
                const float4 sw = *reinterpret_cast<const float4*>(base + OFF_WS + (g * SS + r * 16 + 4 * kb) * 4);
                const v4i32 zero = {0, 0, 0, 0};
                const v4i32 dd = __builtin_amdgcn_mfma_i32_16x16x64_i8(a, b, zero, 0, 0, 0);
                const float swv[4] = {sw.x, sw.y, sw.z, sw.w};
#pragma unroll
                for (int i = 0; i < 4; i++) {
                    const float p = ((float)dd[i] * swv[i]) * sx;
                    col[r][8 * PAR + g][i] = col[r][8 * PAR + g][i] + p;
                }
            }
        }
    };
    const int tokj = tt * 16 + li;
    int q = 0;                                               // stage counter: stage q lives in LDS buffer q & 1
    for (int tile = (int)blockIdx.x; tile < ntiles; tile += (int)gridDim.x) {
        for (int sl = 0; sl < nslab; sl += 2) {
            __syncthreads();                                 // stage q is in LDS
            if (live) compute(q & 1, std::integral_constant<int, 0>());
            q++;
            if (sl + 1 >= nslab) break;
            __syncthreads();
            if (live) compute(q & 1, std::integral_constant<int, 1>());
            q++;
        }
        if (tokj < ntok) {
            const int r0 = tile * R;
#pragma unroll
            for (int r = 0; r < RT; r++) {
                float res[4];
#pragma unroll
                for (int i = 0; i < 4; i++) {
                    float t8[8], t4[4], t2[2];
#pragma unroll
                    for (int c = 0; c < 8; c++) t8[c] = col[r][c][i] + col[r][c + 8][i];
#pragma unroll
                    for (int c = 0; c < 4; c++) t4[c] = t8[c] + t8[c + 4];
#pragma unroll
                    for (int c = 0; c < 2; c++) t2[c] = t4[c] + t4[c + 2];
                    res[i] = t2[0] + t2[1];
                }
                const int row = r0 + r * 16 + 4 * kb;
                if (EPI == EPI_SWIGLU) {
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
        }
#pragma unroll
        for (int r = 0; r < RT; r++)
#pragma unroll
            for (int c = 0; c < 16; c++)
#pragma unroll
                for (int i = 0; i < 4; i++) col[r][c][i] = 0.0f;
    }
}


template <int RT>
static void launch_gemm_lds(const int8_t* W, const float* S, int n, int d, const int8_t* xq, const float* xs, int ntok,
                            float* out, int ldo, Epi epi, hipStream_t st) {
    const int ntiles = (d + 16 * RT - 1) / (16 * RT);
    const dim3 grid(ntiles < cu_count() ? ntiles : cu_count()), block(512);      // one workgroup per CU (LDS), persistent
    if (epi == EPI_STORE) hipLaunchKernelGGL((k_gemm_q8_lds<EPI_STORE, RT>), grid, block, 0, st, W, S, n, d, xq, xs, ntok, out, ldo, ntiles);
    else if (epi == EPI_RESID) hipLaunchKernelGGL((k_gemm_q8_lds<EPI_RESID, RT>), grid, block, 0, st, W, S, n, d, xq, xs, ntok, out, ldo, ntiles);
    else hipLaunchKernelGGL((k_gemm_q8_lds<EPI_SWIGLU, RT>), grid, block, 0, st, W, S, n, d, xq, xs, ntok, out, ldo, ntiles);
}

template <bool AL4>
static void launch_gemm(const int8_t* W, const float* S, int n, int d, const int8_t* xq, const float* xs, int ntok,
                        float* out, int ldo, Epi epi, hipStream_t st) {
    const dim3 grid((d + 15) / 16), block(64 * ((ntok + 15) / 16));
    if (epi == EPI_STORE) hipLaunchKernelGGL((k_gemm_q8<EPI_STORE, AL4>), grid, block, 0, st, W, S, n, d, xq, xs, ntok, out, ldo);
    else if (epi == EPI_RESID) hipLaunchKernelGGL((k_gemm_q8<EPI_RESID, AL4>), grid, block, 0, st, W, S, n, d, xq, xs, ntok, out, ldo);
    else hipLaunchKernelGGL((k_gemm_q8<EPI_SWIGLU, AL4>), grid, block, 0, st, W, S, n, d, xq, xs, ntok, out, ldo);
}
void gemm_q8(const int8_t* W, const float* S, int n, int d, const int8_t* xq, const float* xs, int ntok, float* out,
             int ldo, Epi epi, hipStream_t st) {
    if (n % 64 || d % 2 || ntok < 1 || ntok > 64 || (long long)d * n >= (1ll << 31)) {
        fprintf(stderr, "[q3hip] gemm_q8: bad shape (n=%d d=%d tokens=%d)\n", n, d, ntok);
        exit(EXIT_FAILURE);
    }
    static const int force_gather = getenv("Q3_GEMM_GATHER") ? atoi(getenv("Q3_GEMM_GATHER")) : 0;   // diagnosis: the k_gemm_q8 path
    if (n % 512 == 0 && !force_gather) {
        // enough 32-row tiles to fill the chip twice over -> the wider tile (half the activation traffic)
        if (d >= 2 * 256 * 32) launch_gemm_lds<2>(W, S, n, d, xq, xs, ntok, out, ldo, epi, st);
        else launch_gemm_lds<1>(W, S, n, d, xq, xs, ntok, out, ldo, epi, st);
        return;
    }
    if ((n >> 6) % 4 == 0) launch_gemm<true>(W, S, n, d, xq, xs, ntok, out, ldo, epi, st);
    else launch_gemm<false>(W, S, n, d, xq, xs, ntok, out, ldo, epi, st);
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
