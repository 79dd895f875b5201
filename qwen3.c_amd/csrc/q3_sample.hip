// q3_sample.hip -- device-side temperature / top-p sampling: the reference's
// sample() (src/sampler.c:189-201) without the 608 KB logits download and the host qsort of
// 151,936 entries per token.
//
//   scale + softmax   three launches over ceil(V/4096) workgroups (max, exp + chunk sums, divide):
//                     logits/temperature by true division, then the softmax of q3_numerics.h
//   sort              rocprim::radix_sort_pairs_desc on (probability, index): stable, so equal
//                     probabilities keep their index order -- what the reference's qsort (glibc's
//                     merge sort) yields with its comparator (sampler.c:138-148)
//   nucleus + draw    ONE wave walks the sorted distribution with the reference's own sequential
//                     fp32 sums (sampler.c:88-136: smallest prefix whose mass exceeds top_p, the
//                     "healing" of tiny masses, inverse-CDF draw, fallbacks included) and stops
//                     at the prefix -- a few tokens for a trained model, the whole vocabulary only
//                     for a flat distribution
// The coin comes from xorshift64* (src/xorshift.c:7-16), on the host (one draw per call) or on
// the device (generation loop, state in device memory).  Everything is bit-exact against
// oracle orc_sample() in tree mode; tests/test_gpu_sample.py.
#include <cstring>

#include <rocprim/device/device_radix_sort.hpp>

#include <cstdio>
#include <cstdlib>

#include "q3_device.hpp"
#include "q3_kernels.hpp"

namespace q3k {

// x <- x / temperature, chunk maxima
__global__ __launch_bounds__(256) void k_sm_scale_max(float* x, int n, float temperature, float* pmax) {
    __shared__ float red[4];
    const int c0 = blockIdx.x * Q3_SM_CHUNK;
    float m = -3.4e38f;
    for (int i = c0 + threadIdx.x; i < n && i < c0 + Q3_SM_CHUNK; i += 256) {
        const float v = x[i] / temperature;
        x[i] = v;
        m = fmaxf(m, v);
    }
    m = wave_max(m);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = m;
    __syncthreads();
    if (threadIdx.x == 0) pmax[blockIdx.x] = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
}

// e = q3_expf(x - max), chunk sums (SUM256 of the chunk, from LDS, by wave 0)
__global__ __launch_bounds__(256) void k_sm_exp_sum(float* x, int n, const float* pmax, int nchunks, float* psum) {
    __shared__ __attribute__((aligned(16))) float e[Q3_SM_CHUNK];
    float m = pmax[0];
    for (int c = 1; c < nchunks; c++) m = fmaxf(m, pmax[c]);
    const int c0 = blockIdx.x * Q3_SM_CHUNK;
    const int cn = n - c0 < Q3_SM_CHUNK ? n - c0 : Q3_SM_CHUNK;
    for (int i = threadIdx.x; i < Q3_SM_CHUNK; i += 256) {
        float v = 0.0f;
        if (i < cn) {
            v = q3_expf(x[c0 + i] - m);
            x[c0 + i] = v;
        }
        e[i] = v;               // zeros past the end: adding +0 to the non-negative partials is exact
    }
    __syncthreads();
    if (threadIdx.x < 64) {
        const int lane = threadIdx.x;
        float c0s = 0.f, c1s = 0.f, c2s = 0.f, c3s = 0.f;
        for (int i = 4 * lane; i < cn; i += 256) {
            const float4 v = *reinterpret_cast<const float4*>(e + i);
            c0s = c0s + v.x;
            c1s = c1s + v.y;
            c2s = c2s + v.z;
            c3s = c3s + v.w;
        }
        const float s = bfly64((c0s + c1s) + (c2s + c3s));
        if (lane == 0) psum[blockIdx.x] = s;
    }
}

// p = e / sum, sum = the chunk sums in chunk order
__global__ __launch_bounds__(256) void k_sm_divide(float* x, int n, const float* psum, int nchunks) {
    float sum = psum[0];
    if (nchunks > 1) {
        sum = 0.0f;
        for (int c = 0; c < nchunks; c++) sum = sum + psum[c];
    }
    const int c0 = blockIdx.x * Q3_SM_CHUNK;
    for (int i = c0 + threadIdx.x; i < n && i < c0 + Q3_SM_CHUNK; i += 256) x[i] = x[i] / sum;
}

__global__ void k_iota(int* idx, int n) {
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) idx[i] = i;
}

// reference src/xorshift.c:7-16
__device__ __forceinline__ float xorshift_float(unsigned long long* state) {
    unsigned long long s = *state;
    s ^= s >> 12;
    s ^= s << 25;
    s ^= s >> 27;
    *state = s;
    const unsigned r = (unsigned)((s * 0x2545F4914F6CDD1Dull) >> 32);
    return (float)(r >> 8) / 16777216.0f;
}

// One wave.  `prob` / `index` are the sorted distribution.  The sums are the reference's: strictly
// sequential in sorted order, so the wave loads 64 entries at a time (one per lane) and lane-uniform
// code adds them one after the other.
__global__ __launch_bounds__(64) void k_nucleus(const float* __restrict__ prob, const int* __restrict__ index, int n,
                                                float top_p, float coin_host, unsigned long long* seed, int* out,
                                                int* out2) {
    const int lane = threadIdx.x;
    float coin = coin_host;
    if (seed) {                          // generation loop: the RNG state lives on the device
        unsigned long long s = *seed;
        coin = xorshift_float(&s);
        if (lane == 0) *seed = s;
    }
    // The walks below add 64 entries per step (one per lane, broadcast with v_readlane) in straight-line
    // code.  The running sum is the only dependency chain; lane k keeps the sum as it stood after entry k,
    // so "the first entry at which the condition held" is ONE vector compare + ballot per 64 entries.
    // sampler_mass_index (sampler.c:88-113)
    float mass = 0.0f;
    int id = n - 1;
    for (int b = 0; b < n; b += 64) {
        const float v = (b + lane < n) ? prob[b + lane] : 0.0f;      // +0 past the end: exact
        float run = mass, mine = 0.0f;
#pragma unroll
        for (int k = 0; k < 64; k++) {
            run = run + __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), k));
            mine = (lane == k) ? run : mine;
        }
        const unsigned long long hit = __builtin_amdgcn_ballot_w64(mine > top_p);
        if (hit) {
            const int first = __builtin_ctzll(hit);
            id = b + first;
            mass = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, mine), first));
            break;
        }
        mass = run;
    }
    if (mass < 1e-3f) {                  // "heal the sampled distribution"
        for (int b = 0; b <= id; b += 64) {
            const float v = (b + lane <= id) ? prob[b + lane] : 0.0f;
#pragma unroll
            for (int k = 0; k < 64; k++)
                mass = mass + __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), k));
        }
    }
    // sampler_cdf_index (sampler.c:126-136), called with n = id: entries 0..id, fallback dist[id-1]
    const float r = coin * mass;
    float cdf = 0.0f;
    int pick = -1;
    for (int b = 0; b <= id; b += 64) {
        const float v = (b + lane <= id) ? prob[b + lane] : 0.0f;
        float mine = 0.0f;
#pragma unroll
        for (int k = 0; k < 64; k++) {
            cdf = cdf + __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), k));
            mine = (lane == k) ? cdf : mine;
        }
        const unsigned long long hit = __builtin_amdgcn_ballot_w64((r < mine) && (b + lane <= id));
        if (hit) {
            pick = b + __builtin_ctzll(hit);
            break;
        }
    }
    int tok;
    if (pick >= 0) tok = index[pick];
    else if (id > 0) tok = index[id - 1];
    else tok = 0;        // the reference reads dist[-1] here (see oracle orc_sample): token 0 with glibc
    if (lane == 0) {
        *out = tok;
        if (out2) *out2 = tok;
    }
}

size_t sample_temp_bytes(int n) {
    size_t bytes = 0;
    (void)rocprim::radix_sort_pairs_desc(nullptr, bytes, (const float*)nullptr, (float*)nullptr, (const int*)nullptr,
                                         (int*)nullptr, (size_t)n, 0, 32, (hipStream_t) nullptr);
    return bytes;
}

void sample_init(const SampleBufs& b, int n, hipStream_t st) {
    hipLaunchKernelGGL(k_iota, dim3(256), dim3(256), 0, st, b.idx_in, n);
}

void sample(float* logits, int n, float temperature, float top_p, float coin, unsigned long long* seed_dev,
            const SampleBufs& b, int* out, int* out2, hipStream_t st) {
    const int nchunks = (n + Q3_SM_CHUNK - 1) / Q3_SM_CHUNK;
    if (nchunks > Q3_SAMPLE_MAX_CHUNKS) {
        fprintf(stderr, "[q3hip] sample: vocabulary of %d entries is beyond %d\n", n, Q3_SAMPLE_MAX_CHUNKS * Q3_SM_CHUNK);
        exit(EXIT_FAILURE);
    }
    hipLaunchKernelGGL(k_sm_scale_max, dim3(nchunks), dim3(256), 0, st, logits, n, temperature, b.pmax);
    hipLaunchKernelGGL(k_sm_exp_sum, dim3(nchunks), dim3(256), 0, st, logits, n, b.pmax, nchunks, b.psum);
    hipLaunchKernelGGL(k_sm_divide, dim3(nchunks), dim3(256), 0, st, logits, n, b.psum, nchunks);
    size_t bytes = b.tmp_bytes;
    const hipError_t e = rocprim::radix_sort_pairs_desc(b.tmp, bytes, (const float*)logits, b.key_out, (const int*)b.idx_in,
                                                        b.idx_out, (size_t)n, 0, 32, st);
    if (e != hipSuccess) {
        fprintf(stderr, "[q3hip] sample: radix sort failed: %s\n", hipGetErrorString(e));
        exit(EXIT_FAILURE);
    }
    hipLaunchKernelGGL(k_nucleus, dim3(1), dim3(64), 0, st, b.key_out, b.idx_out, n, top_p, coin, seed_dev, out, out2);
}

}  // namespace q3k
