// What does the PLATFORM charge for the decode step's dependency structure, with the arithmetic taken out?
// A hipGraph of the step's dependent launches per token in which every launch only streams the bytes its stage reads
// (nt 16-byte loads, a rolling window of loads per thread) and then writes one small vector the next launch reads
// first (so the launches are truly dependent, as the residual makes them).  No norm, no quantise, no dot products,
// no reductions.  Qwen3-4B byte counts.
//
//   chain_floor.bin [4|5]      launches per layer: 4 = {qkv, attention+Wo (Wo's bytes), gate/up, down} (the round-3/4
//                              step, 146 launches), 5 = {qkv, attention, wo, gate/up, down} (round 2, 182 launches)
//
// Round 4 additions: the ARGS template parameter passes the stage's operands either as scalars (which the build flag
// `-mllvm -amdgpu-kernarg-preload-count=16` turns into SGPRs preloaded at wave launch) or inside a by-value struct
// (never preloaded: what the product kernels did until round 4); `empty` rows = the same graph with launches that
// read one dependent word and write one (the boundary alone).
// Build: hipcc --offload-arch=gfx950 -O3 [-mllvm -amdgpu-kernarg-preload-count=16] tools/micro/chain_floor.hip -o tools/micro/chain_floor[_pl].bin
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CHK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)
typedef int v4i __attribute__((ext_vector_type(4)));

struct StageArgs {                // by-value struct: lands in the kernarg segment, fetched by s_load
    const v4i* w; size_t nvec; const float* dep_in; float* dep_out;
    const float* pad0; const float* pad1; int d, n, ntasks, tw;     // the size of the product's Gemv struct
};

template <int DEPTH, int NT>
__device__ __forceinline__ void stage_body(const v4i* __restrict__ w, size_t nvec, const float* dep_in, float* dep_out) {
    const int tid = threadIdx.x;
    // the dependent read every stage starts with
    float d = dep_in[(blockIdx.x * 7 + tid) % 2560];
    const size_t per = (nvec + gridDim.x - 1) / gridDim.x;
    const size_t lo = (size_t)blockIdx.x * per;
    const size_t hi = lo + per < nvec ? lo + per : nvec;
    v4i acc = {0, 0, 0, 0};
    for (size_t i = lo + tid; i < hi; i += (size_t)NT * DEPTH) {
        v4i v[DEPTH];
#pragma unroll
        for (int k = 0; k < DEPTH; k++) {
            const size_t j = i + (size_t)k * NT;
            v[k] = j < hi ? __builtin_nontemporal_load(&w[j]) : acc;
        }
#pragma unroll
        for (int k = 0; k < DEPTH; k++) acc ^= v[k];
    }
    const int r = acc.x ^ acc.y ^ acc.z ^ acc.w;
    if (tid < 10) dep_out[(blockIdx.x * 10 + tid) % 2560] = d + (float)(r & 1);
}

template <int DEPTH, int NT>
__global__ __launch_bounds__(NT) void k_stage(const v4i* __restrict__ w, size_t nvec, const float* dep_in, float* dep_out) {
    stage_body<DEPTH, NT>(w, nvec, dep_in, dep_out);
}
template <int DEPTH, int NT>
__global__ __launch_bounds__(NT) void k_stage_struct(StageArgs a) {
    stage_body<DEPTH, NT>(a.w, a.nvec, a.dep_in, a.dep_out);
}
// the boundary alone: one dependent word in, one out
__global__ __launch_bounds__(1024) void k_empty(const float* dep_in, float* dep_out) {
    if (threadIdx.x == 0) dep_out[blockIdx.x] = dep_in[blockIdx.x] + 1.0f;
}

int main(int argc, char** argv) {
    const int L = 36;
    const int per_layer_launches = argc > 1 ? atoi(argv[1]) : 4;
    // bytes each stage reads (Qwen3-4B, Q8_0 codes + scales); attention at short context reads ~0.2 MB
    const size_t B_QKV = 16711680 + 0, B_ATT = 200000, B_WO = 11141120, B_GU = 52920320, B_DN = 26460160, B_CLS = 413265920;
    const size_t per_layer = B_QKV + B_ATT + B_WO + B_GU + B_DN;
    const size_t total = per_layer * L + B_CLS;
    char* buf; float *dA, *dB;
    CHK(hipMalloc(&buf, total + 4096)); CHK(hipMemset(buf, 1, total));
    CHK(hipMalloc(&dA, 2560 * 4)); CHK(hipMalloc(&dB, 2560 * 4));
    CHK(hipMemset(dA, 0, 2560 * 4)); CHK(hipMemset(dB, 0, 2560 * 4));
    hipStream_t st; CHK(hipStreamCreate(&st));
    hipEvent_t e0, e1; CHK(hipEventCreate(&e0)); CHK(hipEventCreate(&e1));
    float ms;
    struct V { int depth, nt, grid, strct; };
    const V vs[] = {{8, 1024, 256, 0}, {8, 1024, 256, 1}, {8, 512, 512, 0}, {8, 512, 512, 1}, {4, 1024, 256, 0}, {16, 512, 256, 0}};
    for (int variant = 0; variant < (int)(sizeof(vs) / sizeof(vs[0])); variant++) {
        const int grid = vs[variant].grid;
        hipGraph_t g; hipGraphExec_t ge;
        CHK(hipStreamBeginCapture(st, hipStreamCaptureModeGlobal));
        size_t off = 0; int n = 0;
        auto stage = [&](size_t bytes) {
            const float* in = (n & 1) ? dB : dA; float* out = (n & 1) ? dA : dB;
            const v4i* w = (const v4i*)(buf + off);
            StageArgs sa = {w, bytes / 16, in, out, in, in, 1, 2, 3, 4};
            switch (variant) {
                case 0: hipLaunchKernelGGL((k_stage<8, 1024>), dim3(grid), dim3(1024), 0, st, w, bytes / 16, in, out); break;
                case 1: hipLaunchKernelGGL((k_stage_struct<8, 1024>), dim3(grid), dim3(1024), 0, st, sa); break;
                case 2: hipLaunchKernelGGL((k_stage<8, 512>), dim3(grid), dim3(512), 0, st, w, bytes / 16, in, out); break;
                case 3: hipLaunchKernelGGL((k_stage_struct<8, 512>), dim3(grid), dim3(512), 0, st, sa); break;
                case 4: hipLaunchKernelGGL((k_stage<4, 1024>), dim3(grid), dim3(1024), 0, st, w, bytes / 16, in, out); break;
                default: hipLaunchKernelGGL((k_stage<16, 512>), dim3(grid), dim3(512), 0, st, w, bytes / 16, in, out); break;
            }
            off += bytes & ~(size_t)15; n++;
        };
        stage(10240);                                   // begin
        for (int l = 0; l < L; l++) {
            stage(B_QKV);
            if (per_layer_launches == 5) { stage(B_ATT); stage(B_WO); } else stage(B_ATT + B_WO);
            stage(B_GU); stage(B_DN);
        }
        stage(B_CLS);
        CHK(hipStreamEndCapture(st, &g));
        CHK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
        for (int i = 0; i < 5; i++) CHK(hipGraphLaunch(ge, st));
        CHK(hipStreamSynchronize(st));
        const int reps = 50;
        CHK(hipEventRecord(e0, st));
        for (int i = 0; i < reps; i++) CHK(hipGraphLaunch(ge, st));
        CHK(hipEventRecord(e1, st)); CHK(hipEventSynchronize(e1));
        CHK(hipEventElapsedTime(&ms, e0, e1));
        const double us = ms * 1e3 / reps;
        printf("%d workgroups x %d threads, %d loads in flight per thread, %s args: %d launches, %.1f us per token-equivalent = %.1f tok/s, %.2f TB/s = %.1f %% of 8 TB/s; per layer %.2f us\n",
               grid, vs[variant].nt, vs[variant].depth, vs[variant].strct ? "struct" : "scalar", n, us, 1e6 / us, total / us / 1e6, total / us / 1e6 / 8 * 100, (us - (double)B_CLS / 6.2e6) / L);
        // per-stage: the same launches one at a time, 36 dependent launches of one size per graph
        if (variant <= 1) {
            const size_t sizes[6] = {B_QKV, B_ATT, B_WO, B_ATT + B_WO, B_GU, B_DN}; const char* names[6] = {"qkv", "attn-sized", "wo", "attn+wo", "gate/up", "down"};
            for (int s = 0; s < 6; s++) {
                hipGraph_t g2; hipGraphExec_t ge2;
                CHK(hipStreamBeginCapture(st, hipStreamCaptureModeGlobal));
                size_t o2 = 0;
                for (int i = 0; i < 36; i++) {
                    const v4i* w = (const v4i*)(buf + o2);
                    const float* in = (i & 1) ? dB : dA; float* out = (i & 1) ? dA : dB;
                    StageArgs sa = {w, sizes[s] / 16, in, out, in, in, 1, 2, 3, 4};
                    if (variant == 0) hipLaunchKernelGGL((k_stage<8, 1024>), dim3(grid), dim3(1024), 0, st, w, sizes[s] / 16, in, out);
                    else hipLaunchKernelGGL((k_stage_struct<8, 1024>), dim3(grid), dim3(1024), 0, st, sa);
                    o2 += per_layer & ~(size_t)15;
                }
                CHK(hipStreamEndCapture(st, &g2));
                CHK(hipGraphInstantiate(&ge2, g2, nullptr, nullptr, 0));
                CHK(hipGraphLaunch(ge2, st)); CHK(hipStreamSynchronize(st));
                CHK(hipEventRecord(e0, st));
                for (int i = 0; i < 20; i++) CHK(hipGraphLaunch(ge2, st));
                CHK(hipEventRecord(e1, st)); CHK(hipEventSynchronize(e1));
                CHK(hipEventElapsedTime(&ms, e0, e1));
                const double u = ms * 1e3 / (20 * 36);
                printf("  %-10s %9zu B: %.2f us per dependent launch in a graph = %.2f TB/s\n", names[s], sizes[s], u, sizes[s] / u / 1e6);
            }
        }
    }
    // the boundary alone
    for (int wg = 256; wg <= 256; wg *= 2) {
        for (int nt = 64; nt <= 1024; nt *= 4) {
            hipGraph_t g; hipGraphExec_t ge;
            CHK(hipStreamBeginCapture(st, hipStreamCaptureModeGlobal));
            for (int i = 0; i < 146; i++) hipLaunchKernelGGL(k_empty, dim3(wg), dim3(nt), 0, st, (i & 1) ? dB : dA, (i & 1) ? dA : dB);
            CHK(hipStreamEndCapture(st, &g));
            CHK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
            CHK(hipGraphLaunch(ge, st)); CHK(hipStreamSynchronize(st));
            CHK(hipEventRecord(e0, st));
            for (int i = 0; i < 50; i++) CHK(hipGraphLaunch(ge, st));
            CHK(hipEventRecord(e1, st)); CHK(hipEventSynchronize(e1));
            CHK(hipEventElapsedTime(&ms, e0, e1));
            printf("empty: 146 dependent launches of %d workgroups x %d threads: %.2f us per launch\n", wg, nt, ms * 1e3 / (50 * 146));
        }
    }
    return 0;
}
