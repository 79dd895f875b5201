// Can a launch whose CUs sit idle (the short-context attention stage) warm the XCDs' L2s with the
// first part of the NEXT launch's weight stream?  Two questions, one program:
//   (1) placement: which XCD (HW_REG_XCC_ID) do blocks b of consecutive graph-captured launches of
//       different grid sizes land on -- is "block b of launch A shares an XCD with block b of launch B"
//       stable enough to aim a prefetch by block index?  (speed only: nothing depends on it)
//   (2) payoff: a consumer launch (256 workgroups x 1024 threads, each streaming its own S bytes once
//       with nt loads, as the gate/up GEMV does) behind a prefetch launch in which workgroup b reads the
//       first F bytes of region (b + shift) with default-policy loads; shift = 0 aims at the same
//       block index, shift = 1 at a neighbour (another XCD under round-robin placement).
// Every layer has its own weights (16 x 53 MB > Infinity Cache), so nothing is re-read across layers.
// Build: hipcc --offload-arch=gfx950 -O3 tools/micro/l2_prefetch.hip -o tools/micro/l2_prefetch.bin
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CHK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)
typedef int v4i __attribute__((ext_vector_type(4)));

__device__ __forceinline__ int xcc_id() { return __builtin_amdgcn_s_getreg((3 << 11) | (0 << 6) | 20) & 15; }

// prefetch: workgroup b touches the first nvec 16-byte units of region (b + shift) % nreg, default cache policy
__global__ __launch_bounds__(256) void k_prefetch(const v4i* __restrict__ base, size_t region_v, int nreg, int nvec, int shift,
                                                  int* sink, int* xcc) {
    const int reg = ((int)blockIdx.x + shift) % nreg;
    const v4i* p = base + (size_t)reg * region_v;
    v4i acc = {0, 0, 0, 0};
    for (int i = threadIdx.x; i < nvec; i += 256 * 8) {
        v4i v[8];
#pragma unroll
        for (int k = 0; k < 8; k++) v[k] = (i + k * 256 < nvec) ? p[i + k * 256] : acc;
#pragma unroll
        for (int k = 0; k < 8; k++) acc ^= v[k];
    }
    if ((acc.x ^ acc.y ^ acc.z ^ acc.w) == 0x12345678) sink[0] = 1;
    if (xcc && threadIdx.x == 0) xcc[blockIdx.x] = xcc_id();
}

// consumer: workgroup b streams region b once, nt loads, 8 in flight per thread; leaves its clocks
__global__ __launch_bounds__(1024) void k_consume(const v4i* __restrict__ base, size_t region_v, int nvec, int* sink, int* xcc,
                                                  unsigned long long* clk) {
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    const v4i* p = base + (size_t)blockIdx.x * region_v;
    v4i acc = {0, 0, 0, 0};
    for (int i = threadIdx.x; i < nvec; i += 1024 * 8) {
        v4i v[8];
#pragma unroll
        for (int k = 0; k < 8; k++) v[k] = (i + k * 1024 < nvec) ? __builtin_nontemporal_load(p + i + k * 1024) : acc;
#pragma unroll
        for (int k = 0; k < 8; k++) acc ^= v[k];
    }
    if ((acc.x ^ acc.y ^ acc.z ^ acc.w) == 0x12345678) sink[0] = 1;
    __syncthreads();
    if (threadIdx.x == 0) {
        if (xcc) xcc[blockIdx.x] = xcc_id();
        if (clk) {
            atomicMin(clk, t0);
            atomicMax(clk + 1, __builtin_amdgcn_s_memrealtime());
        }
    }
}

// a small dependent stage between prefetch and consumer (stands for the Wo epilogue): grid g
__global__ void k_small(int* sink, int* xcc) {
    if (xcc && threadIdx.x == 0) xcc[blockIdx.x] = xcc_id();
    if (threadIdx.x == 12345) sink[1] = 1;
}

int main() {
    const int L = 16, NREG = 256;
    const size_t S = 207 * 1024;                   // bytes per region: the gate/up share of one CU at Qwen3-4B (52.9 MB / 256)
    const size_t layer_bytes = S * NREG;
    v4i* buf; int* sink; int* xcc; unsigned long long* clk;
    CHK(hipMalloc(&buf, layer_bytes * L)); CHK(hipMalloc(&sink, 64)); CHK(hipMalloc(&xcc, 4 * 4096)); CHK(hipMalloc(&clk, 16 * L));
    CHK(hipMemset(buf, 1, layer_bytes * L)); CHK(hipMemset(xcc, 0xff, 4 * 4096));
    hipStream_t st; CHK(hipStreamCreate(&st));

    // ---- (1) placement across a graph of launches with the decode layer's grid sizes -------------
    {
        const int grids[6] = {256, 8, 256, 244, 256, 264};
        hipGraph_t g; hipGraphExec_t ex;
        CHK(hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal));
        for (int rep = 0; rep < 2; rep++)
            for (int k = 0; k < 6; k++) hipLaunchKernelGGL(k_small, dim3(grids[k]), dim3(256), 0, st, sink, xcc + (rep * 6 + k) * 272);
        CHK(hipStreamEndCapture(st, &g)); CHK(hipGraphInstantiate(&ex, g, nullptr, nullptr, 0));
        std::vector<int> h(4096);
        for (int run = 0; run < 3; run++) {
            CHK(hipGraphLaunch(ex, st)); CHK(hipStreamSynchronize(st));
            CHK(hipMemcpy(h.data(), xcc, 4 * 4096, hipMemcpyDeviceToHost));
            printf("placement, graph replay %d: per launch, XCC_ID of blocks 0..15, and #blocks whose XCC_ID == (XCC_ID of block 0 + b) %% 8\n", run);
            for (int k = 0; k < 12; k++) {
                const int* x = h.data() + k * 272; const int n = grids[k % 6];
                int rr = 0; for (int b = 0; b < n; b++) rr += (x[b] == (x[0] + b) % 8);
                printf("  launch %2d grid %3d:", k, n);
                for (int b = 0; b < 16 && b < n; b++) printf(" %d", x[b]);
                printf("   round-robin %d/%d\n", rr, n);
            }
        }
        CHK(hipGraphExecDestroy(ex)); CHK(hipGraphDestroy(g));
    }

    // ---- (2) payoff ---------------------------------------------------------------------------------
    auto run = [&](const char* name, double frac, int shift, bool small_between) {
        hipGraph_t g; hipGraphExec_t ex;
        const int nvec = (int)(S / 16), pvec = (int)(S * frac / 16);
        CHK(hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal));
        for (int l = 0; l < L; l++) {
            const v4i* base = buf + (size_t)l * (layer_bytes / 16);
            if (pvec > 0) hipLaunchKernelGGL(k_prefetch, dim3(NREG), dim3(256), 0, st, base, S / 16, NREG, pvec, shift, sink, (int*)nullptr);
            if (small_between) hipLaunchKernelGGL(k_small, dim3(256), dim3(256), 0, st, sink, (int*)nullptr);
            hipLaunchKernelGGL(k_consume, dim3(NREG), dim3(1024), 0, st, base, S / 16, nvec, sink, (int*)nullptr, clk + 2 * l);
        }
        CHK(hipStreamEndCapture(st, &g)); CHK(hipGraphInstantiate(&ex, g, nullptr, nullptr, 0));
        double best = 1e30, best_all = 1e30;
        hipEvent_t e0, e1; CHK(hipEventCreate(&e0)); CHK(hipEventCreate(&e1));
        for (int it = 0; it < 6; it++) {
            std::vector<unsigned long long> init(2 * L);
            for (int l = 0; l < L; l++) { init[2 * l] = ~0ull; init[2 * l + 1] = 0; }
            CHK(hipMemcpy(clk, init.data(), 16 * L, hipMemcpyHostToDevice));
            CHK(hipEventRecord(e0, st)); CHK(hipGraphLaunch(ex, st)); CHK(hipEventRecord(e1, st)); CHK(hipStreamSynchronize(st));
            float ms; CHK(hipEventElapsedTime(&ms, e0, e1));
            CHK(hipMemcpy(init.data(), clk, 16 * L, hipMemcpyDeviceToHost));
            double sum = 0; for (int l = 0; l < L; l++) sum += (double)(init[2 * l + 1] - init[2 * l]) * 0.01;
            if (it > 0 && sum / L < best) best = sum / L;
            if (it > 0 && ms * 1e3 / L < best_all) best_all = ms * 1e3 / L;
        }
        printf("%-64s consumer in-kernel %6.2f us   (prefetch + consumer per layer, events: %6.2f us)\n", name, best, best_all);
        CHK(hipGraphExecDestroy(ex)); CHK(hipGraphDestroy(g));
    };
    run("no prefetch", 0.0, 0, false);
    run("prefetch 10 % of each region, same block index", 0.10, 0, false);
    run("prefetch 20 %, same block index", 0.20, 0, false);
    run("prefetch 30 %, same block index", 0.30, 0, false);
    run("prefetch 40 %, same block index", 0.40, 0, false);
    run("prefetch 50 %, same block index", 0.50, 0, false);
    run("prefetch 30 %, neighbour block (shift 1)", 0.30, 1, false);
    run("prefetch 30 %, shift 8 (same XCD label, other block)", 0.30, 8, false);
    run("prefetch 30 %, same index, a small launch in between", 0.30, 0, true);
    run("no prefetch, a small launch in between", 0.0, 0, true);
    return 0;
}
