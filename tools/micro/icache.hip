// Does a dependent launch start with a cold instruction cache?  k_probe times (s_memtime, shader clock) one pass over a
// straight-line block of 1024 four-byte instructions (4 KiB = 64 cache lines of code) right after entry, and a second
// pass over the SAME block; between two probes the graph runs k_other, whose workgroups execute ~40 KiB of different
// code (so a 64-KiB instruction cache shared by a CU pair would have to be thrashed to lose the block).
// Build: hipcc --offload-arch=gfx950 -O3 tools/micro/icache.hip -o tools/micro/icache.bin
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CHK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

__global__ __launch_bounds__(64) void k_probe(unsigned long long* out, int slot) {
    unsigned long long t0, t1, t2;
    int pass = 0;
    t0 = __builtin_amdgcn_s_memtime();
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    unsigned long long ta = t0, tb = 0, tc = 0;
again:
    asm volatile(".rept 1024\n s_nop 0\n .endr" ::: "memory");
    t1 = __builtin_amdgcn_s_memtime();
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    if (pass == 0) { tb = t1; pass = 1; asm volatile("" : "+s"(pass)); goto again; }
    tc = t1;
    (void)t2;
    if (threadIdx.x == 0) {
        out[(slot * gridDim.x + blockIdx.x) * 2 + 0] = tb - ta;
        out[(slot * gridDim.x + blockIdx.x) * 2 + 1] = tc - tb;
    }
}
// ~40 KiB of other straight-line code
__global__ __launch_bounds__(64) void k_other(float* x) {
    float v = x[threadIdx.x];
    asm volatile(".rept 10000\n v_add_f32 %0, %0, %0\n .endr" : "+v"(v));
    x[threadIdx.x] = v;
}
__global__ __launch_bounds__(64) void k_small(float* x) { x[threadIdx.x] += 1.0f; }

int main() {
    const int G = 256, NP = 16;
    unsigned long long* out; float* x;
    CHK(hipMalloc(&out, G * NP * 2 * 8 * 3)); CHK(hipMemset(out, 0, G * NP * 2 * 8 * 3));
    CHK(hipMalloc(&x, 4096)); CHK(hipMemset(x, 0, 4096));
    hipStream_t st; CHK(hipStreamCreate(&st));
    const char* names[3] = {"probe after a launch of 40 KiB of other code", "probe after a tiny other kernel", "probe after probe (same code)"};
    for (int mode = 0; mode < 3; mode++) {
        hipGraph_t g; hipGraphExec_t ge;
        CHK(hipStreamBeginCapture(st, hipStreamCaptureModeGlobal));
        for (int i = 0; i < NP; i++) {
            if (mode == 0) hipLaunchKernelGGL(k_other, dim3(G), dim3(64), 0, st, x);
            if (mode == 1) hipLaunchKernelGGL(k_small, dim3(G), dim3(64), 0, st, x);
            hipLaunchKernelGGL(k_probe, dim3(G), dim3(64), 0, st, out + (size_t)mode * G * NP * 2, i);
        }
        CHK(hipStreamEndCapture(st, &g));
        CHK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
        for (int r = 0; r < 3; r++) CHK(hipGraphLaunch(ge, st));
        CHK(hipStreamSynchronize(st));
        static unsigned long long h[256 * 16 * 2];
        CHK(hipMemcpy(h, out + (size_t)mode * G * NP * 2, sizeof(h), hipMemcpyDeviceToHost));
        double a = 0, b = 0; unsigned long long amax = 0, bmax = 0;
        for (int i = G * 2; i < G * NP; i++) { a += h[2 * i]; b += h[2 * i + 1]; if (h[2 * i] > amax) amax = h[2 * i]; if (h[2 * i + 1] > bmax) bmax = h[2 * i + 1]; }
        const int n = G * NP - G * 2;
        printf("%-48s first pass %.0f cycles (max %llu), second pass %.0f (max %llu): cold cost %.0f cycles per 4 KiB of code\n", names[mode], a / n, amax, b / n, bmax, (a - b) / n);
    }
    return 0;
}
