// Per-CU ingest rate of 16-byte vector loads: distinct HBM data vs a shared, L2-resident block.
// Build: hipcc --offload-arch=gfx950 -O3 tools/micro/ingest.hip -o gpurun_out/ingest ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CHK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)
typedef int v4i __attribute__((ext_vector_type(4)));

// every workgroup reads `bytes_per_wg` bytes starting at base + wg * stride (stride 0 = shared block),
// `reps` times over; 16 loads in flight per thread
template <int NT>
__global__ __launch_bounds__(NT) void k_read(const v4i* __restrict__ base, size_t stride_v, int nvec, int reps, int* out) {
    extern __shared__ int lds[];
    const v4i* p = base + (size_t)blockIdx.x * stride_v;
    v4i acc = {0, 0, 0, 0};
    for (int r = 0; r < reps; r++) {
        for (int i = threadIdx.x; i < nvec; i += NT * 8) {
            v4i v[8];
#pragma unroll
            for (int k = 0; k < 8; k++) v[k] = (i + k * NT < nvec) ? p[i + k * NT] : acc;
#pragma unroll
            for (int k = 0; k < 8; k++) acc ^= v[k];
        }
    }
    if ((acc.x ^ acc.y ^ acc.z ^ acc.w) == 0x12345678) out[0] = 1;
    if (threadIdx.x == 0) lds[0] = 0;
}

int main() {
    const size_t total = (size_t)1 << 30;
    v4i* buf; int* out;
    CHK(hipMalloc(&buf, total)); CHK(hipMalloc(&out, 4));
    CHK(hipMemset(buf, 1, total));
    hipEvent_t e0, e1; CHK(hipEventCreate(&e0)); CHK(hipEventCreate(&e1));
    auto run = [&](const char* name, int nt, int grid, size_t stride_bytes, size_t bytes_per_wg, int reps, int lds_bytes) {
        float best = 1e9f;
        for (int it = 0; it < 5; it++) {
            CHK(hipEventRecord(e0));
            if (nt == 256) hipLaunchKernelGGL(k_read<256>, dim3(grid), dim3(256), lds_bytes, 0, buf, stride_bytes / 16, (int)(bytes_per_wg / 16), reps, out);
            else if (nt == 512) hipLaunchKernelGGL(k_read<512>, dim3(grid), dim3(512), lds_bytes, 0, buf, stride_bytes / 16, (int)(bytes_per_wg / 16), reps, out);
            else hipLaunchKernelGGL(k_read<1024>, dim3(grid), dim3(1024), lds_bytes, 0, buf, stride_bytes / 16, (int)(bytes_per_wg / 16), reps, out);
            CHK(hipEventRecord(e1)); CHK(hipEventSynchronize(e1));
            float ms; CHK(hipEventElapsedTime(&ms, e0, e1));
            if (ms < best) best = ms;
        }
        const double us = best * 1e3, per_wg_kb = (double)bytes_per_wg * reps / 1024.0;
        printf("%-58s %8.1f us  %7.1f KB/us per workgroup  %6.2f TB/s chip\n", name, us, per_wg_kb / us, per_wg_kb * 1024.0 * grid / us / 1e6);
    };
    // distinct data from HBM: 256 workgroups (one per CU, 128 KB of LDS each), 2 MB each
    run("HBM distinct 2 MB/wg, 256 thr, 1 wg/CU", 256, 256, 2 << 20, 2 << 20, 1, 128 << 10);
    run("HBM distinct 2 MB/wg, 512 thr, 1 wg/CU", 512, 256, 2 << 20, 2 << 20, 1, 128 << 10);
    run("HBM distinct 2 MB/wg, 1024 thr, 1 wg/CU", 1024, 256, 2 << 20, 2 << 20, 1, 128 << 10);
    run("HBM distinct 1 MB/wg, 256 thr, 2 wg/CU", 256, 512, 1 << 20, 1 << 20, 1, 64 << 10);
    // shared block: L2 hits after the first touch
    run("shared 160 KB x 16 reps, 256 thr, 1 wg/CU", 256, 256, 0, 160 << 10, 16, 128 << 10);
    run("shared 160 KB x 16 reps, 512 thr, 1 wg/CU", 512, 256, 0, 160 << 10, 16, 128 << 10);
    run("shared 160 KB x 16 reps, 1024 thr, 1 wg/CU", 1024, 256, 0, 160 << 10, 16, 128 << 10);
    run("shared 32 KB x 64 reps (fits L1), 256 thr, 1 wg/CU", 256, 256, 0, 32 << 10, 64, 128 << 10);
    run("shared 16 KB x 128 reps (fits L1), 256 thr, 1 wg/CU", 256, 256, 0, 16 << 10, 128, 128 << 10);
    run("shared 640 KB x 4 reps, 256 thr, 1 wg/CU", 256, 256, 0, 640 << 10, 4, 128 << 10);
    run("own 160 KB x 16 reps (L2 per wg), 256 thr, 1 wg/CU", 256, 256, 160 << 10, 160 << 10, 16, 128 << 10);
    return 0;
}
