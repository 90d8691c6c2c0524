// What does a plain streaming READ of 2 GB reach on this chip?  (The ceiling the stand-alone residual kernel -- 7.2 KB read and
// 0.8 KB written per element -- is measured against: SURVEY.md 8d prices it at 8 TB/s.)  Every thread sums 16-byte loads with U
// of them in flight; grid-stride over the array; variants: loads in flight, workgroups per CU, non-temporal loads, and LDS-DMA
// (global_load_lds_dwordx4 into a ring, nothing read back).
// Build: hipcc --offload-arch=gfx950 -O3 -w scripts/hbm_read_probe.hip -o scripts/hbm_read_probe.bin
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double v2d __attribute__((ext_vector_type(2)));

template <int U, bool NT>
__global__ void __launch_bounds__(256) k_read(const v2d* __restrict__ a, long n, double* out) {
    const long stride = (long)gridDim.x * 256;
    double s = 0.0;
    long i = (long)blockIdx.x * 256 + threadIdx.x;
    for (; i + (U - 1) * stride < n; i += U * stride) {
        v2d v[U];
#pragma unroll
        for (int u = 0; u < U; ++u) v[u] = NT ? __builtin_nontemporal_load(a + i + u * stride) : a[i + u * stride];
#pragma unroll
        for (int u = 0; u < U; ++u) s += v[u][0] + v[u][1];
    }
    if (s == 1.2345e300) out[0] = s;
}

// LDS-DMA: every wave streams its slices into a private 8 KB ring (8 x 1 KB), waiting only so that at most 6 are outstanding
__global__ void __launch_bounds__(256) k_dma(const v2d* __restrict__ a, long n, double* out, int aux_nt) {
    extern __shared__ __attribute__((aligned(16))) double lds[];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    double* ring = lds + wv * 1024;                    // 8 KB per wave
    const long nw = (long)gridDim.x * 4, w = (long)blockIdx.x * 4 + wv;
    long i = w * 64;
    int k = 0;
    for (; i + 64 <= n; i += nw * 64, ++k) {
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(a + i + lane),
                                         (__attribute__((address_space(3))) void*)(ring + (k & 7) * 128), 16, 0, 0);
        asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (ring[lane] == 1.2345e300) out[0] = ring[lane];
}

template <typename F>
static void timeit(const char* name, F launch, double bytes) {
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    launch(); launch();
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0, 0);
    for (int r = 0; r < 10; ++r) launch();
    (void)hipEventRecord(e1, 0);
    (void)hipEventSynchronize(e1);
    float ms = 0;
    (void)hipEventElapsedTime(&ms, e0, e1);
    printf("%-64s %7.3f ms  %6.2f TB/s\n", name, ms / 10, bytes / (ms / 10) / 1e9);
}

int main() {
    const long n = 1L << 27;       // 16-byte elements: 2 GiB
    v2d* a; double* out;
    (void)hipMalloc(&a, n * 16);
    (void)hipMalloc(&out, 8);
    (void)hipMemset(a, 0, n * 16);
    const double bytes = (double)n * 16;
    for (int wgs : {2, 4, 8}) {
        const int grid = 256 * wgs;
        char nm[128];
        snprintf(nm, sizeof nm, "b128 loads, 4 in flight, %d workgroups of 256 per CU", wgs);
        timeit(nm, [&] { hipLaunchKernelGGL((k_read<4, false>), dim3(grid), dim3(256), 0, 0, a, n, out); }, bytes);
        snprintf(nm, sizeof nm, "b128 loads, 8 in flight, %d workgroups of 256 per CU", wgs);
        timeit(nm, [&] { hipLaunchKernelGGL((k_read<8, false>), dim3(grid), dim3(256), 0, 0, a, n, out); }, bytes);
        snprintf(nm, sizeof nm, "b128 loads, 16 in flight, %d workgroups of 256 per CU", wgs);
        timeit(nm, [&] { hipLaunchKernelGGL((k_read<16, false>), dim3(grid), dim3(256), 0, 0, a, n, out); }, bytes);
        snprintf(nm, sizeof nm, "b128 nt loads, 8 in flight, %d workgroups of 256 per CU", wgs);
        timeit(nm, [&] { hipLaunchKernelGGL((k_read<8, true>), dim3(grid), dim3(256), 0, 0, a, n, out); }, bytes);
    }
    for (int wgs : {1, 2, 4}) {
        char nm[128];
        snprintf(nm, sizeof nm, "LDS-DMA 1 KB pieces, 6 in flight per wave, %d workgroups per CU", wgs);
        timeit(nm, [&] { hipLaunchKernelGGL(k_dma, dim3(256 * wgs), dim3(256), 32768, 0, a, n, out, 0); }, bytes);
    }
    return 0;
}
