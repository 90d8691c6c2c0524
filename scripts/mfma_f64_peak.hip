// Micro-benchmark: issue rate of v_mfma_f64_16x16x4_f64, v_mfma_f64_4x4x4_4b_f64 and v_fma_f64 on gfx950.
// Build: hipcc --offload-arch=gfx950 -O3 scripts/mfma_f64_peak.hip -o scripts/mfma_f64_peak.bin
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double v4d __attribute__((ext_vector_type(4)));

template <int NACC>
__global__ void __launch_bounds__(256) k_mfma(double* out, int iters, double a0) {
    v4d acc[NACC];
    for (int i = 0; i < NACC; ++i) acc[i] = v4d{0, 0, 0, 0};
    double a = a0 + threadIdx.x * 1e-9, b = 1.0 + threadIdx.x * 1e-9;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[i], 0, 0, 0);
    }
    double s = 0;
    for (int i = 0; i < NACC; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

__global__ void __launch_bounds__(256) k_mfma4(double* out, int iters, double a0) {
    double acc[8];
    for (int i = 0; i < 8; ++i) acc[i] = 0.0;
    double a = a0 + threadIdx.x * 1e-9, b = 1.0 + threadIdx.x * 1e-9;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 8; ++i) acc[i] = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, acc[i], 0, 0, 0);
    }
    double s = 0;
    for (int i = 0; i < 8; ++i) s += acc[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

__global__ void __launch_bounds__(256) k_fma(double* out, int iters, double a0) {
    double x[8];
    for (int i = 0; i < 8; ++i) x[i] = a0 + i + threadIdx.x * 1e-9;
    const double m = 1.0000001, c = 1e-9;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 8; ++i) x[i] = fma(x[i], m, c);
    }
    double s = 0;
    for (int i = 0; i < 8; ++i) s += x[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

int main() {
    double* d;
    hipMalloc(&d, 256 * 8 * 256 * sizeof(double));
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    const int iters = 20000;
    for (int blocks_per_cu = 1; blocks_per_cu <= 2; ++blocks_per_cu) {
        int grid = 256 * blocks_per_cu;
        for (int nacc = 1; nacc <= 8; nacc *= 2) {
            for (int rep = 0; rep < 2; ++rep) {
                hipEventRecord(e0);
                if (nacc == 1) hipLaunchKernelGGL(k_mfma<1>, dim3(grid), dim3(256), 0, 0, d, iters, 1.0);
                if (nacc == 2) hipLaunchKernelGGL(k_mfma<2>, dim3(grid), dim3(256), 0, 0, d, iters, 1.0);
                if (nacc == 4) hipLaunchKernelGGL(k_mfma<4>, dim3(grid), dim3(256), 0, 0, d, iters, 1.0);
                if (nacc == 8) hipLaunchKernelGGL(k_mfma<8>, dim3(grid), dim3(256), 0, 0, d, iters, 1.0);
                hipEventRecord(e1);
                hipEventSynchronize(e1);
            }
            float ms;
            hipEventElapsedTime(&ms, e0, e1);
            double n = (double)grid * 4 * iters * nacc;   // MFMA instructions
            double tf = n * 2048.0 / (ms * 1e-3) / 1e12;
            // cycles per MFMA per SIMD at 2.4 GHz: each SIMD runs blocks_per_cu waves
            double cyc = (ms * 1e-3) * 2.4e9 / ((double)iters * nacc * blocks_per_cu);
            printf("mfma_f64_16x16x4 waves/SIMD=%d nacc=%d : %.3f ms  %.1f TFLOP/s  ~%.1f cyc/MFMA/SIMD @2.4GHz\n", blocks_per_cu, nacc, ms, tf, cyc);
        }
    }
    for (int blocks_per_cu = 1; blocks_per_cu <= 2; ++blocks_per_cu) {
        int grid = 256 * blocks_per_cu;
        for (int rep = 0; rep < 2; ++rep) {
            hipEventRecord(e0);
            hipLaunchKernelGGL(k_mfma4, dim3(grid), dim3(256), 0, 0, d, iters, 1.0);
            hipEventRecord(e1);
            hipEventSynchronize(e1);
        }
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        double n = (double)grid * 4 * iters * 8;   // 4x4x4_4b instructions: 4 blocks x 4x4x4 x 2 = 512 flop each
        double cyc = (ms * 1e-3) * 2.4e9 / ((double)iters * 8 * blocks_per_cu);
        printf("mfma_f64_4x4x4_4b waves/SIMD=%d nacc=8 : %.3f ms  %.1f TFLOP/s  ~%.1f cyc/MFMA/SIMD @2.4GHz\n", blocks_per_cu, ms,
               n * 512.0 / (ms * 1e-3) / 1e12, cyc);
    }
    for (int blocks_per_cu = 1; blocks_per_cu <= 4; blocks_per_cu *= 2) {
        int grid = 256 * blocks_per_cu;
        for (int rep = 0; rep < 2; ++rep) {
            hipEventRecord(e0);
            hipLaunchKernelGGL(k_fma, dim3(grid), dim3(256), 0, 0, d, iters, 1.0);
            hipEventRecord(e1);
            hipEventSynchronize(e1);
        }
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        double n = (double)grid * 256 * iters * 8;
        printf("v_fma_f64 waves/SIMD=%d : %.3f ms  %.1f TFLOP/s\n", blocks_per_cu, ms, n * 2 / (ms * 1e-3) / 1e12);
    }
    return 0;
}
