// Do fp64 MFMA and fp64 VALU share execution resources on gfx950?  Each SIMD hosts one wave of an
// "MFMA" block and one wave of an "FMA" block; compare the mixed run with each stream alone.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double v4d __attribute__((ext_vector_type(4)));
__global__ void __launch_bounds__(256) k(double* out, int iters, int mode) {
    // mode 0: all blocks MFMA; 1: all FMA; 2: even blocks MFMA, odd blocks FMA
    const bool mf = mode == 0 || (mode == 2 && (blockIdx.x & 1) == 0);
    double s = 0;
    if (mf) {
        v4d acc[4];
        for (int i = 0; i < 4; ++i) acc[i] = v4d{0, 0, 0, 0};
        double a = 1.0 + threadIdx.x * 1e-9, b = 1.0 - threadIdx.x * 1e-9;
        for (int it = 0; it < iters; ++it)
#pragma unroll
            for (int i = 0; i < 4; ++i) acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[i], 0, 0, 0);
        for (int i = 0; i < 4; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    } else {
        double x[8];
        for (int i = 0; i < 8; ++i) x[i] = 1.0 + i + threadIdx.x * 1e-9;
        for (int it = 0; it < iters * 8; ++it)   // 64 FMAs per outer MFMA-iteration-equivalent
#pragma unroll
            for (int i = 0; i < 8; ++i) x[i] = fma(x[i], 1.0000001, 1e-9);
        for (int i = 0; i < 8; ++i) s += x[i];
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
int main() {
    double* d; (void)hipMalloc(&d, 512 * 256 * sizeof(double));
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    const int iters = 20000;
    const char* names[3] = {"MFMA only (2 waves/SIMD)", "FMA only (2 waves/SIMD)", "1 MFMA wave + 1 FMA wave per SIMD"};
    for (int mode = 0; mode < 3; ++mode) {
        float ms = 0;
        for (int rep = 0; rep < 2; ++rep) {
            (void)hipEventRecord(e0);
            hipLaunchKernelGGL(k, dim3(512), dim3(256), 0, 0, d, iters, mode);
            (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
            (void)hipEventElapsedTime(&ms, e0, e1);
        }
        double mf_blocks = mode == 0 ? 512 : mode == 2 ? 256 : 0, fm_blocks = mode == 1 ? 512 : mode == 2 ? 256 : 0;
        double fl = mf_blocks * 4 * (double)iters * 4 * 2048 + fm_blocks * 256 * (double)iters * 8 * 8 * 2;
        printf("%-36s %.3f ms  %.1f TFLOP/s total\n", names[mode], ms, fl / (ms * 1e-3) / 1e12);
    }
    return 0;
}
