// Instruction-level probe of the gfx950 fp64 datapath, ONE wave per SIMD (the regime of k_iter_fused): shader cycles (s_memtime)
// and wall clock (s_memrealtime, 100 MHz) per instruction for short bursts of
//   v_mfma_f64_16x16x4_f64 / v_mfma_f64_4x4x4_4b_f64 / v_fma_f64, independent and dependent chains, and mixes of them,
// with the whole chip busy (256 workgroups) or one CU alone (no power limit in play).
// Build: hipcc --offload-arch=gfx950 -O3 scripts/f64_issue_probe.hip -o scripts/f64_issue_probe.bin
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef double v4d __attribute__((ext_vector_type(4)));

#define REPS 32

struct Stamp { long long cyc, wall; };

template <int MODE>
__global__ void __launch_bounds__(256, 1) k_probe(double* out, Stamp* st, int iters, double a0) {
    const int lane = threadIdx.x & 63;
    double a = a0 + lane * 1e-9, b = 1.0 + lane * 1e-9;
    v4d acc[6];
    double s4[8], x[10];
    int ix[8];
    __shared__ double sm[2048];
    for (int i = threadIdx.x; i < 2048; i += 256) sm[i] = 1e-9 * i;
    for (int i = 0; i < 8; ++i) ix[i] = lane + i;
    for (int i = 0; i < 6; ++i) acc[i] = v4d{0, 0, 0, 0};
    for (int i = 0; i < 8; ++i) s4[i] = 0.0;
    for (int i = 0; i < 10; ++i) x[i] = a0 + i + lane * 1e-9;
    const double m = 1.0000001, c = 1e-9;
    __syncthreads();
    const long long t0 = clock64(), w0 = wall_clock64();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int r = 0; r < REPS; ++r) {
            if constexpr (MODE == 0) {          // 16x16x4, 4 independent accumulators
#pragma unroll
                for (int i = 0; i < 4; ++i) acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[i], 0, 0, 0);
            } else if constexpr (MODE == 1) {   // 16x16x4, ONE dependent chain (x4 per rep)
#pragma unroll
                for (int i = 0; i < 4; ++i) acc[0] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[0], 0, 0, 0);
            } else if constexpr (MODE == 2) {   // 4x4x4_4b, 8 independent
#pragma unroll
                for (int i = 0; i < 8; ++i) s4[i] = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, s4[i], 0, 0, 0);
            } else if constexpr (MODE == 3) {   // 4x4x4_4b, one dependent chain (x8)
#pragma unroll
                for (int i = 0; i < 8; ++i) s4[0] = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, s4[0], 0, 0, 0);
            } else if constexpr (MODE == 4) {   // v_fma_f64, 10 independent
#pragma unroll
                for (int i = 0; i < 10; ++i) x[i] = fma(x[i], m, c);
            } else if constexpr (MODE == 5) {   // v_fma_f64, one dependent chain (x10)
#pragma unroll
                for (int i = 0; i < 10; ++i) x[0] = fma(x[0], m, c);
            } else if constexpr (MODE == 6) {   // v_fma_f64, two dependent chains interleaved (x5 each)
#pragma unroll
                for (int i = 0; i < 5; ++i) { x[0] = fma(x[0], m, c); x[1] = fma(x[1], m, c); }
            } else if constexpr (MODE == 7) {   // 1 large MFMA + 8 independent FMAs
                acc[r & 3] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[r & 3], 0, 0, 0);
#pragma unroll
                for (int i = 0; i < 8; ++i) x[i] = fma(x[i], m, c);
            } else if constexpr (MODE == 8) {   // 1 large MFMA + 16 independent FMAs
                acc[r & 3] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[r & 3], 0, 0, 0);
#pragma unroll
                for (int i = 0; i < 8; ++i) x[i] = fma(x[i], m, c);
#pragma unroll
                for (int i = 0; i < 8; ++i) x[i] = fma(x[i], m, c);
            } else if constexpr (MODE == 9) {   // 1 large + 1 small MFMA (the kernels' 16 + 4 split)
                acc[r & 3] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[r & 3], 0, 0, 0);
                s4[r & 7] = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, s4[r & 7], 0, 0, 0);
            } else if constexpr (MODE == 10) {  // 1 small MFMA + 2 independent FMAs
                s4[r & 7] = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, s4[r & 7], 0, 0, 0);
                x[r & 7] = fma(x[r & 7], m, c);
                x[(r + 4) & 7] = fma(x[(r + 4) & 7], m, c);
            } else if constexpr (MODE == 11) {  // v_fma_f64, 4 chains interleaved
#pragma unroll
                for (int i = 0; i < 2; ++i) { x[0] = fma(x[0], m, c); x[1] = fma(x[1], m, c); x[2] = fma(x[2], m, c); x[3] = fma(x[3], m, c); }
            } else if constexpr (MODE == 12) {  // 32-bit moves (v_accvgpr traffic stands in as v_mov of two halves)
#pragma unroll
                for (int i = 0; i < 10; ++i) {
                    int lo = __double2loint(x[i]);
                    asm volatile("v_accvgpr_write_b32 a0, %0\n\ts_nop 0\n\tv_accvgpr_read_b32 %0, a0" : "+v"(lo)::"a0");
                    x[i] = __hiloint2double(__double2hiint(x[i]), lo);
                }
            } else if constexpr (MODE == 13) {  // v_rcp_f64 independent x8
#pragma unroll
                for (int i = 0; i < 8; ++i) x[i] = __builtin_amdgcn_rcp(x[i]);
            } else if constexpr (MODE == 14) {  // v_mul_f64 dependent on MFMA result: MFMA -> read -> FMA -> MFMA operand (full latency exposure)
                acc[0] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[0], 0, 0, 0);
                b = fma(acc[0][0], 1e-30, b);
            } else if constexpr (MODE == 15) {  // same with the small MFMA
                s4[0] = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, s4[0], 0, 0, 0);
                b = fma(s4[0], 1e-30, b);
            } else if constexpr (MODE == 16) {  // one channel-layer product as fz_layer writes it: L S L S L S L S L S on one accumulator pair
#pragma unroll
                for (int i = 0; i < 5; ++i) {
                    acc[0] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[0], 0, 0, 0);
                    s4[0] = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, s4[0], 0, 0, 0);
                }
            } else if constexpr (MODE == 17) {  // the same product grouped: L L L L L then S S S S S
#pragma unroll
                for (int i = 0; i < 5; ++i) acc[0] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[0], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int i = 0; i < 5; ++i) s4[0] = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, s4[0], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
            } else if constexpr (MODE == 18) {  // three channels: 15 L (chains interleaved) then 15 S (chains interleaved)
#pragma unroll
                for (int i = 0; i < 5; ++i)
#pragma unroll
                    for (int c = 0; c < 3; ++c) acc[c] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[c], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int i = 0; i < 5; ++i)
#pragma unroll
                    for (int c = 0; c < 3; ++c) s4[c] = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, s4[c], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
            } else if constexpr (MODE == 19) {  // three channels, pairs interleaved: (L0 S0 L1 S1 L2 S2) x 5
#pragma unroll
                for (int i = 0; i < 5; ++i)
#pragma unroll
                    for (int c = 0; c < 3; ++c) {
                        acc[c] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[c], 0, 0, 0);
                        s4[c] = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, s4[c], 0, 0, 0);
                    }
            } else if constexpr (MODE == 20) {  // 1 L + 8 independent 32-bit VALU (v_mov / integer): do they hide under the MFMA?
                acc[r & 3] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[r & 3], 0, 0, 0);
#pragma unroll
                for (int i = 0; i < 8; ++i) ix[i] = ix[i] * 3 + 1;
            } else if constexpr (MODE == 21) {  // 8 independent 32-bit VALU alone
#pragma unroll
                for (int i = 0; i < 8; ++i) ix[i] = ix[i] * 3 + 1;
            } else if constexpr (MODE == 22) {  // 1 L + 4 independent ds_read_b64 (waited for at the end of the rep)
                acc[r & 3] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[r & 3], 0, 0, 0);
#pragma unroll
                for (int i = 0; i < 4; ++i) x[i] += sm[(lane + 64 * i + r) & 1023];
            } else if constexpr (MODE == 23) {  // 4 independent ds_read_b64 + 4 adds alone
#pragma unroll
                for (int i = 0; i < 4; ++i) x[i] += sm[(lane + 64 * i + r) & 1023];
            } else if constexpr (MODE == 24) {  // 1 L + 8 v_accvgpr_read of other accumulators into VGPRs that nothing waits for
                acc[0] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[0], 0, 0, 0);
#pragma unroll
                for (int i = 0; i < 8; ++i) asm volatile("v_accvgpr_read_b32 %0, a[%1]" : "=v"(ix[i]) : "n"(200 + i));
            } else if constexpr (MODE == 25) {  // 8 v_accvgpr_read alone
#pragma unroll
                for (int i = 0; i < 8; ++i) asm volatile("v_accvgpr_read_b32 %0, a[%1]" : "=v"(ix[i]) : "n"(200 + i));
            } else if constexpr (MODE == 26) {  // 1 S + 2 v_accvgpr_read
                s4[r & 7] = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, s4[r & 7], 0, 0, 0);
#pragma unroll
                for (int i = 0; i < 2; ++i) asm volatile("v_accvgpr_read_b32 %0, a[%1]" : "=v"(ix[i]) : "n"(200 + i));
            } else if constexpr (MODE == 27) {  // 1 L + 4 ds_write_b64
                acc[r & 3] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[r & 3], 0, 0, 0);
#pragma unroll
                for (int i = 0; i < 4; ++i) sm[lane + 64 * i + 1024 * (r & 1)] = x[i];
            } else if constexpr (MODE == 28) {  // 1 L + 8 dependent-free v_mul_f64 by constants from SGPRs (the Horner shape)
                acc[r & 3] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[r & 3], 0, 0, 0);
#pragma unroll
                for (int i = 0; i < 8; ++i) x[i] = fma(x[i], x[i], 0.5);
            }
        }
    }
    const long long t1 = clock64(), w1 = wall_clock64();
    double s = b;
    for (int i = 0; i < 6; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    for (int i = 0; i < 8; ++i) s += s4[i];
    for (int i = 0; i < 10; ++i) s += x[i];
    for (int i = 0; i < 8; ++i) s += ix[i];
    if (MODE >= 24 && MODE <= 26) asm volatile("" ::: "a255");
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (lane == 0) st[blockIdx.x * 4 + (threadIdx.x >> 6)] = Stamp{t1 - t0, w1 - w0};
}

static const char* NAMES[] = {"mfma16 x4 indep", "mfma16 dependent chain", "mfma4_4b x8 indep", "mfma4_4b dependent chain", "fma_f64 x10 indep",
                              "fma_f64 dependent chain", "fma_f64 2 chains", "1 mfma16 + 8 fma", "1 mfma16 + 16 fma", "1 mfma16 + 1 mfma4",
                              "1 mfma4 + 2 fma", "fma_f64 4 chains", "accvgpr write+read pairs x10", "rcp_f64 x8 indep",
                              "mfma16 -> fma -> mfma16 round trip", "mfma4 -> fma -> mfma4 round trip",
                              "channel product L S L S.. (per product)", "channel product LLLLL SSSSS", "3 channels 15 L then 15 S (per 3)",
                              "3 channels (L S) pairs interleaved (per 3)", "1 mfma16 + 8 int VALU", "8 int VALU alone", "1 mfma16 + 4 ds_read_b64",
                              "4 ds_read_b64 + add alone", "1 mfma16 + 8 accvgpr_read", "8 accvgpr_read alone", "1 mfma4 + 2 accvgpr_read",
                              "1 mfma16 + 4 ds_write_b64", "1 mfma16 + 8 fma(x,x,c)"};
static const int PER_REP[] = {4, 4, 8, 8, 10, 10, 10, 1, 1, 1, 1, 8, 10, 8, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1};   // "units" per rep the per-unit figures refer to

template <int MODE>
static void run(double* d, Stamp* dst, int grid, int iters) {
    hipLaunchKernelGGL(k_probe<MODE>, dim3(grid), dim3(256), 0, 0, d, dst, iters, 1.0);
    hipLaunchKernelGGL(k_probe<MODE>, dim3(grid), dim3(256), 0, 0, d, dst, iters, 1.0);
    (void)hipDeviceSynchronize();
    std::vector<Stamp> h(grid * 4);
    (void)hipMemcpy(h.data(), dst, h.size() * sizeof(Stamp), hipMemcpyDeviceToHost);
    double cyc = 0, wall = 0;
    for (auto& s : h) { cyc += (double)s.cyc; wall += (double)s.wall; }
    cyc /= h.size(); wall /= h.size();
    const double units = (double)iters * REPS * PER_REP[MODE];
    printf("%-38s grid=%3d iters=%3d : %8.1f cycles/unit  %7.2f ns/unit  clock %.2f GHz  (burst %.1f us)\n", NAMES[MODE], grid, iters, cyc / units,
           wall * 10.0 / units, cyc / (wall * 10.0), wall * 0.01);
}

int main() {
    double* d;
    Stamp* st;
    (void)hipMalloc(&d, 256 * 256 * sizeof(double));
    (void)hipMalloc(&st, 256 * 4 * sizeof(Stamp));
    for (int grid : {256}) {
        for (int iters : {8}) {
            run<16>(d, st, grid, iters); run<17>(d, st, grid, iters); run<18>(d, st, grid, iters); run<19>(d, st, grid, iters);
            run<20>(d, st, grid, iters); run<21>(d, st, grid, iters); run<22>(d, st, grid, iters); run<23>(d, st, grid, iters);
            run<24>(d, st, grid, iters); run<25>(d, st, grid, iters); run<26>(d, st, grid, iters); run<27>(d, st, grid, iters);
            run<28>(d, st, grid, iters);
            run<0>(d, st, grid, iters); run<1>(d, st, grid, iters); run<2>(d, st, grid, iters); run<3>(d, st, grid, iters);
            run<4>(d, st, grid, iters); run<5>(d, st, grid, iters); run<6>(d, st, grid, iters); run<11>(d, st, grid, iters);
            run<7>(d, st, grid, iters); run<8>(d, st, grid, iters); run<9>(d, st, grid, iters); run<10>(d, st, grid, iters);
            run<12>(d, st, grid, iters); run<13>(d, st, grid, iters); run<14>(d, st, grid, iters); run<15>(d, st, grid, iters);
            printf("\n");
        }
    }
    return 0;
}
