// Verdict round 5, item 3: would a fixed-order row reduction inside each XCD -- before the gradient rows leave the iteration kernel --
// pay?  Today k_iter_fused's 256 workgroups each store one gradient row (P = 921 doubles: 1.89 MB per iteration) and k_finalize<256>
// reads them back (16 columns per block, 256 rows).  The proposal: the 32 workgroups that share an L2 hand each other their rows as
// tagged granules (hpv_fused_dev.h, xg_*: the only ordering-free publication this chip offers), each reduces P/32 columns of its XCD's
// row in a fixed order, and the finalize kernel reads 8 rows instead of 256.
//
// This probe runs exactly the two tails, without the iteration in front of them, and times kernel PAIRS end to end:
//   A  k_rows<0>: every workgroup stores its row (plain stores)                          + k_fin over 256 rows (+ an Adam-shaped update)
//   B  k_rows<1>: publish the row as granules, gather my 29 columns from my 32 XCD        + k_fin over   8 rows
//                 partners (blockIdx % 8 equal), sum in member order, store into row[xcd]
// 256 workgroups of 256 threads with 100 KB of LDS each (one per CU, all co-resident: the hand-off needs that).  Also printed: the
// bytes each variant moves through device memory per iteration, and that both variants produce the same sums (to summation order).
// Build + run: hipcc --offload-arch=gfx950 -O3 -w scripts/rowreduce_probe.hip -o /tmp/rowreduce_probe && /tmp/rowreduce_probe
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <vector>

#define P 921
#define NWG 256
#define NX 8                    // XCDs
#define NM (NWG / NX)           // members of an XCD group
#define SLICE ((P + NM - 1) / NM)     // 29 columns per member

__device__ __forceinline__ double row_value(int b, int idx, int it) { return 1e-3 * (double)((b * 37 + idx * 11 + it) % 1009) - 0.5; }

template <int MODE>
__global__ void __launch_bounds__(256, 1) k_rows(double* GPART, unsigned long long* xg, double* ROW8, int it, int* err) {
    extern __shared__ double lds[];
    const int tid = threadIdx.x, b = blockIdx.x;
    for (int idx = tid; idx < P; idx += 256) lds[idx] = row_value(b, idx, it);
    __syncthreads();
    if constexpr (MODE == 0) {
        for (int idx = tid; idx < P; idx += 256) GPART[(long)b * P + idx] = lds[idx];
    } else {
        const unsigned long long tag = (unsigned long long)(unsigned)it << 32;
        for (int idx = tid; idx < P; idx += 256) {
            const double v = lds[idx];
            unsigned long long* s = xg + ((long)b * P + idx) * 2;
            __hip_atomic_store(s, tag | (unsigned)__double2loint(v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(s + 1, tag | (unsigned)__double2hiint(v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        const int group = b % NX, member = b / NX;
        const int c0 = member * SLICE, nc = min(SLICE, P - c0) > 0 ? min(SLICE, P - c0) : 0;
        // gather: NM partners x nc columns x 2 granules into lds[P + m * SLICE + c] (as 32-bit halves)
        unsigned* dst = (unsigned*)(lds + 1024);
        const int nwords = NM * nc * 2;
        constexpr int NIT = (NM * SLICE * 2 + 255) / 256;
        bool done[NIT];
        int left = 0;
#pragma unroll
        for (int k = 0; k < NIT; ++k) { done[k] = k * 256 + tid >= nwords; left += done[k] ? 0 : 1; }
        const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
        bool failed = false;
        while (left > 0) {
            unsigned long long w[NIT];
#pragma unroll
            for (int k = 0; k < NIT; ++k) {
                const int i = k * 256 + tid;
                if (!done[k]) {
                    const int m = i / (nc * 2), r = i % (nc * 2);
                    w[k] = __hip_atomic_load(xg + ((long)(m * NX + group) * P + c0) * 2 + r, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
            }
#pragma unroll
            for (int k = 0; k < NIT; ++k) {
                const int i = k * 256 + tid;
                if (!done[k] && (w[k] >> 32) == (unsigned long long)(unsigned)it) { dst[i] = (unsigned)w[k]; done[k] = true; --left; }
            }
            if (left > 0) {
                if (__builtin_amdgcn_s_memrealtime() - t0 > 5000000ULL) { failed = true; break; }      // 50 ms
                __builtin_amdgcn_s_sleep(2);
            }
        }
        if (failed) atomicExch(err, 1);
        __syncthreads();
        if (tid < nc) {
            const double* g = lds + 1024;
            double acc = 0.0;
            for (int m = 0; m < NM; ++m) acc += g[m * nc + tid];       // fixed order: member 0 .. 31
            ROW8[(long)group * P + c0 + tid] = acc;
        }
    }
}

// the finalize step's shape: 16 columns per block, 16 row groups in parallel, fixed-order combination, an Adam-shaped update
template <int ROWS>
__global__ void __launch_bounds__(256) k_fin(const double* __restrict__ G, double* RB, double* th, double* m, double* v) {
    __shared__ double red[256];
    const int c = threadIdx.x & 15, part = threadIdx.x >> 4, idx = blockIdx.x * 16 + c;
    double m0 = 0, v0 = 0, t0 = 0;
    const bool upd = part == 0 && idx < P;
    if (upd) { m0 = m[idx]; v0 = v[idx]; t0 = th[idx]; }
    double acc = 0.0;
    if (idx < P)
        for (int r = part; r < ROWS; r += 16) acc += G[(long)r * P + idx];
    red[part * 16 + c] = acc;
    __syncthreads();
    if (upd) {
        double t = 0.0;
        for (int k = 0; k < 16; ++k) t += red[k * 16 + c];
        RB[idx] = t;
        const double m1 = 0.9 * m0 + 0.1 * t, v1 = 0.999 * v0 + 0.001 * t * t;
        m[idx] = m1; v[idx] = v1; th[idx] = t0 - 1e-3 * m1 / (sqrt(v1) + 1e-8);
    }
}

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at line %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)

int main() {
    double *GPART, *ROW8, *RBa, *RBb, *th, *m, *v;
    unsigned long long* xg;
    int* err;
    CK(hipMalloc(&GPART, sizeof(double) * NWG * P)); CK(hipMalloc(&ROW8, sizeof(double) * NX * P));
    CK(hipMalloc(&RBa, sizeof(double) * P)); CK(hipMalloc(&RBb, sizeof(double) * P));
    CK(hipMalloc(&th, sizeof(double) * P)); CK(hipMalloc(&m, sizeof(double) * P)); CK(hipMalloc(&v, sizeof(double) * P));
    CK(hipMalloc(&xg, sizeof(unsigned long long) * NWG * P * 2)); CK(hipMemset(xg, 0, sizeof(unsigned long long) * NWG * P * 2));
    CK(hipMalloc(&err, sizeof(int))); CK(hipMemset(err, 0, sizeof(int)));
    CK(hipMemset(th, 0, sizeof(double) * P)); CK(hipMemset(m, 0, sizeof(double) * P)); CK(hipMemset(v, 0, sizeof(double) * P));
    const size_t lds = 100 * 1024;
    CK(hipFuncSetAttribute((const void*)k_rows<0>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    CK(hipFuncSetAttribute((const void*)k_rows<1>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipStream_t s;
    CK(hipStreamCreate(&s));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const int fin_blocks = (P + 15) / 16;
    int it = 0;
    auto run = [&](int mode, int n, bool with_fin, bool with_rows) {
        for (int i = 0; i < n; ++i) {
            ++it;
            if (with_rows) {
                if (mode == 0) hipLaunchKernelGGL(k_rows<0>, dim3(NWG), dim3(256), lds, s, GPART, xg, ROW8, it, err);
                else hipLaunchKernelGGL(k_rows<1>, dim3(NWG), dim3(256), lds, s, GPART, xg, ROW8, it, err);
            }
            if (with_fin) {
                if (mode == 0) hipLaunchKernelGGL(k_fin<NWG>, dim3(fin_blocks), dim3(256), 0, s, GPART, RBa, th, m, v);
                else hipLaunchKernelGGL(k_fin<NX>, dim3(fin_blocks), dim3(256), 0, s, ROW8, RBb, th, m, v);
            }
        }
    };
    auto timed = [&](int mode, bool with_fin, bool with_rows) -> double {
        run(mode, 100, with_fin, with_rows);
        double best = 1e30;
        for (int rep = 0; rep < 5; ++rep) {
            hipEventRecord(e0, s);
            run(mode, 1000, with_fin, with_rows);
            hipEventRecord(e1, s);
            hipEventSynchronize(e1);
            float ms = 0;
            hipEventElapsedTime(&ms, e0, e1);
            best = std::min(best, (double)ms);
        }
        return best;     // microseconds per iteration (ms per 1000)
    };
    // correctness: the same iteration number through both variants
    it = 7000; run(0, 1, true, true); const int itA = it;
    it = itA - 1; run(1, 1, true, true);
    CK(hipStreamSynchronize(s));
    std::vector<double> a(P), b(P);
    CK(hipMemcpy(a.data(), RBa, sizeof(double) * P, hipMemcpyDeviceToHost)); CK(hipMemcpy(b.data(), RBb, sizeof(double) * P, hipMemcpyDeviceToHost));
    double worst = 0, big = 0;
    for (int i = 0; i < P; ++i) { worst = std::max(worst, fabs(a[i] - b[i])); big = std::max(big, fabs(a[i])); }
    int herr = 0;
    CK(hipMemcpy(&herr, err, sizeof(int), hipMemcpyDeviceToHost));
    printf("sums of 256 rows, plain rows + 256-row finalize against XCD hand-off + 8-row finalize: max |difference| %.2e of %.2e (summation order); timeouts: %d\n", worst, big, herr);
    it = 10000;
    const double A = timed(0, true, true), B = timed(1, true, true);
    const double Ar = timed(0, false, true), Br = timed(1, false, true);
    const double Af = timed(0, true, false), Bf = timed(1, true, false);
    CK(hipMemcpy(&herr, err, sizeof(int), hipMemcpyDeviceToHost));
    printf("| variant | rows kernel + finalize kernel, us per pair | rows kernel alone | finalize kernel alone | bytes through device memory per iteration |\n|---|---|---|---|---|\n");
    printf("| A: every workgroup stores its row; finalize reads 256 rows | %.2f | %.2f | %.2f | %.2f MB written + %.2f MB read |\n", A, Ar, Af, NWG * P * 8 / 1e6, NWG * P * 8 / 1e6);
    printf("| B: tagged-granule hand-off inside each XCD; finalize reads 8 rows | %.2f | %.2f | %.2f | %.2f MB written (granules) + %.2f MB polled + %.2f MB rows |\n", B, Br, Bf,
           NWG * P * 16 / 1e6, NWG * P * 16 / 1e6, 2 * NX * P * 8 / 1e6);
    printf("B - A = %+.2f us per iteration (timeouts: %d)\n", B - A, herr);
    return 0;
}
