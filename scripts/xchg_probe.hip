// How long does a tagged-granule exchange among co-resident workgroups take, and does the memory scope of its stores / polls or the
// placement of the partners change that?  (k_iter_tall waits ~10 k cycles per launch for 25 partial sums of 31 partners; the SPLIT
// mode of k_iter_fused for 800 channel values.)
//
// 256 workgroups (one per CU: 100 KB of LDS each), groups of 32 partners.  Every round all workgroups wait for a common deadline on
// the 100 MHz real-time counter, publish 25 doubles as {32 data bits | 32-bit tag} granule pairs and poll their 32 partners' 1 600
// granules until all carry the round's tag; the time from the deadline to the last arrival is recorded per workgroup.
//   stores / polls: agent scope (what the kernels use), system scope, workgroup scope (sc0: served by the XCD's L2)
//   partners:       blockIdx % 8 equal (the same XCD if workgroups are dealt round-robin)  |  32 consecutive blockIdx (all XCDs)
// Build: hipcc --offload-arch=gfx950 -O3 -w scripts/xchg_probe.hip -o scripts/xchg_probe.bin
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <vector>

#define NV 25
#define NPART 32
#define ROUNDS 40
#define PERIOD 3000ULL          // 30 us between rounds (100 MHz ticks)
#define TIMEOUT 2000ULL         // 20 us: a poll that has not seen everything by then gives up (counted)

template <int SC>
__device__ __forceinline__ void st(unsigned long long* p, unsigned long long v) {
    if constexpr (SC == 0) __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    else if constexpr (SC == 1) __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    else __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}
template <int SC>
__device__ __forceinline__ unsigned long long ld(const unsigned long long* p) {
    if constexpr (SC == 0) return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    else if constexpr (SC == 1) return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    else return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}

// GROUPING 0: group = blockIdx % 8, member = blockIdx / 8;  1: group = blockIdx / 32, member = blockIdx % 32
template <int SST, int SLD, int GROUPING>
__global__ void __launch_bounds__(256, 1) k_xchg(unsigned long long* gran, unsigned long long* gbase, float* out, int* xcc) {
    extern __shared__ double lds[];
    const int tid = threadIdx.x, b = blockIdx.x;
    const int group = GROUPING == 0 ? b % 8 : b / NPART, member = GROUPING == 0 ? b / 8 : b % NPART;
    if (tid == 0) {
        int x;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(x));
        xcc[b] = x & 0xf;
        const unsigned long long now = __builtin_amdgcn_s_memrealtime();
        atomicCAS(gbase, 0ULL, now + 5000ULL);        // the first workgroup fixes the time base: 50 us from now
    }
    __syncthreads();
    const unsigned long long base = __hip_atomic_load(gbase, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    unsigned long long* mine = gran + ((long)group * NPART + member) * NV * 2;
    const unsigned long long* all = gran + (long)group * NPART * NV * 2;
    constexpr int NW = NPART * NV * 2, NIT = (NW + 255) / 256;
    for (int r = 1; r <= ROUNDS; ++r) {
        const unsigned long long t0 = base + (unsigned long long)r * PERIOD;
        while (__builtin_amdgcn_s_memrealtime() < t0) __builtin_amdgcn_s_sleep(1);
        const unsigned long long tag = (unsigned long long)r << 32;
        if (tid < NV) {
            const double v = (double)(b * 100 + tid) + 0.5 * r;
            st<SST>(mine + 2 * tid, tag | (unsigned)__double2loint(v));
            st<SST>(mine + 2 * tid + 1, tag | (unsigned)__double2hiint(v));
        }
        bool done[NIT];
        int left = 0, sweeps = 0;
#pragma unroll
        for (int it = 0; it < NIT; ++it) { done[it] = it * 256 + tid >= NW; left += done[it] ? 0 : 1; }
        bool failed = false;
        while (left > 0) {
            unsigned long long w[NIT];
#pragma unroll
            for (int it = 0; it < NIT; ++it) if (!done[it]) w[it] = ld<SLD>(all + it * 256 + tid);
#pragma unroll
            for (int it = 0; it < NIT; ++it) if (!done[it] && (w[it] >> 32) == (unsigned long long)r) { done[it] = true; --left; }
            ++sweeps;
            if (left > 0 && __builtin_amdgcn_s_memrealtime() - t0 > TIMEOUT) { failed = true; break; }
        }
        const int any_failed = __syncthreads_or(failed ? 1 : 0);
        const unsigned long long t1 = __builtin_amdgcn_s_memrealtime();
        const int max_sweeps = __syncthreads_or(0) * 0 + sweeps;      // (thread 0's count is representative enough)
        if (tid == 0) {
            out[((long)r * 256 + b) * 2] = any_failed ? -1.0f : (float)(t1 - t0) * 0.01f;     // microseconds
            out[((long)r * 256 + b) * 2 + 1] = (float)max_sweeps;
        }
    }
}

template <int SST, int SLD, int GROUPING>
static void run(const char* name, unsigned long long* gran, unsigned long long* gbase, float* out, int* xcc) {
    const size_t lds = 100 * 1024;
    (void)hipFuncSetAttribute((const void*)k_xchg<SST, SLD, GROUPING>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    (void)hipMemset(gran, 0, 8 * NPART * NV * 2 * 8);
    (void)hipMemset(gbase, 0, 8);
    hipLaunchKernelGGL((k_xchg<SST, SLD, GROUPING>), dim3(256), dim3(256), lds, 0, gran, gbase, out, xcc);
    hipError_t e = hipDeviceSynchronize();
    if (e != hipSuccess) { printf("%s: %s\n", name, hipGetErrorString(e)); return; }
    std::vector<float> h((size_t)(ROUNDS + 1) * 256 * 2);
    std::vector<int> hx(256);
    (void)hipMemcpy(h.data(), out, h.size() * 4, hipMemcpyDeviceToHost);
    (void)hipMemcpy(hx.data(), xcc, 256 * 4, hipMemcpyDeviceToHost);
    std::vector<float> lat;
    int fails = 0;
    double sw = 0;
    for (int r = 5; r <= ROUNDS; ++r)
        for (int b = 0; b < 256; ++b) {
            const float v = h[((size_t)r * 256 + b) * 2];
            if (v < 0) ++fails; else { lat.push_back(v); sw += h[((size_t)r * 256 + b) * 2 + 1]; }
        }
    std::sort(lat.begin(), lat.end());
    int same = 0;      // workgroups whose XCC id equals blockIdx % 8
    for (int b = 0; b < 256; ++b) same += hx[b] == b % 8;
    if (lat.empty()) { printf("%-58s  every poll timed out (%d)\n", name, fails); return; }
    printf("%-58s  median %5.2f us  p10 %5.2f  p90 %5.2f  max %5.2f  sweeps %.1f  timed out %d  (XCC id == blockIdx %% 8: %d / 256)\n", name,
           lat[lat.size() / 2], lat[lat.size() / 10], lat[lat.size() * 9 / 10], lat.back(), sw / lat.size(), fails, same);
}

int main() {
    unsigned long long *gran, *gbase;
    float* out; int* xcc;
    (void)hipMalloc(&gran, 8 * NPART * NV * 2 * 8);
    (void)hipMalloc(&gbase, 8);
    (void)hipMalloc(&out, (size_t)(ROUNDS + 1) * 256 * 2 * 4);
    (void)hipMalloc(&xcc, 256 * 4);
    for (int rep = 0; rep < 2; ++rep) {
        run<0, 0, 0>("store agent,     poll agent,     partners b % 8 equal", gran, gbase, out, xcc);
        run<0, 0, 1>("store agent,     poll agent,     partners consecutive", gran, gbase, out, xcc);
        run<1, 1, 0>("store system,    poll system,    partners b % 8 equal", gran, gbase, out, xcc);
        run<0, 2, 0>("store agent,     poll workgroup, partners b % 8 equal", gran, gbase, out, xcc);
        run<2, 2, 0>("store workgroup, poll workgroup, partners b % 8 equal", gran, gbase, out, xcc);
        run<2, 0, 0>("store workgroup, poll agent,     partners b % 8 equal", gran, gbase, out, xcc);
        run<2, 2, 1>("store workgroup, poll workgroup, partners consecutive", gran, gbase, out, xcc);
    }
    return 0;
}
