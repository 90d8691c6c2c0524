// Device-side helpers shared by the element-resident whole-iteration kernels (kernels_fused.hip: 20x20-point elements, one or
// several workgroups per element; kernels_tall.hip: tall elements split over many workgroups): the hand-managed AGPR stash,
// the 16 + 4 layer product, the barrier among the workgroups that share an element.
#pragma once
#include "hpv_mfma_dev.h"

// ---- explicit AGPR stash -------------------------------------------------------------------------------------------
// At one wave per SIMD a wave owns 512 registers: 256 architectural VGPRs + 256 accumulation registers (AGPRs).  The s
// values of the wave's tiles 1..6 (6 x L x 5 doubles per lane) are parked in the TOP AGPRs a[FZ_ABASE..255] by hand
// (v_accvgpr_write/read through inline asm): left to the register allocator, the same values end up behind PHI copies
// that move hundreds of registers per tile.  The compiler does not know these registers hold live values -- it only sees
// that a255 is clobbered, which makes the kernel descriptor reserve all 256 AGPRs -- so csrc/build.sh runs
// scripts/check_agpr.py on the generated assembly: if compiler-generated code touches a[FZ_ABASE..255] the library is built WITHOUT
// that kernel / instantiation (-DHPV_AGPR_GUARD_TRIPPED[_QT], reported by hpv_build_info); the build FAILS when the check itself
// cannot run (symbol renamed, no assembly).
template <int IDX>
__device__ __forceinline__ void acc_put(double v) {
    const int lo = __double2loint(v), hi = __double2hiint(v);
    asm volatile("v_accvgpr_write_b32 a[%2], %0\n\tv_accvgpr_write_b32 a[%3], %1" ::"v"(lo), "v"(hi), "n"(IDX), "n"(IDX + 1));
}
template <int IDX>
__device__ __forceinline__ double acc_get() {
    int lo, hi;
    asm volatile("v_accvgpr_read_b32 %0, a[%2]\n\tv_accvgpr_read_b32 %1, a[%3]" : "=v"(lo), "=v"(hi) : "n"(IDX), "n"(IDX + 1));
    return __hiloint2double(hi, lo);
}
template <int BASE, int N, int J = 0>
__device__ __forceinline__ void acc_put_all(const double (&sv)[N]) {
    if constexpr (J < N) { acc_put<BASE + 2 * J>(sv[J]); acc_put_all<BASE, N, J + 1>(sv); }
}
template <int BASE, int N, int J = 0>
__device__ __forceinline__ void acc_get_all(double (&sv)[N]) {
    if constexpr (J < N) { sv[J] = acc_get<BASE + 2 * J>(); acc_get_all<BASE, N, J + 1>(sv); }
}
// (the doubles J0 .. N - 1 only, at BASE: a stash place whose first J0 doubles live elsewhere)
template <int BASE, int N, int J0, int J = J0>
__device__ __forceinline__ void acc_put_from(const double (&sv)[N]) {
    if constexpr (J < N) { acc_put<BASE + 2 * (J - J0)>(sv[J]); acc_put_from<BASE, N, J0, J + 1>(sv); }
}
template <int BASE, int N, int J0, int J = J0>
__device__ __forceinline__ void acc_get_from(double (&sv)[N]) {
    if constexpr (J < N) { sv[J] = acc_get<BASE + 2 * (J - J0)>(); acc_get_from<BASE, N, J0, J + 1>(sv); }
}

// N doubles p[0], p[64], p[128], .. (one 512-byte wave row each) straight INTO the hand-managed AGPRs a[BASE + 2j : BASE + 2j + 1]:
// all loads in flight at once without a single compiler-allocated register (k_iter_fused<.., MULTI>: the spilled gradient sums of
// the workgroup's earlier elements come back in ONE memory round trip; through compiler registers it was either six serialized
// round trips or 90 more registers -- inside the stash).  Groups of eight share a base address (13-bit offset field).  The caller
// waits (acc_load_wait) before acc_get.
template <int BASE, int N, int J = 0>
__device__ __forceinline__ void acc_load_all(const double* p) {
    if constexpr (J < N) {
        asm volatile("global_load_dwordx2 a[%1:%2], %0, off offset:%3" ::"v"(p + (J / 8) * 512), "n"(BASE + 2 * J), "n"(BASE + 2 * J + 1),
                     "n"((J % 8) * 512)
                     : "memory");
        acc_load_all<BASE, N, J + 1>(p);
    }
}
__device__ __forceinline__ void acc_load_wait() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }

// hidden -> hidden product of one channel: z^T = W^T h^T (+ bias fragment for the value channel), 16 + 4 split
template <bool BIAS>
__device__ __forceinline__ void fz_layer(const double* WTl, const double* WRl, const double* BHl, int lofs,
                                         const double (&h)[MF_KS], double (&z)[MF_KS]) {
    v4d acc = BIAS ? v4d{BHl[lofs], BHl[64 + lofs], BHl[128 + lofs], BHl[192 + lofs]} : v4d{0.0, 0.0, 0.0, 0.0};
    double z16 = BIAS ? BHl[256 + lofs] : 0.0;
    const double* wrl = WRl + (lofs >> 4) * 4 + (lofs & 3);
#pragma unroll
    for (int s = 0; s < MF_KS; ++s) {
        acc = __builtin_amdgcn_mfma_f64_16x16x4f64(WTl[s * 64 + lofs], h[s], acc, 0, 0, 0);
        z16 = __builtin_amdgcn_mfma_f64_4x4x4f64(wrl[s * 16], h[s], z16, 0, 0, 0);
    }
#pragma unroll
    for (int s = 0; s < 4; ++s) z[s] = acc[s];
    z[4] = z16;
}

// An opaque use of a computed value: keeps the compiler from sinking its computation into one arm of a later per-lane select --
// which would turn the select into an exec-masked block of its own and cut the straight-line schedule into pieces (the packed
// quarter tiles choose per lane between the value-slot and the tangent-slot formulas all the time).
__device__ __forceinline__ void fz_keep(double& x) { asm volatile("" : "+v"(x)); }

// the same with the bias fragment scaled per lane (bm = 1 or 0): the packed quarter tile of k_iter_fused<.., QT> carries the value
// channel in some of its 16 point slots and the tangent channels, which take no bias, in the others
__device__ __forceinline__ void fz_layer_m(const double* WTl, const double* WRl, const double* BHl, int lofs, double bm,
                                           const double (&h)[MF_KS], double (&z)[MF_KS]) {
    v4d acc = v4d{BHl[lofs] * bm, BHl[64 + lofs] * bm, BHl[128 + lofs] * bm, BHl[192 + lofs] * bm};
    double z16 = BHl[256 + lofs] * bm;
    const double* wrl = WRl + (lofs >> 4) * 4 + (lofs & 3);
#pragma unroll
    for (int s = 0; s < MF_KS; ++s) {
        acc = __builtin_amdgcn_mfma_f64_16x16x4f64(WTl[s * 64 + lofs], h[s], acc, 0, 0, 0);
        z16 = __builtin_amdgcn_mfma_f64_4x4x4f64(wrl[s * 16], h[s], z16, 0, 0, 0);
    }
#pragma unroll
    for (int s = 0; s < 4; ++s) z[s] = acc[s];
    z[4] = z16;
}

// ---- tagged exchange among the workgroups that share an element ------------------------------------------------------------
// Workgroups that share an element (SPLIT mode of k_iter_fused: small shards of a multi-GPU run; k_iter_tall: few tall elements)
// hand each other a few hundred doubles in the middle of the launch.  Rounds 2 / 3a did it with write-through payload stores, a
// monotonic arrival counter per element and a reload: four dependent memory round trips (payload store acknowledged ->
// fetch-add -> last poll -> payload loads; 8.7 k cycles in SPLIT mode, 10 k = 4.4 us in k_iter_tall).  Here every exchanged
// double travels as TWO 8-byte
// granules {32 bits of the value | 32-bit launch tag}, each written by ONE write-through store (cdna_hip_programming.md
// Guideline 16, form R2: a naturally aligned 8-byte granule needs no ordering): a consumer polls the granules themselves until
// all carry this launch's tag, so the data's arrival is its own notification -- one one-way trip plus a poll sweep.
// The tag is *xiter + 1, read at kernel start.  xiter is advanced by the kernel that FOLLOWS the launch on the stream (k_finalize,
// thread 0 of its last block: `xiter_bump`), i.e. strictly after EVERY workgroup of this launch has ended -- also one that was
// dispatched late on a shared GPU, the case the time-out exists for.  (Round 3 let workgroup 0 advance it at its own end; a
// workgroup of another element dispatched after that could then publish with the next launch's tag: advisor, round 3.)  All partners are co-resident (the grid is
// at most one workgroup per CU); the wait is nevertheless bounded by wall clock.  A failed exchange must leave the replica
// intact (round-2 verdict / advisor): the workgroup that gives up sets the handle's sticky flag *xerr and EVERY thread of it
// leaves the kernel before it has written R, loss_e or its gradient row; k_finalize / k_adam / k_p2p_exchange read the flag
// (directly and, on the multi-GPU path, through the pad slot of the all-reduced buffer) and skip the update, the loss history
// and the beta powers; every later launch of the handle sees the flag at kernel start and stays away from the exchange, until
// the host has reported the failure (HpvError -7) and cleared it (hpv_api.hip, sync_check).
__device__ __forceinline__ void xg_publish(unsigned long long* slot, double v, unsigned tag) {
    const unsigned long long t = (unsigned long long)tag << 32;
    __hip_atomic_store(slot, t | (unsigned)__double2loint(v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __hip_atomic_store(slot + 1, t | (unsigned)__double2hiint(v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
// Gather `nwords` granules (nwords <= NIT * BLOCK) starting at `src` into the LDS array `dst` (the low halves, consecutively:
// granules 2i, 2i+1 become the double dst[i]).  Returns false when a granule did not show this launch's tag within ~0.2 s.
template <int NIT, int BLOCK>
__device__ __forceinline__ bool xg_gather(const unsigned long long* src, int nwords, unsigned tag, unsigned* dst, int tid) {
    unsigned long long w[NIT];
    bool done[NIT];
    int left = 0;
#pragma unroll
    for (int it = 0; it < NIT; ++it) { done[it] = it * BLOCK + tid >= nwords; left += done[it] ? 0 : 1; }
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    bool failed = false;
    while (left > 0) {
#pragma unroll
        for (int it = 0; it < NIT; ++it)
            if (!done[it]) w[it] = __hip_atomic_load(src + it * BLOCK + tid, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#pragma unroll
        for (int it = 0; it < NIT; ++it)
            if (!done[it] && (unsigned)(w[it] >> 32) == tag) { dst[it * BLOCK + tid] = (unsigned)w[it]; done[it] = true; --left; }
        if (left > 0) {
            if (__builtin_amdgcn_s_memrealtime() - t0 > 20000000ULL) { failed = true; break; }    // 0.2 s at 100 MHz
            __builtin_amdgcn_s_sleep(2);
        }
    }
    return !failed;
}
