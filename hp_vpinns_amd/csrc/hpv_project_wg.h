// Workgroup-per-element projection as a device function (shared by kernels_project.hip and the fused
// forward+projection kernel in kernels_mfma.hip).
#pragma once
#include "hpv_internal.h"

// Lane exchanges without an LDS round trip (__shfl_xor compiles to ds_bpermute_b32 pairs, ~100 cycles of latency each in the
// dependent chain of a reduction): inside a row of 16 lanes DPP moves (one VALU instruction per half), across rows gfx950's
// permlane swaps (v_permlane16_swap exchanges the odd rows of one register with the even rows of another, v_permlane32_swap
// the upper half of one with the lower half of the other; with both registers = v the two results add up to v[l] + v[l ^ 16]
// resp. v[l] + v[l ^ 32] in every lane).
template <int CTRL>
__device__ __forceinline__ double pj_dpp(double v) {
    // (bound_ctrl: a lane whose source lies outside its row reads 0 -- the row shifts rely on it -- and no `old` operand has to
    //  be zeroed first)
    const int lo = __builtin_amdgcn_mov_dpp(__double2loint(v), CTRL, 0xf, 0xf, true);
    const int hi = __builtin_amdgcn_mov_dpp(__double2hiint(v), CTRL, 0xf, 0xf, true);
    return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double pj_xrow16(double v) {
    const auto lo = __builtin_amdgcn_permlane16_swap(__double2loint(v), __double2loint(v), false, false);
    const auto hi = __builtin_amdgcn_permlane16_swap(__double2hiint(v), __double2hiint(v), false, false);
    return __hiloint2double(hi[0], lo[0]) + __hiloint2double(hi[1], lo[1]);
}
__device__ __forceinline__ double pj_xrow32(double v) {
    const auto lo = __builtin_amdgcn_permlane32_swap(__double2loint(v), __double2loint(v), false, false);
    const auto hi = __builtin_amdgcn_permlane32_swap(__double2hiint(v), __double2hiint(v), false, false);
    return __hiloint2double(hi[0], lo[0]) + __hiloint2double(hi[1], lo[1]);
}
// sum over every aligned group of SP adjacent lanes (SP a power of two), in every lane of the group; all 64 lanes must be active
template <int SP>
__device__ __forceinline__ double pj_group_sum(double v) {
    static_assert(SP >= 1 && SP <= 64 && (SP & (SP - 1)) == 0, "power of two");
    if constexpr (SP >= 2) v += pj_dpp<0xB1>(v);     // quad_perm [1,0,3,2]
    if constexpr (SP >= 4) v += pj_dpp<0x4E>(v);     // quad_perm [2,3,0,1]
    if constexpr (SP >= 8) v += pj_dpp<0x141>(v);    // row_half_mirror
    if constexpr (SP >= 16) v += pj_dpp<0x140>(v);   // row_mirror
    if constexpr (SP >= 32) v = pj_xrow16(v);
    if constexpr (SP >= 64) v = pj_xrow32(v);
    return v;
}
__device__ __forceinline__ double pj_wave_sum(double v) { return pj_group_sum<64>(v); }
__device__ __forceinline__ double pj_quad_sum(double v) { return pj_group_sum<4>(v); }
__device__ __forceinline__ double pj_wave_sum_dpp(double v) { return pj_group_sum<64>(v); }
// Workgroup barrier for LDS hand-offs that leaves global loads IN FLIGHT: __syncthreads() waits for the whole vector-memory
// queue (s_waitcnt vmcnt(0)), which puts the round trip of every load requested ahead of it -- the projection tables the
// whole-iteration kernels request early and park late -- in front of the barrier (cdna_hip_programming.md section 5:
// "raw s_barrier + lgkmcnt(0) only").  The "memory" clobber keeps the compiler from moving LDS accesses across it.
__device__ __forceinline__ void pj_lds_barrier() {
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}
__device__ __forceinline__ void pj_wave_sync() {
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
}

// All arguments of one projection launch (plain pointers; passed by value in the kernarg segment).
struct ProjArgs {
    ProjDesc pd;
    const double* OUT;
    double* GBAR;
    double* R;
    const double* F;
    const double* coef;
    long coef_stride;
    const double* wtx;
    const double* wty;
    const double* eps_ptr;
    double* loss_e;
    double* deps_e;
    long N;
    int do_adjoint;
    const double* edge_u;
    const double* edge_dphi;
    const double* edge_coef;
    double* edge_gbar;
};

// lanes per output of a contraction of length K with n_out outputs on `block` threads: the largest power of two
// (<= 64, <= K) that still gives every output its own lane group in one pass
constexpr int pj_splitk(int n_out, int K, int block) {
    int s = 1;
    while (2 * s <= 64 && 2 * s <= K && n_out * 2 * s <= block) s *= 2;
    return s >= 4 ? s : 1;   // a 2-way split measured slower than none (80x80 element: 30.5 vs 28.0 us)
}

template <int QX, int QY, int NTX, int NTY>
constexpr int project_wg_lds_doubles() {
    return QY * (QX + 1) + HPV_MAXT * NTX * (QX + 1) + HPV_MAXT * NTY * QY + QY * NTX + NTX * NTY + HPV_MAXT * NTY * QX + 64;
}

// Table staging of project_element_wg<.., PRE = true> split in two, so that a caller can put its own prologue loads between
// the global reads and the LDS stores: load (registers) ... store (into the scratch `sm`).
template <int QX, int QY, int NTX, int NTY, int PW_BLOCK>
struct ProjTableRegs {
    static constexpr int ITX = (NTX * QX + PW_BLOCK - 1) / PW_BLOCK, ITY = (NTY * QY + PW_BLOCK - 1) / PW_BLOCK;
    static constexpr int NR = NTX * NTY, ITR = (NR + PW_BLOCK - 1) / PW_BLOCK;
    static constexpr int AXLD = QX + 1;          // padded rows: the x-contraction reads 16 different rows per wave
    double ax[HPV_MAXT][ITX], by[HPV_MAXT][ITY], fr[ITR], sc4;
    int nterms_loaded = HPV_MAXT;
    // besides the tables: -F of the element (the initial value of U) and, lane t < 4 of the block, one of the four scalars
    // {coef[0][e], coef[1][e], epsilon, active test count}
    // (the run may have FEWER test functions per direction than the instantiation -- N_test is a free hyper-parameter, P2:283-286:
    //  the tables of the missing ones are zero, so their residuals are exactly 0; F, R and the means use the run's counts)
    __device__ __forceinline__ void load(const ProjArgs& pa, long e) {
        nterms_loaded = pa.pd.nterms;
        const int rnx = pa.pd.ntx, rny = pa.pd.nty;
#pragma unroll
        for (int t = 0; t < HPV_MAXT; ++t) {
            const bool on = t < pa.pd.nterms;           // (workgroup-uniform: an unused term costs no loads)
            const int dx = on ? pa.pd.t[t].dx : 0, dy = on ? pa.pd.t[t].dy : 0;
#pragma unroll
            for (int it = 0; it < ITX; ++it) {
                const int i = it * PW_BLOCK + (int)threadIdx.x;
                ax[t][it] = (on && i < NTX * QX && i / QX < rnx) ? pa.wtx[((long)dx * rnx + i / QX) * QX + i % QX] : 0.0;
            }
#pragma unroll
            for (int it = 0; it < ITY; ++it) {
                const int i = it * PW_BLOCK + (int)threadIdx.x;
                by[t][it] = (on && i < NTY * QY && i / QY < rny) ? pa.wty[((long)dy * rny + i / QY) * QY + i % QY] : 0.0;
            }
        }
#pragma unroll
        for (int it = 0; it < ITR; ++it) {
            const int idx = it * PW_BLOCK + (int)threadIdx.x;
            const int k_ = idx / NTX, r_ = idx % NTX;
            fr[it] = (pa.F && idx < NR && k_ < rny && r_ < rnx) ? -pa.F[e * (rnx * rny) + k_ * rnx + r_] : 0.0;
        }
        const int t4 = (int)threadIdx.x;
        sc4 = 0.0;
        if (t4 < HPV_MAXT) sc4 = pa.coef[(long)(t4 < pa.pd.nterms ? t4 : 0) * pa.coef_stride + e];
        else if (t4 == HPV_MAXT) sc4 = pa.eps_ptr ? pa.eps_ptr[0] : 0.0;
        else if (t4 == HPV_MAXT + 1) sc4 = pa.pd.nact ? (double)pa.pd.nact[e] : (double)NTX;
    }
    __device__ __forceinline__ void store(double* sm) const {
        double* AXl = sm + QY * (QX + 1);
        double* BYl = AXl + HPV_MAXT * NTX * AXLD;
        double* U = BYl + HPV_MAXT * NTY * QY + QY * NTX;
        double* red = U + NR + HPV_MAXT * NTY * QX;
#pragma unroll
        for (int t = 0; t < HPV_MAXT; ++t) {
            if (t >= nterms_loaded) break;      // (workgroup-uniform: the tables of an unused term are never read either)
#pragma unroll
            for (int it = 0; it < ITX; ++it) {
                const int i = it * PW_BLOCK + (int)threadIdx.x;
                if (i < NTX * QX) AXl[t * NTX * AXLD + (i / QX) * AXLD + (i % QX)] = ax[t][it];
            }
#pragma unroll
            for (int it = 0; it < ITY; ++it) {
                const int i = it * PW_BLOCK + (int)threadIdx.x;
                if (i < NTY * QY) BYl[t * NTY * QY + i] = by[t][it];
            }
        }
#pragma unroll
        for (int it = 0; it < ITR; ++it) {
            const int idx = it * PW_BLOCK + (int)threadIdx.x;
            if (idx < NR) U[idx] = fr[it];
        }
        if (threadIdx.x < HPV_MAXT + 2) red[48 + threadIdx.x] = sc4;
    }
};

// Projection (+ adjoint) of ONE element by a whole workgroup of PW_BLOCK threads; `sm` = its LDS scratch of
// project_wg_lds_doubles<...>() doubles.  Called by k_project_wg (one element per workgroup) and, fused, at
// the end of the forward kernel's element block (kernels_mfma.hip).
// PRE (the whole-iteration tile kernel): the caller has already staged every term's tables at their places in `sm`
// (ProjTableRegs, issued with its own prologue loads and followed by a barrier) and keeps the element's channels [C][NQ] and
// their adjoints in LDS (out_lds / gbar_lds) -- the function then has no table or channel traffic to global memory.
template <int QX, int QY, int NTX, int NTY, int PW_BLOCK, bool PRE = false>
__device__ __forceinline__ void project_element_wg(const ProjArgs& pa, const long e, double* sm, const double* out_lds = nullptr,
                                                   double* gbar_lds = nullptr) {
    const ProjDesc& pd = pa.pd;
    // out_lds / gbar_lds: the element's channels [C][NQ] and their adjoints live in LDS instead of the global rows
    // (addressed below as OUT[ch * N + base + q]: N = NQ and base = 0 there)
    const double* __restrict__ OUT = PRE ? out_lds : pa.OUT;
    double* __restrict__ GBAR = PRE ? gbar_lds : pa.GBAR;
    double* __restrict__ R = pa.R;
    const double* __restrict__ F = pa.F;
    const double* __restrict__ coef = pa.coef;
    const long coef_stride = pa.coef_stride;
    const double* __restrict__ wtx = pa.wtx;
    const double* __restrict__ wty = pa.wty;
    const double* __restrict__ eps_ptr = pa.eps_ptr;
    double* __restrict__ loss_e = pa.loss_e;
    double* __restrict__ deps_e = pa.deps_e;
    const long N = PRE ? (long)(QX * QY) : pa.N;
    const int do_adjoint = pa.do_adjoint;
    const double* __restrict__ edge_u = pa.edge_u;
    const double* __restrict__ edge_dphi = pa.edge_dphi;
    const double* __restrict__ edge_coef = pa.edge_coef;
    double* __restrict__ edge_gbar = pa.edge_gbar;
    constexpr int NWV = PW_BLOCK / 64;
    static_assert(3 * NWV <= 64, "per-wave reduction scratch");
    constexpr int NQ = QX * QY, NR = NTX * NTY, LDG = QX + 1;
    constexpr int NIT = (NQ + PW_BLOCK - 1) / PW_BLOCK;
    constexpr int AXLD = PRE ? QX + 1 : QX;      // (the pre-staged tables have padded rows, see ProjTableRegs)
    const int rnx = pd.ntx, rny = pd.nty, rnr = rnx * rny;     // the run's test functions per direction (<= NTX, NTY: see ProjTableRegs)
    double* G = sm;                              // [QY][LDG]
    double* AXl = G + QY * LDG;                  // [HPV_MAXT][NTX][AXLD]  every term's w_x phi^(dx)
    double* BYl = AXl + HPV_MAXT * NTX * (QX + 1);   // [HPV_MAXT][NTY][QY]  every term's w_y phi^(dy)
    double* T = BYl + HPV_MAXT * NTY * QY;       // [QY][NTX]
    double* U = T + QY * NTX;                    // [NR]
    double* S = U + NR;                          // [HPV_MAXT][NTY][QX]
    double* red = S + HPV_MAXT * NTY * QX;       // [64]: three arrays of one entry per wave (<= 16 waves)
    const long base = PRE ? 0 : e * NQ;
    const int tid = threadIdx.x;
    const int nterms = pd.nterms, C = pd.C;
    static_assert(!PRE || (HPV_MAXT + 2 <= 16 && 3 * NWV <= 48), "pre-staged scalars sit at red[48..]");
    const double eps = PRE ? red[48 + HPV_MAXT] : (eps_ptr ? eps_ptr[0] : 0.0);
#ifdef HPV_PJ_TIMING
    if (PRE && threadIdx.x == 0) pa.GBAR[e * 16 + 0] = (double)clock64();
#endif

    // ---- every global read of the forward half is issued up front (ONE memory round trip): the right-hand
    //      side, both tables of every term, and every term's integrand at this thread's points ----
    constexpr int ITX = (NTX * QX + PW_BLOCK - 1) / PW_BLOCK, ITY = (NTY * QY + PW_BLOCK - 1) / PW_BLOCK;
    constexpr int ITR = (NR + PW_BLOCK - 1) / PW_BLOCK;
    double gv[HPV_MAXT][NIT], tax[PRE ? 1 : HPV_MAXT][PRE ? 1 : ITX], tby[PRE ? 1 : HPV_MAXT][PRE ? 1 : ITY], fr[ITR], cf[HPV_MAXT];
    if constexpr (PRE) {      // -F is already in U, the coefficients next to the reduction scratch
#pragma unroll
        for (int it = 0; it < ITR; ++it) fr[it] = 0.0;
#pragma unroll
        for (int t = 0; t < HPV_MAXT; ++t) cf[t] = red[48 + t];
    } else {
#pragma unroll
        for (int it = 0; it < ITR; ++it) {
            const int idx = it * PW_BLOCK + tid;
            const int k_ = idx / NTX, r_ = idx % NTX;
            fr[it] = (F && idx < NR && k_ < rny && r_ < rnx) ? -F[e * rnr + k_ * rnx + r_] : 0.0;
        }
#pragma unroll
        for (int t = 0; t < HPV_MAXT; ++t) cf[t] = coef[(long)(t < nterms ? t : 0) * coef_stride + e];
    }
#pragma unroll
    for (int t = 0; t < HPV_MAXT; ++t) {
        const int dx = t < nterms ? pd.t[t].dx : 0, dy = t < nterms ? pd.t[t].dy : 0;
        if constexpr (!PRE) {
#pragma unroll
            for (int it = 0; it < ITX; ++it) {
                const int i = it * PW_BLOCK + tid;
                tax[t][it] = (i < NTX * QX && i / QX < rnx) ? wtx[((long)dx * rnx + i / QX) * QX + i % QX] : 0.0;
            }
#pragma unroll
            for (int it = 0; it < ITY; ++it) {
                const int i = it * PW_BLOCK + tid;
                tby[t][it] = (i < NTY * QY && i / QY < rny) ? wty[((long)dy * rny + i / QY) * QY + i % QY] : 0.0;
            }
        }
#pragma unroll
        for (int it = 0; it < NIT; ++it) gv[t][it] = 0.0;
        if (t < nterms) {
#pragma unroll
            for (int ch = 0; ch < HPV_MAXC; ++ch) {
                const double al = (ch < C) ? pd.t[t].a0[ch] + eps * pd.t[t].a1[ch] : 0.0;
                if (al != 0.0) {     // block-uniform: whole channels are skipped, never single loads (a branch per load
                                     // would serialise them: measured 60 us for the 80x80 element)
#pragma unroll
                    for (int it = 0; it < NIT; ++it) {
                        const int qd = it * PW_BLOCK + tid;
                        const double v = OUT[(long)ch * N + base + (qd < NQ ? qd : NQ - 1)];
                        gv[t][it] = fma(al, v, gv[t][it]);
                    }
                }
            }
        }
    }
    if constexpr (!PRE) {
#pragma unroll
        for (int it = 0; it < ITR; ++it) {
            const int idx = it * PW_BLOCK + tid;
            if (idx < NR) U[idx] = fr[it];
        }
    }
    if constexpr (!PRE) {
#pragma unroll
        for (int t = 0; t < HPV_MAXT; ++t) {
#pragma unroll
            for (int it = 0; it < ITX; ++it) {
                const int i = it * PW_BLOCK + tid;
                if (i < NTX * QX) AXl[t * NTX * QX + i] = tax[t][it];
            }
#pragma unroll
            for (int it = 0; it < ITY; ++it) {
                const int i = it * PW_BLOCK + tid;
                if (i < NTY * QY) BYl[t * NTY * QY + i] = tby[t][it];
            }
        }
    }
#pragma unroll
    for (int t = 0; t < HPV_MAXT; ++t) {
        if (t >= nterms) break;
        const TermDesc& td = pd.t[t];
        __syncthreads();                         // tables staged / previous users of G and T are done
#ifdef HPV_PJ_TIMING
    if (PRE && threadIdx.x == 0) pa.GBAR[e * 16 + 1] = (double)clock64();
#endif
#pragma unroll
        for (int it = 0; it < NIT; ++it) {
            const int qd = it * PW_BLOCK + tid;
            if (qd < NQ) G[(qd / QX) * LDG + (qd % QX)] = gv[t][it];
        }
        __syncthreads();
#ifdef HPV_PJ_TIMING
    if (PRE && threadIdx.x == 0) pa.GBAR[e * 16 + 2] = (double)clock64();
#endif
        // split-K: SPX adjacent lanes share one output and a shuffle tree adds their partial sums -- with few
        // outputs (QY*NTX, then NR) and long contractions the phases are latency chains, not throughput
        {
            constexpr int SPX = pj_splitk(QY * NTX, QX, PW_BLOCK);
            for (int o0 = 0; o0 < QY * NTX; o0 += PW_BLOCK / SPX) {
                const int o = o0 + tid / SPX, part = tid % SPX;
                const bool ok = o < QY * NTX;
                const int j = ok ? o / NTX : 0, r = ok ? o % NTX : 0;
                double acc = 0.0;
                constexpr int KIT = (QX + SPX - 1) / SPX;        // compile-time trip count: every operand read is issued
                if constexpr (KIT <= 24) {                       // before the first fma (the phase is one LDS latency, not KIT)
                    double av[KIT], gq[KIT];
#pragma unroll
                    for (int it = 0; it < KIT; ++it) {
                        const int i = part + it * SPX, ic = i < QX ? i : 0;
                        av[it] = AXl[t * NTX * AXLD + r * AXLD + ic];
                        gq[it] = i < QX ? G[j * LDG + ic] : 0.0;
                    }
#pragma unroll
                    for (int it = 0; it < KIT; ++it) acc = fma(av[it], gq[it], acc);
                } else {
#pragma unroll 8
                    for (int i = part; i < QX; i += SPX) acc = fma(AXl[t * NTX * AXLD + r * AXLD + i], G[j * LDG + i], acc);
                }
                acc = pj_group_sum<SPX>(acc);
                if (ok && part == 0) T[o] = acc;
            }
        }
        __syncthreads();
#ifdef HPV_PJ_TIMING
    if (PRE && threadIdx.x == 0) pa.GBAR[e * 16 + 3] = (double)clock64();
#endif
        const double c = cf[t] * (td.eps_mult ? eps : 1.0);
        {
            constexpr int SPY = pj_splitk(NR, QY, PW_BLOCK);
            for (int o0 = 0; o0 < NR; o0 += PW_BLOCK / SPY) {
                const int o = o0 + tid / SPY, part = tid % SPY;
                const bool ok = o < NR;
                const int k = ok ? o / NTX : 0, r = ok ? o % NTX : 0;
                double acc = 0.0;
#pragma unroll 8
                for (int j = part; j < QY; j += SPY) acc = fma(BYl[t * NTY * QY + k * QY + j], T[j * NTX + r], acc);
                acc = pj_group_sum<SPY>(acc);
                if (ok && part == 0) U[o] = fma(c, acc, U[o]);
            }
        }
    }
    __syncthreads();
#ifdef HPV_PJ_TIMING
    if (PRE && threadIdx.x == 0) pa.GBAR[e * 16 + 4] = (double)clock64();
#endif
    if (pd.edge) {   // P1:90: + 1/J [u(x_R) phi'_k(1) - u(x_L) phi'_k(-1)]
        const double uL = edge_u[2 * e], uR = edge_u[2 * e + 1], ce = edge_coef[e];
        for (int o = tid; o < NR; o += PW_BLOCK) U[o] += ce * (uR * edge_dphi[2 * o + 1] - uL * edge_dphi[2 * o]);
        __syncthreads();
    }
    const int nax = PRE ? (int)red[48 + HPV_MAXT + 1] : (pd.nact ? pd.nact[e] : NTX);   // active test functions (p-refinement, P1:67)
    const bool counted = pd.nact != nullptr;     // (per-element counts, P1:66-67: 1-D, rnx = NTX)
    const double NRa = counted ? (double)(nax * NTY) : (double)rnr;
    double sq = 0.0;
    for (int o = tid; o < NR; o += PW_BLOCK) {
        const int k_ = o / NTX, r_ = o % NTX;
        const double u = r_ < nax ? U[o] : 0.0;
        U[o] = u;                                         // (each entry is read and written by its own thread only)
        if (k_ < rny && r_ < rnx) R[e * rnr + k_ * rnx + r_] = u;
        sq = fma(u, u, sq);
    }
    sq = pj_wave_sum(sq);
    if ((tid & 63) == 0) red[tid >> 6] = sq;
    __syncthreads();
#ifdef HPV_PJ_TIMING
    if (PRE && threadIdx.x == 0) pa.GBAR[e * 16 + 5] = (double)clock64();
#endif
    if (tid == 0) { double t = 0.0; for (int w = 0; w < NWV; ++w) t += red[w]; loss_e[e] = t / NRa; }
    if (!do_adjoint) return;

    const double sc = 2.0 / NRa;
    {
        constexpr int SPS = pj_splitk(NTY * QX, NTX, PW_BLOCK);    // lanes per output (80 outputs of 60 terms: 4 x 15)
        for (int t = 0; t < nterms; ++t) {
            for (int o0 = 0; o0 < NTY * QX; o0 += PW_BLOCK / SPS) {
                const int o = o0 + tid / SPS, part = tid % SPS;
                const bool ok = o < NTY * QX;
                const int k = ok ? o / QX : 0, i = ok ? o % QX : 0;
                double acc = 0.0;
                constexpr int KIT = (NTX + SPS - 1) / SPS;
                if constexpr (KIT <= 24) {
                    double av[KIT], uq[KIT];
#pragma unroll
                    for (int it = 0; it < KIT; ++it) {
                        const int r = part + it * SPS, rc = r < NTX ? r : 0;
                        av[it] = AXl[t * NTX * AXLD + rc * AXLD + i];
                        uq[it] = r < NTX ? U[k * NTX + rc] : 0.0;
                    }
#pragma unroll
                    for (int it = 0; it < KIT; ++it) acc = fma(av[it], uq[it], acc);
                } else {
#pragma unroll 4
                    for (int r = part; r < NTX; r += SPS) acc = fma(AXl[t * NTX * AXLD + r * AXLD + i], U[k * NTX + r], acc);
                }
                acc = pj_group_sum<SPS>(acc);
                if (ok && part == 0) S[t * NTY * QX + o] = acc * sc;
            }
        }
    }
    __syncthreads();
#ifdef HPV_PJ_TIMING
    if (PRE && threadIdx.x == 0) pa.GBAR[e * 16 + 6] = (double)clock64();
#endif
    double deps = 0.0;
    constexpr int CHK = 5;                       // points per thread whose channel re-reads travel together
#pragma unroll 1
    for (int it0 = 0; it0 < NIT; it0 += CHK) {
        double o[CHK][HPV_MAXC];
        if (pd.has_eps) {                        // d/d(eps) needs the channels again: one round trip per chunk
#pragma unroll
            for (int u = 0; u < CHK; ++u) {
                const int qd = (it0 + u) * PW_BLOCK + tid;
#pragma unroll
                for (int ch = 0; ch < HPV_MAXC; ++ch)
                    o[u][ch] = OUT[(long)(ch < C ? ch : 0) * N + base + (qd < NQ ? qd : NQ - 1)];
            }
        }
#pragma unroll
        for (int u = 0; u < CHK; ++u) {
            const int qd = (it0 + u) * PW_BLOCK + tid;
            if (it0 + u < NIT && qd < NQ) {
                const int j = qd / QX, i = qd % QX;
                double gb[HPV_MAXC];
#pragma unroll
                for (int ch = 0; ch < HPV_MAXC; ++ch) gb[ch] = 0.0;
                for (int t = 0; t < nterms; ++t) {
                    const TermDesc& td = pd.t[t];
                    const double* by = BYl + t * NTY * QY;
                    double gh = 0.0;
#pragma unroll 4
                    for (int k = 0; k < NTY; ++k) gh = fma(by[k * QY + j], S[t * NTY * QX + k * QX + i], gh);
                    gh *= cf[t];
                    const double m = td.eps_mult ? eps : 1.0;
                    double g1 = 0.0, gt = 0.0;
#pragma unroll
                    for (int ch = 0; ch < HPV_MAXC; ++ch) {
                        const double al = td.a0[ch] + eps * td.a1[ch];
                        gb[ch] = fma(al, m * gh, gb[ch]);
                        if (pd.has_eps) {
                            g1 = fma(td.a1[ch], o[u][ch], g1);
                            gt = fma(al, o[u][ch], gt);
                        }
                    }
                    deps = fma(gh, m * g1 + (td.eps_mult ? gt : 0.0), deps);
                }
#pragma unroll
                for (int ch = 0; ch < HPV_MAXC; ++ch)
                    if (ch < C) GBAR[(long)ch * N + base + qd] = gb[ch];
            }
        }
    }
    if (pd.edge) {
        double sl = 0.0, sr = 0.0;
        for (int o = tid; o < NR; o += PW_BLOCK) {
            sl = fma(U[o], edge_dphi[2 * o], sl);
            sr = fma(U[o], edge_dphi[2 * o + 1], sr);
        }
        sl = pj_wave_sum(sl);
        sr = pj_wave_sum(sr);
        __syncthreads();
        if ((tid & 63) == 0) { red[tid >> 6] = sl; red[NWV + (tid >> 6)] = sr; }
        __syncthreads();
        if (tid == 0) {
            double tl = 0.0, tr = 0.0;
            for (int w = 0; w < NWV; ++w) { tl += red[w]; tr += red[NWV + w]; }
            edge_gbar[2 * e] = -edge_coef[e] * sc * tl;
            edge_gbar[2 * e + 1] = edge_coef[e] * sc * tr;
        }
    }
    if (pd.has_eps) {
        deps = pj_wave_sum(deps);
        __syncthreads();
        if ((tid & 63) == 0) red[2 * NWV + (tid >> 6)] = deps;
        __syncthreads();
        if (tid == 0) { double t = 0.0; for (int w = 0; w < NWV; ++w) t += red[2 * NWV + w]; deps_e[e] = t; }
    }
}


// ------------------------------------------------------------------------------------------------
// The same projection for a 1-D element (QY = NTY = 1: P1:82-100) inside the whole-iteration tile kernel (tables pre-staged by
// ProjTableRegs, channels and adjoints in LDS, no edge term -- the caller's host side excludes var_form 3).  The general
// function above spends seven workgroup barriers and two LDS hand-offs per term on what is, in 1-D, two matrix-vector
// products with one NTX x QX table per term; here every term's integrand goes to LDS at once and the element takes two
// barriers (+ the caller's) (LDS-only ones, pj_lds_barrier: a __syncthreads would also wait for the acknowledgement of the R / loss stores):
// integrands | residual (one pass over all terms, -F, active-count mask, R, squared sum) | adjoint contraction and the
// point's adjoint channels by the same lane quad.  Every sum keeps the order of the general function (per-lane chain, then the xor tree, terms in
// sequence), so the residuals, S and the adjoint channels are bit-identical to it; only the element's squared sum is added up
// in another lane order.
// ------------------------------------------------------------------------------------------------
template <int QX, int NTX, int PW_BLOCK>
__device__ __forceinline__ void project_element_1d(const ProjArgs& pa, const long e, double* sm, const double* __restrict__ OUT,
                                                   double* __restrict__ GBAR) {
    const ProjDesc& pd = pa.pd;
    constexpr int NWV = PW_BLOCK / 64, AXLD = QX + 1, SP = 4;     // (SP = 4: the lanes of one output are a DPP quad)
    static_assert(NTX * SP <= PW_BLOCK && QX * SP <= PW_BLOCK, "one pass per contraction");
    static_assert(3 * NWV <= 48, "pre-staged scalars sit at red[48..]");
    // the scratch as ProjTableRegs<QX, 1, NTX, 1, PW_BLOCK>::store laid it out
    double* AXl = sm + (QX + 1);                 // [HPV_MAXT][NTX][AXLD]
    double* BYl = AXl + HPV_MAXT * NTX * AXLD;   // [HPV_MAXT]  (w_y phi^(dy) of the single y point)
    double* U = BYl + HPV_MAXT + NTX;            // [NTX]  -F on entry
    double* S = U + NTX;                         // [HPV_MAXT][QX]  first every term's integrand, then its adjoint contraction
    double* red = S + HPV_MAXT * QX;             // [64]
    const int tid = threadIdx.x;
    const int nterms = pd.nterms, C = pd.C;
    const double eps = red[48 + HPV_MAXT];
    const int nax = (int)red[48 + HPV_MAXT + 1];
    double cf[HPV_MAXT], byv[HPV_MAXT];
#pragma unroll
    for (int t = 0; t < HPV_MAXT; ++t) { cf[t] = red[48 + t]; byv[t] = BYl[t]; }
#ifdef HPV_PJ_TIMING
    if (threadIdx.x == 0) pa.GBAR[e * 16 + 0] = (double)clock64();
#endif
    // every kernarg scalar the phases below need, fetched in one batch (a scalar load in the middle of a latency chain costs its
    // whole round trip): the destination pointers, and per term the channel coefficients alpha = a0 + eps a1 (kept in registers
    // for the adjoint channels), the epsilon multiplier
    double* const Rg = pa.R;
    double* const loss_g = pa.loss_e;
    double* const deps_g = pa.deps_e;
    const int do_adjoint = pa.do_adjoint, has_eps = pd.has_eps;
    asm volatile("" ::"s"(Rg), "s"(loss_g), "s"(deps_g), "s"(do_adjoint), "s"(has_eps));
    double al[HPV_MAXT][HPV_MAXC], a1v[HPV_MAXT][HPV_MAXC], mt[HPV_MAXT];
    bool em[HPV_MAXT];
#pragma unroll
    for (int t = 0; t < HPV_MAXT; ++t) {
        const bool on = t < nterms;
        em[t] = on && pd.t[t].eps_mult;
        mt[t] = em[t] ? eps : 1.0;
#pragma unroll
        for (int ch = 0; ch < HPV_MAXC; ++ch) {
            a1v[t][ch] = (on && ch < C) ? pd.t[t].a1[ch] : 0.0;
            al[t][ch] = (on && ch < C) ? pd.t[t].a0[ch] + eps * a1v[t][ch] : 0.0;
        }
    }
    // ---- integrands of every term at this thread's point ----
    double ov[HPV_MAXC];
    {
        const int q = tid < QX ? tid : QX - 1;
#pragma unroll
        for (int ch = 0; ch < HPV_MAXC; ++ch) ov[ch] = OUT[(ch < C ? ch : 0) * QX + q];
#pragma unroll
        for (int t = 0; t < HPV_MAXT; ++t) {
            if (t >= nterms) break;
            double gsum = 0.0;
#pragma unroll
            for (int ch = 0; ch < HPV_MAXC; ++ch) gsum = fma(al[t][ch], ov[ch], gsum);
            if (tid < QX) S[t * QX + tid] = gsum;
        }
    }
    pj_lds_barrier();
#ifdef HPV_PJ_TIMING
    if (threadIdx.x == 0) pa.GBAR[e * 16 + 1] = (double)clock64();
#endif
    // ---- residual: SP adjacent lanes share one test function ----
    {
        // lanes of a 32-lane LDS group take rows 4 apart: with AXLD = QX + 1 = 17 (mod 32) their ds_read_b64 then cover all
        // 32 eight-byte bank pairs (rows r0 .. r0 + 7 would collide two by two)
        const int part = tid % SP, grp = tid >> 5;
        const int r = (grp & 3) + 4 * ((tid >> 2) & 7) + 32 * (grp >> 2);
        const bool ok = r < NTX && tid < 256;
        const int rc = ok ? r : 0;
        constexpr int KIT = (QX + SP - 1) / SP;
        static_assert(NTX <= 64 && (QX + 1) % 32 == 17, "lane mapping of the 1-D residual / adjoint contractions");
        double u = U[rc];
#pragma unroll
        for (int t = 0; t < HPV_MAXT; ++t) {
            if (t >= nterms) break;
            double av[KIT], gq[KIT];
#pragma unroll
            for (int it = 0; it < KIT; ++it) {
                const int i = part + it * SP, ic = i < QX ? i : 0;
                av[it] = AXl[t * NTX * AXLD + rc * AXLD + ic];
                gq[it] = i < QX ? S[t * QX + ic] : 0.0;
            }
            double acc = 0.0;
#pragma unroll
            for (int it = 0; it < KIT; ++it) acc = fma(av[it], gq[it], acc);
            acc = pj_quad_sum(acc);
            const double c = cf[t] * mt[t];
            u = fma(c, fma(byv[t], acc, 0.0), u);
        }
        u = rc < nax ? u : 0.0;
        double sq = 0.0;
        if (ok && part == 0) {
            U[r] = u;
            Rg[e * NTX + r] = u;
            sq = u * u;
        }
        sq = pj_wave_sum_dpp(sq);
        if ((tid & 63) == 0) red[tid >> 6] = sq;
    }
    pj_lds_barrier();
#ifdef HPV_PJ_TIMING
    if (threadIdx.x == 0) pa.GBAR[e * 16 + 2] = (double)clock64();
#endif
    const double NRa = (double)nax;
    if (tid == 0) { double t = 0.0; for (int w = 0; w < NWV; ++w) t += red[w]; loss_g[e] = t / NRa; }
    if (!do_adjoint) return;
    const double sc = 2.0 / NRa;
    // ---- adjoint: S_t[i] = sc sum_r AX_t[r][i] U[r] by a lane quad per point, which goes straight on to the point's adjoint
    //      channels (+ its share of d loss / d epsilon): no hand-off through LDS, no barrier between the two ----
    double deps = 0.0;
    {
        // (columns 4 apart per LDS group: rows part + 4 it are 0, 17, 2, 19 (mod 32) bank pairs apart, see above)
        const int part = tid % SP, grp = tid >> 5;
        const int i = (grp & 3) + 4 * ((tid >> 2) & 7) + 32 * (grp >> 2);
        const bool ok = i < QX;
        const int ic = ok ? i : 0;
        constexpr int KIT = (NTX + SP - 1) / SP;
        static_assert(QX <= 32 * (PW_BLOCK / 128), "every column has its lane quad");
        double uq[KIT];
#pragma unroll
        for (int it = 0; it < KIT; ++it) {
            const int r = part + it * SP;
            uq[it] = r < NTX ? U[r] : 0.0;
        }
        double oi[HPV_MAXC];                 // the point's channels again (d/d eps only)
#pragma unroll
        for (int ch = 0; ch < HPV_MAXC; ++ch) oi[ch] = has_eps ? OUT[(ch < C ? ch : 0) * QX + ic] : 0.0;
        double gb[HPV_MAXC];
#pragma unroll
        for (int ch = 0; ch < HPV_MAXC; ++ch) gb[ch] = 0.0;
#pragma unroll
        for (int t = 0; t < HPV_MAXT; ++t) {
            if (t >= nterms) break;
            double av[KIT];
#pragma unroll
            for (int it = 0; it < KIT; ++it) {
                const int r = part + it * SP, rc = r < NTX ? r : 0;
                av[it] = AXl[t * NTX * AXLD + rc * AXLD + ic];
            }
            double acc = 0.0;
#pragma unroll
            for (int it = 0; it < KIT; ++it) acc = fma(av[it], uq[it], acc);
            acc = pj_quad_sum(acc);
            double gh = fma(byv[t], acc * sc, 0.0);
            gh *= cf[t];
            const double m = mt[t];
            double g1 = 0.0, gt = 0.0;
#pragma unroll
            for (int ch = 0; ch < HPV_MAXC; ++ch) {
                gb[ch] = fma(al[t][ch], m * gh, gb[ch]);
                if (has_eps) {
                    g1 = fma(a1v[t][ch], oi[ch], g1);
                    gt = fma(al[t][ch], oi[ch], gt);
                }
            }
            if (ok && part == 0) deps = fma(gh, m * g1 + (em[t] ? gt : 0.0), deps);
        }
        if (ok && part == 0) {
#pragma unroll
            for (int ch = 0; ch < HPV_MAXC; ++ch)
                if (ch < C) GBAR[ch * QX + i] = gb[ch];
        }
    }
#ifdef HPV_PJ_TIMING
    if (threadIdx.x == 0) pa.GBAR[e * 16 + 3] = (double)clock64();
    if (threadIdx.x == 0) pa.GBAR[e * 16 + 4] = (double)clock64();
#endif
    if (has_eps) {
        deps = pj_wave_sum_dpp(deps);
        pj_lds_barrier();
        if ((tid & 63) == 0) red[2 * NWV + (tid >> 6)] = deps;
        pj_lds_barrier();
        if (tid == 0) { double t = 0.0; for (int w = 0; w < NWV; ++w) t += red[2 * NWV + w]; deps_g[e] = t; }
    }
}


// ------------------------------------------------------------------------------------------------
// Row-split projection for FEW, TALL elements (AdvDiff with the 80x80 rule: 8 elements of 6 400 points would
// otherwise keep 8 CUs busy for 25 us).  PJ_SPLIT workgroups per element, workgroup (e, c) owns the QY/PJ_SPLIT rows
// j0 .. j0+RB-1 of the element:
//   phase A  k_project_rows_fwd : its rows' integrands, x-contraction, PARTIAL y-contraction
//                                 Upart[e][c][k][r] = sum_t m_t c_t sum_{j in chunk} BY_t[k][j] sum_i AX_t[r][i] G_t[j][i]
//   phase B  k_project_rows_adj : U = sum_c Upart - F (every workgroup, 25..100 values), R / loss_e by chunk 0, then the
//                                 adjoint of its own rows: GBAR and the chunk's share of d loss / d epsilon.
// loss_e / deps_e are written per (e, c) (zeros where a chunk has nothing to add); the finalize kernel sums
// n_elem * PJ_SPLIT entries in a fixed order.
// ------------------------------------------------------------------------------------------------
#define PJ_SPLIT 8
#define PJ_RBLOCK 512

template <int QX, int QY, int NTX, int NTY>
constexpr int project_rows_lds_doubles() {
    constexpr int RB = QY / PJ_SPLIT;
    return RB * (QX + 1) + HPV_MAXT * NTX * QX + HPV_MAXT * NTY * RB + RB * NTX + NTX * NTY + HPV_MAXT * NTY * QX + 64;
}

template <int QX, int QY, int NTX, int NTY>
__global__ void __launch_bounds__(PJ_RBLOCK) k_project_rows_fwd(ProjArgs pa, double* __restrict__ Upart) {
    static_assert(QY % PJ_SPLIT == 0, "rows per chunk");
    constexpr int RB = QY / PJ_SPLIT, NR = NTX * NTY, LDG = QX + 1, NP = RB * QX;
    constexpr int NIT = (NP + PJ_RBLOCK - 1) / PJ_RBLOCK;
    extern __shared__ __attribute__((aligned(16))) double sm[];
    double* G = sm;                              // [RB][LDG]
    double* AXl = G + RB * LDG;                  // [HPV_MAXT][NTX][QX]
    double* BYl = AXl + HPV_MAXT * NTX * QX;     // [HPV_MAXT][NTY][RB]   columns j0..j0+RB-1 of w_y phi^(dy)
    double* T = BYl + HPV_MAXT * NTY * RB;       // [RB][NTX]
    double* U = T + RB * NTX;                    // [NR]
    const ProjDesc& pd = pa.pd;
    const long e = blockIdx.x / PJ_SPLIT;
    const int c = blockIdx.x % PJ_SPLIT, j0 = c * RB, tid = threadIdx.x;
    const long base = e * (long)(QX * QY) + (long)j0 * QX;
    const int nterms = pd.nterms, C = pd.C;
    const double eps = pa.eps_ptr ? pa.eps_ptr[0] : 0.0;
    // all global reads up front: tables and every term's integrand at this thread's points
    double gv[HPV_MAXT][NIT];
#pragma unroll
    for (int t = 0; t < HPV_MAXT; ++t) {
        const int dx = t < nterms ? pd.t[t].dx : 0, dy = t < nterms ? pd.t[t].dy : 0;
        for (int i = tid; i < NTX * QX; i += PJ_RBLOCK) AXl[t * NTX * QX + i] = pa.wtx[(long)dx * NTX * QX + i];
        for (int i = tid; i < NTY * RB; i += PJ_RBLOCK) BYl[t * NTY * RB + i] = pa.wty[(long)dy * NTY * QY + (i / RB) * QY + j0 + i % RB];
#pragma unroll
        for (int it = 0; it < NIT; ++it) gv[t][it] = 0.0;
        if (t < nterms) {
#pragma unroll
            for (int ch = 0; ch < HPV_MAXC; ++ch) {
                const double al = (ch < C) ? pd.t[t].a0[ch] + eps * pd.t[t].a1[ch] : 0.0;
                if (al != 0.0) {     // block-uniform: whole channels are skipped
#pragma unroll
                    for (int it = 0; it < NIT; ++it) {
                        const int qd = it * PJ_RBLOCK + tid;
                        gv[t][it] = fma(al, pa.OUT[(long)ch * pa.N + base + (qd < NP ? qd : NP - 1)], gv[t][it]);
                    }
                }
            }
        }
    }
    for (int o = tid; o < NR; o += PJ_RBLOCK) U[o] = 0.0;
#pragma unroll
    for (int t = 0; t < HPV_MAXT; ++t) {
        if (t >= nterms) break;
        __syncthreads();
#pragma unroll
        for (int it = 0; it < NIT; ++it) {
            const int qd = it * PJ_RBLOCK + tid;
            if (qd < NP) G[(qd / QX) * LDG + (qd % QX)] = gv[t][it];
        }
        __syncthreads();
        {
            constexpr int SPX = pj_splitk(RB * NTX, QX, PJ_RBLOCK);
            for (int o0 = 0; o0 < RB * NTX; o0 += PJ_RBLOCK / SPX) {
                const int o = o0 + tid / SPX, part = tid % SPX;
                const bool ok = o < RB * NTX;
                const int j = ok ? o / NTX : 0, r = ok ? o % NTX : 0;
                double acc = 0.0;
#pragma unroll 8
                for (int i = part; i < QX; i += SPX) acc = fma(AXl[t * NTX * QX + r * QX + i], G[j * LDG + i], acc);
                acc = pj_group_sum<SPX>(acc);
                if (ok && part == 0) T[o] = acc;
            }
        }
        __syncthreads();
        const double cf = pa.coef[(long)t * pa.coef_stride + e] * (pd.t[t].eps_mult ? eps : 1.0);
        for (int o = tid; o < NR; o += PJ_RBLOCK) {
            const int k = o / NTX, r = o % NTX;
            double acc = 0.0;
#pragma unroll
            for (int j = 0; j < RB; ++j) acc = fma(BYl[t * NTY * RB + k * RB + j], T[j * NTX + r], acc);
            U[o] = fma(cf, acc, U[o]);
        }
    }
    __syncthreads();
    for (int o = tid; o < NR; o += PJ_RBLOCK) Upart[(long)blockIdx.x * NR + o] = U[o];
}

template <int QX, int QY, int NTX, int NTY>
__global__ void __launch_bounds__(PJ_RBLOCK) k_project_rows_adj(ProjArgs pa, const double* __restrict__ Upart) {
    constexpr int RB = QY / PJ_SPLIT, NR = NTX * NTY, NP = RB * QX, NWV = PJ_RBLOCK / 64;
    constexpr int NIT = (NP + PJ_RBLOCK - 1) / PJ_RBLOCK;
    extern __shared__ __attribute__((aligned(16))) double sm[];
    double* AXl = sm;                            // [HPV_MAXT][NTX][QX]
    double* BYl = AXl + HPV_MAXT * NTX * QX;     // [HPV_MAXT][NTY][RB]
    double* U = BYl + HPV_MAXT * NTY * RB;       // [NR]
    double* S = U + NR;                          // [HPV_MAXT][NTY][QX]
    double* red = S + HPV_MAXT * NTY * QX;       // [64]
    const ProjDesc& pd = pa.pd;
    const long e = blockIdx.x / PJ_SPLIT;
    const int c = blockIdx.x % PJ_SPLIT, j0 = c * RB, tid = threadIdx.x;
    const long base = e * (long)(QX * QY) + (long)j0 * QX;
    const int nterms = pd.nterms, C = pd.C;
    const double eps = pa.eps_ptr ? pa.eps_ptr[0] : 0.0;
    for (int t = 0; t < nterms; ++t) {
        for (int i = tid; i < NTX * QX; i += PJ_RBLOCK) AXl[t * NTX * QX + i] = pa.wtx[(long)pd.t[t].dx * NTX * QX + i];
        for (int i = tid; i < NTY * RB; i += PJ_RBLOCK)
            BYl[t * NTY * RB + i] = pa.wty[(long)pd.t[t].dy * NTY * QY + (i / RB) * QY + j0 + i % RB];
    }
    double sq = 0.0;
    for (int o = tid; o < NR; o += PJ_RBLOCK) {
        double u = pa.F ? -pa.F[e * NR + o] : 0.0;
#pragma unroll
        for (int cc = 0; cc < PJ_SPLIT; ++cc) u += Upart[(e * PJ_SPLIT + cc) * (long)NR + o];   // fixed order: every chunk agrees
        U[o] = u;
        if (c == 0) { pa.R[e * NR + o] = u; sq = fma(u, u, sq); }
    }
    sq = pj_wave_sum(sq);
    if ((tid & 63) == 0) red[tid >> 6] = sq;
    __syncthreads();
    if (tid == 0) { double t = 0.0; for (int w = 0; w < NWV; ++w) t += red[w]; pa.loss_e[blockIdx.x] = c == 0 ? t / (double)NR : 0.0; }
    if (!pa.do_adjoint) return;
    const double sc = 2.0 / (double)NR;
    for (int t = 0; t < nterms; ++t)
        for (int o = tid; o < NTY * QX; o += PJ_RBLOCK) {
            const int k = o / QX, i = o % QX;
            double acc = 0.0;
#pragma unroll
            for (int r = 0; r < NTX; ++r) acc = fma(AXl[t * NTX * QX + r * QX + i], U[k * NTX + r], acc);
            S[t * NTY * QX + o] = acc * sc;
        }
    __syncthreads();
    double deps = 0.0;
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
        const int qd = it * PJ_RBLOCK + tid;
        double o[HPV_MAXC];
        if (pd.has_eps) {
#pragma unroll
            for (int ch = 0; ch < HPV_MAXC; ++ch) o[ch] = pa.OUT[(long)(ch < C ? ch : 0) * pa.N + base + (qd < NP ? qd : NP - 1)];
        }
        if (qd < NP) {
            const int j = qd / QX, i = qd % QX;
            double gb[HPV_MAXC];
#pragma unroll
            for (int ch = 0; ch < HPV_MAXC; ++ch) gb[ch] = 0.0;
            for (int t = 0; t < nterms; ++t) {
                const TermDesc& td = pd.t[t];
                double gh = 0.0;
#pragma unroll
                for (int k = 0; k < NTY; ++k) gh = fma(BYl[t * NTY * RB + k * RB + j], S[t * NTY * QX + k * QX + i], gh);
                gh *= pa.coef[(long)t * pa.coef_stride + e];
                const double m = td.eps_mult ? eps : 1.0;
                double g1 = 0.0, gt = 0.0;
#pragma unroll
                for (int ch = 0; ch < HPV_MAXC; ++ch) {
                    const double al = td.a0[ch] + eps * td.a1[ch];
                    gb[ch] = fma(al, m * gh, gb[ch]);
                    if (pd.has_eps) {
                        g1 = fma(td.a1[ch], o[ch], g1);
                        gt = fma(al, o[ch], gt);
                    }
                }
                deps = fma(gh, m * g1 + (td.eps_mult ? gt : 0.0), deps);
            }
#pragma unroll
            for (int ch = 0; ch < HPV_MAXC; ++ch)
                if (ch < C) pa.GBAR[(long)ch * pa.N + base + qd] = gb[ch];
        }
    }
    if (pd.has_eps) {
        deps = pj_wave_sum(deps);
        __syncthreads();
        if ((tid & 63) == 0) red[16 + (tid >> 6)] = deps;
        __syncthreads();
        if (tid == 0) { double t = 0.0; for (int w = 0; w < NWV; ++w) t += red[16 + w]; pa.deps_e[blockIdx.x] = t; }
    }
}
