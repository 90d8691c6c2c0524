// Tensor-product projection kernel for the hot element shapes: one wavefront per element,
// test-function tables staged once per workgroup in LDS, sum-factorised contractions
// (x then y), wave-level synchronisation only, shuffle reductions for the element loss.
//
//   forward :  T_t[j][r] = sum_i AX_t[r][i] G_t[j][i]            AX = w_x * phi^(dx)      (P2:94-105)
//              U[k][r]  += m_t c_t sum_j BY_t[k][j] T_t[j][r]    BY = w_y * phi^(dy)
//              R = U - F,  loss_e = mean(R^2)                                             (P2:118-119)
//   adjoint :  S_t[k][i] = sum_r AX_t[r][i] (2/NR) R[k][r]
//              Ghat_t[j][i] = c_t sum_k BY_t[k][j] S_t[k][i],   GBAR[ch] = sum_t alpha_t[ch] m_t Ghat_t
//
// HBM traffic per element = read the integrated channels + F, write the adjoint channels + R (channels no
// term uses are neither read nor written; their GBAR rows are zeroed once by the host).  This is the kernel
// judged against the HBM roofline on the scaled synthetic batch.
#include <algorithm>
#include <cstdlib>

#include "hpv_internal.h"
#include "hpv_project_wg.h"


// Work decomposition: "a lane owns a line".  LPE = max(QX,QY) lanes serve one element, EPW = 64/LPE elements
// share a wavefront (3 for 20x20 points, 6 for 10x10).  In every contraction the lane keeps its line of
// element data in registers and the test-function table entry is WAVE-UNIFORM (same index for all lanes
// at the same instruction): tables are staged once per workgroup in LDS and read with broadcast reads.
// The y-contraction comes first with the lane owning COLUMN i of the integrand (all j): those loads are
// coalesced straight from HBM into registers (q = j*QX + i, i fastest), so no LDS staging of the
// integrand is needed; one small LDS transpose (NTY x QX per element) separates the two contractions,
// in the forward and again in the adjoint.  Residual rows stay in registers between forward and adjoint.
//   forward :  T[k][i]  = sum_j BY[k][j] G[j][i]      (lane = i)     ->LDS->
//              U[k][r] += m c sum_i AX[r][i] T[k][i]   (lane = k)
//   adjoint :  V[k][i]  = sum_r AX[r][i] Rs[k][r]      (lane = k)     ->LDS->
//              Gh[j][i] = c sum_k BY[k][j] V[k][i]     (lane = i)     -> coalesced stores
// The one-hot streaming instantiation exists in two plans (A/B at run time: HPV_PJ_PIPE=1 selects the second):
//   default: 8 waves per workgroup, 4 waves per SIMD (126 VGPRs), a term's channel column loaded when the term starts;
//   PIPE:    4 waves per workgroup, 3 workgroups per CU (168 VGPRs), the column of the NEXT term / next element group requested
//            while this one's contractions run.  Measured SLOWER on the 2^18-element batch (round 4: 2.99-3.05 TB/s against
//            3.48-3.50): at 168 registers the kernel spills 8 doubles, scratch reloads count in vmcnt like every vector load,
//            so each reload waits for the whole prefetch (s_waitcnt vmcnt(0) in the middle of the contractions) -- the overlap
//            the plan exists for does not happen, and three waves per SIMD hide less than four.  Kept for A/B runs.
// HPV_PJ_NT (compile time, scripts/build_variant.sh <name> -DHPV_PJ_NT=1): the streamed channel columns are read, and R
// written, with non-temporal hints (each byte is touched once).  Measured (round 4): no gain (46-47 % vs 47.5-48.7 % of 8 TB/s
// without) and +5 % written bytes by PMC (8-byte nt stores with an 80-byte lane stride are not merged into whole lines): off.
// HPV_PJ_SGPR (compile time, default 1): k_project_tp reads its test-function tables from
// global memory at wave-uniform addresses -- scalar loads, an SGPR operand per FMA -- instead of LDS broadcast reads (800 ds_read
// per element group with an s_waitcnt in front of the FMAs that use them).
#ifndef HPV_PJ_SGPR
#define HPV_PJ_SGPR 1
#endif
#ifndef HPV_PJ_NT
#define HPV_PJ_NT 0
#endif
__device__ __forceinline__ double pj_stream_load(const double* p) {
#if HPV_PJ_NT
    return __builtin_nontemporal_load(p);
#else
    return *p;
#endif
}
__device__ __forceinline__ void pj_stream_store(double* p, double v) {
#if HPV_PJ_NT
    __builtin_nontemporal_store(v, p);
#else
    *p = v;
#endif
}

struct ActiveCh {
    int n;                      // number of channels some term integrates
    int id[HPV_MAXC];           // their channel indices
};

// OH ("one-hot"): term t integrates exactly the active channel t and nothing else (Poisson-2D var_form 1: u_x, u_y).
// The integrand column then IS the prefetched channel column (alpha folds into the term coefficient) and the
// adjoint column of term t is stored straight to channel t -- no gcol / gacc copies.  With 8 waves per workgroup the
// channel column of a term is loaded when the term starts (124 VGPRs, 4 waves per SIMD, 2 x 63.6 KB LDS per CU):
// measured 3.6 TB/s against 3.15 TB/s for the 202-VGPR / 2-waves-per-SIMD general variant on the 2^18-element batch
// (5 waves per SIMD spills: 2.5 TB/s).
template <int QX, int QY, int NTX, int NTY, int NA, bool EPS, int PJ_WAVES, bool OH = false, bool PIPE = false>
__global__ void __launch_bounds__(PJ_WAVES * 64, (OH && PJ_WAVES == 8) ? 4 : (PIPE ? 3 : 1)) k_project_tp(ProjDesc pd, ActiveCh ac, const double* __restrict__ OUT,
                                                        double* __restrict__ GBAR, double* __restrict__ R,
                                                        const double* __restrict__ F, const double* __restrict__ coef,
                                                        long coef_stride, const double* __restrict__ wtx,
                                                        const double* __restrict__ wty, const double* __restrict__ eps_ptr,
                                                        double* __restrict__ loss_e, double* __restrict__ deps_e, long N,
                                                        long n_elem, int do_adjoint) {
    constexpr int NQ = QX * QY, NR = NTX * NTY;
    constexpr int LPE = QX > QY ? QX : QY;
    static_assert(NTX <= LPE && NTY <= LPE && LPE <= 64, "element shape");
    constexpr int EPW = 64 / LPE;
    constexpr int LDT = QX + 1;
    constexpr int TB_D = EPW * NTY * LDT;
    constexpr int WAVE_DOUBLES = TB_D + 64;
    constexpr int PJ_BLOCK = PJ_WAVES * 64;
    extern __shared__ __attribute__((aligned(16))) double sm[];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    // tables in both orientations, so that every contraction walks its table contiguously in the OUTPUT
    // index (independent accumulators, wide broadcast reads, many LDS reads in flight).  The transposed copies
    // are derived here; ALL global loads of the staging are issued before the first LDS store (one L2 round
    // trip instead of one per loop iteration -- the staging was most of the kernel's latency at 256 elements).
    constexpr bool SGT = !PIPE && (HPV_PJ_SGPR != 0);   // tables as SGPR operands (scalar loads at uniform addresses), not LDS reads
    double* AXs = sm;                      // [3][NTX][QX]  w_x phi^(d)[r][i]
    double* BYs = AXs + 3 * NTX * QX;      // [3][NTY][QY]  w_y phi^(d)[k][j]
    double* AXT = BYs + 3 * NTY * QY;      // [3][QX][NTX]
    double* BYT = AXT + 3 * NTX * QX;      // [3][QY][NTY]
    if constexpr (!SGT) {      // (with SGPR tables nothing is staged: the kernel starts with its first element group's loads)
        constexpr int NAX = 3 * NTX * QX, NBY = 3 * NTY * QY;
        constexpr int ITA = (NAX + PJ_BLOCK - 1) / PJ_BLOCK, ITB = (NBY + PJ_BLOCK - 1) / PJ_BLOCK;
        double va[ITA], vb[ITB];
#pragma unroll
        for (int it = 0; it < ITA; ++it) { const int i = it * PJ_BLOCK + threadIdx.x; va[it] = i < NAX ? wtx[i] : 0.0; }
#pragma unroll
        for (int it = 0; it < ITB; ++it) { const int i = it * PJ_BLOCK + threadIdx.x; vb[it] = i < NBY ? wty[i] : 0.0; }
#pragma unroll
        for (int it = 0; it < ITA; ++it) {
            const int i = it * PJ_BLOCK + threadIdx.x;
            if (i < NAX) {
                AXs[i] = va[it];
                const int d = i / (NTX * QX), r = (i / QX) % NTX, c = i % QX;
                AXT[d * (NTX * QX) + c * NTX + r] = va[it];
            }
        }
#pragma unroll
        for (int it = 0; it < ITB; ++it) {
            const int i = it * PJ_BLOCK + threadIdx.x;
            if (i < NBY) {
                BYs[i] = vb[it];
                const int d = i / (NTY * QY), k = (i / QY) % NTY, c = i % QY;
                BYT[d * (NTY * QY) + c * NTY + k] = vb[it];
            }
        }
    }
    if constexpr (!SGT) __syncthreads();
    double* Tb = BYT + 3 * NTY * QY + wv * WAVE_DOUBLES;   // [EPW][NTY][LDT]  transpose tile (T, then V)
    double* Rd = Tb + TB_D;                                // [64] slot-wise reductions
    const int slot = lane / LPE, li = lane % LPE;
    const bool lane_ok = slot < EPW;

    const int nterms = pd.nterms;
    const double eps = eps_ptr ? eps_ptr[0] : 0.0;
    const double sc = 2.0 / (double)NR;

    const long ngroups = (n_elem + EPW - 1) / EPW;
    constexpr bool LATE = OH && PJ_WAVES == 8;   // 4 waves/SIMD (124 VGPRs): the other waves hide the per-term round trip

    static_assert(!PIPE || OH, "the pipelined plan is written for the one-hot term / channel structure");
    const long gstride = (long)gridDim.x * PJ_WAVES;
    // PIPE: `nxt` always holds the column the NEXT (term, group) step consumes; its loads were issued one step earlier
    double nxt[PIPE ? QY : 1];
    auto request = [&](long grp_, int t_) {
        const long e_ = grp_ * EPW + slot;
        const bool c_ = lane_ok && e_ < n_elem && li < QX && grp_ < ngroups;
        const double* __restrict__ p_ = OUT + (c_ ? e_ : 0) * NQ + li + (long)ac.id[t_] * N;
#pragma unroll
        for (int j = 0; j < QY; ++j) nxt[PIPE ? j : 0] = c_ ? pj_stream_load(p_ + j * QX) : 0.0;
    };
    if constexpr (PIPE) request((long)blockIdx.x * PJ_WAVES + wv, 0);
    for (long grp = (long)blockIdx.x * PJ_WAVES + wv; grp < ngroups; grp += gstride) {
        const long e = grp * EPW + slot;
        const bool ev = lane_ok && e < n_elem;
        const bool col = ev && li < QX;     // this lane owns quadrature column i = li
        const bool row = ev && li < NTY;    // this lane owns residual row k = li
        const double* __restrict__ Oe = OUT + e * NQ + li;
        // all HBM reads of the group are issued up front (one memory round trip): the right-hand-side row
        // and the quadrature column of every integrated channel
        double u[NTX];
#pragma unroll
        for (int r = 0; r < NTX; ++r) u[r] = (row && F) ? -F[e * NR + li * NTX + r] : 0.0;
        // the term coefficients are requested HERE, ahead of any prefetch of the next column: vector loads return in order, and a
        // wait for a coefficient issued behind the prefetch (s_waitcnt vmcnt(0)) would wait for the prefetch as well
        double cf[OH ? NA : HPV_MAXT];
#pragma unroll
        for (int t = 0; t < (OH ? NA : HPV_MAXT); ++t) cf[t] = (ev && (OH || t < nterms)) ? coef[(long)t * coef_stride + e] : 0.0;
        double o[NA][QY];
        if constexpr (!LATE) {
#pragma unroll
            for (int a = 0; a < NA; ++a)
#pragma unroll
                for (int j = 0; j < QY; ++j) o[a][j] = col ? Oe[(long)ac.id[a] * N + j * QX] : 0.0;
        }
        double gacc[OH ? 1 : NA][OH ? 1 : QY];
        if constexpr (!OH) {
#pragma unroll
            for (int a = 0; a < NA; ++a)
#pragma unroll
                for (int j = 0; j < QY; ++j) gacc[a][j] = 0.0;
        }

#pragma unroll
        for (int t = 0; t < (OH ? NA : HPV_MAXT); ++t) {
            if (!OH && t >= nterms) break;
            const TermDesc& td = pd.t[t];
            // (a) integrand column of this term from the prefetched channels
            double gcol[QY];
            double alpha_t = 1.0;
            if constexpr (OH) {
                alpha_t = td.a0[ac.id[t]] + eps * td.a1[ac.id[t]];
                if constexpr (PIPE) {
#pragma unroll
                    for (int j = 0; j < QY; ++j) gcol[j] = nxt[PIPE ? j : 0];
                    if (t + 1 < NA) request(grp, t + 1); else request(grp + gstride, 0);     // in flight during this term's contractions
                } else if constexpr (LATE) {
#pragma unroll
                    for (int j = 0; j < QY; ++j) gcol[j] = col ? Oe[(long)ac.id[t] * N + j * QX] : 0.0;
                } else {
#pragma unroll
                    for (int j = 0; j < QY; ++j) gcol[j] = o[t][j];
                }
            } else {
#pragma unroll
                for (int j = 0; j < QY; ++j) gcol[j] = 0.0;
#pragma unroll
                for (int a = 0; a < NA; ++a) {
                    const double al = td.a0[ac.id[a]] + eps * td.a1[ac.id[a]];
#pragma unroll
                    for (int j = 0; j < QY; ++j) gcol[j] = fma(al, o[a][j], gcol[j]);
                }
            }
            pj_wave_sync();   // previous readers of Tb are done
            // (b) y-contraction, lane = column i
            if (col) {
                const double* byt = BYT + td.dy * (NTY * QY);
                const double* __restrict__ byg = wty + (long)td.dy * (NTY * QY);      // [k][j], wave-uniform
                double acc[NTY];
#pragma unroll
                for (int k = 0; k < NTY; ++k) acc[k] = 0.0;
                if constexpr (SGT) {
#pragma unroll
                    for (int k = 0; k < NTY; ++k)
#pragma unroll
                        for (int j = 0; j < QY; ++j) acc[k] = fma(byg[k * QY + j], gcol[j], acc[k]);
                } else {
#pragma unroll
                for (int j = 0; j < QY; ++j)
#pragma unroll
                    for (int k = 0; k < NTY; ++k) acc[k] = fma(byt[j * NTY + k], gcol[j], acc[k]);
                }
#pragma unroll
                for (int k = 0; k < NTY; ++k) Tb[slot * (NTY * LDT) + k * LDT + li] = acc[k];
            }
            pj_wave_sync();
            // (c) x-contraction, lane = residual row k
            if (row) {
                const double* axt = AXT + td.dx * (NTX * QX);
                const double* __restrict__ axg = wtx + (long)td.dx * (NTX * QX);      // [r][i], wave-uniform
                const double c = cf[t] * (td.eps_mult ? eps : 1.0) * alpha_t;
                double trow[QX], acc[NTX];
#pragma unroll
                for (int i = 0; i < QX; ++i) trow[i] = Tb[slot * (NTY * LDT) + li * LDT + i];
#pragma unroll
                for (int r = 0; r < NTX; ++r) acc[r] = 0.0;
                if constexpr (SGT) {
#pragma unroll
                    for (int r = 0; r < NTX; ++r)
#pragma unroll
                        for (int i = 0; i < QX; ++i) acc[r] = fma(axg[r * QX + i], trow[i], acc[r]);
                } else {
#pragma unroll
                for (int i = 0; i < QX; ++i)
#pragma unroll
                    for (int r = 0; r < NTX; ++r) acc[r] = fma(axt[i * NTX + r], trow[i], acc[r]);
                }
#pragma unroll
                for (int r = 0; r < NTX; ++r) u[r] = fma(c, acc[r], u[r]);
            }
        }
        // residual row, element loss
        double sq = 0.0;
        if (row) {
#pragma unroll
            for (int r = 0; r < NTX; ++r) {
                if constexpr (OH) pj_stream_store(R + e * NR + li * NTX + r, u[r]); else R[e * NR + li * NTX + r] = u[r];
                sq = fma(u[r], u[r], sq);
                u[r] *= sc;
            }
        }
        Rd[lane] = sq;
        pj_wave_sync();
        if (ev && li == 0) {
            double s = 0.0;
#pragma unroll
            for (int k = 0; k < NTY; ++k) s += Rd[slot * LPE + k];
            loss_e[e] = s / (double)NR;
        }
        if (!do_adjoint) continue;

        double deps = 0.0;
#pragma unroll
        for (int t = 0; t < (OH ? NA : HPV_MAXT); ++t) {
            if (!OH && t >= nterms) break;
            const TermDesc& td = pd.t[t];
            pj_wave_sync();
            // (d) V[k][i] = sum_r AX[r][i] Rs[k][r], lane = row k
            if (row) {
                const double* ax = SGT ? wtx + (long)td.dx * (NTX * QX) : AXs + td.dx * (NTX * QX);
                double acc[QX];
#pragma unroll
                for (int i = 0; i < QX; ++i) acc[i] = 0.0;
#pragma unroll
                for (int r = 0; r < NTX; ++r)
#pragma unroll
                    for (int i = 0; i < QX; ++i) acc[i] = fma(ax[r * QX + i], u[r], acc[i]);
#pragma unroll
                for (int i = 0; i < QX; ++i) Tb[slot * (NTY * LDT) + li * LDT + i] = acc[i];
            }
            pj_wave_sync();
            // (e) Gh[j][i] = c sum_k BY[k][j] V[k][i], lane = column i; scattered onto the integrated channels
            if (col) {
                const double* by = SGT ? wty + (long)td.dy * (NTY * QY) : BYs + td.dy * (NTY * QY);
                const double c = cf[t];
                const double m = td.eps_mult ? eps : 1.0;
                double vcol[NTY], gh[QY];
#pragma unroll
                for (int k = 0; k < NTY; ++k) vcol[k] = Tb[slot * (NTY * LDT) + k * LDT + li];
#pragma unroll
                for (int j = 0; j < QY; ++j) gh[j] = 0.0;
#pragma unroll
                for (int k = 0; k < NTY; ++k)
#pragma unroll
                    for (int j = 0; j < QY; ++j) gh[j] = fma(by[k * QY + j], vcol[k], gh[j]);
                if constexpr (OH) {   // term t <-> channel t: its adjoint column goes out directly, coalesced
                    const double al = (td.a0[ac.id[t]] + eps * td.a1[ac.id[t]]) * (m * c);
                    double* __restrict__ Ge = GBAR + e * NQ + li + (long)ac.id[t] * N;
#pragma unroll
                    for (int j = 0; j < QY; ++j) Ge[j * QX] = al * gh[j];
                } else {
#pragma unroll
                    for (int a = 0; a < NA; ++a) {
                        const double al = (td.a0[ac.id[a]] + eps * td.a1[ac.id[a]]) * (m * c);
#pragma unroll
                        for (int j = 0; j < QY; ++j) gacc[a][j] = fma(al, gh[j], gacc[a][j]);
                    }
                }
                if constexpr (EPS) {   // d loss / d epsilon (P3:63): through alpha(eps) and eps-multiplied terms
#pragma unroll
                    for (int j = 0; j < QY; ++j) {
                        double g1 = 0.0, gt = 0.0;
#pragma unroll
                        for (int a = 0; a < NA; ++a) {
                            g1 = fma(td.a1[ac.id[a]], o[a][j], g1);
                            gt = fma(td.a0[ac.id[a]] + eps * td.a1[ac.id[a]], o[a][j], gt);
                        }
                        deps = fma(c * gh[j], m * g1 + (td.eps_mult ? gt : 0.0), deps);
                    }
                }
            }
        }
        // (f) adjoint of the integrated channels, coalesced (channels no term uses are never touched)
        if (!OH && col) {
            double* __restrict__ Ge = GBAR + e * NQ + li;
#pragma unroll
            for (int a = 0; a < NA; ++a)
#pragma unroll
                for (int j = 0; j < QY; ++j) Ge[(long)ac.id[a] * N + j * QX] = gacc[a][j];
        }
        if constexpr (EPS) {
            pj_wave_sync();
            Rd[lane] = deps;
            pj_wave_sync();
            if (ev && li == 0) {
                double s = 0.0;
#pragma unroll
                for (int i = 0; i < QX; ++i) s += Rd[slot * LPE + i];
                deps_e[e] = s;
            }
        }
    }
}

#ifdef HPV_EXPERIMENTS   // measured-slower plans of the residual stream (profiles/r04_notes.md 2, 10): libhpvpinn_testhooks.so only
// ------------------------------------------------------------------------------------------------
// Streaming residual kernel (round 4): the one-hot two-term form (Poisson-2D var_form 1) on LARGE batches, residual only
// (R = U - F and the element loss: the launch whose bytes SURVEY.md 8(d) counts, 8 (C_u N + 2 N_R)).
//
// k_project_tp keeps a lane's quadrature column in REGISTERS: 124 VGPRs cap it at 4 waves per SIMD, every wave alternates
// "20 loads -> wait the full loaded latency -> 1 600 cycles of contractions", and 16 B... per-lane 8-byte loads of 160-byte
// row segments touch 8-10 cache lines per instruction: 3.5 TB/s.  Here the data path and the arithmetic are decoupled:
//   * a workgroup (4 waves) owns batches of NB consecutive elements = two contiguous runs of NB Q doubles (one per channel);
//     every thread fetches them with 16-byte loads (fully coalesced, whole cache lines) into REGISTERS for the NEXT batch while
//     the contractions of the current one run from LDS -- the registers are the second buffer, nothing waits in the middle of a batch;
//   * contractions read the integrand from LDS ("a lane owns a column" still: lane (e, t, i) takes T_t[.][i], conflict-free),
//     tables broadcast from LDS, one LDS hand-off T between the contractions, lane (e, k, r-half) finishes both terms of its
//     residual entries and stores R straight from registers (a batch's R is one contiguous run);
//   * 75 KB of LDS per workgroup -> two workgroups per CU cover each other's barriers.
// ------------------------------------------------------------------------------------------------
template <int QX, int QY, int NTX, int NTY, int NB>
struct RsLds {
    static constexpr int NQ = QX * QY, NR = NTX * NTY, LDT = QX + 2;          // (even leading dimension: 16-byte row reads)
    static constexpr int G = 0;                              // [2 channels][NB][NQ] the batch's integrand channels
    static constexpr int T = G + 2 * NB * NQ;                // [NB][2][NTY][LDT]
    static constexpr int SQ = T + NB * 2 * NTY * LDT;        // [NB][NTY][2] partial squares
    static constexpr int TOTAL = SQ + NB * NTY * 2 + 16;
};

// Tables: NOT in LDS.  Every table value a wave needs in a contraction step is wave-uniform (the wave's term in the y-contraction,
// its half of the r range in the x-contraction are functions of the wave index), so the tables are read from global memory at
// uniform addresses -- scalar loads into SGPRs, an SGPR operand per FMA: no LDS read, no VGPR, no wait in front of every FMA (the
// LDS-table version of this kernel spent 19 k cycles per batch in 195 exposed ds_read -> s_waitcnt -> fma round trips).
template <int QX, int QY, int NTX, int NTY, int NB>
__global__ void __launch_bounds__(256, 2) k_residual_stream(ProjDesc pd, int ch0, int ch1, const double* __restrict__ OUT,
                                                            double* __restrict__ R, const double* __restrict__ F,
                                                            const double* __restrict__ coef, long coef_stride,
                                                            const double* __restrict__ wtx, const double* __restrict__ wty,
                                                            double* __restrict__ loss_e, long N, long n_elem) {
    using M = RsLds<QX, QY, NTX, NTY, NB>;
    constexpr int NQ = QX * QY, NR = NTX * NTY, LDT = M::LDT, BT = 256;
    constexpr int RUN = NB * NQ;                              // doubles per channel and batch (contiguous in memory)
    constexpr int NLD = (2 * RUN / 2 + BT - 1) / BT;          // 16-byte loads per thread and batch
    constexpr int RH = NTX / 2;
    static_assert(NQ % 2 == 0 && NB * QX <= 128 && NB * NTY <= 64 && NTX % 2 == 0 && QX % 2 == 0, "lane maps");
    extern __shared__ __attribute__((aligned(16))) double sm[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    typedef double v2d __attribute__((ext_vector_type(2)));
    const double al0 = pd.t[0].a0[ch0], al1 = pd.t[1].a0[ch1];
    const long nbatch = (n_elem + NB - 1) / NB;
    const double* __restrict__ C0 = OUT + (long)ch0 * N;
    const double* __restrict__ C1 = OUT + (long)ch1 * N;
    const long ntot = n_elem * NQ;
    // lane maps.  y-contraction: waves 0, 1 take term 0, waves 2, 3 term 1; the 128 lanes of a pair = (element, column) of NB x QX.
    // x-contraction: wave 0 takes the first half of the r range, wave 1 the second; its lanes = (element, row k); waves 2, 3 rest.
    const int yt = wv >> 1;
    const int yl = (wv & 1) * 64 + lane, ye = yl / QX, yi = yl % QX;
    const bool yon = yl < NB * QX;
    const int xh = wv & 1, xe = lane / NTY, xk = lane % NTY;
    const bool xon = wv < 2 && lane < NB * NTY;
    const double* __restrict__ byt = wty + (long)pd.t[yt].dy * NTY * QY;                 // [k][j]  (wave-uniform address)
    const double* __restrict__ ax0 = wtx + (long)pd.t[0].dx * NTX * QX + xh * RH * QX;   // [r][i] of this wave's r range
    const double* __restrict__ ax1 = wtx + (long)pd.t[1].dx * NTX * QX + xh * RH * QX;
    v2d nx[NLD];
    double nf[RH], nc0, nc1;            // this lane's right-hand-side entries and term coefficients of the prefetched batch
    auto request = [&](long b) {        // the two runs of batch b, 16 bytes per thread and load (clamped at the end of the arrays)
        const long base = b * RUN;
        {   // unconditional loads from clamped addresses: a conditional load is a branch with a wait of its own
            long e_ = b * NB + (xon ? xe : 0);
            e_ = e_ < n_elem ? e_ : n_elem - 1;
            const double* fp = (F ? F : OUT) + e_ * NR + (xon ? xk : 0) * NTX + xh * RH;
#pragma unroll
            for (int r = 0; r < RH; ++r) nf[r] = fp[r];
            nc0 = coef[e_];
            nc1 = coef[coef_stride + e_];
        }
#pragma unroll
        for (int p = 0; p < NLD; ++p) {
            const int idx = 2 * (p * BT + tid);              // 0 .. 2 RUN - 2: first run = channel 0, second = channel 1
            const bool second = idx >= RUN;
            long o = base + (second ? idx - RUN : idx);
            if (o > ntot - 2) o = ntot - 2;
            nx[p] = (idx < 2 * RUN && b < nbatch) ? *(const v2d*)((second ? C1 : C0) + o) : v2d{0.0, 0.0};
        }
    };
    long b = blockIdx.x;
    request(b);
    for (; b < nbatch; b += gridDim.x) {
        // park the batch in LDS (the previous batch's readers are behind the barrier at the loop's end)
#pragma unroll
        for (int p = 0; p < NLD; ++p) {
            const int idx = 2 * (p * BT + tid);
            if (idx < 2 * RUN) *(v2d*)(sm + M::G + idx) = nx[p];
        }
        const long e_x = b * NB + xe;
        const bool xv = xon && e_x < n_elem;
        double u[RH];
#pragma unroll
        for (int r = 0; r < RH; ++r) u[r] = F ? -nf[r] : 0.0;
        const double c0 = nc0 * al0, c1 = nc1 * al1;
        pj_lds_barrier();      // LDS hand-off only (s_waitcnt lgkmcnt(0) + s_barrier): __syncthreads() would also wait for the prefetch
        request(b + gridDim.x);                              // in flight during everything below
        // y-contraction: T_t[k][i] = sum_j BY_t[k][j] G_t[j][i], lane = (element, column) of the wave pair's term
        if (yon) {
            const double* g = sm + M::G + yt * RUN + ye * NQ + yi;
            double gv[QY];
#pragma unroll
            for (int j = 0; j < QY; ++j) gv[j] = g[j * QX];
            double* tt = sm + M::T + ((ye * 2 + yt) * NTY) * LDT + yi;
#pragma unroll
            for (int k = 0; k < NTY; ++k) {
                double acc = 0.0;
#pragma unroll
                for (int j = 0; j < QY; ++j) acc = fma(byt[k * QY + j], gv[j], acc);      // table value: SGPR operand
                tt[k * LDT] = acc;
            }
        }
        pj_lds_barrier();
        // x-contraction, both terms: U[k][r] = sum_t c_t sum_i AX_t[r][i] T_t[k][i], lane = (element, row k), wave = half of r
        if (xon) {
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                const double* tr = sm + M::T + ((xe * 2 + t) * NTY + xk) * LDT;
                const double* __restrict__ ax = t == 0 ? ax0 : ax1;
                double tv[QX];
#pragma unroll
                for (int i = 0; i < QX; i += 2) { const v2d w = *(const v2d*)(tr + i); tv[i] = w[0]; tv[i + 1] = w[1]; }
                const double c = t == 0 ? c0 : c1;
#pragma unroll
                for (int r = 0; r < RH; ++r) {
                    double acc = 0.0;
#pragma unroll
                    for (int i = 0; i < QX; ++i) acc = fma(ax[r * QX + i], tv[i], acc);   // table value: SGPR operand
                    u[r] = fma(c, acc, u[r]);
                }
            }
            double sq = 0.0;
            if (xv) {
#pragma unroll
                for (int r = 0; r < RH; ++r) {
                    R[e_x * NR + xk * NTX + xh * RH + r] = u[r];
                    sq = fma(u[r], u[r], sq);
                }
            }
            sm[M::SQ + (xe * NTY + xk) * 2 + xh] = sq;
        }
        pj_lds_barrier();
        if (tid < NB && b * NB + tid < n_elem) {
            double s = 0.0;
#pragma unroll
            for (int i = 0; i < NTY * 2; ++i) s += sm[M::SQ + tid * (NTY * 2) + i];
            loss_e[b * NB + tid] = s / (double)NR;
        }
    }
}

// ------------------------------------------------------------------------------------------------------------------------------
// The same residual-only stream with LDS-DMA staging (round 4): `global_load_lds_dwordx4` moves a batch's two channel runs, its
// right-hand side and its coefficients straight from global memory into one of THREE LDS buffers -- no register holds a load in
// flight, so two whole batches (86 KB per CU) travel while the third is contracted; the register-staged kernel above keeps one.
// Every load of the kernel is a DMA: an ordinary global load beside them would make the compiler drain the DMA queue at its use
// (vmcnt(0)).  Order of a buffer's life: waves issue their shares of DMA(i) in iteration i - 2; at the top of iteration i every wave
// waits until at most its NEWER operations are outstanding (the DMA of batch i + 1 and a few stores), then the barrier makes all
// shares visible; the buffer is refilled in iteration i + 1, behind the barrier that ended its readers' iteration.
// One workgroup per CU (152 KB of LDS), persistent over the batches.
template <int QX, int QY, int NTX, int NTY, int NB>
struct RdLds {
    static constexpr int NQ = QX * QY, NR = NTX * NTY, LDT = QX + 2;
    static constexpr int RUN = NB * NQ;
    static constexpr int FO = 2 * RUN;                       // inside a buffer: [channel 0 run | channel 1 run | F of the batch | coefficients]
    static constexpr int C0O = FO + NB * NR;                 // four arrays of 64 floats: low / high halves of c_0, of c_1 (lane e = element e of the batch)
    static constexpr int BUF = C0O + 128;                    // doubles per buffer (16-byte multiple)
    static constexpr int T = 3 * BUF;                        // [NB][2][NTY][LDT]
    static constexpr int SQ = T + NB * 2 * NTY * LDT;
    static constexpr int TOTAL = SQ + NB * NTY * 2 + 16;
    static_assert(NB <= 8 && (2 * RUN) % 2 == 0 && (NB * NR) % 2 == 0, "16-byte units");
};

template <int QX, int QY, int NTX, int NTY, int NB>
__global__ void __launch_bounds__(256, 1) k_residual_dma(ProjDesc pd, int ch0, int ch1, const double* __restrict__ OUT,
                                                         double* __restrict__ R, const double* __restrict__ F,
                                                         const double* __restrict__ coef, long coef_stride,
                                                         const double* __restrict__ wtx, const double* __restrict__ wty,
                                                         double* __restrict__ loss_e, long N, long n_elem) {
    using M = RdLds<QX, QY, NTX, NTY, NB>;
    constexpr int NQ = QX * QY, NR = NTX * NTY, LDT = M::LDT, RUN = M::RUN;
    constexpr int RH = NTX / 2;
    constexpr int UNITS = (2 * RUN + NB * NR) / 2;           // 16-byte units of a batch: both runs and F (contiguous in the buffer)
    constexpr int NDI = (UNITS + 255) / 256;                 // DMA instructions per wave and batch (4 waves x 64 lanes x 16 B)
    constexpr int NWAIT = (UNITS - 192 + 255) / 256 - 0;     // ... the FEWEST a wave issues (wave 3's last one may be empty)
    static_assert(NQ % 2 == 0 && NB * QX <= 128 && NB * NTY <= 64 && NTX % 2 == 0 && QX % 2 == 0, "lane maps");
    extern __shared__ __attribute__((aligned(16))) double sm[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    typedef double v2d __attribute__((ext_vector_type(2)));
    typedef const __attribute__((address_space(1))) void* gptr_t;
    typedef __attribute__((address_space(3))) void* lptr_t;
    const double al0 = pd.t[0].a0[ch0], al1 = pd.t[1].a0[ch1];
    const long nbatch = (n_elem + NB - 1) / NB;
    const double* __restrict__ C0 = OUT + (long)ch0 * N;
    const double* __restrict__ C1 = OUT + (long)ch1 * N;
    const double* __restrict__ Fp = F ? F : OUT;             // (no right-hand side: anything readable, the values are not used)
    const long ntot = n_elem * NQ, nftot = n_elem * NR;
    const int yt = wv >> 1;
    const int yl = (wv & 1) * 64 + lane, ye = yl / QX, yi = yl % QX;
    const bool yon = yl < NB * QX;
    const int xh = wv & 1, xe = lane / NTY, xk = lane % NTY;
    const bool xon = wv < 2 && lane < NB * NTY;
    const double* __restrict__ byt = wty + (long)pd.t[yt].dy * NTY * QY;
    const double* __restrict__ ax0 = wtx + (long)pd.t[0].dx * NTX * QX + xh * RH * QX;
    const double* __restrict__ ax1 = wtx + (long)pd.t[1].dx * NTX * QX + xh * RH * QX;
    // this wave's share of the DMA of batch b into buffer `buf` (b >= nbatch: nothing)
    auto dma = [&](long b, int buf) {
        if (b >= nbatch) return;
        double* dst = sm + buf * M::BUF;
#pragma unroll
        for (int p = 0; p < NDI; ++p) {
            const int u0 = p * 256 + wv * 64;                // wave-uniform first unit of this instruction
            if (u0 < UNITS) {
                const int u = u0 + lane;
                if (u < UNITS) {
                    const int d = 2 * u;                     // double index inside the buffer
                    const double* src;
                    if (d < RUN) { long o = b * RUN + d; src = C0 + (o > ntot - 2 ? ntot - 2 : o); }
                    else if (d < 2 * RUN) { long o = b * RUN + (d - RUN); src = C1 + (o > ntot - 2 ? ntot - 2 : o); }
                    else { long o = b * (NB * NR) + (d - 2 * RUN); src = Fp + (o > nftot - 2 ? nftot - 2 : o); }
                    __builtin_amdgcn_global_load_lds((gptr_t)src, (lptr_t)(dst + 2 * u0), 16, 0, 0);
                }
            }
        }
        if (wv == 3 && lane < NB) {       // the term coefficients of the batch's elements: 8 bytes per lane
            long e_ = b * NB + lane;
            e_ = e_ < n_elem ? e_ : n_elem - 1;
            float* cd = (float*)(dst + M::C0O);
            __builtin_amdgcn_global_load_lds((gptr_t)(coef + e_), (lptr_t)cd, 4, 0, 0);
            __builtin_amdgcn_global_load_lds((gptr_t)((const float*)(coef + e_) + 1), (lptr_t)(cd + 64), 4, 0, 0);
            __builtin_amdgcn_global_load_lds((gptr_t)(coef + coef_stride + e_), (lptr_t)(cd + 128), 4, 0, 0);
            __builtin_amdgcn_global_load_lds((gptr_t)((const float*)(coef + coef_stride + e_) + 1), (lptr_t)(cd + 192), 4, 0, 0);
        }
    };
    long b = blockIdx.x;
    dma(b, 0);
    dma(b + gridDim.x, 1);
    int it = 0;
    for (; b < nbatch; b += gridDim.x, ++it) {
        const int buf = it % 3;
        // batch `b` has landed once at most the NEWER operations of this wave are outstanding; then everybody's share is visible
        // (no newer batch in flight: everything)
        if (b + gridDim.x < nbatch) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NWAIT) : "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        pj_lds_barrier();
        dma(b + 2 * (long)gridDim.x, (it + 2) % 3);          // refills the buffer whose readers passed the barrier above
        const double* bs = sm + buf * M::BUF;
        const long e_x = b * NB + xe;
        const bool xv = xon && e_x < n_elem;
        double u[RH];
        {
            const double* fp = bs + M::FO + (xon ? xe : 0) * NR + (xon ? xk : 0) * NTX + xh * RH;
#pragma unroll
            for (int r = 0; r < RH; ++r) u[r] = F ? -fp[r] : 0.0;
        }
        double c0, c1;
        {   // coefficients were DMA'd as two 4-byte halves per element: [lo halves: 64 floats][hi halves: 64 floats]
            const float* cl = (const float*)(bs + M::C0O);
            const int xi_ = xon ? xe : 0;
            c0 = __hiloint2double(__float_as_int(cl[64 + xi_]), __float_as_int(cl[xi_])) * al0;
            c1 = __hiloint2double(__float_as_int(cl[192 + xi_]), __float_as_int(cl[128 + xi_])) * al1;
        }
        if (yon) {
            const double* g = bs + yt * RUN + ye * NQ + yi;
            double gv[QY];
#pragma unroll
            for (int j = 0; j < QY; ++j) gv[j] = g[j * QX];
            double* tt = sm + M::T + ((ye * 2 + yt) * NTY) * LDT + yi;
#pragma unroll
            for (int k = 0; k < NTY; ++k) {
                double acc = 0.0;
#pragma unroll
                for (int j = 0; j < QY; ++j) acc = fma(byt[k * QY + j], gv[j], acc);
                tt[k * LDT] = acc;
            }
        }
        pj_lds_barrier();
        if (xon) {
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                const double* tr = sm + M::T + ((xe * 2 + t) * NTY + xk) * LDT;
                const double* __restrict__ ax = t == 0 ? ax0 : ax1;
                double tv[QX];
#pragma unroll
                for (int i = 0; i < QX; i += 2) { const v2d w = *(const v2d*)(tr + i); tv[i] = w[0]; tv[i + 1] = w[1]; }
                const double c = t == 0 ? c0 : c1;
#pragma unroll
                for (int r = 0; r < RH; ++r) {
                    double acc = 0.0;
#pragma unroll
                    for (int i = 0; i < QX; ++i) acc = fma(ax[r * QX + i], tv[i], acc);
                    u[r] = fma(c, acc, u[r]);
                }
            }
            double sq = 0.0;
            if (xv) {
#pragma unroll
                for (int r = 0; r < RH; ++r) {
                    R[e_x * NR + xk * NTX + xh * RH + r] = u[r];
                    sq = fma(u[r], u[r], sq);
                }
            }
            sm[M::SQ + (xe * NTY + xk) * 2 + xh] = sq;
        }
        pj_lds_barrier();
        if (tid < NB && b * NB + tid < n_elem) {
            double s = 0.0;
#pragma unroll
            for (int i = 0; i < NTY * 2; ++i) s += sm[M::SQ + tid * (NTY * 2) + i];
            loss_e[b * NB + tid] = s / (double)NR;
        }
    }
}

template <int QX, int QY, int NTX, int NTY>
static bool launch_residual_dma(const ProjDesc& pd, const ActiveCh& ac, const double* OUT, double* R, const double* F, const double* coef,
                                long coef_stride, const double* wtx, const double* wty, double* loss_e, long N, long n_elem, hipStream_t s) {
    constexpr int NB = 6;
    using M = RdLds<QX, QY, NTX, NTY, NB>;
    constexpr size_t lds = (size_t)M::TOTAL * sizeof(double);
    static_assert(lds <= 160 * 1024, "three batch buffers fit the LDS");
    static bool attr_set = false;
    if (!attr_set) {
        if (hipFuncSetAttribute((const void*)k_residual_dma<QX, QY, NTX, NTY, NB>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) {
            (void)hipGetLastError();
            return false;
        }
        attr_set = true;
    }
    const long nbatch = (n_elem + NB - 1) / NB;
    const unsigned blocks = (unsigned)std::min<long>(nbatch, 256);          // one resident workgroup per CU, each streams its batches
    hipLaunchKernelGGL((k_residual_dma<QX, QY, NTX, NTY, NB>), dim3(blocks), dim3(256), lds, s, pd, ac.id[0], ac.id[1], OUT, R, F, coef,
                       coef_stride, wtx, wty, loss_e, N, n_elem);
    return true;
}

// ------------------------------------------------------------------------------------------------------------------------------
// Per-WAVE LDS-DMA stream (round 4, second structure): every wave is its own loader and consumer, no workgroup barrier anywhere.
// A wave owns groups of 64 / QX elements exactly like k_project_tp ("a lane owns a line"), but its group's channel runs and
// right-hand side arrive by `global_load_lds_dwordx4` in its private LDS block: at the top of a trip the wave waits for its own
// DMA (vmcnt -- the issuing wave's count is all that orders its own reads), copies its columns / rows into registers, requests the
// NEXT group into the same block and contracts.  The residual stores of a group are issued one trip late, right behind the next
// request, so that the wait at the top (which also covers them: loads and stores share the counter) never meets a young store.
// Five waves per CU keep 5 x 21.6 KB in flight; each request is a linear run of 1 KB pieces.
template <int QX, int QY, int NTX, int NTY, int WAVES>
__global__ void __launch_bounds__(WAVES * 64, 1) k_residual_wdma(ProjDesc pd, int ch0, int ch1, const double* __restrict__ OUT,
                                                                 double* __restrict__ R, const double* __restrict__ F,
                                                                 const double* __restrict__ coef, long coef_stride,
                                                                 const double* __restrict__ wtx, const double* __restrict__ wty,
                                                                 double* __restrict__ loss_e, long N, long n_elem) {
    constexpr int NQ = QX * QY, NR = NTX * NTY, LPE = QX, EPW = 64 / LPE, LDT = QX + 1;
    constexpr int RUNW = EPW * NQ, FW = EPW * NR;            // doubles of a group per channel / of its right-hand side
    constexpr int GU = RUNW / 2, FU = FW / 2;                // ... in 16-byte units
    constexpr int NG = (GU + 63) / 64, NF = (FU + 63) / 64;  // DMA instructions per run
    constexpr int FO = 2 * NG * 128;                         // block layout: [run 0 | run 1 | F | transpose tile | sums], runs padded to whole instructions
    constexpr int TO = FO + NF * 128, RO = TO + EPW * NTY * LDT;
    constexpr int WAVE_D = (RO + 64 + 1) / 2 * 2;
    static_assert(QX == QY && RUNW % 2 == 0 && FW % 2 == 0 && NTX <= LPE && NTY <= LPE, "shape");
    extern __shared__ __attribute__((aligned(16))) double sm[];
    typedef const __attribute__((address_space(1))) void* gptr_t;
    typedef __attribute__((address_space(3))) void* lptr_t;
    const int lane = threadIdx.x & 63;
    const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    double* blk = sm + wv * WAVE_D;
    const int slot = lane / LPE, li = lane % LPE;
    const bool lane_ok = slot < EPW;
    const double al0 = pd.t[0].a0[ch0], al1 = pd.t[1].a0[ch1];
    const double* __restrict__ C0 = OUT + (long)ch0 * N;
    const double* __restrict__ C1 = OUT + (long)ch1 * N;
    const double* __restrict__ Fp = F ? F : OUT;
    const long ntot = n_elem * NQ, nftot = n_elem * NR;
    const long ngroups = (n_elem + EPW - 1) / EPW, gstride = (long)gridDim.x * WAVES;
    const double* __restrict__ by0 = wty + (long)pd.t[0].dy * (NTY * QY);
    const double* __restrict__ by1 = wty + (long)pd.t[1].dy * (NTY * QY);
    const double* __restrict__ ax0 = wtx + (long)pd.t[0].dx * (NTX * QX);
    const double* __restrict__ ax1 = wtx + (long)pd.t[1].dx * (NTX * QX);
    auto request = [&](long grp) {
        if (grp >= ngroups) return;
#pragma unroll
        for (int p = 0; p < 2 * NG + NF; ++p) {
            const int run = p < NG ? 0 : (p < 2 * NG ? 1 : 2), q = run == 0 ? p : (run == 1 ? p - NG : p - 2 * NG);
            const int u = q * 64 + lane;
            if (u < (run == 2 ? FU : GU)) {
                const long o = (run == 2 ? grp * FW : grp * RUNW) + 2 * u;
                const double* src = run == 0 ? C0 + (o > ntot - 2 ? ntot - 2 : o)
                                             : (run == 1 ? C1 + (o > ntot - 2 ? ntot - 2 : o) : Fp + (o > nftot - 2 ? nftot - 2 : o));
                __builtin_amdgcn_global_load_lds((gptr_t)src, (lptr_t)(blk + p * 128), 16, 0, 0);
            }
        }
    };
    long grp = (long)blockIdx.x * WAVES + wv;
    request(grp);
    double up[NTX];                      // the previous group's residual row of this lane, stored one trip late
    long ep = -1;
    bool rowp = false;
    double lossp = 0.0;
#pragma unroll
    for (int r = 0; r < NTX; ++r) up[r] = 0.0;
    for (; grp < ngroups; grp += gstride) {
        const long e = grp * EPW + slot;
        const bool ev = lane_ok && e < n_elem;
        const bool col = ev, row = ev && li < NTY;
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");       // this wave's request has landed (and the stores of two trips ago)
        double g0[QY], g1[QY], u[NTX];
        {
            const double* c0p = blk + (lane_ok ? slot : 0) * NQ + li;
            const double* c1p = blk + NG * 128 + (lane_ok ? slot : 0) * NQ + li;
#pragma unroll
            for (int j = 0; j < QY; ++j) { g0[j] = c0p[j * QX]; g1[j] = c1p[j * QX]; }
            const double* fp = blk + FO + (lane_ok ? slot : 0) * NR + (li < NTY ? li : 0) * NTX;
#pragma unroll
            for (int r = 0; r < NTX; ++r) u[r] = F ? -fp[r] : 0.0;
        }
        // (term coefficients: wave-uniform addresses -> scalar loads, the vector-memory counter never sees them)
        double cw0[EPW], cw1[EPW];
#pragma unroll
        for (int s_ = 0; s_ < EPW; ++s_) {
            long ee = grp * EPW + s_;
            ee = ee < n_elem ? ee : n_elem - 1;
            cw0[s_] = coef[ee]; cw1[s_] = coef[coef_stride + ee];
        }
        double c0 = 0.0, c1 = 0.0;
#pragma unroll
        for (int s_ = 0; s_ < EPW; ++s_) { c0 = slot == s_ ? cw0[s_] : c0; c1 = slot == s_ ? cw1[s_] : c1; }
        c0 *= al0; c1 *= al1;
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");     // the block has been read: refill it
        request(grp + gstride);
        if (rowp) {                                            // the previous group's residual row (see the header)
#pragma unroll
            for (int r = 0; r < NTX; ++r) R[ep * NR + li * NTX + r] = up[r];
        }
        if (ep >= 0 && li == 0 && lane_ok) loss_e[ep] = lossp;
        double* Tb = blk + TO;
        double* Rd = blk + RO;
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            const double* __restrict__ byg = t == 0 ? by0 : by1;
            const double* __restrict__ axg = t == 0 ? ax0 : ax1;
            pj_wave_sync();
            if (col) {
                double acc[NTY];
#pragma unroll
                for (int k = 0; k < NTY; ++k) {
                    acc[k] = 0.0;
#pragma unroll
                    for (int j = 0; j < QY; ++j) acc[k] = fma(byg[k * QY + j], t == 0 ? g0[j] : g1[j], acc[k]);
                }
#pragma unroll
                for (int k = 0; k < NTY; ++k) Tb[slot * (NTY * LDT) + k * LDT + li] = acc[k];
            }
            pj_wave_sync();
            if (row) {
                double trow[QX];
#pragma unroll
                for (int i = 0; i < QX; ++i) trow[i] = Tb[slot * (NTY * LDT) + li * LDT + i];
                const double c = t == 0 ? c0 : c1;
#pragma unroll
                for (int r = 0; r < NTX; ++r) {
                    double acc = 0.0;
#pragma unroll
                    for (int i = 0; i < QX; ++i) acc = fma(axg[r * QX + i], trow[i], acc);
                    u[r] = fma(c, acc, u[r]);
                }
            }
        }
        double sq = 0.0;
        if (row) {
#pragma unroll
            for (int r = 0; r < NTX; ++r) sq = fma(u[r], u[r], sq);
        }
        Rd[lane] = sq;
        pj_wave_sync();
        double ls = 0.0;
        if (ev && li == 0) {
#pragma unroll
            for (int k = 0; k < NTY; ++k) ls += Rd[slot * LPE + k];
        }
#pragma unroll
        for (int r = 0; r < NTX; ++r) up[r] = u[r];
        ep = ev ? e : -1; rowp = row; lossp = ls / (double)NR;
    }
    if (rowp) {
#pragma unroll
        for (int r = 0; r < NTX; ++r) R[ep * NR + li * NTX + r] = up[r];
    }
    if (ep >= 0 && li == 0 && lane_ok) loss_e[ep] = lossp;
}

template <int QX, int QY, int NTX, int NTY>
static bool launch_residual_wdma(const ProjDesc& pd, const ActiveCh& ac, const double* OUT, double* R, const double* F, const double* coef,
                                 long coef_stride, const double* wtx, const double* wty, double* loss_e, long N, long n_elem, hipStream_t s) {
    constexpr int WAVES = 5, EPW = 64 / QX, NQ = QX * QY, NR = NTX * NTY;
    constexpr int NG = (EPW * NQ / 2 + 63) / 64, NF = (EPW * NR / 2 + 63) / 64;
    constexpr int WAVE_D = (2 * NG * 128 + NF * 128 + EPW * NTY * (QX + 1) + 64 + 1) / 2 * 2;
    constexpr size_t lds = (size_t)WAVES * WAVE_D * sizeof(double);
    static_assert(lds <= 160 * 1024, "five wave blocks fit the LDS");
    static bool attr_set = false;
    if (!attr_set) {
        if (hipFuncSetAttribute((const void*)k_residual_wdma<QX, QY, NTX, NTY, WAVES>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) {
            (void)hipGetLastError();
            return false;
        }
        attr_set = true;
    }
    const long ngroups = (n_elem + EPW - 1) / EPW;
    const unsigned blocks = (unsigned)std::min<long>((ngroups + WAVES - 1) / WAVES, 256);
    hipLaunchKernelGGL((k_residual_wdma<QX, QY, NTX, NTY, WAVES>), dim3(blocks), dim3(WAVES * 64), lds, s, pd, ac.id[0], ac.id[1], OUT, R, F,
                       coef, coef_stride, wtx, wty, loss_e, N, n_elem);
    return true;
}

template <int QX, int QY, int NTX, int NTY>
static bool launch_residual_stream(const ProjDesc& pd, const ActiveCh& ac, const double* OUT, double* R, const double* F, const double* coef,
                                   long coef_stride, const double* wtx, const double* wty, double* loss_e, long N, long n_elem, hipStream_t s) {
    constexpr int NB = 6;
    using M = RsLds<QX, QY, NTX, NTY, NB>;
    constexpr size_t lds = (size_t)M::TOTAL * sizeof(double);
    static bool attr_set = false;
    if (!attr_set) {
        if (hipFuncSetAttribute((const void*)k_residual_stream<QX, QY, NTX, NTY, NB>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) {
            (void)hipGetLastError();
            return false;
        }
        attr_set = true;
    }
    const long nbatch = (n_elem + NB - 1) / NB;
    const unsigned blocks = (unsigned)std::min<long>(nbatch, 512);          // two resident workgroups per CU, each streams its batches
    hipLaunchKernelGGL((k_residual_stream<QX, QY, NTX, NTY, NB>), dim3(blocks), dim3(256), lds, s, pd, ac.id[0], ac.id[1], OUT, R, F, coef,
                       coef_stride, wtx, wty, loss_e, N, n_elem);
    return true;
}

#endif  // HPV_EXPERIMENTS

template <int QX, int QY, int NTX, int NTY, int NA, bool EPS, int PJ_WAVES, bool OH = false, bool PIPE = false>
static void launch_tp3(const ProjDesc& pd, const ActiveCh& ac, const double* OUT, double* GBAR, double* R, const double* F,
                       const double* coef, long coef_stride, const double* wtx, const double* wty, const double* eps_ptr,
                       double* loss_e, double* deps_e, long N, long n_elem, int do_adjoint, long ngroups, hipStream_t s) {
    constexpr int LPE = QX > QY ? QX : QY;
    constexpr int EPW = 64 / LPE;
    constexpr int WAVE_DOUBLES = EPW * NTY * (QX + 1) + 64;
    size_t lds = (size_t)(2 * (3 * NTX * QX + 3 * NTY * QY) + PJ_WAVES * WAVE_DOUBLES) * sizeof(double);
    long blocks = (ngroups + PJ_WAVES - 1) / PJ_WAVES;
    // (A/B knobs of the stand-alone bandwidth measurement: HPV_PJ_OCC_PAD = bytes of unused LDS per workgroup, i.e. fewer resident
    //  workgroups per CU; HPV_PJ_GRID = resident-grid cap in workgroups per CU.  scripts/hbm_read_probe.hip: a plain read stream is
    //  FASTER with fewer waves and loads in flight -- 6.3-6.6 TB/s at 2 workgroups per CU against 5.0-5.5 at 4-8)
#ifdef HPV_EXPERIMENTS
    static const long occ_pad = getenv("HPV_PJ_OCC_PAD") ? atol(getenv("HPV_PJ_OCC_PAD")) : 0;
    static const long grid_cap = getenv("HPV_PJ_GRID") ? atol(getenv("HPV_PJ_GRID")) : 16;
#else
    constexpr long occ_pad = 0, grid_cap = 16;
#endif
    lds += (size_t)occ_pad;
    if (blocks > 256 * grid_cap) blocks = 256 * grid_cap;   // grid-stride beyond that
    if (lds > 65536) {
        static bool attr_set = false;
        if (!attr_set) {
            (void)hipFuncSetAttribute((const void*)k_project_tp<QX, QY, NTX, NTY, NA, EPS, PJ_WAVES, OH, PIPE>,
                                      hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
            attr_set = true;
        }
    }
    hipLaunchKernelGGL((k_project_tp<QX, QY, NTX, NTY, NA, EPS, PJ_WAVES, OH, PIPE>), dim3((unsigned)blocks), dim3(PJ_WAVES * 64), lds, s,
                       pd, ac, OUT, GBAR, R, F, coef, coef_stride, wtx, wty, eps_ptr, loss_e, deps_e, N, n_elem, do_adjoint);
}

template <int QX, int QY, int NTX, int NTY, int NA, bool EPS>
static bool launch_tp2(const ProjDesc& pd, const ActiveCh& ac, const double* OUT, double* GBAR, double* R, const double* F,
                       const double* coef, long coef_stride, const double* wtx, const double* wty, const double* eps_ptr,
                       double* loss_e, double* deps_e, long N, long n_elem, int do_adjoint, hipStream_t s) {
    constexpr int LPE = QX > QY ? QX : QY;
    constexpr int EPW = 64 / LPE;
    const long ngroups = (n_elem + EPW - 1) / EPW;
    // few element groups (config-4 scale): one wavefront per workgroup spreads the groups over as many CUs as
    // possible (each wave then has a CU's LDS port to itself for its ~800 broadcast table reads); large batches:
    // four waves per workgroup amortise the table staging
    // one-hot term/channel structure (see k_project_tp): term t integrates active channel t only
    bool onehot = !EPS && pd.nterms == NA && NA >= 2;
    for (int t = 0; t < pd.nterms && onehot; ++t)
        for (int a = 0; a < NA; ++a)
            if (a != t && (pd.t[t].a0[ac.id[a]] != 0.0 || pd.t[t].a1[ac.id[a]] != 0.0)) onehot = false;
#define HPV_GO(W_, OH_, PIPE_)                                                                                           \
    launch_tp3<QX, QY, NTX, NTY, NA, EPS, W_, OH_, PIPE_>(pd, ac, OUT, GBAR, R, F, coef, coef_stride, wtx, wty, eps_ptr, loss_e, deps_e, \
                                                          N, n_elem, do_adjoint, ngroups, s)
    if constexpr (!EPS && NA >= 2) {
        if (onehot) {
#ifdef HPV_EXPERIMENTS
            const bool pipe = getenv("HPV_PJ_PIPE") && getenv("HPV_PJ_PIPE")[0] == '1';     // (A/B switch)
            // (opt-in A/B switch: with its tables as SGPR operands k_project_tp reaches the same 4.4-4.5 TB/s as the streaming kernel
            //  on the 2^18-element batch -- and serves the adjoint half as well)
            const bool no_stream = !(getenv("HPV_PJ_STREAM") && getenv("HPV_PJ_STREAM")[0] == '1');
            if constexpr (NA == 2 && QX == 20 && QY == 20 && NTX == 10 && NTY == 10) {
                // large batches, residual only: the LDS-DMA stream (HPV_PJ_DMA=1: A/B switch)
                const bool dma_on = getenv("HPV_PJ_DMA") && getenv("HPV_PJ_DMA")[0] == '1';
                const bool wdma_on = getenv("HPV_PJ_DMA") && getenv("HPV_PJ_DMA")[0] == '2';       // per-wave loader + consumer
                if (!do_adjoint && wdma_on && n_elem >= 4096 && N == n_elem * (long)(QX * QY) && pd.t[0].a1[ac.id[0]] == 0.0 &&
                    pd.t[1].a1[ac.id[1]] == 0.0 && !pd.t[0].eps_mult && !pd.t[1].eps_mult &&
                    launch_residual_wdma<QX, QY, NTX, NTY>(pd, ac, OUT, R, F, coef, coef_stride, wtx, wty, loss_e, N, n_elem, s))
                    return true;
                if (!do_adjoint && dma_on && n_elem >= 4096 && N == n_elem * (long)(QX * QY) && pd.t[0].a1[ac.id[0]] == 0.0 &&
                    pd.t[1].a1[ac.id[1]] == 0.0 && !pd.t[0].eps_mult && !pd.t[1].eps_mult &&
                    launch_residual_dma<QX, QY, NTX, NTY>(pd, ac, OUT, R, F, coef, coef_stride, wtx, wty, loss_e, N, n_elem, s))
                    return true;
                // large batches, residual only, unit channel weights: the streaming kernel (LDS-staged, register double-buffered)
                if (!do_adjoint && !no_stream && n_elem >= 4096 && N == n_elem * (long)(QX * QY) && pd.t[0].a1[ac.id[0]] == 0.0 &&
                    pd.t[1].a1[ac.id[1]] == 0.0 && !pd.t[0].eps_mult && !pd.t[1].eps_mult &&
                    launch_residual_stream<QX, QY, NTX, NTY>(pd, ac, OUT, R, F, coef, coef_stride, wtx, wty, loss_e, N, n_elem, s))
                    return true;
            }
            if (ngroups <= 1024) HPV_GO(1, true, false);
            else if (!pipe) HPV_GO(8, true, false);
            else HPV_GO(4, true, true);
#else
            if (ngroups <= 1024) HPV_GO(1, true, false); else HPV_GO(8, true, false);
#endif
            return true;
        }
    }
    if (ngroups <= 1024) HPV_GO(1, false, false); else HPV_GO(4, false, false);
#undef HPV_GO
    return true;
}

template <int QX, int QY, int NTX, int NTY>
static bool launch_tp(const ProjDesc& pd, const double* OUT, double* GBAR, double* R, const double* F, const double* coef,
                      long coef_stride, const double* wtx, const double* wty, const double* eps_ptr, double* loss_e,
                      double* deps_e, long N, long n_elem, int do_adjoint, hipStream_t s) {
    ActiveCh ac{};
    for (int ch = 0; ch < pd.C; ++ch) {
        bool used = false;
        for (int t = 0; t < pd.nterms; ++t) used |= (pd.t[t].a0[ch] != 0.0 || pd.t[t].a1[ch] != 0.0);
        if (used) ac.id[ac.n++] = ch;
    }
#define HPV_NA(NA_, EPS_)                                                                                              \
    if (ac.n == NA_ && (pd.has_eps != 0) == EPS_)                                                                      \
        return launch_tp2<QX, QY, NTX, NTY, NA_, EPS_>(pd, ac, OUT, GBAR, R, F, coef, coef_stride, wtx, wty, eps_ptr, loss_e,   \
                                                       deps_e, N, n_elem, do_adjoint, s);
    HPV_NA(1, false) HPV_NA(2, false) HPV_NA(2, true) HPV_NA(3, true)
#undef HPV_NA
    return false;
}

// ------------------------------------------------------------------------------------------------
// Workgroup-per-element variant for TALL elements (80 points per direction: the 1-D rule of P1:238 and the
// AdvDiff 80x80 rule of BASELINE config 5), where a line of the element no longer fits the lanes of one wave.
// Same sum-factorised algorithm as k_project (x-contraction, then y), but with compile-time shapes, the
// term's two tables staged in LDS, the integrand staged once in LDS from batched coalesced loads, and the
// Poisson-1D element-edge term (P1:90) supported.
// ------------------------------------------------------------------------------------------------
#define PW_BLOCK 1024
template <int QX, int QY, int NTX, int NTY>
__global__ void __launch_bounds__(PW_BLOCK) k_project_wg(ProjArgs pa) {
    extern __shared__ __attribute__((aligned(16))) double sm[];
    project_element_wg<QX, QY, NTX, NTY, PW_BLOCK>(pa, (long)blockIdx.x, sm);
}

template <int QX, int QY, int NTX, int NTY>
static bool launch_wg(const ProjDesc& pd, const double* OUT, double* GBAR, double* R, const double* F, const double* coef,
                      long coef_stride, const double* wtx, const double* wty, const double* eps_ptr, double* loss_e,
                      double* deps_e, long N, long n_elem, int do_adjoint, const double* edge_u, const double* edge_dphi,
                      const double* edge_coef, double* edge_gbar, hipStream_t s) {
    constexpr size_t lds = (size_t)project_wg_lds_doubles<QX, QY, NTX, NTY>() * sizeof(double);
    static bool attr_set = false;
    if (!attr_set) {   // > 64 KB of dynamic LDS needs the opt-in
        (void)hipFuncSetAttribute((const void*)k_project_wg<QX, QY, NTX, NTY>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        attr_set = true;
    }
    ProjArgs pa{pd, OUT, GBAR, R, F, coef, coef_stride, wtx, wty, eps_ptr, loss_e, deps_e, N, do_adjoint, edge_u, edge_dphi,
                edge_coef, edge_gbar};
    hipLaunchKernelGGL((k_project_wg<QX, QY, NTX, NTY>), dim3((unsigned)n_elem), dim3(PW_BLOCK), lds, s, pa);
    return true;
}

// Workgroups per element of the row-split projection (1: not applicable); loss_e / deps_e then hold n_elem * split entries.
int project_row_split(const ProjDesc& pd, long n_elem, int backend_generic) {
    if (backend_generic || pd.edge || n_elem <= 0 || n_elem * PJ_SPLIT > 4096) return 1;
    if (pd.qx == 80 && pd.qy == 80 && pd.ntx == 5 && pd.nty == 5) return PJ_SPLIT;
    return 1;
}

template <int QX, int QY, int NTX, int NTY>
static void launch_rows(const ProjArgs& pa, long n_elem, double* upart, hipStream_t s) {
    constexpr size_t lds = (size_t)project_rows_lds_doubles<QX, QY, NTX, NTY>() * sizeof(double);
    static bool attr_set = false;
    if (!attr_set) {
        (void)hipFuncSetAttribute((const void*)k_project_rows_fwd<QX, QY, NTX, NTY>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        (void)hipFuncSetAttribute((const void*)k_project_rows_adj<QX, QY, NTX, NTY>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        attr_set = true;
    }
    const unsigned blocks = (unsigned)(n_elem * PJ_SPLIT);
    hipLaunchKernelGGL((k_project_rows_fwd<QX, QY, NTX, NTY>), dim3(blocks), dim3(PJ_RBLOCK), lds, s, pa, upart);
    hipLaunchKernelGGL((k_project_rows_adj<QX, QY, NTX, NTY>), dim3(blocks), dim3(PJ_RBLOCK), lds, s, pa, upart);
}

bool launch_project_wg(const ProjDesc& pd, const double* OUT, double* GBAR, double* R, const double* F, const double* coef,
                       long coef_stride, const double* wtx, const double* wty, const double* eps_ptr, double* loss_e,
                       double* deps_e, long N, long n_elem, int do_adjoint, const double* edge_u, const double* edge_dphi,
                       const double* edge_coef, double* edge_gbar, hipStream_t s, double* upart) {
    if (n_elem <= 0) return false;
    if (upart && !pd.nact && project_row_split(pd, n_elem, 0) > 1) {   // few tall elements: PJ_SPLIT workgroups per element, two phases
        ProjArgs pa{pd, OUT, GBAR, R, F, coef, coef_stride, wtx, wty, eps_ptr, loss_e, deps_e, N, do_adjoint, nullptr, nullptr,
                    nullptr, nullptr};
        launch_rows<80, 80, 5, 5>(pa, n_elem, upart, s);
        return true;
    }
#define HPV_WG(QX_, QY_, NTX_, NTY_, EXACT_)                                                                              \
    if (pd.qx == QX_ && pd.qy == QY_ && (EXACT_ ? pd.ntx == NTX_ && pd.nty == NTY_ : pd.ntx >= 1 && pd.ntx <= NTX_ && pd.nty >= 1 && pd.nty <= NTY_))   \
        return launch_wg<QX_, QY_, NTX_, NTY_>(pd, OUT, GBAR, R, F, coef, coef_stride, wtx, wty, eps_ptr, loss_e, deps_e, N, \
                                               n_elem, do_adjoint, edge_u, edge_dphi, edge_coef, edge_gbar, s);
    HPV_WG(80, 1, 60, 1, true)     // Poisson-1D reference rule: N_Quad = 80, N_testfcn = 60 (P1:237-238; BASELINE configs 1, 2)
    HPV_WG(80, 80, 5, 5, true)     // AdvDiff with the 80-point rule per direction (BASELINE config 5)
    // small grids of the 2-D shapes (round 4): one 1024-thread workgroup per element spreads a few hundred elements over all CUs,
    // where "a lane owns a line" (k_project_tp: 3-6 elements per WAVE) leaves most of the chip idle -- the caller prefers this
    // launch when the shard has at most two elements per CU
    // (these take any smaller test-function counts at run time: project_element_wg stages the missing functions' tables as zeros)
    HPV_WG(20, 20, 10, 10, false)
    HPV_WG(10, 10, 5, 5, false)
    HPV_WG(16, 16, 8, 8, false)
    HPV_WG(12, 12, 6, 6, false)
    // larger rules (no whole-iteration kernel takes them: forward -> this -> reverse): 35 -> ~10 us of a 120 us iteration at 24x24 points
    HPV_WG(24, 24, 12, 12, false)
    HPV_WG(28, 28, 14, 14, false)
    HPV_WG(32, 32, 16, 16, false)
    HPV_WG(36, 36, 18, 18, false)
    HPV_WG(40, 40, 20, 20, false)
#undef HPV_WG
    return false;
}

// Returns false when the element shape has no specialised instantiation (caller falls back to k_project).
bool launch_project_tp(const ProjDesc& pd, const double* OUT, double* GBAR, double* R, const double* F, const double* coef,
                       long coef_stride, const double* wtx, const double* wty, const double* eps_ptr, double* loss_e,
                       double* deps_e, long N, long n_elem, int do_adjoint, hipStream_t s) {
    if (pd.nact) return false;   // per-element active test counts: the general projections only
    if (pd.edge || n_elem <= 0) return false;
#define HPV_TP(QX_, QY_, NTX_, NTY_)                                                                              \
    if (pd.qx == QX_ && pd.qy == QY_ && pd.ntx == NTX_ && pd.nty == NTY_)                                          \
        return launch_tp<QX_, QY_, NTX_, NTY_>(pd, OUT, GBAR, R, F, coef, coef_stride, wtx, wty, eps_ptr, loss_e,  \
                                               deps_e, N, n_elem, do_adjoint, s);
    HPV_TP(20, 20, 10, 10)   // BASELINE config 4
    HPV_TP(10, 10, 5, 5)     // BASELINE config 3, the Poisson-2D and AdvDiff reference defaults (P2:282-286, P3:47-51)
#undef HPV_TP
    return false;
}
