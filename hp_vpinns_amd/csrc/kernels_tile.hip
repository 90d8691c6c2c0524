// Whole training pass of a shard of SMALL elements in one launch, for every channel set of the MFMA path.
//
// The 1-D drivers (P1: 80 quadrature points = 5 tiles per element, sin activation, u'' or u' integrated against 60 test
// functions) and the 10x10-point elements of the 2-D / AdvDiff reference defaults (P2:275-283, P3:46-51: 7 tiles) are
// latency problems: one 16-point tile per wave, and on the separate path four dependent launches (forward, projection,
// reverse, finalize) that each pay a launch and a weight-staging prologue for that one tile.  Here ONE workgroup owns an
// element: wave w runs tile w of the element forward, keeps the tile's saved state (s, [cos], z_c, z_cc of every hidden
// layer: 40..60 doubles per lane) in REGISTERS, the workgroup projects the element (project_element_wg, the general
// TermDesc projection incl. the trainable epsilon of P3:63), and every wave reverses its tile from the registers.
// Waves beyond the element's tiles adopt the boundary / data tiles behind the elements (P1:98, P3:184); workgroups
// beyond the elements take the rest of those, eight per workgroup.  Nothing is stored for the reverse pass and nothing is
// recomputed; the channel values and their adjoints cross the projection in LDS, its tables are staged with the weights.
// k_iter_small (kernels_fused.hip) is the hand-tuned special case of this for Poisson-2D var_form 1.
//
// Arithmetic: the forward and reverse tile bodies are those of k_fwd_mfma / k_bwd_mfma (kernels_mfma.hip) with the
// activation store replaced by registers and one tile per wave (so the gradient "accumulators" are plain values).
#include <cstdio>
#include <cstdlib>
#include <type_traits>

#include "hpv_mfma_dev.h"

// -DHPV_FZ_TIMING: per-wave phase durations (staging, forward, wait, projection, reverse, wait, epilogue; shader cycles) and
// the wave's wall-clock total / start / end into the (otherwise unused) activation store; scripts/fz_timing.py t1|t2|t5b
#ifdef HPV_FZ_TIMING
#define TL_STAMP(I) tl_t[I] = clock64()
#else
#define TL_STAMP(I)
#endif

template <int L, int WAVES, int QX, int QY, int NTX, int NTY, int D>
struct TlLds {
    static constexpr int LH = L - 1;
    static constexpr int WT = 0;                               // forward A fragments  W^T[out = pt][in = 4s+q]   [LH][5][64]
    static constexpr int BH = WT + LH * MF_KS * 64;            // bias fragments                                  [LH][5][64]
    static constexpr int WR = BH + LH * MF_KS * 64;            // forward remainder (neurons 16..19)              [LH][5][16]
    static constexpr int WN = WR + LH * MF_KS * 16;            // reverse A fragments  W[in = pt][out = 4s+q]     [LH][5][64]
    static constexpr int WRB = WN + LH * MF_KS * 64;           // reverse remainder                               [LH][5][16]
    static constexpr int W1O = WRB + LH * MF_KS * 16;          // first-layer rows, head weights, first bias      [D+2][5][64]
    static constexpr int TAB = W1O + (D + 2) * MF_KS * 64;     // per-wave transpose pair, one channel at a time  [WAVES][2][20*17]
    static constexpr int CHN = TAB + WAVES * 2 * MF_TRB * MF_LD;   // the element's channel values, then their adjoints [2][C][NQ]
    static constexpr int RA = CHN + 2 * HPV_MAXC * QX * QY;        // projection scratch (tables staged up front), then the per-wave gradient rows
    static constexpr int PROJ = project_wg_lds_doubles<QX, QY, NTX, NTY>();
    static constexpr int total(int P) { return RA + (PROJ > WAVES * P ? PROJ : WAVES * P); }
};

// PERSIST (one-workgroup grids only -- the reference's own 1-element 1-D default, BASELINE config 1; verdict round 3, next 7): ONE
// launch runs g.persist_iters whole iterations (forward, projection, reverse, TF1 Adam, loss history) back to back.  The iteration
// is the SAME function as the one-iteration kernel's body (tile_body), called from the loop through a __noinline__ wrapper:
// a loop around the inlined body lets the compiler keep the body's invariants (46 fp64 literals of sincos / tanh, ~100 uniform
// addresses) in registers across the back edge -- 256 VGPRs (the cap of six waves on four SIMDs) + 0.5-1 KB of scratch per lane,
// whose reloads serialise on vmcnt: 34.4 us per iteration against 22.9 for one launch per iteration (round 3, and round 4 again
// with the thread id and every pointer argument laundered through empty asm per trip, and with -disable-machine-licm /
// -disable-constant-hoisting).  Behind a call the body is compiled on its own.  Its arguments live in LDS (tl_args: the kernarg
// segment is not addressable from a callee), the parameter pointer loses its __restrict__ (the Adam update of trip k writes what
// trip k + 1 stages), and all hand-offs between trips go through global memory of ONE CU (stores, s_waitcnt vmcnt(0),
// workgroup barrier, loads: coherent within a CU's L1).
// MEASURED (round 4, config 1, same box, 4 000 iterations): 24.6 us per iteration persistent against 24.0 with one launch per
// iteration -- the call removes the 34 us disaster but not the launch's worth: uniform arguments read from LDS occupy VGPRs
// instead of SGPRs (256 VGPRs + 484 B of scratch in the callee), and the parameters still make the round trip through L2
// between the Adam update and the next trip's staging.  The launch boundary it saves is ~1.5 us.  Hence OPT-IN (HPV_PERSIST=1).
template <bool PERSIST> struct TlParamPtr { typedef const double* __restrict__ type; };
template <> struct TlParamPtr<true> { typedef const double* type; };

template <int D, int NT1, int NT2, int ACT, int L, int QX, int QY, int NTX, int NTY, int WAVES, bool PERSIST>
__device__ __forceinline__ void tile_body(const MfmaArgs& g, double* lds) {
    constexpr int C = 1 + NT1 + NT2, NQ = QX * QY, TPE = (NQ + 15) / 16, BT = WAVES * 64, LH = L - 1, FREE = WAVES - TPE;
    static_assert(L >= 2 && TPE <= WAVES, "one tile per wave");
    using M = TlLds<L, WAVES, QX, QY, NTX, NTY, D>;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int q = lane >> 4, pt = lane & 15;
    typename TlParamPtr<PERSIST>::type th = g.theta;
    const ProjArgs& pa = g.pa;
    const long n_elem = g.proj_n_elem;
    const bool elem_wg = (long)blockIdx.x < n_elem;
    const long e = blockIdx.x;
    // the projection reads its term descriptors from the kernarg segment half a launch from now, in the middle of latency
    // chains: touch their cache lines here (scalar loads nobody waits for), so that those reads hit the scalar cache
#pragma unroll
    for (int t_ = 0; t_ < HPV_MAXT; ++t_) {
        asm volatile("" ::"s"(__double2loint(pa.pd.t[t_].a0[0])));
        asm volatile("" ::"s"(__double2loint(pa.pd.t[t_].a1[HPV_MAXC - 1])));
    }
    asm volatile("" ::"s"(pa.R), "s"(pa.loss_e));
#ifdef HPV_FZ_TIMING
    long long tl_t[8];
    TL_STAMP(0);
    const long long tl_wall = wall_clock64();
#endif

    // ---- the tile of this wave ----
    const long n_dt = g.data_off >= 0 ? (g.N - g.data_off + 15) / 16 : 0;          // boundary / data tiles behind the elements
    const long d0 = elem_wg ? e * FREE : n_elem * FREE + ((long)blockIdx.x - n_elem) * WAVES;   // first one this workgroup adopts
    const bool is_el = elem_wg && wv < TPE;
    long di = is_el ? -1 : d0 + (elem_wg ? wv - TPE : wv);
    if (di >= n_dt) di = -1;
    const bool active = is_el || di >= 0;
    // (data tiles are adopted in order, so the active waves of a workgroup are a prefix: the epilogue sums that many rows)
    const long n_dmine = n_dt - d0 < 0 ? 0 : (n_dt - d0 < (elem_wg ? FREE : WAVES) ? n_dt - d0 : (elem_wg ? FREE : WAVES));
    const int n_act = (elem_wg ? TPE : 0) + (int)n_dmine;
    const long p = is_el ? e * NQ + 16 * wv + pt : g.data_off + 16 * (di < 0 ? 0 : di) + pt;
    const bool valid = is_el ? (16 * wv + pt < NQ) : (di >= 0 && p < g.N);
    const long pc = (valid && p < g.N) ? p : 0;

    // the tile's coordinates and target values travel with the weight staging (one round trip)
    double x[D];
#pragma unroll
    for (int c = 0; c < D; ++c) x[c] = g.X[(long)c * g.N + pc];
    const double udv = (di >= 0 && valid) ? g.ud[pc - g.data_off] : 0.0;

    // ---- stage the weight fragments of both passes and the projection's tables: every global read before the first LDS store ----
    ProjTableRegs<QX, QY, NTX, NTY, BT> ptab;
    {
        constexpr int ITW = (MF_KS * 64 + BT - 1) / BT;
        constexpr int N1 = (D + 2) * MF_KS * 64, IT1 = (N1 + BT - 1) / BT;
        static_assert(MF_KS * 16 <= BT, "one remainder fragment per thread");
        double vw[LH][ITW], vb[LH][ITW], vn[LH][ITW], vr[LH], vq[LH], v1[IT1];
        const int w0o = g.woff[0], wLo = g.woff[L], b0o = g.boff[0];
#pragma unroll
        for (int i_ = 1; i_ < L; ++i_) {
            const int wo_ = g.woff[i_], bo_ = g.boff[i_];
#pragma unroll
            for (int it = 0; it < ITW; ++it) {
                const int f = it * BT + tid, fc = f < MF_KS * 64 ? f : 0;
                const int ln = fc & 63, s_ = fc >> 6;
                vw[i_ - 1][it] = th[wo_ + (4 * s_ + (ln >> 4)) * MF_H + (ln & 15)];
                vb[i_ - 1][it] = th[bo_ + 4 * s_ + (ln >> 4)];
                vn[i_ - 1][it] = th[wo_ + (ln & 15) * MF_H + 4 * s_ + (ln >> 4)];
            }
            const int fr = tid < MF_KS * 16 ? tid : 0;
            vr[i_ - 1] = th[wo_ + (4 * (fr >> 4) + ((fr >> 2) & 3)) * MF_H + 16 + (fr & 3)];
            vq[i_ - 1] = th[wo_ + (16 + (fr & 3)) * MF_H + 4 * (fr >> 4) + ((fr >> 2) & 3)];
        }
#pragma unroll
        for (int it = 0; it < IT1; ++it) {
            const int f = it * BT + tid, fc = f < N1 ? f : 0;
            const int ln = fc & 63, s_ = (fc >> 6) % MF_KS, c_ = fc / (64 * MF_KS);
            const int j = 4 * s_ + (ln >> 4);
            v1[it] = th[(c_ < D ? w0o + c_ * MF_H : (c_ == D ? wLo : b0o)) + j];
        }
        if (elem_wg) ptab.load(pa, e);   // requested LAST (loads return in order): the weight stores below do not wait for them
#pragma unroll
        for (int i_ = 1; i_ < L; ++i_) {
#pragma unroll
            for (int it = 0; it < ITW; ++it) {
                const int f = it * BT + tid;
                if (f < MF_KS * 64) {
                    lds[M::WT + (i_ - 1) * MF_KS * 64 + f] = vw[i_ - 1][it];
                    lds[M::BH + (i_ - 1) * MF_KS * 64 + f] = vb[i_ - 1][it];
                    lds[M::WN + (i_ - 1) * MF_KS * 64 + f] = vn[i_ - 1][it];
                }
            }
            if (tid < MF_KS * 16) {
                lds[M::WR + (i_ - 1) * MF_KS * 16 + tid] = vr[i_ - 1];
                lds[M::WRB + (i_ - 1) * MF_KS * 16 + tid] = vq[i_ - 1];
            }
        }
#pragma unroll
        for (int it = 0; it < IT1; ++it) { const int f = it * BT + tid; if (f < N1) lds[M::W1O + f] = v1[it]; }
    }

    const double bo = th[g.boff[L]];
    pj_lds_barrier();        // the weight fragments are in LDS; the table loads (ptab) and bo stay in flight behind it
    TL_STAMP(1);

    const double* WT = lds + M::WT;
    const double* BH = lds + M::BH;
    const double* WR = lds + M::WR;
    const double* WN = lds + M::WN;
    const double* WRB = lds + M::WRB;
    const double* W1O = lds + M::W1O;

    // saved state of the tile: per hidden layer and lane the 5 neurons' s, [cos], z_c, z_cc (layer 1: z_c = W1, z_cc = 0)
    struct Slots {
        double a[MF_KS], a1s[MF_KS];
        double zc[NT1 > 0 ? NT1 : 1][MF_KS];
        double zcc[NT2 > 0 ? NT2 : 1][MF_KS];
    };
    Slots st[L];
    double gdat = 0.0;            // adjoint of u at a data tile's point (P1:98, P2:122, P3:184)

    // =============================================================================================
    // forward
    // =============================================================================================
    if (active) {
        double h[C][MF_KS];
        // the activation loops exist twice: branch-free sincos (FAST) when the whole wave's pre-activations are in its range --
        // the five chains of a lane then interleave -- and the guarded one otherwise (tanh: FAST is the only version)
        {
            double z1[MF_KS];
#pragma unroll
            for (int s = 0; s < MF_KS; ++s) {
                double z = W1O[((D + 1) * MF_KS + s) * 64 + lane];
#pragma unroll
                for (int c = 0; c < D; ++c) z = fma(x[c], W1O[(c * MF_KS + s) * 64 + lane], z);
                z1[s] = z;
            }
            auto act1 = [&](auto fast) {
#pragma unroll
                for (int s = 0; s < MF_KS; ++s) {
                    double a, a1, a2;
                    act_fwd<ACT, decltype(fast)::value>(z1[s], a, a1, a2);
                    h[0][s] = a;
                    st[0].a[s] = a;
                    st[0].a1s[s] = a1;
#pragma unroll
                    for (int t = 0; t < NT1; ++t) h[1 + t][s] = a1 * W1O[((t < D ? t : 0) * MF_KS + s) * 64 + lane];
#pragma unroll
                    for (int b = 0; b < NT2; ++b) {
                        const double zc = W1O[((b < D ? b : 0) * MF_KS + s) * 64 + lane];
                        if constexpr (T2Mix<NT1, NT2>::value) h[1 + NT1 + b][s] = a2 * t2_square<NT1, NT2>(g.t2w, b, W1O[s * 64 + lane], W1O[((D > 1 ? 1 : 0) * MF_KS + s) * 64 + lane]);
                        else h[1 + NT1 + b][s] = a2 * zc * zc;
                    }
                }
            };
            if (act_wave_needs_safe<ACT>(z1)) act1(std::false_type{}); else act1(std::true_type{});
        }
#pragma unroll
        for (int i = 1; i < L; ++i) {
            v4d acc[C];
            const double* bhl = BH + (i - 1) * MF_KS * 64 + lane;
            double z16[C];
#pragma unroll
            for (int ch = 0; ch < C; ++ch)
                acc[ch] = (ch == 0) ? v4d{bhl[0], bhl[64], bhl[128], bhl[192]} : v4d{0.0, 0.0, 0.0, 0.0};
#pragma unroll
            for (int s = 0; s < MF_KS; ++s)
#pragma unroll
                for (int ch = 0; ch < C; ++ch)
                    acc[ch] = __builtin_amdgcn_mfma_f64_16x16x4f64(WT[((i - 1) * MF_KS + s) * 64 + lane], h[ch][s], acc[ch], 0, 0, 0);
            {
                const double* wrl = WR + (i - 1) * MF_KS * 16 + q * 4 + (lane & 3);
                double wr[MF_KS];
#pragma unroll
                for (int s = 0; s < MF_KS; ++s) wr[s] = wrl[s * 16];
#pragma unroll
                for (int ch = 0; ch < C; ++ch) {
                    double zz = (ch == 0) ? bhl[256] : 0.0;
#pragma unroll
                    for (int s = 0; s < MF_KS; ++s) zz = __builtin_amdgcn_mfma_f64_4x4x4f64(wr[s], h[ch][s], zz, 0, 0, 0);
                    z16[ch] = zz;
                }
            }
            double zv[MF_KS];
#pragma unroll
            for (int s = 0; s < MF_KS; ++s) zv[s] = s < 4 ? acc[0][s & 3] : z16[0];
            auto acti = [&](auto fast) {
#pragma unroll
                for (int s = 0; s < MF_KS; ++s) {
                    double a, a1, a2;
                    act_fwd<ACT, decltype(fast)::value>(zv[s], a, a1, a2);
                    h[0][s] = a;
                    st[i].a[s] = a;
                    st[i].a1s[s] = a1;
                    double zc[NT1 > 0 ? NT1 : 1];
#pragma unroll
                    for (int u = 0; u < NT1; ++u) {
                        zc[u] = s < 4 ? acc[1 + u][s & 3] : z16[1 + u];
                        st[i].zc[u][s] = zc[u];
                        h[1 + u][s] = a1 * zc[u];
                    }
#pragma unroll
                    for (int b = 0; b < NT2; ++b) {
                        const double zcc = s < 4 ? acc[1 + NT1 + b][s & 3] : z16[1 + NT1 + b];
                        const double z1 = zc[b < NT1 ? b : 0];
                        st[i].zcc[b][s] = zcc;
                        if constexpr (T2Mix<NT1, NT2>::value) h[1 + NT1 + b][s] = a2 * t2_square<NT1, NT2>(g.t2w, b, zc[0], zc[NT1 > 1 ? 1 : 0]) + a1 * zcc;
                        else h[1 + NT1 + b][s] = a2 * z1 * z1 + a1 * zcc;
                    }
                }
            };
            if (act_wave_needs_safe<ACT>(zv)) acti(std::false_type{}); else acti(std::true_type{});
        }
        // linear head: every lane ends up with the full sum of its point
#pragma unroll
        for (int ch = 0; ch < C; ++ch) {
            double v = 0.0;
#pragma unroll
            for (int s = 0; s < MF_KS; ++s) v = fma(h[ch][s], W1O[(D * MF_KS + s) * 64 + lane], v);
            v = xrow_sum16(v);
            v = xrow_sum32(v);
            if (ch == 0) v += bo;
            if (is_el) {
                if (q == 0 && valid) lds[M::CHN + ch * NQ + 16 * wv + pt] = v;
            } else if (ch == 0) {
                const double dd = valid ? udv - v : 0.0;
                gdat = g.data_scale * dd;
                const double sq = row_sum16(dd * dd);
                if (lane == 0) g.data_part[di] = sq;
            }
        }
    }

    // =============================================================================================
    // projection of the element (residual, element loss, adjoint channels): the general TermDesc device function
    // =============================================================================================
    // the projection's tables were requested with the weights; they are parked in LDS only now, so that their round trip
    // (38..77 KB per workgroup) hides behind the forward pass instead of sitting in front of it
    if (elem_wg) ptab.store(lds + M::RA);
    TL_STAMP(2);
    __syncthreads();
    TL_STAMP(3);
    if (elem_wg) {
        // (the element's channels and their adjoints stay in LDS)
        if constexpr (QY == 1 && NTY == 1)
            project_element_1d<QX, NTX, BT>(pa, e, lds + M::RA, lds + M::CHN, lds + M::CHN + HPV_MAXC * NQ);
        else
            project_element_wg<QX, QY, NTX, NTY, BT, true>(pa, e, lds + M::RA, lds + M::CHN, lds + M::CHN + HPV_MAXC * NQ);
#ifdef HPV_PJ_TIMING
        if (threadIdx.x == 0) pa.GBAR[e * 16 + 7] = (double)clock64();
#endif
        pj_lds_barrier();      // (adjoint channels in LDS; the R / loss stores need not have been acknowledged)
    }
    TL_STAMP(4);

    // =============================================================================================
    // reverse
    // =============================================================================================
    double* WP = lds + M::RA + (long)wv * g.P;            // this wave's gradient row (the projection scratch is dead)
    if (active) {
        double gb[C];
#pragma unroll
        for (int ch = 0; ch < C; ++ch)
            gb[ch] = is_el ? (valid ? lds[M::CHN + (HPV_MAXC + ch) * NQ + 16 * wv + pt] : 0.0) : (ch == 0 ? gdat : 0.0);
        double* TA = lds + M::TAB + wv * (2 * MF_TRB * MF_LD);
        double* TB = TA + MF_TRB * MF_LD;
        auto zc_of = [&](int i, int u, int s) -> double {   // z_c of layer i (compile-time i after unrolling)
            return i == 0 ? W1O[((u < D ? u : 0) * MF_KS + s) * 64 + lane] : st[i].zc[u < NT1 ? u : 0][s];
        };
        auto outputs_of = [&](int i, int ch, double (&hv)[MF_KS]) {   // channel ch of layer i's outputs
#pragma unroll
            for (int s = 0; s < MF_KS; ++s) {
                double a1, a2, a3;
                act_saved<ACT>(st[i].a[s], st[i].a1s[s], a1, a2, a3);
                if (ch == 0) hv[s] = st[i].a[s];
                else if (ch <= NT1) hv[s] = a1 * zc_of(i, ch - 1, s);
                else {
                    const int b = ch - 1 - NT1;
                    const double z1 = zc_of(i, b, s);
                    if constexpr (T2Mix<NT1, NT2>::value) hv[s] = a2 * t2_square<NT1, NT2>(g.t2w, b, zc_of(i, 0, s), zc_of(i, NT1 > 1 ? 1 : 0, s)) + (i == 0 ? 0.0 : a1 * st[i].zcc[0][s]);
                    else hv[s] = a2 * z1 * z1 + (i == 0 ? 0.0 : a1 * st[i].zcc[b < NT2 ? b : 0][s]);
                }
            }
        };
        double hbar[C][MF_KS], zbar[C][MF_KS];
        // linear head
        {
            double dwo[MF_KS];
#pragma unroll
            for (int s = 0; s < MF_KS; ++s) dwo[s] = 0.0;
#pragma unroll
            for (int ch = 0; ch < C; ++ch) {
                double hv[MF_KS];
                outputs_of(L - 1, ch, hv);
#pragma unroll
                for (int s = 0; s < MF_KS; ++s) {
                    dwo[s] = fma(hv[s], gb[ch], dwo[s]);
                    hbar[ch][s] = gb[ch] * W1O[(D * MF_KS + s) * 64 + lane];
                }
            }
#pragma unroll
            for (int s = 0; s < MF_KS; ++s) {
                const double t = row_sum16(dwo[s]);
                if (pt == 0) WP[g.woff[L] + 4 * s + q] = t;
            }
            const double t = row_sum16(q == 0 ? gb[0] : 0.0);
            if (lane == 0) WP[g.boff[L]] = t;
        }
#pragma unroll
        for (int i = L - 1; i >= 0; --i) {
#pragma unroll
            for (int s = 0; s < MF_KS; ++s) {
                double a1, a2, a3;
                act_saved<ACT>(st[i].a[s], st[i].a1s[s], a1, a2, a3);
                double zb = hbar[0][s] * a1;
#pragma unroll
                for (int u = 0; u < NT1; ++u) {
                    zbar[1 + u][s] = hbar[1 + u][s] * a1;
                    zb += hbar[1 + u][s] * a2 * zc_of(i, u, s);
                }
#pragma unroll
                for (int b = 0; b < NT2; ++b) {
                    const int u = b < NT1 ? b : 0;
                    const double hb = hbar[1 + NT1 + b][s];
                    const double z1 = zc_of(i, u, s);
                    zbar[1 + NT1 + b][s] = hb * a1;
                    if constexpr (T2Mix<NT1, NT2>::value) {      // the mixed second tangent rides on both first tangents
                        const double y0 = zc_of(i, 0, s), y1 = zc_of(i, NT1 > 1 ? 1 : 0, s);
                        zbar[1][s] += 2.0 * hb * a2 * g.t2w[0] * y0;
                        zbar[2][s] += 2.0 * hb * a2 * g.t2w[1] * y1;
                        zb += hb * (a3 * t2_square<NT1, NT2>(g.t2w, b, y0, y1) + (i == 0 ? 0.0 : a2 * st[i].zcc[b][s]));
                    } else {
                        zbar[1 + u][s] += 2.0 * hb * a2 * z1;
                        zb += hb * (a3 * z1 * z1 + (i == 0 ? 0.0 : a2 * st[i].zcc[b][s]));
                    }
                }
                zbar[0][s] = zb;
                const double t = row_sum16(zb);
                if (pt == 0) WP[g.boff[i] + 4 * s + q] = t;
            }
            if (i == 0) {
                // dW1[c][j] = sum_pt x_c zbar[j] + [c in T1] zbar_c[j]
#pragma unroll
                for (int s = 0; s < MF_KS; ++s) {
#pragma unroll
                    for (int c = 0; c < D; ++c) {
                        double v = x[c] * zbar[0][s];
                        if (c < NT1) v += zbar[1 + (c < NT1 ? c : 0)][s];
                        const double t = row_sum16(valid ? v : 0.0);
                        if (pt == 0) WP[g.woff[0] + c * MF_H + 4 * s + q] = t;
                    }
                }
            } else {
                // weight gradient: contraction over the tile's 16 points and the channels; ONE channel's transpose pair in LDS
                // at a time (the pairs of all channels would be 8 x 16 KB)
                v4d dWacc = v4d{0.0, 0.0, 0.0, 0.0};
                double dS10 = 0.0, dS01 = 0.0, accC = 0.0;
#pragma unroll
                for (int ch = 0; ch < C; ++ch) {
                    double hv[MF_KS];
                    outputs_of(i - 1, ch, hv);
                    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
                    __builtin_amdgcn_wave_barrier();
#pragma unroll
                    for (int s = 0; s < MF_KS; ++s) {
                        TA[(4 * s + q) * MF_LD + pt] = hv[s];
                        TB[(4 * s + q) * MF_LD + pt] = zbar[ch][s];
                    }
                    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
                    __builtin_amdgcn_wave_barrier();
                    double aF[4], bF[4], aS[4], bS[4];
#pragma unroll
                    for (int kk = 0; kk < 4; ++kk) {
                        aF[kk] = TA[pt * MF_LD + 4 * kk + q];
                        bF[kk] = TB[pt * MF_LD + 4 * kk + q];
                        aS[kk] = TA[(16 + (lane & 3)) * MF_LD + 4 * kk + q];
                        bS[kk] = TB[(16 + (lane & 3)) * MF_LD + 4 * kk + q];
                    }
                    const double cA = TA[(16 + (lane & 3)) * MF_LD + (pt & 12) + q];
                    const double cB = TB[(16 + (lane & 3)) * MF_LD + (pt & 12) + q];
#pragma unroll
                    for (int kk = 0; kk < 4; ++kk) {
                        dWacc = __builtin_amdgcn_mfma_f64_16x16x4f64(aF[kk], bF[kk], dWacc, 0, 0, 0);
                        dS10 = __builtin_amdgcn_mfma_f64_4x4x4f64(aS[kk], bF[kk], dS10, 0, 0, 0);
                        dS01 = __builtin_amdgcn_mfma_f64_4x4x4f64(bS[kk], aF[kk], dS01, 0, 0, 0);
                    }
                    accC = __builtin_amdgcn_mfma_f64_4x4x4f64(cA, cB, accC, 0, 0, 0);
                }
#pragma unroll
                for (int r = 0; r < 4; ++r) WP[g.woff[i] + (4 * r + q) * MF_H + pt] = dWacc[r];
                WP[g.woff[i] + (16 + q) * MF_H + pt] = dS10;
                WP[g.woff[i] + pt * MF_H + 16 + q] = dS01;
                {
                    double t = accC;
                    t = quad4_sum(t);
                    if (pt < 4) WP[g.woff[i] + (16 + q) * MF_H + 16 + pt] = t;
                }
                // hbar_in^T = W zbar^T
#pragma unroll
                for (int ch = 0; ch < C; ++ch) {
                    v4d acc = v4d{0.0, 0.0, 0.0, 0.0};
                    double h4 = 0.0;
                    const double* wrl = WRB + (i - 1) * MF_KS * 16 + q * 4 + (lane & 3);
#pragma unroll
                    for (int s = 0; s < MF_KS; ++s) {
                        acc = __builtin_amdgcn_mfma_f64_16x16x4f64(WN[((i - 1) * MF_KS + s) * 64 + lane], zbar[ch][s], acc, 0, 0, 0);
                        h4 = __builtin_amdgcn_mfma_f64_4x4x4f64(wrl[s * 16], zbar[ch][s], h4, 0, 0, 0);
                    }
#pragma unroll
                    for (int s = 0; s < 4; ++s) hbar[ch][s] = acc[s];
                    hbar[ch][4] = h4;
                }
            }
        }
    }

    // ---- epilogue: the active waves' rows -> one gradient row per workgroup ----
    TL_STAMP(5);
    __syncthreads();
    TL_STAMP(6);
    const double* W0 = lds + M::RA;
    if (g.fin_mode == 0) {
        double* row = g.GPART + (long)blockIdx.x * g.P;
        for (int idx = tid; idx < g.P; idx += BT) {
            double acc = 0.0;
            for (int w = 0; w < n_act; ++w) acc += W0[(long)w * g.P + idx];
            row[idx] = acc;
        }
    } else {
        // ONE workgroup is the whole grid: its row IS the gradient -- write the packed buffer, apply the TF1 Adam rule and record
        // the loss here (k_finalize's arithmetic, kernels_generic.hip), no dependent launch
        const AdamArgs& ad = g.fin_ad;
        const bool upd = g.fin_mode == 2;
        const int P = g.P, Ptot = P + (g.fin_has_eps ? 1 : 0);
        const double b1p = upd ? ad.state[0] : 0.0, b2p = upd ? ad.state[1] : 0.0;
        const double lr_t = upd ? hpv_adam_lr_t(ad.lr, b1p, b2p) : 0.0;
        // (every operand of the update is requested before the first sum: one memory round trip, not one per 384 parameters)
        constexpr int FIT = (1341 + BT - 1) / BT;      // P <= 1341 for the networks this path takes (L <= 4)
        double m0[FIT], v0[FIT], th0[FIT];
#pragma unroll
        for (int it = 0; it < FIT; ++it) {
            const int idx = it * BT + tid, ic = idx < P ? idx : 0;
            m0[it] = upd ? ad.m[ic] : 0.0; v0[it] = upd ? ad.v[ic] : 0.0; th0[it] = upd ? ad.theta[ic] : 0.0;
        }
#pragma unroll
        for (int it = 0; it < FIT; ++it) {
            const int idx = it * BT + tid;
            if (idx < P) {
                double t = 0.0;
                for (int w = 0; w < n_act; ++w) t += W0[(long)w * P + idx];
                g.fin_RB[idx] = t;
                if (upd) {
                    double mi, vi, ti;
                    hpv_adam_one_lr(lr_t, ad.b1, ad.b2, ad.eps, t, m0[it], v0[it], th0[it], mi, vi, ti);
                    ad.m[idx] = mi;
                    ad.v[idx] = vi;
                    ad.theta[idx] = ti;
                }
            }
        }
        for (int idx = FIT * BT + tid; idx < P; idx += BT) {      // (wider networks than expected: plain loop)
            double t = 0.0;
            for (int w = 0; w < n_act; ++w) t += W0[(long)w * P + idx];
            g.fin_RB[idx] = t;
            if (upd) {
                double mi, vi, ti;
                hpv_adam_one_lr(lr_t, ad.b1, ad.b2, ad.eps, t, ad.m[idx], ad.v[idx], ad.theta[idx], mi, vi, ti);
                ad.m[idx] = mi;
                ad.v[idx] = vi;
                ad.theta[idx] = ti;
            }
        }
        if (tid == 0) {      // the scalars: loss_e / deps_e of the one element were written by this thread (project_element_wg)
            const double lv = pa.loss_e[0];
            const double de = (g.fin_has_eps && pa.deps_e) ? pa.deps_e[0] : 0.0;
            double sq = 0.0;
            for (int i = 0; i < g.fin_n_data_part; ++i) sq += g.data_part[i];     // (written before the barrier above, this CU)
            const double msq = g.fin_n_data > 0 ? sq / (double)g.fin_n_data : 0.0;
            const double eps_now = (g.fin_has_eps && upd) ? ad.theta[P] : 0.0;
            if (g.fin_has_eps) {
                g.fin_RB[P] = de;
                if (upd) {
                    double mi, vi, ti;
                    hpv_adam_one_lr(lr_t, ad.b1, ad.b2, ad.eps, de, ad.m[P], ad.v[P], eps_now, mi, vi, ti);
                    ad.m[P] = mi;
                    ad.v[P] = vi;
                    ad.theta[P] = ti;
                }
            }
            if (upd && ad.n_upd) *ad.n_upd += 1;
            if (upd && ad.hist) {
                const int i = *ad.hist_idx;
                if (i >= 0 && i < ad.hist_cap) {
                    ad.hist[4 * i] = lv; ad.hist[4 * i + 1] = g.fin_lossb_weight * msq; ad.hist[4 * i + 2] = msq; ad.hist[4 * i + 3] = eps_now;
                    *ad.hist_idx = i + 1;
                }
            }
            g.fin_RB[Ptot + 0] = lv;
            g.fin_RB[Ptot + 1] = g.fin_lossb_weight * msq;
            g.fin_RB[Ptot + 2] = msq;
            g.fin_RB[Ptot + 3] = 0.0;
        }
        if (upd) {           // every replicated copy of the running beta powers advances (k_finalize keeps one per block, k_adam all)
            __syncthreads();                       // (all reads of state[0..1] above are done)
            for (int c = tid; c < g.fin_ncopies; c += BT) { ad.state[2 * c] = b1p * ad.b1; ad.state[2 * c + 1] = b2p * ad.b2; }
        }
    }
#ifdef HPV_FZ_TIMING
    if (lane == 0 && g.ACTS) {
        TL_STAMP(7);
        double* o = g.ACTS + ((long)blockIdx.x * WAVES + wv) * 10;
        for (int i = 0; i < 7; ++i) o[i] = (double)(tl_t[i + 1] - tl_t[i]);
        const long long tl_end = wall_clock64();
        o[7] = (double)(tl_end - tl_wall) * 0.01;
        o[8] = (double)(tl_wall & 0xffffffffffll) * 0.01;
        o[9] = (double)(tl_end & 0xffffffffffll) * 0.01;
    }
#endif
}

template <int D, int NT1, int NT2, int ACT, int L, int QX, int QY, int NTX, int NTY, int WAVES>
__global__ void __launch_bounds__(WAVES * 64, 1) k_iter_tile(MfmaArgs g) {
    extern __shared__ __attribute__((aligned(16))) double lds[];
    tile_body<D, NT1, NT2, ACT, L, QX, QY, NTX, NTY, WAVES, false>(g, lds);
}

#ifdef HPV_EXPERIMENTS   // measured no faster (24.6 against 24.0 us, see tile_body): libhpvpinn_testhooks.so only, HPV_PERSIST=1 there
// the persistent launch: arguments in (static) LDS, the body behind a call
__shared__ MfmaArgs tl_args;
template <int D, int NT1, int NT2, int ACT, int L, int QX, int QY, int NTX, int NTY, int WAVES>
__device__ __noinline__ void tile_body_call() {
    extern __shared__ __attribute__((aligned(16))) double lds[];
    tile_body<D, NT1, NT2, ACT, L, QX, QY, NTX, NTY, WAVES, true>(tl_args, lds);
}
template <int D, int NT1, int NT2, int ACT, int L, int QX, int QY, int NTX, int NTY, int WAVES>
__global__ void __launch_bounds__(WAVES * 64, 1) k_iter_tile_persist(MfmaArgs g) {
    static_assert(sizeof(MfmaArgs) % 4 == 0, "word-wise copy");
    for (int i = threadIdx.x; i < (int)(sizeof(MfmaArgs) / 4); i += WAVES * 64) ((int*)&tl_args)[i] = ((const int*)&g)[i];
    __syncthreads();
    const int n_trips = g.persist_iters;
    for (int trip = 0; trip < n_trips; ++trip) {
        tile_body_call<D, NT1, NT2, ACT, L, QX, QY, NTX, NTY, WAVES>();
        __syncthreads();        // the next trip stages the updated parameters (s_waitcnt vmcnt(0) + barrier) and reuses the LDS
    }
}

#endif  // HPV_EXPERIMENTS

// ------------------------------------------------------------------------------------------------
// host side
// ------------------------------------------------------------------------------------------------
template <int D, int NT1, int NT2, int ACT, int L, int QX, int QY, int NTX, int NTY, int WAVES, bool PERSIST = false>
static bool launch_iter_tile(const MfmaArgs& a, int blocks, hipStream_t s) {
#ifdef HPV_EXPERIMENTS
    if constexpr (!PERSIST && QY == 1) {      // persistent loop: instantiated for the 1-D rule (config 1)
        if (a.persist_iters > 1) return launch_iter_tile<D, NT1, NT2, ACT, L, QX, QY, NTX, NTY, WAVES, true>(a, blocks, s);
    }
#else
    static_assert(!PERSIST, "the persistent launch exists in -DHPV_EXPERIMENTS builds only");
#endif
    static_assert(!PERSIST || QY == 1, "the persistent launch is instantiated for the 1-D rule only");
    using M = TlLds<L, WAVES, QX, QY, NTX, NTY, D>;
    const size_t bytes = (size_t)M::total(a.P) * sizeof(double);
#ifdef HPV_EXPERIMENTS
    static const bool dbg = getenv("HPV_TILE_DEBUG") != nullptr;
#else
    constexpr bool dbg = false;
#endif
    if (bytes + (PERSIST ? sizeof(MfmaArgs) + 64 : 0) > 160 * 1024) {
        if (dbg) fprintf(stderr, "hpv_mfma_iter_tile: %zu bytes of LDS needed\n", bytes);
        return false;
    }
    static bool attr_set = false;
    if (!attr_set) {
        const void* kfn;
#ifdef HPV_EXPERIMENTS
        if constexpr (PERSIST) kfn = (const void*)k_iter_tile_persist<D, NT1, NT2, ACT, L, QX, QY, NTX, NTY, WAVES>;
        else
#endif
        kfn = (const void*)k_iter_tile<D, NT1, NT2, ACT, L, QX, QY, NTX, NTY, WAVES>;
        const hipError_t e = hipFuncSetAttribute(kfn,
                                                 hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
        if (e != hipSuccess) {
            if (dbg) fprintf(stderr, "hpv_mfma_iter_tile: hipFuncSetAttribute(%zu bytes): %s\n", bytes, hipGetErrorString(e));
            (void)hipGetLastError();
            return false;
        }
        attr_set = true;
    }
#ifdef HPV_EXPERIMENTS
    if constexpr (PERSIST) hipLaunchKernelGGL((k_iter_tile_persist<D, NT1, NT2, ACT, L, QX, QY, NTX, NTY, WAVES>), dim3(blocks), dim3(WAVES * 64), bytes, s, a);
    else
#endif
    hipLaunchKernelGGL((k_iter_tile<D, NT1, NT2, ACT, L, QX, QY, NTX, NTY, WAVES>), dim3(blocks), dim3(WAVES * 64), bytes, s, a);
    return true;
}

template <int D, int NT1, int NT2, int ACT, int QX, int QY, int NTX, int NTY, int WAVES, int MAXL>
static bool launch_iter_tile_L(int L, const MfmaArgs& a, int blocks, hipStream_t s) {
    if (L == 2) return launch_iter_tile<D, NT1, NT2, ACT, 2, QX, QY, NTX, NTY, WAVES>(a, blocks, s);
    if (L == 3) return launch_iter_tile<D, NT1, NT2, ACT, 3, QX, QY, NTX, NTY, WAVES>(a, blocks, s);
    if constexpr (MAXL >= 4) {
        if (L == 4) return launch_iter_tile<D, NT1, NT2, ACT, 4, QX, QY, NTX, NTY, WAVES>(a, blocks, s);
    }
    return false;
}

// Whole training pass (forward, projection, reverse) of a shard of small elements in one launch.  Returns false when the
// element shape / channel set / layout is not covered; the caller then runs the separate kernels.
#ifdef HPV_EXPERIMENTS
#define TL_WHY(K) do { static const bool dbg_ = getenv("HPV_TILE_DEBUG") != nullptr; if (dbg_) fprintf(stderr, "hpv_mfma_iter_tile: not applicable (check %d)\n", K); } while (0)
#else
#define TL_WHY(K) do { } while (0)
#endif
bool hpv_mfma_iter_tile(HpvMfma* m, const double* theta, const double* X, double* GPART, int* rows, hipStream_t s,
                        const MfmaDataTerm* dt, const ProjArgs& pa, long n_elem, const MfmaFinalize* fin, bool* fin_done) {
    if (fin_done) *fin_done = false;
    const ProjDesc& pd = pa.pd;
    const NetDesc& nd = m->nd;
    if (!m->iter_fused_ok || pd.edge || n_elem <= 0 || m->L < 2 || m->L > 4 || m->H != MF_H) { TL_WHY(1); return false; }
    const int key = nd.d * 100 + nd.nT1 * 10 + nd.nT2;
    const bool shape1d = pd.qx == 80 && pd.qy == 1 && pd.ntx == 60 && pd.nty == 1 && nd.act == HPV_ACT_SIN && (key == 111 || key == 110);
    const bool shape2d = pd.qx == 10 && pd.qy == 10 && pd.ntx >= 1 && pd.ntx <= 5 && pd.nty >= 1 && pd.nty <= 5 && nd.act == HPV_ACT_TANH &&
                         (key == 200 || key == 220 || key == 221 || key == 222) && m->L <= 3;
    if (!shape1d && !shape2d) { TL_WHY(2); return false; }
    // (thousands of small 2-D elements: the separate launches stream, one workgroup per element does not -- scripts/grid_sweep.py)
    if (shape2d && n_elem > hpv_elem_resident_max(2, 10, m->n_cus) && !m->iter_fused_force) { TL_WHY(5); return false; }
    const int waves = shape1d ? 6 : 8, nq = pd.qx * pd.qy, tpe = (nq + 15) / 16;
    // batch layout [element points | pad to 16 | data points]
    const long npad = (n_elem * nq + 15) / 16 * 16;
    const bool has_data = dt && dt->n_data > 0;
    if (has_data ? (dt->data_off != npad || m->N != npad + dt->n_data) : (m->N != npad && m->N != n_elem * nq)) { TL_WHY(3); return false; }
    const long n_dt = has_data ? (dt->n_data + 15) / 16 : 0;
    const long left = n_dt - n_elem * (waves - tpe);
    const long blocks = n_elem + (left > 0 ? (left + waves - 1) / waves : 0);
    if (blocks > hpv_mfma_grad_rows(m) && blocks > m->max_rows) { TL_WHY(4); return false; }
    MfmaArgs a = m->base;
    a.theta = theta; a.X = X; a.GPART = GPART;
    a.OUT = const_cast<double*>(pa.OUT);
    a.data_off = -1;
    if (has_data) {
        a.data_off = dt->data_off; a.ud = dt->ud; a.gbar0 = dt->gbar0; a.data_part = dt->data_part;
        a.data_scale = dt->scale; a.data_write_gbar = dt->write_gbar;
    }
    a.proj_n_elem = n_elem;
    a.proj_split = 1;
    a.pa = pa;
#ifdef HPV_EXPERIMENTS
    const bool fin_off = getenv("HPV_NO_INKERNEL_FINALIZE") != nullptr;            // (A/B switch of libhpvpinn_testhooks.so, read per launch / capture)
#else
    constexpr bool fin_off = false;
#endif
    const bool fin_here = fin && blocks == 1 && n_elem == 1 && !fin_off;
    a.fin_mode = 0;
    a.persist_iters = 1;
    // persistent loop (k_iter_tile<.., PERSIST>): only where the kernel finishes the iteration itself AND applies the update
    // (opt-in, HPV_PERSIST=1: measured 24.6 us per iteration against 24.0 for one launch per iteration -- see tile_body)
#ifdef HPV_EXPERIMENTS
    const bool persist_off = !(getenv("HPV_PERSIST") && getenv("HPV_PERSIST")[0] == '1');
#else
    constexpr bool persist_off = true;
#endif
    if (fin_here && fin->ad.theta && fin->n_iters > 1 && shape1d && !persist_off) a.persist_iters = fin->n_iters;
    if (fin_here) {
        a.fin_mode = fin->ad.theta ? 2 : 1;
        a.fin_ad = fin->ad; a.fin_RB = fin->RB; a.fin_lossb_weight = fin->lossb_weight;
        a.fin_n_data = fin->n_data; a.fin_n_data_part = fin->n_data_part; a.fin_has_eps = fin->has_eps; a.fin_ncopies = fin->ncopies;
    }
    m->last_split = false;
    snprintf(m->variant, sizeof m->variant, "k_iter_tile<D=%d,NT1=%d,NT2=%d,%s,L=%d,%dx%d/%dx%d,waves=%d>%s%s", nd.d, nd.nT1, nd.nT2,
             nd.act == HPV_ACT_SIN ? "sin" : "tanh", m->L, pd.qx, pd.qy, pd.ntx, pd.nty, waves, fin_here ? " +finalize" : "",
             (fin_here && fin->ad.theta && fin->n_iters > 1 && shape1d && !persist_off) ? " persistent" : "");
    bool ok = false;
    if (shape1d) {
        if (key == 111) ok = launch_iter_tile_L<1, 1, 1, HPV_ACT_SIN, 80, 1, 60, 1, 6, 4>(m->L, a, (int)blocks, s);
        else ok = launch_iter_tile_L<1, 1, 0, HPV_ACT_SIN, 80, 1, 60, 1, 6, 4>(m->L, a, (int)blocks, s);
    } else {
        if (key == 221) ok = launch_iter_tile_L<2, 2, 1, HPV_ACT_TANH, 10, 10, 5, 5, 8, 3>(m->L, a, (int)blocks, s);
        else if (key == 222) ok = launch_iter_tile_L<2, 2, 2, HPV_ACT_TANH, 10, 10, 5, 5, 8, 3>(m->L, a, (int)blocks, s);
        else if (key == 200) ok = launch_iter_tile_L<2, 0, 0, HPV_ACT_TANH, 10, 10, 5, 5, 8, 3>(m->L, a, (int)blocks, s);
        else ok = launch_iter_tile_L<2, 2, 0, HPV_ACT_TANH, 10, 10, 5, 5, 8, 3>(m->L, a, (int)blocks, s);
    }
    if (ok && rows) *rows = (int)blocks;
    if (ok && fin_done) *fin_done = fin_here;
    if (ok && fin && fin->iters_done) *fin->iters_done = a.persist_iters;
    return ok;
}
