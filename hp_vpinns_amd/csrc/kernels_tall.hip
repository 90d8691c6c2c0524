// Element-resident whole-iteration kernel for FEW, TALL elements (BASELINE config 5: AdvDiff-Identification, 8 elements of
// 80 x 80 Gauss-Lobatto points = 400 16-point tiles each, 5 x 5 test functions, trainable epsilon; P3:108-187).
//
// The round-2 path of this shape was forward -> activation store (74 MB) -> row-split projection (two launches) -> reverse:
// ~148 MB of HBM traffic per iteration for a 0.9 MB problem.  Here S workgroups (S = 2^k, n_elem S <= CUs: 32 at config 5)
// share an element; workgroup (e, part) owns the tiles [400 part / S, 400 (part + 1) / S) -- 12 or 13 of them, 3..4 per wave,
// one wave per SIMD -- and runs the whole iteration for those points without touching HBM in between:
//   phase F  Taylor-mode forward of its tiles (channels u, u_x, u_t[, u_xx]); only s = tanh(z) of every hidden layer is kept,
//            in the hand-managed top AGPRs (hpv_fused_dev.h); the channel values of its points go to LDS.
//   phase P  the projection is LINEAR in the channels: the workgroup projects ITS points onto the 25 test-function pairs
//            (general TermDesc integrands, P3:161-174) and publishes the 25 partial sums as tagged granules (hpv_fused_dev.h,
//            xg_*: the data's arrival is its own notification), every partner gathers and adds the S x 25 partials in a fixed order
//            (bitwise identical everywhere), forms R = U - F and the element loss (P3:176-182), and evaluates the adjoint of
//            the channels and its share of d loss / d epsilon at its own points.
//   phase R  reverse pass of its tiles, tangent pre-activations (first AND second order) recomputed from s on the MFMA pipe.
// Per iteration the kernel reads the coordinates twice, the parameters, F, the S x 25 partial sums of its element, and writes
// the partial sums, R, loss / d-epsilon entries and one gradient row per workgroup.
//
// Lane layout, MFMA formulation and the 16 + 4 split of a 20-wide layer are those of kernels_mfma.hip; the reverse-pass algebra
// for second tangents is that of k_bwd_mfma / k_iter_tile (hand-derived third-order reverse pass).
#include <cstdio>
#include <cstdlib>
#include <type_traits>

#include "hpv_fused_dev.h"

#define TA_WAVES 4
#define TA_BLOCK (TA_WAVES * 64)
#define TA_MAXT 4          // tiles per wave (element tiles + at most one boundary/data tile)
#define TA_SLICES 10       // point slices of the partial projection: 25 outputs x 10 slices = 250 threads

// -DHPV_FZ_TIMING: phase durations per wave (shader cycles) into the adjoint channel buffer; scripts/fz_timing.py c5
#ifdef HPV_FZ_TIMING
#define TA_STAMP(I) ta_t[I] = clock64()
#else
#define TA_STAMP(I)
#endif

template <int L, int C, int QX, int QY, int NTX, int NTY, int MAXP>
struct TaLds {
    static constexpr int LH = L - 1;
    static constexpr int NR = NTX * NTY;
    static constexpr int WT = 0;                               // [LH][5][64]  forward A fragments
    static constexpr int BH = WT + LH * MF_KS * 64;            // [LH][5][64]  bias fragments
    static constexpr int WR = BH + LH * MF_KS * 64;            // [LH][5][16]
    static constexpr int WN = WR + LH * MF_KS * 16;            // [LH][5][64]  reverse A fragments
    static constexpr int WRB = WN + LH * MF_KS * 64;           // [LH][5][16]
    static constexpr int W1O = WRB + LH * MF_KS * 16;          // [4][5][64]   W1[0], W1[1], Wo, b1
    static constexpr int AX = W1O + 4 * MF_KS * 64;            // [HPV_MAXT][NTX][QX]  w_x phi^(dx_t)
    static constexpr int BY = AX + HPV_MAXT * NTX * QX;        // [HPV_MAXT][NTY][QY]  w_y phi^(dy_t)
    static constexpr int CH = BY + HPV_MAXT * NTY * QY;        // [C][MAXP]    channel values of this workgroup's points
    static constexpr int GB = CH + C * MAXP;                   // [C][MAXP]    their adjoints
    static constexpr int UP = GB + C * MAXP;                   // [TA_SLICES][NR] slice partials of the projection
    static constexpr int U = UP + TA_SLICES * NR;              // [NR]         residual of the element
    static constexpr int RED = U + NR;                         // [16]         scalars (last word: the barrier's verdict)
    static constexpr int TR = RED + 16;                        // per-wave transpose tiles of ALL channels | epilogue rows
    static constexpr int TR_WAVE = C * 2 * MF_TRB * MF_LD;
    // QT: the quarters' s behind everything else ([TA_WAVES][L*5][16]: value-slot lanes, compact) -- at the end, so that the offsets
    // of the regions above (and with them the whole-tile code's addressing) are those of the instantiation without quarters
    static constexpr int pq(int P) { return TR + (TA_WAVES * TR_WAVE > TA_WAVES * P ? TA_WAVES * TR_WAVE : TA_WAVES * P); }
    static constexpr int total(int P) { return pq(P) + TA_WAVES * L * MF_KS * 16; }
};

template <int NT1, int NT2, int L, int QX, int QY, int NTX, int NTY, bool QT>
__global__ void __launch_bounds__(TA_BLOCK, 1) k_iter_tall(MfmaArgs g) {
    constexpr int C = 1 + NT1 + NT2, NQ = QX * QY, TPE = NQ / 16, NR = NTX * NTY, LH = L - 1, NSV = L * MF_KS;
    static_assert(NQ % 16 == 0 && L >= 2 && NT2 <= NT1, "whole tiles per element; second tangents ride on first ones");
    static_assert(HPV_MAXT <= C, "the per-term integrands borrow the adjoint array");
    constexpr int MAXP = 16 * (TA_WAVES * TA_MAXT);            // points a workgroup can own (its element tiles)
    using M = TaLds<L, C, QX, QY, NTX, NTY, MAXP>;
    extern __shared__ __attribute__((aligned(16))) double lds[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int q = lane >> 4, pt = lane & 15;
    const int split = g.proj_split, lg = __builtin_ctz(split);
    // workgroup -> (element, part): consecutive workgroups go to DIFFERENT elements, so that the partners of an element sit on
    // few XCDs (block b is observed on XCD b % 8: with 8 elements all 32 partners of element e share XCD e) and their arrival
    // counter, partial sums and polls stay in one L2 -- a speed choice only, nothing depends on the placement
    const int n_el_grid = (int)g.proj_n_elem;
    const long e = (long)(blockIdx.x % n_el_grid);
    const int part = (int)(blockIdx.x / n_el_grid);
    const long wg_slot = e * split + part;                  // this workgroup's slot in the granule buffer / loss_e / deps_e
    const double* __restrict__ th = g.theta;
    const ProjArgs& pa = g.pa;
    const ProjDesc& pd = pa.pd;
#ifdef HPV_FZ_TIMING
    long long ta_t[10];
    const long long ta_wall = wall_clock64();
    TA_STAMP(0);
#endif

    // ---- stage the weight fragments: every global read before the first LDS store.  The projection tables are requested
    //      last (loads return in order) and parked in LDS only after the forward phase, which hides their round trip ----
    constexpr int NTABX = HPV_MAXT * NTX * QX, NTABY = HPV_MAXT * NTY * QY;
    constexpr int ITX = (NTABX + TA_BLOCK - 1) / TA_BLOCK, ITY = (NTABY + TA_BLOCK - 1) / TA_BLOCK;
    double vax[ITX], vby[ITY];
    {
        constexpr int N1 = 4 * MF_KS * 64, IT1 = (N1 + TA_BLOCK - 1) / TA_BLOCK;
        constexpr int ITW = (MF_KS * 64 + TA_BLOCK - 1) / TA_BLOCK;
        double vwt[LH][ITW], vbh[LH][ITW], vwn[LH][ITW], vwr[LH], vwrb[LH], v1[IT1];
#pragma unroll
        for (int i_ = 1; i_ < L; ++i_) {
            const int wo = g.woff[i_], bo_ = g.boff[i_];
#pragma unroll
            for (int it = 0; it < ITW; ++it) {
                const int f = it * TA_BLOCK + tid, fc = f < MF_KS * 64 ? f : 0;
                const int ln = fc & 63, s_ = fc >> 6;
                vwt[i_ - 1][it] = th[wo + (4 * s_ + (ln >> 4)) * MF_H + (ln & 15)];
                vbh[i_ - 1][it] = th[bo_ + 4 * s_ + (ln >> 4)];
                vwn[i_ - 1][it] = th[wo + (ln & 15) * MF_H + 4 * s_ + (ln >> 4)];
            }
            const int fr = tid < MF_KS * 16 ? tid : 0;
            const int a_ = fr & 3, q_ = (fr >> 2) & 3, s_ = fr >> 4;
            vwr[i_ - 1] = th[wo + (4 * s_ + q_) * MF_H + 16 + a_];
            vwrb[i_ - 1] = th[wo + (16 + a_) * MF_H + 4 * s_ + q_];
        }
        const int w0o = g.woff[0], wLo = g.woff[L], b0o = g.boff[0];
#pragma unroll
        for (int it = 0; it < IT1; ++it) {
            const int f = it * TA_BLOCK + tid, fc = f < N1 ? f : 0;
            const int ln = fc & 63, s_ = (fc >> 6) % MF_KS, c_ = fc / (64 * MF_KS);
            const int j = 4 * s_ + (ln >> 4);
            v1[it] = th[(c_ < 2 ? w0o + c_ * MF_H : (c_ == 2 ? wLo : b0o)) + j];
        }
#pragma unroll
        for (int it = 0; it < ITX; ++it) {
            const int f = it * TA_BLOCK + tid, fc = f < NTABX ? f : 0;
            const int t_ = fc / (NTX * QX), i_ = fc % (NTX * QX);
            vax[it] = t_ < pd.nterms ? pa.wtx[(long)pd.t[t_].dx * NTX * QX + i_] : 0.0;
        }
#pragma unroll
        for (int it = 0; it < ITY; ++it) {
            const int f = it * TA_BLOCK + tid, fc = f < NTABY ? f : 0;
            const int t_ = fc / (NTY * QY), i_ = fc % (NTY * QY);
            vby[it] = t_ < pd.nterms ? pa.wty[(long)pd.t[t_].dy * NTY * QY + i_] : 0.0;
        }
#pragma unroll
        for (int i_ = 1; i_ < L; ++i_) {
#pragma unroll
            for (int it = 0; it < ITW; ++it) {
                const int f = it * TA_BLOCK + tid;
                if (f < MF_KS * 64) {
                    lds[M::WT + (i_ - 1) * MF_KS * 64 + f] = vwt[i_ - 1][it];
                    lds[M::BH + (i_ - 1) * MF_KS * 64 + f] = vbh[i_ - 1][it];
                    lds[M::WN + (i_ - 1) * MF_KS * 64 + f] = vwn[i_ - 1][it];
                }
            }
            if (tid < MF_KS * 16) {
                lds[M::WR + (i_ - 1) * MF_KS * 16 + tid] = vwr[i_ - 1];
                lds[M::WRB + (i_ - 1) * MF_KS * 16 + tid] = vwrb[i_ - 1];
            }
        }
#pragma unroll
        for (int it = 0; it < IT1; ++it) { const int f = it * TA_BLOCK + tid; if (f < N1) lds[M::W1O + f] = v1[it]; }
    }
    const double bo = th[g.boff[L]];
    // has a barrier of an EARLIER launch of this handle failed?  A plain load (kernel boundaries make earlier launches' stores
    // visible), requested behind the staging loads -- an agent-scope load ahead of them held every weight load back behind its
    // own memory round trip (loads return in order) -- and consumed at the barrier
    const int xsticky = *g.xerr;
    const unsigned xtag = *g.xiter + 1u;          // this launch's exchange tag (hpv_fused_dev.h, xg_*)
    const double eps = pa.eps_ptr ? pa.eps_ptr[0] : 0.0;
    // per-term scalars of the element, requested now: coefficient x (epsilon if the term carries it)
    double cterm[HPV_MAXT];
#pragma unroll
    for (int t = 0; t < HPV_MAXT; ++t)
        cterm[t] = t < pd.nterms ? pa.coef[(long)t * pa.coef_stride + e] * (pd.t[t].eps_mult ? eps : 1.0) : 0.0;
    const double pF = (pa.F && tid < NR) ? pa.F[e * NR + tid] : 0.0;
    pj_lds_barrier();        // the weight fragments are in LDS; tables, sticky flag and the element's scalars stay in flight
    TA_STAMP(1);

    // ---- tile list of this wave: element tiles tbase + wv, + 4, ..; possibly one boundary/data tile behind the elements ----
    const int tbase = (part * TPE) >> lg, tend = ((part + 1) * TPE) >> lg;
    const int n_mine = tend - tbase;                                    // element tiles of this workgroup (<= 16)
    // Quarter-tile plan (g.tall_qt, decided by the host for the whole grid: every workgroup's tile count is 0 or 1 mod 4).
    // 400 tiles over 32 workgroups are 12 or 13 tiles per workgroup, i.e. 3 + 3 + 3 + 3 or 4 + 3 + 3 + 3 per wave -- and the
    // boundary / data tiles made a 4-tile wave in the others: every launch lasted four tile-times for 3.2 of work.  Here every
    // wave owns n_mine / 4 whole tiles, and the workgroup's one extra tile -- its 13th element tile, or (12-tile workgroups)
    // a boundary / data tile -- is cut in four: a wave takes four of its points as ONE packed operand whose 16 point slots are
    // the C channels x 4 points (kernels_fused.hip, QT: layer products, hbar chain, tangent recompute and dW products once per
    // packed operand, the channels of a point coupled through DPP row shifts in the element-wise steps).
    constexpr bool qt = QT;                 // (the host launches the QT instantiation when its plan holds for the whole grid)
    const int nwq = n_mine / TA_WAVES;
    // one data tile per workgroup, as far as they go -- QT: per workgroup WITHOUT a 13th element tile
    auto dtile_of = [&]() -> long {
        if constexpr (QT) {
            int c0 = 0;
            for (int p_ = 0; p_ < part; ++p_) c0 += ((((p_ + 1) * TPE) >> lg) - ((p_ * TPE) >> lg)) % TA_WAVES == 0 ? 1 : 0;
            return n_mine % TA_WAVES == 0 ? g.proj_n_elem * TPE + (long)c0 * n_el_grid + e : g.ntiles;
        } else {
            return g.proj_n_elem * TPE + blockIdx.x;
        }
    };
    const long dtile = dtile_of();
    const int q_kind = !qt ? 0 : (n_mine % TA_WAVES == 1 ? 1 : (dtile < g.ntiles ? 2 : 0));     // 1 element quarter, 2 data quarter
    const int n_el = qt ? nwq : ((n_mine - wv + TA_WAVES - 1 > 0) ? (n_mine - wv + TA_WAVES - 1) / TA_WAVES : 0);
    (void)wg_slot;
    const bool has_d = !qt && (wv == n_mine % TA_WAVES) && dtile < g.ntiles && n_el < TA_MAXT;
    const int n_own = n_el + (has_d ? 1 : 0);
    auto tile_of = [&](int k) -> long { return k < n_el ? e * TPE + tbase + wv + (long)k * TA_WAVES : dtile; };
    auto lp_of = [&](int k) -> int { return (wv + k * TA_WAVES) * 16 + pt; };      // point index inside this workgroup's range

    // (QT: at most TA_MAXT - 1 whole tiles per wave, the quarter's s lives in LDS: one stash slot less for the compiler's benefit)
    constexpr int NSLOT = QT ? TA_MAXT - 1 : TA_MAXT;
    constexpr int ABASE = 256 - NSLOT * 2 * NSV;
    asm volatile("" ::: "a255");       // the kernel owns all 256 AGPRs; a[ABASE..255] are hand-managed (scripts/check_agpr.py)
    double gdat = 0.0;

    const double* WTl = lds + M::WT;
    const double* WRl = lds + M::WR;
    const double* BHl = lds + M::BH;

    // =============================================================================================
    // phase F: forward
    // =============================================================================================
#pragma unroll 1
    for (int k = 0; k < n_own; ++k) {
        const long tile = tile_of(k);
        const long p = tile * 16 + pt;
        const bool valid = p < g.N;
        const long pc = valid ? p : g.N - 1;
        const double x0 = valid ? g.X[pc] : 0.0, x1 = valid ? g.X[g.N + pc] : 0.0;
        int lofs = lane;
        asm volatile("" : "+v"(lofs));       // opaque: the LDS fragment reads stay inside the loop
        double h[C][MF_KS], sv[NSV];
        // layer 1 (VALU)
#pragma unroll
        for (int s = 0; s < MF_KS; ++s) {
            const double w0 = lds[M::W1O + (0 * MF_KS + s) * 64 + lofs], w1 = lds[M::W1O + (1 * MF_KS + s) * 64 + lofs];
            const double z = lds[M::W1O + (3 * MF_KS + s) * 64 + lofs] + x0 * w0 + x1 * w1;
            double a, a1, a2;
            act_fwd<HPV_ACT_TANH>(z, a, a1, a2);
            sv[s] = a;
            h[0][s] = a;
#pragma unroll
            for (int u = 0; u < NT1; ++u) h[1 + u][s] = a1 * (u == 0 ? w0 : w1);
#pragma unroll
            for (int b = 0; b < NT2; ++b) { const double wb = b == 0 ? w0 : w1; h[1 + NT1 + b][s] = a2 * wb * wb; }
        }
#pragma unroll
        for (int i = 1; i < L; ++i) {
            double z[C][MF_KS];
            fz_layer<true>(WTl + (i - 1) * MF_KS * 64, WRl + (i - 1) * MF_KS * 16, BHl + (i - 1) * MF_KS * 64, lofs, h[0], z[0]);
#pragma unroll
            for (int ch = 1; ch < C; ++ch)
                fz_layer<false>(WTl + (i - 1) * MF_KS * 64, WRl + (i - 1) * MF_KS * 16, nullptr, lofs, h[ch], z[ch]);
#pragma unroll
            for (int s = 0; s < MF_KS; ++s) {
                double a, a1, a2;
                act_fwd<HPV_ACT_TANH>(z[0][s], a, a1, a2);
                sv[i * MF_KS + s] = a;
                h[0][s] = a;
#pragma unroll
                for (int u = 0; u < NT1; ++u) h[1 + u][s] = a1 * z[1 + u][s];
#pragma unroll
                for (int b = 0; b < NT2; ++b) h[1 + NT1 + b][s] = a2 * z[1 + b][s] * z[1 + b][s] + a1 * z[1 + NT1 + b][s];
            }
        }
        // linear head
        double o[C];
#pragma unroll
        for (int ch = 0; ch < C; ++ch) {
            double v = 0.0;
#pragma unroll
            for (int s = 0; s < MF_KS; ++s) v += h[ch][s] * lds[M::W1O + (2 * MF_KS + s) * 64 + lofs];
            v = xrow_sum16(v);
            v = xrow_sum32(v);
            o[ch] = v;
        }
        o[0] += bo;
        if (k < n_el) {
            if (q == 0) {
                const int lp = lp_of(k);
#pragma unroll
                for (int ch = 0; ch < C; ++ch) lds[M::CH + ch * MAXP + lp] = o[ch];
            }
        } else {
            // lossb = w mean((u_d - u)^2) (P3:184): adjoint of u kept in a register, per-tile partial sum to memory
            const double dd = valid ? g.ud[p - g.data_off] - o[0] : 0.0;
            gdat = g.data_scale * dd;
            const double sq = row_sum16(dd * dd);
            if (lane == 0) g.data_part[p / 16 - g.data_off / 16] = sq;
        }
        switch (k) {     // wave-uniform; the stash slot must be a compile-time register index
#define TA_STASH(K) case K: if constexpr (K < NSLOT) acc_put_all<ABASE + K * 2 * NSV, NSV>(sv); break;
            TA_STASH(0) TA_STASH(1) TA_STASH(2) TA_STASH(3)
#undef TA_STASH
        }
    }
    // ---- QT: this wave's packed quarter of the workgroup's extra tile (slot c = pt >> 2 = channel, j = pt & 3 = point) ----
    const int qcs = pt >> 2, qj = pt & 3;
    const bool q_val = qcs == 0, q_first = qcs >= 1 && qcs <= NT1, q_second = qcs > NT1 && qcs < C;
    const int q_lp = (TA_WAVES * nwq) * 16 + 4 * wv + qj;                     // element quarter: the point inside this workgroup's range
    const long q_p = q_kind == 1 ? (e * TPE + tbase + TA_WAVES * nwq) * 16 + 4 * wv + qj : (q_kind == 2 ? dtile * 16 + 4 * wv + qj : 0);
    const bool q_valid = q_kind != 0 && q_p < g.N;
    double gdat_q = 0.0;
    if (QT && q_kind != 0) {               // (wave- and workgroup-uniform)
        const long pc = q_valid ? q_p : 0;
        const double x0 = g.X[pc], x1 = g.X[g.N + pc];
        const double udq = (q_kind == 2 && q_valid) ? g.ud[pc - g.data_off] : 0.0;
        int lofs = lane;
        asm volatile("" : "+v"(lofs));
        // per-lane 0 / 1 masks of the slot kinds and slot indices: the element-wise steps BLEND the candidates arithmetically (exactly
        // one mask is 1, so the blend is exact) -- chains of per-lane selects on the slot index become branches otherwise
        const double mv = q_val ? 1.0 : 0.0, mf = q_first ? 1.0 : 0.0, ms = q_second ? 1.0 : 0.0;
        const double m1 = qcs == 1 ? 1.0 : 0.0, m2 = qcs == 2 ? 1.0 : 0.0, m3 = qcs == 3 ? 1.0 : 0.0;
        double H[MF_KS], AAq[NSV];
#pragma unroll
        for (int s = 0; s < MF_KS; ++s) {
            const double w0 = lds[M::W1O + (0 * MF_KS + s) * 64 + lofs], w1 = lds[M::W1O + (1 * MF_KS + s) * 64 + lofs];
            const double z = lds[M::W1O + (3 * MF_KS + s) * 64 + lofs] + x0 * w0 + x1 * w1;
            double a, a1, a2;
            act_fwd<HPV_ACT_TANH>(z, a, a1, a2);
            AAq[s] = a;
            const double wf = qcs == 1 ? w0 : w1;                              // first tangent u = qcs - 1
            const double wb = (qcs - 1 - NT1) == 0 ? w0 : w1;                  // second tangent b = qcs - 1 - NT1
            H[s] = mv * a + (mf * a1) * wf + (ms * a2) * (wb * wb);
        }
#pragma unroll
        for (int i = 1; i < L; ++i) {
            double Z[MF_KS];
            fz_layer_m(WTl + (i - 1) * MF_KS * 64, WRl + (i - 1) * MF_KS * 16, BHl + (i - 1) * MF_KS * 64, lofs, q_val ? 1.0 : 0.0, H, Z);
#pragma unroll
            for (int s = 0; s < MF_KS; ++s) {
                double a, a1, a2;
                act_fwd<HPV_ACT_TANH>(Z[s], a, a1, a2);                    // (tangent slots: of a tangent pre-activation, not used)
                const double a4 = dpp_move<0x114>(a), a8 = dpp_move<0x118>(a), a12 = dpp_move<0x11C>(a);   // row_shr: the value slot of my point
                const double ab = (mv * a + m1 * a4) + (m2 * a8 + m3 * a12);
                const double zx = dpp_move<0x110 + 4 * NT1>(Z[s]);        // second-tangent slots: z_c of their first tangent
                const double b1 = 1.0 - ab * ab, b2 = -2.0 * ab * b1;
                AAq[i * MF_KS + s] = ab;
                H[s] = mv * ab + ((mf + ms) * b1) * Z[s] + (ms * b2) * (zx * zx);
            }
        }
        double v = 0.0;
#pragma unroll
        for (int s = 0; s < MF_KS; ++s) v += H[s] * lds[M::W1O + (2 * MF_KS + s) * 64 + lofs];
        v = xrow_sum16(v);
        v = xrow_sum32(v);
        if (q_val) v += bo;
        if (q_kind == 1) {
            if (q == 0 && qcs < C) lds[M::CH + qcs * MAXP + q_lp] = v;
        } else {
            // lossb partial of the boundary / data tile (P3:184): the four waves' quarters meet in LDS
            const double dd = (q_valid && q_val) ? udq - v : 0.0;
            gdat_q = g.data_scale * dd;
            const double sq = row_sum16(q == 0 ? dd * dd : 0.0);
            if (lane == 0) lds[M::RED + 8 + wv] = sq;
        }
        if (q_val) {       // the quarter's s: value-slot lanes only (the tangent slots read their point's back)
#pragma unroll
            for (int j = 0; j < NSV; ++j) lds[M::pq(g.P) + (wv * NSV + j) * 16 + q * 4 + qj] = AAq[j];
        }
    }
#pragma unroll
    for (int it = 0; it < ITX; ++it) { const int f = it * TA_BLOCK + tid; if (f < NTABX) lds[M::AX + f] = vax[it]; }
#pragma unroll
    for (int it = 0; it < ITY; ++it) { const int f = it * TA_BLOCK + tid; if (f < NTABY) lds[M::BY + f] = vby[it]; }
    TA_STAMP(2);
    __syncthreads();
    TA_STAMP(3);
    if (q_kind == 2 && tid == 0)
        g.data_part[dtile - g.data_off / 16] = (lds[M::RED + 8] + lds[M::RED + 9]) + (lds[M::RED + 10] + lds[M::RED + 11]);

    // =============================================================================================
    // phase P: partial projection of this workgroup's points, exchange, residual, adjoint at its points
    // =============================================================================================
    // parking place of tile 0's recomputed tangent pre-activations: the transpose region (idle until the reverse pass) behind the
    // gathered partial sums; the four waves' blocks straddle the per-wave transpose tiles, so everybody reads its block back BEFORE
    // the barrier that ends phase P
    // (the quarter-tile instantiations only: the other ones sit too close to the hand-managed AGPR range for a second copy of the
    //  tile body; -DHPV_TALL_NO_EARLY: A/B)
#ifdef HPV_TALL_NO_EARLY
    constexpr bool EARLY0 = false;
#else
    constexpr bool EARLY0 = QT;
#endif
    constexpr int NPARK = (L - 1) * (NT1 + NT2) * MF_KS;
    static_assert(64 * NR + TA_WAVES * NPARK * 64 <= TA_WAVES * M::TR_WAVE, "gathered sums + parked tangents fit the transpose region");
    double* PARK = lds + M::TR + 64 * NR + wv * (NPARK * 64) + lane;     // (behind the gathered sums: at most 64 partners x NR)
    // s of tile k back from the stash and its tangent pre-activations recomputed on the matrix pipe.  Tile 0's are produced BEFORE
    // the gather of the exchange (they need nothing of the projection): the wait for the partners' partial sums (10 k cycles per
    // launch) covers them
    auto fetch_sv = [&](int k, double (&sv)[NSV]) {
        switch (k) {
#define TA_FETCH(K) case K: if constexpr (K < NSLOT) acc_get_all<ABASE + K * 2 * NSV, NSV>(sv); break;
            TA_FETCH(0) TA_FETCH(1) TA_FETCH(2) TA_FETCH(3)
#undef TA_FETCH
            default:
#pragma unroll
                for (int j = 0; j < NSV; ++j) sv[j] = 0.0;
        }
    };
    auto tangents = [&](int lofs, bool recompute, const double (&sv)[NSV], double (&zc)[L][NT1 > 0 ? NT1 : 1][MF_KS], double (&zcc)[L][NT2 > 0 ? NT2 : 1][MF_KS]) {
        // tangent pre-activations of every hidden layer, recomputed: layer 0 has z_c = W1[c,:], z_cc = 0;
        // layer i: z_c = (s' z_c)_{i-1} W_i, z_cc = (s'' z_c^2 + s' z_cc)_{i-1} W_i
#pragma unroll
        for (int s = 0; s < MF_KS; ++s) {
#pragma unroll
            for (int u = 0; u < NT1; ++u) zc[0][u][s] = lds[M::W1O + (u * MF_KS + s) * 64 + lofs];
#pragma unroll
            for (int b = 0; b < NT2; ++b) zcc[0][b][s] = 0.0;
        }
        if (!recompute) return;      // (wave-uniform) tile 0: its tangents were read back from the parking place before the loop
#pragma unroll
        for (int i = 1; i < L; ++i) {
            double hx[NT1 > 0 ? NT1 : 1][MF_KS], hcc[NT2 > 0 ? NT2 : 1][MF_KS];
#pragma unroll
            for (int s = 0; s < MF_KS; ++s) {
                const double a = sv[(i - 1) * MF_KS + s], a1 = 1.0 - a * a, a2 = -2.0 * a * a1;
#pragma unroll
                for (int u = 0; u < NT1; ++u) hx[u][s] = a1 * zc[i - 1][u][s];
#pragma unroll
                for (int b = 0; b < NT2; ++b)
                    hcc[b][s] = a2 * zc[i - 1][b][s] * zc[i - 1][b][s] + (i > 1 ? a1 * zcc[i - 1][b][s] : 0.0);
            }
#pragma unroll
            for (int u = 0; u < NT1; ++u)
                fz_layer<false>(WTl + (i - 1) * MF_KS * 64, WRl + (i - 1) * MF_KS * 16, nullptr, lofs, hx[u], zc[i][u]);
#pragma unroll
            for (int b = 0; b < NT2; ++b)
                fz_layer<false>(WTl + (i - 1) * MF_KS * 64, WRl + (i - 1) * MF_KS * 16, nullptr, lofs, hcc[b], zcc[i][b]);
        }
    };
    double zc[L][NT1 > 0 ? NT1 : 1][MF_KS], zcc[L][NT2 > 0 ? NT2 : 1][MF_KS];       // tile 0's, between the parking place and the reverse pass
    const int np = n_mine * 16;                       // this workgroup's points: qe = 16 tbase + lp, lp < np
    const long qe0 = 16L * tbase;
    {
        // Upart[k][r] = sum_t c_t sum_lp BY_t[k][j] AX_t[r][i] G_t[lp],  G_t = sum_ch (a0 + eps a1)[ch] CH[ch][lp]:
        // the integrands once per point (they overwrite the adjoint array, which is not live yet), then 25 outputs x 10 slices
        for (int lp = tid; lp < np; lp += TA_BLOCK) {
            double ov[C];
#pragma unroll
            for (int ch = 0; ch < C; ++ch) ov[ch] = lds[M::CH + ch * MAXP + lp];
#pragma unroll
            for (int t = 0; t < HPV_MAXT; ++t) {
                double gq = 0.0;
#pragma unroll
                for (int ch = 0; ch < C; ++ch) gq = fma(t < pd.nterms ? pd.t[t].a0[ch] + eps * pd.t[t].a1[ch] : 0.0, ov[ch], gq);
                lds[M::GB + t * MAXP + lp] = cterm[t] * gq;
            }
        }
        __syncthreads();
        if constexpr (QT) {
            // sum-factorised over the rows of the quadrature grid this workgroup's points touch (at most NRW: its <= 15 tiles are a
            // contiguous run of points of a QX-wide grid): T_t[jj][r] = sum_i AX_t[r][i] G_t[(j0 + jj, i)] by a lane quad per output
            // (the points outside the workgroup's range masked), then U[k][r] = sum_t sum_jj BY_t[k][j0 + jj] T_t[jj][r] -- a tenth of
            // the LDS reads of the point-by-point version below, and no slice partials
            constexpr int NRW = (MAXP + QX - 2) / QX + 1;
            static_assert(HPV_MAXT * NRW * NTX * 4 <= TA_BLOCK && HPV_MAXT * NRW * NTX <= TA_SLICES * NR && QX % 4 == 0, "row partials fit the block and the slice scratch");
            const int j0 = (int)qe0 / QX;
            {
                const int o = tid >> 2, part4 = tid & 3;
                const bool ok = o < HPV_MAXT * NRW * NTX;
                const int oc = ok ? o : 0;
                const int t = oc / (NRW * NTX), jj = (oc / NTX) % NRW, r = oc % NTX;
                const int lp0 = (j0 + jj) * QX - (int)qe0;          // lp of the row's first point (may lie outside the range)
                double acc = 0.0;
#pragma unroll
                for (int it = 0; it < QX / 4; ++it) {
                    const int i = part4 + 4 * it, lp = lp0 + i;
                    const bool in = lp >= 0 && lp < np;
                    const double av = lds[M::AX + (t * NTX + r) * QX + i] * (in ? 1.0 : 0.0);
                    acc = fma(av, lds[M::GB + t * MAXP + (in ? lp : 0)], acc);
                }
                acc = pj_group_sum<4>(acc);
                if (ok && part4 == 0) lds[M::UP + oc] = acc;
            }
            __syncthreads();
            if (tid < NR) {
                const int kk = tid / NTX, r = tid % NTX;
                double u = 0.0;
#pragma unroll
                for (int t = 0; t < HPV_MAXT; ++t)
#pragma unroll
                    for (int jj = 0; jj < NRW; ++jj) {
                        const int j = j0 + jj < QY ? j0 + jj : QY - 1;        // (rows past the grid: their T is zero)
                        u = fma(lds[M::BY + (t * NTY + kk) * QY + j], lds[M::UP + (t * NRW + jj) * NTX + r], u);
                    }
                if (!xsticky && !HPV_XDEBUG_SKIP(g, xtag, e, part)) xg_publish(g.xg + (wg_slot * NR + tid) * 2, u, xtag);
            }
        } else {
        if (tid < NR * TA_SLICES) {
            const int o = tid % NR, sl = tid / NR, kk = o / NTX, r = o % NTX;
            double acc = 0.0;
            constexpr int NITP = (MAXP - 16 + TA_SLICES - 1) / TA_SLICES;      // (a workgroup owns at most MAXP / 16 - 1 element tiles)
#pragma unroll
            for (int it = 0; it < NITP; ++it) {       // fixed trip count: the LDS reads of all iterations are in flight together
                const int lp = sl + it * TA_SLICES, lpc = lp < np ? lp : 0;
                const int qe = (int)qe0 + lpc, i = qe % QX, j = qe / QX;
#pragma unroll
                for (int t = 0; t < HPV_MAXT; ++t) {
                    const double w = lds[M::AX + (t * NTX + r) * QX + i] * lds[M::BY + (t * NTY + kk) * QY + j];
                    acc = fma(lp < np ? w : 0.0, lds[M::GB + t * MAXP + lpc], acc);
                }
            }
            lds[M::UP + sl * NR + o] = acc;
        }
        __syncthreads();
        if (tid < NR) {
            double u = 0.0;
#pragma unroll
            for (int sl = 0; sl < TA_SLICES; ++sl) u += lds[M::UP + sl * NR + tid];
            // publish: two tagged granules per value, fire and forget (the partners poll the granules themselves)
            if (!xsticky && !HPV_XDEBUG_SKIP(g, xtag, e, part)) xg_publish(g.xg + (wg_slot * NR + tid) * 2, u, xtag);
        }
        }
        TA_STAMP(4);
        if (EARLY0 && n_own > 0) {         // (wave-uniform)
            double sv_[NSV], zc_[L][NT1 > 0 ? NT1 : 1][MF_KS], zcc_[L][NT2 > 0 ? NT2 : 1][MF_KS];
            int lofs = lane;
            asm volatile("" : "+v"(lofs));
            fetch_sv(0, sv_);
            tangents(lofs, true, sv_, zc_, zcc_);
#pragma unroll
            for (int i = 1; i < L; ++i)
#pragma unroll
                for (int s = 0; s < MF_KS; ++s) {
#pragma unroll
                    for (int u = 0; u < NT1; ++u) PARK[(((i - 1) * (NT1 + NT2) + u) * MF_KS + s) * 64] = zc_[i][u][s];
#pragma unroll
                    for (int b = 0; b < NT2; ++b) PARK[(((i - 1) * (NT1 + NT2) + NT1 + b) * MF_KS + s) * 64] = zcc_[i][b][s];
                }
        }
        {
            // the S x NR partial sums of the element, straight into LDS (the transpose region is idle between the phases)
            constexpr int NITG = (64 * NR * 2 + TA_BLOCK - 1) / TA_BLOCK;
            const bool stay_away = xsticky || HPV_XDEBUG_SKIP(g, xtag, e, part);     // workgroup-uniform
            bool ok = true;
            if (!stay_away) ok = xg_gather<NITG, TA_BLOCK>(g.xg + (long)e * split * NR * 2, split * NR * 2, xtag, (unsigned*)(lds + M::TR), tid);
            const int timed_out = __syncthreads_or(ok ? 0 : 1);       // (also the barrier that makes the gathered sums visible)
            if (timed_out || stay_away) {
                // a partner did not show up (or an earlier launch failed: sticky): nothing of this iteration has been written, the
                // kernels that follow skip the update (kernels_generic.hip), the host reports -7 and resets (hpv_api.hip, sync_check)
                if (timed_out && tid == 0) __hip_atomic_store(g.xerr, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                return;      // (the launch counter is advanced by the kernel that FOLLOWS this launch: k_finalize, hpv_fused_dev.h)
            }
        }
        TA_STAMP(5);
        double sq = 0.0;
        if (tid < NR) {
            double u = -pF;
            for (int c = 0; c < split; ++c) u += lds[M::TR + c * NR + tid];   // fixed order: every partner computes the same bits
            lds[M::U + tid] = u;
            if (part == 0) pa.R[e * NR + tid] = u;
            sq = u * u;
        }
        if (wv == 0) {
            sq = pj_wave_sum(sq);
            if (lane == 0) pa.loss_e[wg_slot] = part == 0 ? sq / (double)NR : 0.0;
        }
        __syncthreads();
        // adjoint at this workgroup's points: gh_t[lp] = c_t (2/NR) sum_kr BY_t[k][j] AX_t[r][i] U[k][r];
        // Gbar[ch][lp] = sum_t (a0 + eps a1)[ch] gh_t;  d loss / d eps += gh_t/c_t-free parts (see k_project_rows_adj)
        const double sc = 2.0 / (double)NR;
        double deps = 0.0;
        for (int lp = tid; lp < np; lp += TA_BLOCK) {
            const int qe = (int)qe0 + lp, i = qe % QX, j = qe / QX;
            double gb[C], ov[C];
#pragma unroll
            for (int ch = 0; ch < C; ++ch) { gb[ch] = 0.0; ov[ch] = lds[M::CH + ch * MAXP + lp]; }
#pragma unroll
            for (int t = 0; t < HPV_MAXT; ++t) {
                if (t >= pd.nterms) break;
                const double* ax = lds + M::AX + t * NTX * QX + i;
                const double* by = lds + M::BY + t * NTY * QY + j;
                double gh = 0.0;
#pragma unroll
                for (int kk = 0; kk < NTY; ++kk) {
                    double sr = 0.0;
#pragma unroll
                    for (int r = 0; r < NTX; ++r) sr = fma(ax[r * QX], lds[M::U + kk * NTX + r], sr);
                    gh = fma(by[kk * QY], sr, gh);
                }
                gh *= sc * pa.coef[(long)t * pa.coef_stride + e];
                const double m = pd.t[t].eps_mult ? eps : 1.0;
                double g1 = 0.0, gt = 0.0;
#pragma unroll
                for (int ch = 0; ch < C; ++ch) {
                    const double al = pd.t[t].a0[ch] + eps * pd.t[t].a1[ch];
                    gb[ch] = fma(al, m * gh, gb[ch]);
                    g1 = fma(pd.t[t].a1[ch], ov[ch], g1);
                    gt = fma(al, ov[ch], gt);
                }
                deps = fma(gh, m * g1 + (pd.t[t].eps_mult ? gt : 0.0), deps);
            }
#pragma unroll
            for (int ch = 0; ch < C; ++ch) lds[M::GB + ch * MAXP + lp] = gb[ch];
        }
        if (pd.has_eps) {
            deps = pj_wave_sum(deps);
            if (lane == 0) lds[M::RED + wv] = deps;
        }
        if (EARLY0 && n_own > 0) {         // tile 0's tangents back from the parking place (before the barrier: the reverse pass reuses the region)
#pragma unroll
            for (int i = 1; i < L; ++i)
#pragma unroll
                for (int s = 0; s < MF_KS; ++s) {
#pragma unroll
                    for (int u = 0; u < NT1; ++u) zc[i][u][s] = PARK[(((i - 1) * (NT1 + NT2) + u) * MF_KS + s) * 64];
#pragma unroll
                    for (int b = 0; b < NT2; ++b) zcc[i][b][s] = PARK[(((i - 1) * (NT1 + NT2) + NT1 + b) * MF_KS + s) * 64];
                }
        }
        __syncthreads();
        if (pd.has_eps && tid == 0)
            pa.deps_e[wg_slot] = (lds[M::RED] + lds[M::RED + 1]) + (lds[M::RED + 2] + lds[M::RED + 3]);
    }

    // =============================================================================================
    // phase R: reverse pass (tangent pre-activations recomputed from s)
    // =============================================================================================
    TA_STAMP(6);
    double* TAB = lds + M::TR + wv * M::TR_WAVE;
    v4d dWacc[LH];
    double dS10[LH], dS01[LH], accC[LH];
#pragma unroll
    for (int i = 0; i < LH; ++i) { dWacc[i] = v4d{0.0, 0.0, 0.0, 0.0}; dS10[i] = 0.0; dS01[i] = 0.0; accC[i] = 0.0; }
    double db[L][MF_KS], dW1[2][MF_KS], dWo[MF_KS], dbo = 0.0;
#pragma unroll
    for (int s = 0; s < MF_KS; ++s) {
        dWo[s] = 0.0; dW1[0][s] = 0.0; dW1[1][s] = 0.0;
#pragma unroll
        for (int i = 0; i < L; ++i) db[i][s] = 0.0;
    }

    // one reverse tile; EARLY_ (tile 0): zc0 / zcc0 hold its tangents of the layers >= 2, read back from the parking place
    auto rev_tile = [&](int k, auto EARLY_, [[maybe_unused]] const double (&zc0)[L][NT1 > 0 ? NT1 : 1][MF_KS],
                        [[maybe_unused]] const double (&zcc0)[L][NT2 > 0 ? NT2 : 1][MF_KS]) {
        constexpr bool EARLY = decltype(EARLY_)::value;
        const long tile = tile_of(k);
        const long p = tile * 16 + pt;
        const bool valid = p < g.N;
        const double x0 = valid ? g.X[p] : 0.0, x1 = valid ? g.X[g.N + p] : 0.0;
        int lofs = lane;
        asm volatile("" : "+v"(lofs));
        double sv[NSV], zc[L][NT1 > 0 ? NT1 : 1][MF_KS], zcc[L][NT2 > 0 ? NT2 : 1][MF_KS];
        fetch_sv(k, sv);
        double gb[C];
        if (k < n_el) {
            const int lp = lp_of(k);
#pragma unroll
            for (int ch = 0; ch < C; ++ch) gb[ch] = lds[M::GB + ch * MAXP + lp];
        } else {
#pragma unroll
            for (int ch = 0; ch < C; ++ch) gb[ch] = 0.0;
            gb[0] = gdat;
        }
        tangents(lofs, !EARLY, sv, zc, zcc);
        if constexpr (EARLY) {
#pragma unroll
            for (int i = 1; i < L; ++i)
#pragma unroll
                for (int s = 0; s < MF_KS; ++s) {
#pragma unroll
                    for (int u = 0; u < NT1; ++u) zc[i][u][s] = zc0[i][u][s];
#pragma unroll
                    for (int b = 0; b < NT2; ++b) zcc[i][b][s] = zcc0[i][b][s];
                }
        }
        // channel ch of layer i's outputs (compile-time i, ch after unrolling)
        auto hv_of = [&](int i, int ch, int s) -> double {
            const double a = sv[i * MF_KS + s], a1 = 1.0 - a * a;
            if (ch == 0) return a;
            if (ch <= NT1) return a1 * zc[i][ch - 1][s];
            const int b = ch - 1 - NT1;
            const double z1 = zc[i][b][s];
            return -2.0 * a * a1 * z1 * z1 + (i == 0 ? 0.0 : a1 * zcc[i][b][s]);
        };

        double hbar[C][MF_KS], zbar[C][MF_KS];
        // ---- linear head ----
#pragma unroll
        for (int s = 0; s < MF_KS; ++s) {
            const double wo = lds[M::W1O + (2 * MF_KS + s) * 64 + lofs];
#pragma unroll
            for (int ch = 0; ch < C; ++ch) {
                dWo[s] = fma(hv_of(L - 1, ch, s), gb[ch], dWo[s]);
                hbar[ch][s] = gb[ch] * wo;
            }
        }
        if (q == 0) dbo += gb[0];

        // ---- hidden layers, last to first ----
#pragma unroll
        for (int i = L - 1; i >= 0; --i) {
#pragma unroll
            for (int s = 0; s < MF_KS; ++s) {
                const double a = sv[i * MF_KS + s];
                const double a1 = 1.0 - a * a, a2 = -2.0 * a * a1, a3 = -2.0 * a1 * (1.0 - 3.0 * a * a);
                double zb = hbar[0][s] * a1;
#pragma unroll
                for (int u = 0; u < NT1; ++u) {
                    zbar[1 + u][s] = hbar[1 + u][s] * a1;
                    zb = fma(hbar[1 + u][s] * a2, zc[i][u][s], zb);
                }
#pragma unroll
                for (int b = 0; b < NT2; ++b) {
                    const double hb = hbar[1 + NT1 + b][s], z1 = zc[i][b][s];
                    zbar[1 + NT1 + b][s] = hb * a1;
                    zbar[1 + b][s] = fma(2.0 * hb * a2, z1, zbar[1 + b][s]);
                    zb = fma(hb, a3 * z1 * z1 + (i == 0 ? 0.0 : a2 * zcc[i][b][s]), zb);
                }
                zbar[0][s] = zb;
                db[i][s] += zb;
            }
            if (i == 0) {
#pragma unroll
                for (int s = 0; s < MF_KS; ++s) {
                    dW1[0][s] += x0 * zbar[0][s] + zbar[1][s];
                    dW1[1][s] += x1 * zbar[0][s] + (NT1 > 1 ? zbar[NT1 > 1 ? 2 : 1][s] : 0.0);
                }
            } else {
                // weight gradient dW_i[in][out] = sum_pt sum_ch h_{i-1,ch}[pt][in] zbar_ch[pt][out]: operands point-major ->
                // per-wave LDS transpose tiles, all channels written first (one wave-level sync)
                pj_wave_sync();
#pragma unroll
                for (int ch = 0; ch < C; ++ch) {
                    double* TA = TAB + (2 * ch) * (MF_TRB * MF_LD);
                    double* TB = TA + MF_TRB * MF_LD;
#pragma unroll
                    for (int s = 0; s < MF_KS; ++s) {
                        TA[(4 * s + q) * MF_LD + pt] = hv_of(i - 1, ch, s);
                        TB[(4 * s + q) * MF_LD + pt] = zbar[ch][s];
                    }
                }
                // hbar_{i-1}^T = W_i zbar^T  (independent of the transposes: issued while the LDS writes above land)
#pragma unroll
                for (int ch = 0; ch < C; ++ch) {
                    v4d acc = v4d{0.0, 0.0, 0.0, 0.0};
                    double h4 = 0.0;
                    const double* wrl = lds + M::WRB + (i - 1) * MF_KS * 16 + q * 4 + (lane & 3);
#pragma unroll
                    for (int s = 0; s < MF_KS; ++s) {
                        acc = __builtin_amdgcn_mfma_f64_16x16x4f64(lds[M::WN + ((i - 1) * MF_KS + s) * 64 + lofs], zbar[ch][s], acc, 0, 0, 0);
                        h4 = __builtin_amdgcn_mfma_f64_4x4x4f64(wrl[s * 16], zbar[ch][s], h4, 0, 0, 0);
                    }
#pragma unroll
                    for (int s = 0; s < 4; ++s) hbar[ch][s] = acc[s];
                    hbar[ch][4] = h4;
                }
                pj_wave_sync();
#pragma unroll
                for (int ch = 0; ch < C; ++ch) {
                    const double* TA = TAB + (2 * ch) * (MF_TRB * MF_LD);
                    const double* TB = TA + MF_TRB * MF_LD;
                    double aF[4], bF[4], aS[4], bS[4];
#pragma unroll
                    for (int kk = 0; kk < 4; ++kk) {
                        aF[kk] = TA[pt * MF_LD + 4 * kk + q];
                        bF[kk] = TB[pt * MF_LD + 4 * kk + q];
                        aS[kk] = TA[(16 + (lane & 3)) * MF_LD + 4 * kk + q];
                        bS[kk] = TB[(16 + (lane & 3)) * MF_LD + 4 * kk + q];
                    }
#pragma unroll
                    for (int kk = 0; kk < 4; ++kk) {
                        dWacc[i - 1] = __builtin_amdgcn_mfma_f64_16x16x4f64(aF[kk], bF[kk], dWacc[i - 1], 0, 0, 0);
                        dS10[i - 1] = __builtin_amdgcn_mfma_f64_4x4x4f64(aS[kk], bF[kk], dS10[i - 1], 0, 0, 0);
                        dS01[i - 1] = __builtin_amdgcn_mfma_f64_4x4x4f64(bS[kk], aF[kk], dS01[i - 1], 0, 0, 0);
                    }
                    accC[i - 1] = __builtin_amdgcn_mfma_f64_4x4x4f64(TA[(16 + (lane & 3)) * MF_LD + (pt & 12) + q],
                                                                   TB[(16 + (lane & 3)) * MF_LD + (pt & 12) + q], accC[i - 1], 0, 0, 0);
                }
            }
        }
    };
    if constexpr (EARLY0) {
        if (n_own > 0) rev_tile(0, std::true_type{}, zc, zcc);
#pragma unroll 1
        for (int k = 1; k < n_own; ++k) rev_tile(k, std::false_type{}, zc, zcc);
    } else {
#pragma unroll 1
        for (int k = 0; k < n_own; ++k) rev_tile(k, std::false_type{}, zc, zcc);
    }

    if (QT && q_kind != 0) {
        // ---- the packed quarter, reverse (branch-free: per-lane selects and unconditional loads, see kernels_fused.hip) ----
        int lofs = lane;
        asm volatile("" : "+v"(lofs));
        const long pc = q_valid ? q_p : 0;
        const double x0 = g.X[pc], x1 = g.X[g.N + pc];
        double AAq[NSV];
#pragma unroll
        for (int j = 0; j < NSV; ++j) AAq[j] = lds[M::pq(g.P) + (wv * NSV + j) * 16 + q * 4 + qj];
        const double gch = lds[M::GB + (qcs < C ? qcs : 0) * MAXP + (q_kind == 1 ? q_lp : 0)];
        const double GB = q_kind == 1 ? (qcs < C ? gch : 0.0) : (q_val ? gdat_q : 0.0);
        const bool q_tan = q_first || q_second;
        // packed layer inputs, own-slot pre-activations z_c / z_cc (0 in the value slot) and, for the second-tangent slots, z_c
        // of their first tangent -- recomputed from s with ONE product per layer
        double Hq[L][MF_KS], ZCq[L][MF_KS], ZXq[L][MF_KS];
#pragma unroll
        for (int s = 0; s < MF_KS; ++s) {
            const double w0 = lds[M::W1O + (0 * MF_KS + s) * 64 + lofs], w1 = lds[M::W1O + (1 * MF_KS + s) * 64 + lofs];
            const double a = AAq[s], a1 = 1.0 - a * a, a2 = -2.0 * a * a1;
            const double wf = qcs == 1 ? w0 : w1, wb = (qcs - 1 - NT1) == 0 ? w0 : w1;
            ZCq[0][s] = q_first ? wf : 0.0;
            ZXq[0][s] = q_second ? wb : 0.0;
            double hf = a1 * wf, hs = a2 * wb * wb;
            fz_keep(hf); fz_keep(hs);
            Hq[0][s] = q_val ? a : (q_first ? hf : (q_second ? hs : 0.0));
        }
#pragma unroll
        for (int i = 1; i < L; ++i) {
            double Z[MF_KS];
            fz_layer<false>(WTl + (i - 1) * MF_KS * 64, WRl + (i - 1) * MF_KS * 16, nullptr, lofs, Hq[i - 1], Z);
#pragma unroll
            for (int s = 0; s < MF_KS; ++s) {
                const double a = AAq[i * MF_KS + s], a1 = 1.0 - a * a, a2 = -2.0 * a * a1;
                const double zx = dpp_move<0x110 + 4 * NT1>(Z[s]);
                ZCq[i][s] = q_tan ? Z[s] : 0.0;
                ZXq[i][s] = q_second ? zx : 0.0;
                double hf = a1 * Z[s], hs = a2 * zx * zx + a1 * Z[s];
                fz_keep(hf); fz_keep(hs);
                Hq[i][s] = q_val ? a : (q_first ? hf : (q_second ? hs : 0.0));
            }
        }
        double HB[MF_KS], ZB[MF_KS];
#pragma unroll
        for (int s = 0; s < MF_KS; ++s) {
            const double wo = lds[M::W1O + (2 * MF_KS + s) * 64 + lofs];
            dWo[s] = fma(Hq[L - 1][s], GB, dWo[s]);
            HB[s] = GB * wo;
        }
        dbo += (q == 0 && q_val) ? GB : 0.0;
        const double mval = q_val ? 1.0 : 0.0;
        const double mfx = (q_first && qcs - 1 < NT2) ? 1.0 : 0.0;       // first tangents that carry a second one
#pragma unroll
        for (int i = L - 1; i >= 0; --i) {
#pragma unroll
            for (int s = 0; s < MF_KS; ++s) {
                const double a = AAq[i * MF_KS + s];
                const double a1 = 1.0 - a * a, a2 = -2.0 * a * a1, a3 = -2.0 * a1 * (1.0 - 3.0 * a * a);
                const double zc = ZCq[i][s], zxx = ZXq[i][s];
                // what the value slot collects from the tangent slots of its point
                double tf = HB[s] * a2 * zc, ts = HB[s] * (a3 * zxx * zxx + a2 * zc);
                fz_keep(tf); fz_keep(ts);
                const double T = q_first ? tf : (q_second ? ts : 0.0);
                const double t4 = dpp_move<0x104>(T), t8 = dpp_move<0x108>(T), t12 = dpp_move<0x10C>(T);
                const double hb2 = dpp_move<0x100 + 4 * NT1>(HB[s]);      // first-tangent slots: hbar of their second tangent
                double zb = HB[s] * a1;
                zb = fma(mval, (t4 + t8) + t12, zb);
                zb = fma(mfx * 2.0 * hb2 * a2, zc, zb);
                ZB[s] = zb;
                db[i][s] = fma(mval, zb, db[i][s]);
            }
            if (i == 0) {
                const double c0 = q_val ? x0 : (qcs == 1 ? 1.0 : 0.0), c1 = q_val ? x1 : ((NT1 > 1 && qcs == 2) ? 1.0 : 0.0);
#pragma unroll
                for (int s = 0; s < MF_KS; ++s) {
                    dW1[0][s] = fma(c0, ZB[s], dW1[0][s]);
                    dW1[1][s] = fma(c1, ZB[s], dW1[1][s]);
                }
            } else {
                pj_wave_sync();
                double* TA = TAB;
                double* TB = TA + MF_TRB * MF_LD;
#pragma unroll
                for (int s = 0; s < MF_KS; ++s) {
                    TA[(4 * s + q) * MF_LD + pt] = Hq[i - 1][s];
                    TB[(4 * s + q) * MF_LD + pt] = ZB[s];
                }
                {
                    v4d acc = v4d{0.0, 0.0, 0.0, 0.0};
                    double h4 = 0.0;
                    const double* wrl = lds + M::WRB + (i - 1) * MF_KS * 16 + q * 4 + (lane & 3);
#pragma unroll
                    for (int s = 0; s < MF_KS; ++s) {
                        acc = __builtin_amdgcn_mfma_f64_16x16x4f64(lds[M::WN + ((i - 1) * MF_KS + s) * 64 + lofs], ZB[s], acc, 0, 0, 0);
                        h4 = __builtin_amdgcn_mfma_f64_4x4x4f64(wrl[s * 16], ZB[s], h4, 0, 0, 0);
                    }
#pragma unroll
                    for (int s = 0; s < 4; ++s) HB[s] = acc[s];
                    HB[4] = h4;
                }
                pj_wave_sync();
                double aF[4], bF[4], aS[4], bS[4];
#pragma unroll
                for (int kk = 0; kk < 4; ++kk) {
                    aF[kk] = TA[pt * MF_LD + 4 * kk + q];
                    bF[kk] = TB[pt * MF_LD + 4 * kk + q];
                    aS[kk] = TA[(16 + (lane & 3)) * MF_LD + 4 * kk + q];
                    bS[kk] = TB[(16 + (lane & 3)) * MF_LD + 4 * kk + q];
                }
#pragma unroll
                for (int kk = 0; kk < 4; ++kk) {
                    dWacc[i - 1] = __builtin_amdgcn_mfma_f64_16x16x4f64(aF[kk], bF[kk], dWacc[i - 1], 0, 0, 0);
                    dS10[i - 1] = __builtin_amdgcn_mfma_f64_4x4x4f64(aS[kk], bF[kk], dS10[i - 1], 0, 0, 0);
                    dS01[i - 1] = __builtin_amdgcn_mfma_f64_4x4x4f64(bS[kk], aF[kk], dS01[i - 1], 0, 0, 0);
                }
                accC[i - 1] = __builtin_amdgcn_mfma_f64_4x4x4f64(TA[(16 + (lane & 3)) * MF_LD + (pt & 12) + q],
                                                               TB[(16 + (lane & 3)) * MF_LD + (pt & 12) + q], accC[i - 1], 0, 0, 0);
            }
        }
    }

    // ---- epilogue: per-wave partials -> LDS -> one gradient row per workgroup ----
    TA_STAMP(7);
    __syncthreads();
    TA_STAMP(8);
    double* WP = lds + M::TR + (long)wv * g.P;     // every one of the P entries is written below
#pragma unroll
    for (int i = 1; i < L; ++i) {
#pragma unroll
        for (int r = 0; r < 4; ++r) WP[g.woff[i] + (4 * r + q) * MF_H + pt] = dWacc[i - 1][r];
        WP[g.woff[i] + (16 + q) * MF_H + pt] = dS10[i - 1];
        WP[g.woff[i] + pt * MF_H + 16 + q] = dS01[i - 1];
    }
#pragma unroll
    for (int i = 1; i < L; ++i) {
        double t = accC[i - 1];
        t = quad4_sum(t);
        if (pt < 4) WP[g.woff[i] + (16 + q) * MF_H + 16 + pt] = t;
    }
#pragma unroll
    for (int s = 0; s < MF_KS; ++s) {
        const int j = 4 * s + q;
        double v[L + 3];
#pragma unroll
        for (int i = 0; i < L; ++i) v[i] = db[i][s];
        v[L] = dW1[0][s]; v[L + 1] = dW1[1][s]; v[L + 2] = dWo[s];
#pragma unroll
        for (int kq = 0; kq < L + 3; ++kq) v[kq] = row_sum16(v[kq]);
        if (pt == 0) {
#pragma unroll
            for (int i = 0; i < L; ++i) WP[g.boff[i] + j] = v[i];
            WP[g.woff[0] + j] = v[L];
            WP[g.woff[0] + MF_H + j] = v[L + 1];
            WP[g.woff[L] + j] = v[L + 2];
        }
    }
    {
        const double t = row_sum16(dbo);
        if (lane == 0) WP[g.boff[L]] = t;
    }
    __syncthreads();
    const double* W0 = lds + M::TR;
    double* row = g.GPART + (long)blockIdx.x * g.P;
    for (int idx = tid; idx < g.P; idx += TA_BLOCK) {
        double acc = 0.0;
#pragma unroll
        for (int w = 0; w < TA_WAVES; ++w) acc += W0[(long)w * g.P + idx];
        row[idx] = acc;
    }
#ifdef HPV_FZ_TIMING
    if (lane == 0 && pa.GBAR) {   // [block][wave][12]: staging, forward, wait, partial projection, barrier, residual+adjoint, reverse, wait, epilogue
        TA_STAMP(9);
        double* o = pa.GBAR + ((long)blockIdx.x * 4 + wv) * 12;
        for (int i = 0; i < 9; ++i) o[i] = (double)(ta_t[i + 1] - ta_t[i]);
        o[9] = (double)(wall_clock64() - ta_wall) * 0.01;
        o[10] = (double)n_own;
    }
#endif
}

// ------------------------------------------------------------------------------------------------
// host side
// ------------------------------------------------------------------------------------------------
template <int NT1, int NT2, int L, int QX, int QY, int NTX, int NTY, bool QT>
static void launch_iter_tall_q(const MfmaArgs& a, int blocks, hipStream_t s) {
    constexpr int C = 1 + NT1 + NT2;
    const size_t bytes = (size_t)TaLds<L, C, QX, QY, NTX, NTY, 16 * TA_WAVES * TA_MAXT>::total(a.P) * sizeof(double);
    static bool attr_set = false;
    if (!attr_set) {
        (void)hipFuncSetAttribute((const void*)k_iter_tall<NT1, NT2, L, QX, QY, NTX, NTY, QT>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
        attr_set = true;
    }
    hipLaunchKernelGGL((k_iter_tall<NT1, NT2, L, QX, QY, NTX, NTY, QT>), dim3(blocks), dim3(TA_BLOCK), bytes, s, a);
}

template <int NT1, int NT2, int L, int QX, int QY, int NTX, int NTY>
static void launch_iter_tall(const MfmaArgs& a, int blocks, hipStream_t s) {
#ifndef HPV_AGPR_GUARD_TRIPPED_QT     // csrc/build.sh: only the quarter-tile instantiations reached their hand-managed AGPR range
    if (a.tall_qt) { launch_iter_tall_q<NT1, NT2, L, QX, QY, NTX, NTY, true>(a, blocks, s); return; }
#endif
    launch_iter_tall_q<NT1, NT2, L, QX, QY, NTX, NTY, false>(a, blocks, s);
}

const char* hpv_tall_build_state() {
#if defined(HPV_AGPR_GUARD_TRIPPED)
    return "absent";
#elif defined(HPV_AGPR_GUARD_TRIPPED_QT)
    return "no-quarter-tile";
#else
    return "ok";
#endif
}
// Workgroups per element of the tall-element kernel (0: not applicable): the largest power of two with n_elem S <= CUs, at most
// 64, such that a workgroup's share fits TA_WAVES x TA_MAXT tiles (with room for a boundary/data tile where one is adopted).
int hpv_mfma_tall_split(HpvMfma* m, const ProjDesc& pd, long n_elem) {
#ifdef HPV_AGPR_GUARD_TRIPPED     // csrc/build.sh: the compiler's registers reached the hand-managed AGPR range of k_iter_tall
    return 0;
#endif
    const NetDesc& nd = m->nd;
    if (!m->iter_fused_ok || !m->iter_split_ok || !m->xerr || !m->xg || !m->xiter || m->H != MF_H) return 0;
    if (!(nd.d == 2 && nd.nT1 == 2 && nd.nT2 <= 1 && nd.act == HPV_ACT_TANH) || m->L < 2 || m->L > 3) return 0;
    if (nd.nT2 == 1 && !(nd.t2w[0] == 1.0 && nd.t2w[1] == 0.0)) return 0;      // (the mixed second tangent, NetDesc::t2w: not in this kernel)
    if (!(pd.qx == 80 && pd.qy == 80 && pd.ntx == 5 && pd.nty == 5) || pd.edge || pd.nact || pd.nterms < 1) return 0;
    if (n_elem <= 0 || n_elem > m->xsync_elems) return 0;
    const int tpe = 80 * 80 / 16;
    int split = 1;
    while (split < 64 && n_elem * split * 2 <= m->n_cus) split *= 2;
    if (n_elem * split > m->n_cus) return 0;                                   // all partners must be resident: one workgroup per CU
    if ((tpe + split - 1) / split > TA_WAVES * TA_MAXT - 1) return 0;          // (one slot per workgroup stays free for a data tile)
    return split;
}

// Whole training pass (forward, projection, reverse) of a shard of tall elements in one launch.  Returns false when the shape /
// variational form / shard is not covered; the caller then runs the separate kernels.
bool hpv_mfma_iter_tall(HpvMfma* m, const double* theta, const double* X, double* GPART, int* rows, hipStream_t s,
                        const MfmaDataTerm* dt, const ProjArgs& pa, long n_elem) {
    const int split = hpv_mfma_tall_split(m, pa.pd, n_elem);
    if (split < 2) return false;
    const long blocks = n_elem * split;
    const long rest = m->ntiles - n_elem * (80 * 80 / 16);           // boundary/data tiles: at most one per workgroup
    if (rest < 0 || rest > blocks) return false;
    if (blocks > hpv_mfma_grad_rows(m) && blocks > m->max_rows) return false;
    MfmaArgs a = m->base;
    a.theta = theta; a.X = X; a.GPART = GPART;
    a.data_off = -1;
    if (dt && dt->n_data > 0) {
        a.data_off = dt->data_off; a.ud = dt->ud; a.gbar0 = dt->gbar0; a.data_part = dt->data_part;
        a.data_scale = dt->scale; a.data_write_gbar = dt->write_gbar;
    } else if (rest > 0) {
        return false;
    }
    a.proj_n_elem = n_elem;
    a.proj_split = split;
    {
        // quarter-tile plan: every workgroup's tile count must be 0 or 1 mod 4 (its one extra tile is cut in four), its whole tiles
        // must leave a stash slot for the quarter, and the workgroups without a 13th element tile must suffice for the data tiles
        const int tpe = 80 * 80 / 16, lg = __builtin_ctz(split);
        bool qt = getenv("HPV_NO_QUARTER_TILE") == nullptr;
        long n0 = 0;
        for (int p_ = 0; p_ < split; ++p_) {
            const int n = (((p_ + 1) * tpe) >> lg) - ((p_ * tpe) >> lg);
            if (n % TA_WAVES > 1 || n / TA_WAVES + 1 > TA_MAXT) qt = false;
            if (n % TA_WAVES == 0) ++n0;
        }
        if (rest > n0 * n_elem) qt = false;
        a.tall_qt = qt ? 1 : 0;
    }
    a.xerr = m->xerr;
    a.xdebug_skip = m->xdebug_skip;
    a.xg = m->xg;
    a.xiter = m->xiter;
    a.pa = pa;
    m->last_split = true;
    m->split_used = true;
    {
#ifdef HPV_AGPR_GUARD_TRIPPED_QT
        const bool qt_run = false;
#else
        const bool qt_run = a.tall_qt != 0;
#endif
        snprintf(m->variant, sizeof m->variant, "k_iter_tall<NT1=2,NT2=%d,L=%d,80,80,5,5,QT=%s> split=%d", m->nd.nT2, m->L,
                 qt_run ? "true" : "false", split);
    }
    const int key = m->nd.nT2 * 10 + m->L;
    switch (key) {
        case 12: launch_iter_tall<2, 1, 2, 80, 80, 5, 5>(a, (int)blocks, s); break;
        case 13: launch_iter_tall<2, 1, 3, 80, 80, 5, 5>(a, (int)blocks, s); break;
        case 2: launch_iter_tall<2, 0, 2, 80, 80, 5, 5>(a, (int)blocks, s); break;
        case 3: launch_iter_tall<2, 0, 3, 80, 80, 5, 5>(a, (int)blocks, s); break;
        default: return false;
    }
    if (rows) *rows = (int)blocks;
    return true;
}
