// Width-generic MFMA tile arithmetic: Taylor-mode MLP layers for ANY uniform hidden width H = 4 m, 8 <= H <= 64.
//
// The reference takes any `Net_layer` (P1:236, P2:280-286, P3:46-51); kernels_mfma.hip / kernels_fused.hip / kernels_tile.hip /
// kernels_tall.hip are written for H = 20 (one 16-row tile + one 4-row remainder, MF_H a #define).  Here the same register
// formulation is written over H as a template parameter:
//     H = 16 NL + 4 NR  (NR = 0..3),   KS = H / 4 k-steps,
// a lane (q = lane >> 4, pt = lane & 15) carries the KS values of neurons 4 s + q, s = 0..KS-1, at point pt, per channel:
//   * output neurons 16 t .. 16 t + 15 (t < NL) come from ONE v_mfma_f64_16x16x4_f64 accumulator (D register r <-> neuron
//     16 t + 4 r + q = 4 (4 t + r) + q, i.e. the lane's value s = 4 t + r): layers chain register-to-register, no LDS, no shuffles;
//   * output neurons 16 NL + 4 u .. + 3 (u < NR) from v_mfma_f64_4x4x4_4b_f64 (four independent 4x4x4 blocks = the four groups of
//     4 points): D lands on the lane's value s = 4 NL + u.  No padded 16-row tiles: fp64 MFMA and fp64 VALU share one datapath
//     on gfx950, padding is pure loss.
// Weight fragments live in LDS, lane-major (conflict-free ds_read_b64), staged once per workgroup:
//   forward   WT[t][s][64]: W[in = 4 s + (ln >> 4)][out = 16 t + (ln & 15)]        WR[u][s][16]: W[in = 4 s + k][out = 16 NL + 4 u + a], index k * 4 + a
//   reverse   WN[t][s][64]: W[in = 16 t + (ln & 15)][out = 4 s + (ln >> 4)]        WB[u][s][16]: W[in = 16 NL + 4 u + a][out = 4 s + k], index k * 4 + a
// Only dW = sum_pt h^T zbar contracts over points and needs the operands in the other orientation: per-wave LDS transpose tiles
// TA / TB [H][17], then NL x NL large tiles, 2 NL NR strips and NR x NR corners on the two MFMA shapes.
#pragma once
#include "hpv_mfma_dev.h"

template <int H>
struct WD {
    static_assert(H % 4 == 0 && H >= 8 && H <= 64, "hidden width: a multiple of 4 in [8, 64]");
    static constexpr int KS = H / 4;
    static constexpr int NL = H / 16;
    static constexpr int NR = KS - 4 * NL;
    static constexpr int LD = 17;                       // padded leading dimension of the transpose tiles
    static constexpr int FRAG = NL * KS * 64 + NR * KS * 16;   // doubles of one layer's fragments in one orientation
    static constexpr int TR = H * LD;                   // one transpose tile
    static constexpr int NACC = NL * NL * 4 + 2 * NL * NR + NR * NR;   // dW accumulator doubles per lane and layer
};

// Stage the fragments of hidden->hidden layer `wo` (offset of W in theta, row-major [in][out], stride H) into LDS.
// FWD: WT | WR (forward orientation); else WN | WB (reverse orientation).  All threads of the block take part; the caller syncs.
template <int H, bool FWD, int BT>
__device__ __forceinline__ void wide_stage_layer(const double* __restrict__ th, int wo, double* dst, int tid) {
    using W = WD<H>;
    constexpr int NBIG = W::NL * W::KS * 64, NSM = W::NR * W::KS * 16;
#pragma unroll 4
    for (int f = tid; f < NBIG; f += BT) {
        const int ln = f & 63, s = (f >> 6) % W::KS, t = f / (64 * W::KS);
        const int a = 4 * s + (ln >> 4), b = 16 * t + (ln & 15);
        dst[f] = FWD ? th[wo + a * H + b] : th[wo + b * H + a];
    }
    if constexpr (W::NR > 0) {
        for (int f = tid; f < NSM; f += BT) {
            const int fr = f & 15, s = (f >> 4) % W::KS, u = f / (16 * W::KS);
            const int a = 4 * s + (fr >> 2), b = 16 * W::NL + 4 * u + (fr & 3);
            dst[NBIG + f] = FWD ? th[wo + a * H + b] : th[wo + b * H + a];
        }
    }
}

// z^T = W^T h^T for C channels (+ bias on the value channel): `frag` = this layer's forward fragments, `bias` = the layer's
// bias vector in LDS (compact, [H]; a lane reads b[4 s + q]: four addresses per instruction, broadcast, conflict-free).
// BIAS = false: no channel takes a bias (the tangent channels alone: the recompute of the element-resident kernel).
template <int H, int C, bool BIAS = true>
__device__ __forceinline__ void wide_fwd_layer(const double* frag, const double* bias, int lofs,
                                               const double (&h)[C][WD<H>::KS], double (&z)[C][WD<H>::KS]) {
    using W = WD<H>;
    const int q = lofs >> 4;
#pragma unroll
    for (int t = 0; t < W::NL; ++t) {
        v4d acc[C];
#pragma unroll
        for (int ch = 0; ch < C; ++ch)
            acc[ch] = (BIAS && ch == 0) ? v4d{bias[16 * t + q], bias[16 * t + 4 + q], bias[16 * t + 8 + q], bias[16 * t + 12 + q]} : v4d{0.0, 0.0, 0.0, 0.0};
#pragma unroll
        for (int s = 0; s < W::KS; ++s) {
            const double a = frag[(t * W::KS + s) * 64 + lofs];
#pragma unroll
            for (int ch = 0; ch < C; ++ch) acc[ch] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, h[ch][s], acc[ch], 0, 0, 0);
        }
#pragma unroll
        for (int ch = 0; ch < C; ++ch)
#pragma unroll
            for (int r = 0; r < 4; ++r) z[ch][4 * t + r] = acc[ch][r];
    }
#pragma unroll
    for (int u = 0; u < W::NR; ++u) {
        const double* wr = frag + W::NL * W::KS * 64 + u * W::KS * 16 + q * 4 + (lofs & 3);
        double zz[C];
#pragma unroll
        for (int ch = 0; ch < C; ++ch) zz[ch] = (BIAS && ch == 0) ? bias[16 * W::NL + 4 * u + q] : 0.0;
#pragma unroll
        for (int s = 0; s < W::KS; ++s) {
            const double a = wr[s * 16];
#pragma unroll
            for (int ch = 0; ch < C; ++ch) zz[ch] = __builtin_amdgcn_mfma_f64_4x4x4f64(a, h[ch][s], zz[ch], 0, 0, 0);
        }
#pragma unroll
        for (int ch = 0; ch < C; ++ch) z[ch][4 * W::NL + u] = zz[ch];
    }
}

// hbar_in^T = W zbar^T for ONE channel: `frag` = this layer's reverse fragments (WN | WB)
template <int H>
__device__ __forceinline__ void wide_hbar(const double* frag, int lofs, const double (&zb)[WD<H>::KS], double (&hb)[WD<H>::KS]) {
    using W = WD<H>;
    const int q = lofs >> 4;
#pragma unroll
    for (int t = 0; t < W::NL; ++t) {
        v4d acc = v4d{0.0, 0.0, 0.0, 0.0};
#pragma unroll
        for (int s = 0; s < W::KS; ++s) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(frag[(t * W::KS + s) * 64 + lofs], zb[s], acc, 0, 0, 0);
#pragma unroll
        for (int r = 0; r < 4; ++r) hb[4 * t + r] = acc[r];
    }
#pragma unroll
    for (int u = 0; u < W::NR; ++u) {
        const double* wr = frag + W::NL * W::KS * 64 + u * W::KS * 16 + q * 4 + (lofs & 3);
        double h4 = 0.0;
#pragma unroll
        for (int s = 0; s < W::KS; ++s) h4 = __builtin_amdgcn_mfma_f64_4x4x4f64(wr[s * 16], zb[s], h4, 0, 0, 0);
        hb[4 * W::NL + u] = h4;
    }
}

// dW accumulators of one hidden->hidden layer (per lane):
//   big[ti][to] (v4d)  lane (q, pt), register r: dW[in = 16 ti + 4 r + q][out = 16 to + pt]
//   s10[u][to]         lane (q, pt):             dW[in = 16 NL + 4 u + q][out = 16 to + pt]
//   s01[ti][u]         lane (q, pt):             dW[in = 16 ti + pt][out = 16 NL + 4 u + q]
//   cor[u][u2]         lane (q, 4 b + j):        partial of point block b for dW[in = 16 NL + 4 u + q][out = 16 NL + 4 u2 + j]
template <int H>
struct WideDW {
    using W = WD<H>;
    v4d big[W::NL > 0 ? W::NL : 1][W::NL > 0 ? W::NL : 1];
    double s10[W::NR > 0 ? W::NR : 1][W::NL > 0 ? W::NL : 1];
    double s01[W::NL > 0 ? W::NL : 1][W::NR > 0 ? W::NR : 1];
    double cor[W::NR > 0 ? W::NR : 1][W::NR > 0 ? W::NR : 1];
    __device__ __forceinline__ void zero() {
#pragma unroll
        for (int a = 0; a < (W::NL > 0 ? W::NL : 1); ++a)
#pragma unroll
            for (int b = 0; b < (W::NL > 0 ? W::NL : 1); ++b) big[a][b] = v4d{0.0, 0.0, 0.0, 0.0};
#pragma unroll
        for (int a = 0; a < (W::NR > 0 ? W::NR : 1); ++a) {
#pragma unroll
            for (int b = 0; b < (W::NL > 0 ? W::NL : 1); ++b) { s10[a][b] = 0.0; s01[b][a] = 0.0; }
#pragma unroll
            for (int b = 0; b < (W::NR > 0 ? W::NR : 1); ++b) cor[a][b] = 0.0;
        }
    }
};

// the tile's (h_in, zbar) of one channel -> this wave's transpose pair (rows = neurons, columns = the 16 points)
template <int H>
__device__ __forceinline__ void wide_transpose_store(double* TA, double* TB, int q, int pt, const double (&hv)[WD<H>::KS],
                                                     const double (&zb)[WD<H>::KS]) {
    using W = WD<H>;
#pragma unroll
    for (int s = 0; s < W::KS; ++s) {
        TA[(4 * s + q) * W::LD + pt] = hv[s];
        TB[(4 * s + q) * W::LD + pt] = zb[s];
    }
}

// dW += h_in^T zbar of one channel, read from the wave's transpose pair (the caller fences around the LDS stores)
template <int H>
__device__ __forceinline__ void wide_dw_accumulate(const double* TA, const double* TB, int lane, WideDW<H>& d) {
    using W = WD<H>;
    const int q = lane >> 4, pt = lane & 15;
    double aF[W::NL > 0 ? W::NL : 1][4], bF[W::NL > 0 ? W::NL : 1][4], aS[W::NR > 0 ? W::NR : 1][4], bS[W::NR > 0 ? W::NR : 1][4];
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) {
#pragma unroll
        for (int t = 0; t < W::NL; ++t) {
            aF[t][kk] = TA[(16 * t + pt) * W::LD + 4 * kk + q];
            bF[t][kk] = TB[(16 * t + pt) * W::LD + 4 * kk + q];
        }
#pragma unroll
        for (int u = 0; u < W::NR; ++u) {
            aS[u][kk] = TA[(16 * W::NL + 4 * u + (lane & 3)) * W::LD + 4 * kk + q];
            bS[u][kk] = TB[(16 * W::NL + 4 * u + (lane & 3)) * W::LD + 4 * kk + q];
        }
    }
    double cA[W::NR > 0 ? W::NR : 1], cB[W::NR > 0 ? W::NR : 1];
#pragma unroll
    for (int u = 0; u < W::NR; ++u) {
        cA[u] = TA[(16 * W::NL + 4 * u + (lane & 3)) * W::LD + (pt & 12) + q];
        cB[u] = TB[(16 * W::NL + 4 * u + (lane & 3)) * W::LD + (pt & 12) + q];
    }
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) {
#pragma unroll
        for (int ti = 0; ti < W::NL; ++ti)
#pragma unroll
            for (int to = 0; to < W::NL; ++to)
                d.big[ti][to] = __builtin_amdgcn_mfma_f64_16x16x4f64(aF[ti][kk], bF[to][kk], d.big[ti][to], 0, 0, 0);
#pragma unroll
        for (int u = 0; u < W::NR; ++u)
#pragma unroll
            for (int t = 0; t < W::NL; ++t) {
                d.s10[u][t] = __builtin_amdgcn_mfma_f64_4x4x4f64(aS[u][kk], bF[t][kk], d.s10[u][t], 0, 0, 0);   // h[., 16 NL + 4 u + i'] x zbar[., out]
                d.s01[t][u] = __builtin_amdgcn_mfma_f64_4x4x4f64(bS[u][kk], aF[t][kk], d.s01[t][u], 0, 0, 0);   // zbar[., 16 NL + 4 u + i'] x h[., in]
            }
    }
    // corners: A_b[i'][k] = h[pt = 4 b + k][16 NL + 4 u + i'], B_b[k][j] = zbar[pt = 4 b + k][16 NL + 4 u2 + j]  (k = q, b = (lane & 15) >> 2)
#pragma unroll
    for (int u = 0; u < W::NR; ++u)
#pragma unroll
        for (int u2 = 0; u2 < W::NR; ++u2) d.cor[u][u2] = __builtin_amdgcn_mfma_f64_4x4x4f64(cA[u], cB[u2], d.cor[u][u2], 0, 0, 0);
}

// Cross-wave reduction of the per-wave gradient accumulators into ONE row of the workgroup (fixed order over the waves: bitwise
// reproducible).  EX = an LDS region of at least WAVES * max(256, (L + D + 1) H + 1) doubles that is free by now (the transpose
// tiles).  One accumulator group at a time: a v4d accumulator is 4 doubles per lane = 256 per wave.
template <int H, int L, int D, int WAVES>
__device__ __forceinline__ void wide_epilogue(double* EX, const WideDW<H> (&dW)[(L > 1 ? L - 1 : 1)], const double (&db)[L][WD<H>::KS],
                                              const double (&dW1)[D][WD<H>::KS], const double (&dWo)[WD<H>::KS], double dbo,
                                              double* __restrict__ row, const int* woff, const int* boff) {
    using W = WD<H>;
    constexpr int KS = W::KS, BT = WAVES * 64;
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6, q = lane >> 4, pt = lane & 15;
    auto xsum = [&](int k, int stride) -> double {
        double t = EX[k];
#pragma unroll
        for (int w = 1; w < WAVES; ++w) t += EX[w * stride + k];
        return t;
    };
    auto reduce_v4 = [&](const v4d& a, auto&& index_of) {
        __syncthreads();
#pragma unroll
        for (int r = 0; r < 4; ++r) EX[wv * 256 + r * 64 + lane] = a[r];
        __syncthreads();
        for (int f = tid; f < 256; f += BT) row[index_of(f >> 6, (f & 63) >> 4, f & 15)] = xsum(f, 256);
    };
    auto reduce_1 = [&](double a, auto&& index_of, bool quad_first) {
        if (quad_first) a = quad4_sum(a);
        __syncthreads();
        EX[wv * 256 + lane] = a;
        __syncthreads();
        if (tid < 64) {
            const int idx = index_of(tid >> 4, tid & 15);
            if (idx >= 0) row[idx] = xsum(tid, 256);
        }
    };
#pragma unroll
    for (int i = 1; i < L; ++i) {
        const int wo = woff[i];
        const WideDW<H>& d = dW[i - 1];
#pragma unroll
        for (int ti = 0; ti < W::NL; ++ti)
#pragma unroll
            for (int to = 0; to < W::NL; ++to)
                reduce_v4(d.big[ti][to], [&](int r, int qq, int pp) { return wo + (16 * ti + 4 * r + qq) * H + 16 * to + pp; });
#pragma unroll
        for (int u = 0; u < W::NR; ++u) {
#pragma unroll
            for (int t = 0; t < W::NL; ++t) {
                reduce_1(d.s10[u][t], [&](int qq, int pp) { return wo + (16 * W::NL + 4 * u + qq) * H + 16 * t + pp; }, false);
                reduce_1(d.s01[t][u], [&](int qq, int pp) { return wo + (16 * t + pp) * H + 16 * W::NL + 4 * u + qq; }, false);
            }
#pragma unroll
            for (int u2 = 0; u2 < W::NR; ++u2)
                reduce_1(d.cor[u][u2], [&](int qq, int pp) { return pp < 4 ? wo + (16 * W::NL + 4 * u + qq) * H + 16 * W::NL + 4 * u2 + pp : -1; }, true);
        }
    }
    // per-lane partials: sum over the 16 point lanes of each neuron group, then over the waves -- all of them in one exchange:
    // wave w parks its (L + D + 1) H + 1 sums at EX[w * NV ..]
    constexpr int NV = (L + D + 1) * H + 1;
    __syncthreads();
#pragma unroll
    for (int s = 0; s < KS; ++s) {
#pragma unroll
        for (int k = 0; k < L + D + 1; ++k) {
            const double v = k < L ? db[k < L ? k : 0][s] : (k < L + D ? dW1[(k >= L && k - L < D) ? (k - L) : 0][s] : dWo[s]);
            const double t = row_sum16(v);
            if (pt == 0) EX[wv * NV + k * H + 4 * s + q] = t;
        }
    }
    {
        const double t = row_sum16(dbo);
        if (lane == 0) EX[wv * NV + (L + D + 1) * H] = t;
    }
    __syncthreads();
    for (int f = tid; f < NV; f += BT) {
        const int k = f / H, j = f - k * H;
        const int idx = k < L ? boff[k] + j : (k < L + D ? woff[0] + (k - L) * H + j : (k == L + D ? woff[L] + j : boff[L]));
        row[idx] = xsum(f, NV);
    }
}
