// Taylor-mode MLP forward / reverse kernels for ANY uniform hidden width H = 4 m (instantiated: WIDE_WIDTHS of csrc/build.sh), every
// channel set of the MFMA path, 1..4 hidden layers -- the MFMA path for the networks kernels_mfma.hip (H = 20) does not take.
//
// The reference's `Net_layer` is a free hyper-parameter (P1:236, P2:280-286, P3:46-51); until round 4 every hidden width other
// than 20 ran on the generic VALU kernels (kernels_generic.hip) -- a 75x cliff behind the same class surface.  Structure =
// the two-kernel MFMA path: k_fwd_wide writes the activation store [tile][layer][slot][KS][64] (512-byte
// coalesced wave accesses), the projection runs as its own launch (k_project_tp / k_project_wg / k_project), k_bwd_wide reads
// the store back, accumulates dW / db per wave in registers over its tiles and reduces them per workgroup into one gradient
// row.  The tile arithmetic (16-row + 4-row MFMA decomposition of H) is hpv_wide_dev.h.
#include <cstdio>
#include <cstdlib>
#include <type_traits>

#include "hpv_wide_dev.h"

#define WF_BLOCK 256
#define WF_WAVES 4

// ------------------------------------------------------------------------------------------------
// LDS maps
// ------------------------------------------------------------------------------------------------
template <int H, int L, int D>
struct WideFwdLds {
    using W = WD<H>;
    static constexpr int LH = L - 1;
    static constexpr int FR = 0;                         // forward fragments           [LH][FRAG]
    static constexpr int BI = FR + LH * W::FRAG;         // hidden->hidden biases       [LH][H]
    static constexpr int W1 = BI + LH * H;               // first-layer rows, first bias, head weights  [D + 2][H]
    static constexpr int TOTAL = W1 + (D + 2) * H;
};
template <int H, int L, int D>
struct WideBwdLds {
    using W = WD<H>;
    static constexpr int LH = L - 1;
    static constexpr int TAB = 0;                               // per-wave transpose pair (one channel at a time) [WAVES][2][H][17]; epilogue: exchange
    static constexpr int FR = TAB + WF_WAVES * 2 * W::TR;       // reverse fragments        [LH][FRAG]
    static constexpr int W1 = FR + LH * W::FRAG;                // first-layer rows, head weights  [D + 1][H]
    static constexpr int TOTAL = W1 + (D + 1) * H;
    static_assert(WF_WAVES * 2 * W::TR >= WF_WAVES * 256, "the epilogue exchanges one accumulator (4 doubles per lane) per wave through the transpose region");
};

// ------------------------------------------------------------------------------------------------
// forward
// ------------------------------------------------------------------------------------------------
// SAVE: 1 every slot of the activation store (s, [cos], z_c, z_cc).  The forward-only evaluations (predict, loss read-back:
// g.save_act == 0) run the SAME instantiation with all tiles' stores aliased onto one tile's block (as a third compile-time variant
// they were a third of this file's forward kernels and of their build time);
// 2 (libhpvpinn_testhooks.so only) only s (and cos): the
// reverse kernel k_bwd_wide_rc then recomputes the tangent pre-activations on the MFMA pipe -- a third of the store's bytes for
// three channels (the forward kernel is HBM-write-bound on the store: 182 MB, 4.2 TB/s at H = 32 on the config-4 grid)
template <int D, int NT1, int NT2, int ACT, int L, int H, int SAVE>
__global__ void __launch_bounds__(WF_BLOCK, (H <= 32 ? 2 : 1)) k_fwd_wide(MfmaArgs g) {
    using W = WD<H>;
    using M = WideFwdLds<H, L, D>;
    constexpr int KS = W::KS, C = 1 + NT1 + NT2;
    constexpr int NS = SAVE == 2 ? (ACT == HPV_ACT_SIN ? 2 : 1) : SlotCount<ACT, NT1, NT2>::value;      // slots per layer in the store
    constexpr int SA1 = 1, SZC = 1 + (ACT == HPV_ACT_SIN ? 1 : 0), SZCC = SZC + NT1;
    constexpr bool SAVE_T = SAVE == 1;       // tangent pre-activations stored too
    extern __shared__ __attribute__((aligned(16))) double lds[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int q = lane >> 4, pt = lane & 15;
    const long wave = ((long)blockIdx.x * blockDim.x + tid) >> 6;
    const long nwaves = ((long)gridDim.x * blockDim.x) >> 6;
    const double* __restrict__ th = g.theta;

    // the coordinates of the first tile are requested before the weight staging (one round trip for both)
    double xn[D];
    {
        long p0 = wave * 16 + pt;
        p0 = p0 < g.N ? p0 : g.N - 1;
#pragma unroll
        for (int c = 0; c < D; ++c) xn[c] = g.X[(long)c * g.N + p0];
    }
#pragma unroll
    for (int i = 1; i < L; ++i) {
        wide_stage_layer<H, true, WF_BLOCK>(th, g.woff[i], lds + M::FR + (i - 1) * W::FRAG, tid);
        for (int j = tid; j < H; j += WF_BLOCK) lds[M::BI + (i - 1) * H + j] = th[g.boff[i] + j];
    }
    for (int f = tid; f < (D + 2) * H; f += WF_BLOCK) {
        const int c = f / H, j = f - c * H;
        lds[M::W1 + f] = th[(c < D ? g.woff[0] + c * H : (c == D ? g.boff[0] : g.woff[L])) + j];
    }
    __syncthreads();
    const double bo = th[g.boff[L]];
    const double* W1 = lds + M::W1;

    const long tstride = (SAVE == 2 || g.save_act) ? (long)L * (NS * KS * 64) : 0;     // doubles per tile of the activation store
    for (long tile = wave; tile < g.ntiles; tile += nwaves) {
        const long p = tile * 16 + pt;
        const bool valid = p < g.N;
        double x[D];
#pragma unroll
        for (int c = 0; c < D; ++c) x[c] = valid ? xn[c] : 0.0;
        {
            long pn = (tile + nwaves) * 16 + pt;
            pn = pn < g.N ? pn : g.N - 1;
#pragma unroll
            for (int c = 0; c < D; ++c) xn[c] = g.X[(long)c * g.N + pn];
        }
        int lofs = lane;                       // opaque per tile: keeps the LDS fragment reads inside the loop
        asm volatile("" : "+v"(lofs));
        double h[C][KS];
        // forward-only launches (predict, loss read-back: g.save_act == 0) run THIS instantiation with every tile's stores aliased onto
        // tile 0's block of the store (L2-resident, never read) instead of a third compile-time variant or a branch around each store
        // group (tried: the branch cost the training launch 4-12 %, profiles/r05_notes.md)
        double* sv = g.ACTS + (tile * tstride) + lane;

        // ---- layer 1 (VALU): z = b + x W, z_c = W[c,:], z_cc = 0 ----
        {
            double z1[KS];
#pragma unroll
            for (int s = 0; s < KS; ++s) {
                double z = W1[D * H + 4 * s + (lofs >> 4)];
#pragma unroll
                for (int c = 0; c < D; ++c) z = fma(x[c], W1[c * H + 4 * s + (lofs >> 4)], z);
                z1[s] = z;
            }
            auto act1 = [&](auto fast) {
#pragma unroll
                for (int s = 0; s < KS; ++s) {
                    double a, a1, a2;
                    act_fwd<ACT, decltype(fast)::value>(z1[s], a, a1, a2);
                    h[0][s] = a;
                    {   // (stored always: forward-only launches alias ONE tile of the store, see sv)
                        sv[(0 * KS + s) * 64] = a;
                        if constexpr (ACT == HPV_ACT_SIN) sv[(SA1 * KS + s) * 64] = a1;
                    }
#pragma unroll
                    for (int t = 0; t < NT1; ++t) h[1 + t][s] = a1 * W1[(t < D ? t : 0) * H + 4 * s + (lofs >> 4)];   // T1 = coordinates 0..NT1-1
#pragma unroll
                    for (int b = 0; b < NT2; ++b) {
                        const double zc = W1[(b < D ? b : 0) * H + 4 * s + (lofs >> 4)];                            // T2 = coordinates 0..NT2-1
                        if constexpr (T2Mix<NT1, NT2>::value) h[1 + NT1 + b][s] = a2 * t2_square<NT1, NT2>(g.t2w, b, W1[4 * s + (lofs >> 4)], W1[(D > 1 ? 1 : 0) * H + 4 * s + (lofs >> 4)]);
                        else h[1 + NT1 + b][s] = a2 * zc * zc;
                    }
                }
            };
            if (act_wave_needs_safe<ACT>(z1)) act1(std::false_type{}); else act1(std::true_type{});
        }
        // ---- hidden -> hidden layers (MFMA) ----
#pragma unroll
        for (int i = 1; i < L; ++i) {
            double z[C][KS];
            wide_fwd_layer<H, C>(lds + M::FR + (i - 1) * W::FRAG, lds + M::BI + (i - 1) * H, lofs, h, z);
            double* svl = sv + (long)i * (NS * KS * 64);
            auto acti = [&](auto fast) {
#pragma unroll
                for (int s = 0; s < KS; ++s) {
                    double a, a1, a2;
                    act_fwd<ACT, decltype(fast)::value>(z[0][s], a, a1, a2);
                    h[0][s] = a;
                    {   // (stored always: forward-only launches alias ONE tile of the store, see sv)
                        svl[(0 * KS + s) * 64] = a;
                        if constexpr (ACT == HPV_ACT_SIN) svl[(SA1 * KS + s) * 64] = a1;
                    }
#pragma unroll
                    for (int u = 0; u < NT1; ++u) {
                        if constexpr (SAVE_T) svl[((SZC + u) * KS + s) * 64] = z[1 + u][s];
                        h[1 + u][s] = a1 * z[1 + u][s];
                    }
#pragma unroll
                    for (int b = 0; b < NT2; ++b) {
                        const double zcc = z[1 + NT1 + b][s], zc1 = z[1 + (b < NT1 ? b : 0)][s];
                        if constexpr (SAVE_T) svl[((SZCC + b) * KS + s) * 64] = zcc;
                        if constexpr (T2Mix<NT1, NT2>::value) h[1 + NT1 + b][s] = a2 * t2_square<NT1, NT2>(g.t2w, b, z[1][s], z[NT1 > 1 ? 2 : 1][s]) + a1 * zcc;
                        else h[1 + NT1 + b][s] = a2 * zc1 * zc1 + a1 * zcc;
                    }
                }
            };
            if (act_wave_needs_safe<ACT>(z[0])) acti(std::false_type{}); else acti(std::true_type{});
        }
        // ---- linear head (VALU + the cross-row sums over the 4 neuron groups) ----
#pragma unroll
        for (int ch = 0; ch < C; ++ch) {
            double v = 0.0;
#pragma unroll
            for (int s = 0; s < KS; ++s) v = fma(h[ch][s], W1[(D + 1) * H + 4 * s + (lofs >> 4)], v);
            v = xrow_sum16(v);
            v = xrow_sum32(v);
            if (ch == 0) v += bo;
            if (q == 0 && valid) g.OUT[(long)ch * g.N + p] = v;
            if (ch == 0 && g.data_off >= 0 && tile * 16 >= g.data_off) {
                // lossb = w mean((u_d - u)^2) (P1:98, P2:122, P3:184): adjoint + per-tile partial sum, no extra launch
                double dd = 0.0;
                if (q == 0 && valid) {
                    dd = g.ud[p - g.data_off] - v;
                    if (g.data_write_gbar) g.gbar0[p] = g.data_scale * dd;
                }
                const double sq = row_sum16(dd * dd);
                if (lane == 0) g.data_part[tile - g.data_off / 16] = sq;
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------
// reverse
// ------------------------------------------------------------------------------------------------
template <int D, int NT1, int NT2, int ACT, int L, int H>
__global__ void __launch_bounds__(WF_BLOCK, 1) k_bwd_wide(MfmaArgs g) {
    using W = WD<H>;
    using M = WideBwdLds<H, L, D>;
    constexpr int KS = W::KS, C = 1 + NT1 + NT2, LH = L > 1 ? L - 1 : 1;
    constexpr int NS = SlotCount<ACT, NT1, NT2>::value;
    constexpr int SZC = 1 + (ACT == HPV_ACT_SIN ? 1 : 0), SZCC = SZC + NT1;
    extern __shared__ __attribute__((aligned(16))) double lds[];
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int q = lane >> 4, pt = lane & 15;
    const long wave = ((long)blockIdx.x * blockDim.x + tid) >> 6;
    const long nwaves = ((long)gridDim.x * blockDim.x) >> 6;
    const double* __restrict__ th = g.theta;
#pragma unroll
    for (int i = 1; i < L; ++i) wide_stage_layer<H, false, WF_BLOCK>(th, g.woff[i], lds + M::FR + (i - 1) * W::FRAG, tid);
    for (int f = tid; f < (D + 1) * H; f += WF_BLOCK) {
        const int c = f / H, j = f - c * H;
        lds[M::W1 + f] = th[(c < D ? g.woff[0] + c * H : g.woff[L]) + j];
    }
    __syncthreads();
    const double* W1 = lds + M::W1;
    double* TA = lds + M::TAB + wv * (2 * W::TR);
    double* TB = TA + W::TR;

    // gradient accumulators of this wave over all its tiles
    WideDW<H> dW[LH];
#pragma unroll
    for (int i = 0; i < LH; ++i) dW[i].zero();
    double db[L][KS], dW1[D][KS], dWo[KS], dbo = 0.0;
#pragma unroll
    for (int s = 0; s < KS; ++s) {
        dWo[s] = 0.0;
#pragma unroll
        for (int i = 0; i < L; ++i) db[i][s] = 0.0;
#pragma unroll
        for (int c = 0; c < D; ++c) dW1[c][s] = 0.0;
    }
    struct Slots {
        double a[KS], a1s[KS];
        double zc[NT1 > 0 ? NT1 : 1][KS];
        double zcc[NT2 > 0 ? NT2 : 1][KS];
    };
    int lofs = lane;
    auto load_slots = [&](const double* svl, bool first_layer, Slots& S) {
#pragma unroll
        for (int s = 0; s < KS; ++s) {
            S.a[s] = svl[(0 * KS + s) * 64];
            if constexpr (ACT == HPV_ACT_SIN) S.a1s[s] = svl[(1 * KS + s) * 64]; else S.a1s[s] = 0.0;
#pragma unroll
            for (int u = 0; u < NT1; ++u) S.zc[u][s] = first_layer ? W1[(u < D ? u : 0) * H + 4 * s + (lofs >> 4)] : svl[((SZC + u) * KS + s) * 64];
#pragma unroll
            for (int b = 0; b < NT2; ++b) S.zcc[b][s] = first_layer ? 0.0 : svl[((SZCC + b) * KS + s) * 64];
        }
    };
    auto outputs_of = [&](const Slots& S, int ch, double (&hv)[KS]) {   // channel ch of the layer's outputs
#pragma unroll
        for (int s = 0; s < KS; ++s) {
            double a1, a2, a3;
            act_saved<ACT>(S.a[s], S.a1s[s], a1, a2, a3);
            if (ch == 0) hv[s] = S.a[s];
            else if (ch <= NT1) hv[s] = a1 * S.zc[(ch - 1) < NT1 ? (ch - 1) : 0][s];
            else {
                const int b = ch - 1 - NT1;
                const double z1 = S.zc[b < NT1 ? b : 0][s];
                if constexpr (T2Mix<NT1, NT2>::value) hv[s] = a2 * t2_square<NT1, NT2>(g.t2w, b, S.zc[0][s], S.zc[NT1 > 1 ? 1 : 0][s]) + a1 * S.zcc[0][s];
                else hv[s] = a2 * z1 * z1 + a1 * S.zcc[b < NT2 ? b : 0][s];
            }
        }
    };

    for (long tile = wave; tile < g.ntiles; tile += nwaves) {
        lofs = lane;
        asm volatile("" : "+v"(lofs));   // opaque per tile: the LDS weight reads stay inside the loop
        const long p = tile * 16 + pt;
        const bool valid = p < g.N;
        double x[D], gb[C];
#pragma unroll
        for (int c = 0; c < D; ++c) x[c] = valid ? g.X[(long)c * g.N + p] : 0.0;
#pragma unroll
        for (int ch = 0; ch < C; ++ch) gb[ch] = valid ? g.GBAR[(long)ch * g.N + p] : 0.0;
        const double* sv = g.ACTS + (tile * L) * (long)(NS * KS * 64) + lane;
        Slots cur;
        load_slots(sv + (long)(L - 1) * (NS * KS * 64), L == 1, cur);

        double hbar[C][KS], zbar[C][KS];
        // ---- linear head ----
#pragma unroll
        for (int ch = 0; ch < C; ++ch) {
            double hv[KS];
            outputs_of(cur, ch, hv);
#pragma unroll
            for (int s = 0; s < KS; ++s) {
                dWo[s] = fma(hv[s], gb[ch], dWo[s]);
                hbar[ch][s] = gb[ch] * W1[D * H + 4 * s + (lofs >> 4)];
            }
        }
        if (q == 0) dbo += gb[0];
        // ---- hidden layers, last to first ----
#pragma unroll
        for (int i = L - 1; i >= 0; --i) {
            Slots prev;
            if (i > 0) load_slots(sv + (long)(i - 1) * (NS * KS * 64), i == 1, prev);   // one layer ahead
#pragma unroll
            for (int s = 0; s < KS; ++s) {
                double a1, a2, a3;
                act_saved<ACT>(cur.a[s], cur.a1s[s], a1, a2, a3);
                double zb = hbar[0][s] * a1;
#pragma unroll
                for (int u = 0; u < NT1; ++u) {
                    zbar[1 + u][s] = hbar[1 + u][s] * a1;
                    zb += hbar[1 + u][s] * a2 * cur.zc[u][s];
                }
#pragma unroll
                for (int b = 0; b < NT2; ++b) {
                    const int u = b < NT1 ? b : 0;
                    const double hb = hbar[1 + NT1 + b][s];
                    zbar[1 + NT1 + b][s] = hb * a1;
                    if constexpr (T2Mix<NT1, NT2>::value) {      // the mixed second tangent rides on both first tangents
                        zbar[1][s] += 2.0 * hb * a2 * g.t2w[0] * cur.zc[0][s];
                        zbar[2][s] += 2.0 * hb * a2 * g.t2w[1] * cur.zc[NT1 > 1 ? 1 : 0][s];
                        zb += hb * (a3 * t2_square<NT1, NT2>(g.t2w, b, cur.zc[0][s], cur.zc[NT1 > 1 ? 1 : 0][s]) + a2 * cur.zcc[b][s]);
                    } else {
                        zbar[1 + u][s] += 2.0 * hb * a2 * cur.zc[u][s];
                        zb += hb * (a3 * cur.zc[u][s] * cur.zc[u][s] + a2 * cur.zcc[b][s]);
                    }
                }
                zbar[0][s] = zb;
                db[i][s] += zb;
            }
            if (i == 0) {
                // dW1[c][j] += x_c zbar[j] + [c in T1] zbar_c[j]
#pragma unroll
                for (int s = 0; s < KS; ++s) {
#pragma unroll
                    for (int c = 0; c < D; ++c) dW1[c][s] += x[c] * zbar[0][s];
#pragma unroll
                    for (int u = 0; u < NT1; ++u) dW1[u < D ? u : 0][s] += zbar[1 + u][s];
                }
            } else {
                // weight gradient: contraction over the tile's 16 points and over the channels; one channel's transpose pair in
                // LDS at a time (the pairs of all channels of a 64-wide layer would be 4 x 87 KB)
#pragma unroll
                for (int ch = 0; ch < C; ++ch) {
                    double hv[KS];
                    outputs_of(prev, ch, hv);
                    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
                    __builtin_amdgcn_wave_barrier();
                    wide_transpose_store<H>(TA, TB, q, pt, hv, zbar[ch]);
                    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
                    __builtin_amdgcn_wave_barrier();
                    wide_dw_accumulate<H>(TA, TB, lane, dW[i - 1]);
                }
                // hbar_in^T = W zbar^T
#pragma unroll
                for (int ch = 0; ch < C; ++ch) wide_hbar<H>(lds + M::FR + (i - 1) * W::FRAG, lofs, zbar[ch], hbar[ch]);
                cur = prev;
            }
        }
    }

    // ---- epilogue: per-wave partials -> (LDS exchange, one accumulator group at a time) -> one gradient row per workgroup ----
    // (per-wave rows in LDS, as k_bwd_mfma keeps them, do not fit: 4 x P doubles is 100 KB at H = 40 and 400 KB at H = 64)
    wide_epilogue<H, L, D, WF_WAVES>(lds + M::TAB, dW, db, dW1, dWo, dbo, g.GPART + (long)blockIdx.x * g.P, g.woff, g.boff);
}

// ------------------------------------------------------------------------------------------------
// reverse, tangents recomputed (the store holds s -- and cos -- only; k_fwd_wide<.., SAVE = 2>)
// ------------------------------------------------------------------------------------------------
template <int H, int L, int D>
struct WideBwdRcLds {
    using W = WD<H>;
    static constexpr int LH = L - 1;
    static constexpr int TAB = 0;                               // per-wave transpose pair [WAVES][2][H][17]; epilogue: exchange
    static constexpr int FR = TAB + WF_WAVES * 2 * W::TR;       // reverse fragments        [LH][FRAG]
    static constexpr int FF = FR + LH * W::FRAG;                // forward fragments (the recompute)  [LH][FRAG]
    static constexpr int W1 = FF + LH * W::FRAG;                // first-layer rows, head weights  [D + 1][H]
    static constexpr int TOTAL = W1 + (D + 1) * H;
};

template <int D, int NT1, int NT2, int ACT, int L, int H>
__global__ void __launch_bounds__(WF_BLOCK, 1) k_bwd_wide_rc(MfmaArgs g) {
    using W = WD<H>;
    using M = WideBwdRcLds<H, L, D>;
    constexpr int KS = W::KS, C = 1 + NT1 + NT2, CT = NT1 + NT2, LH = L > 1 ? L - 1 : 1;
    constexpr int NS = ACT == HPV_ACT_SIN ? 2 : 1;
    extern __shared__ __attribute__((aligned(16))) double lds[];
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int q = lane >> 4, pt = lane & 15;
    const long wave = ((long)blockIdx.x * blockDim.x + tid) >> 6;
    const long nwaves = ((long)gridDim.x * blockDim.x) >> 6;
    const double* __restrict__ th = g.theta;
#pragma unroll
    for (int i = 1; i < L; ++i) {
        wide_stage_layer<H, false, WF_BLOCK>(th, g.woff[i], lds + M::FR + (i - 1) * W::FRAG, tid);
        wide_stage_layer<H, true, WF_BLOCK>(th, g.woff[i], lds + M::FF + (i - 1) * W::FRAG, tid);
    }
    for (int f = tid; f < (D + 1) * H; f += WF_BLOCK) {
        const int c = f / H, j = f - c * H;
        lds[M::W1 + f] = th[(c < D ? g.woff[0] + c * H : g.woff[L]) + j];
    }
    __syncthreads();
    const double* W1 = lds + M::W1;
    double* TA = lds + M::TAB + wv * (2 * W::TR);
    double* TB = TA + W::TR;
    WideDW<H> dW[LH];
#pragma unroll
    for (int i = 0; i < LH; ++i) dW[i].zero();
    double db[L][KS], dW1[D][KS], dWo[KS], dbo = 0.0;
#pragma unroll
    for (int s = 0; s < KS; ++s) {
        dWo[s] = 0.0;
#pragma unroll
        for (int i = 0; i < L; ++i) db[i][s] = 0.0;
#pragma unroll
        for (int c = 0; c < D; ++c) dW1[c][s] = 0.0;
    }
    for (long tile = wave; tile < g.ntiles; tile += nwaves) {
        int lofs = lane;
        asm volatile("" : "+v"(lofs));   // opaque per tile: the LDS weight reads stay inside the loop
        const int ql = lofs >> 4;
        const long p = tile * 16 + pt;
        const bool valid = p < g.N;
        double x[D], gb[C];
#pragma unroll
        for (int c = 0; c < D; ++c) x[c] = valid ? g.X[(long)c * g.N + p] : 0.0;
#pragma unroll
        for (int ch = 0; ch < C; ++ch) gb[ch] = valid ? g.GBAR[(long)ch * g.N + p] : 0.0;
        // s (and cos) of every hidden layer: ONE batch of coalesced loads per tile
        const double* sv = g.ACTS + (tile * L) * (long)(NS * KS * 64) + lane;
        double S[L][KS], S1[ACT == HPV_ACT_SIN ? L : 1][KS];
#pragma unroll
        for (int i = 0; i < L; ++i)
#pragma unroll
            for (int s = 0; s < KS; ++s) {
                S[i][s] = sv[(long)i * (NS * KS * 64) + s * 64];
                if constexpr (ACT == HPV_ACT_SIN) S1[i][s] = sv[(long)i * (NS * KS * 64) + (KS + s) * 64];
            }
        // tangent pre-activations of every layer, recomputed on the MFMA pipe
        double zc[L][NT1 > 0 ? NT1 : 1][KS], zcc[L][NT2 > 0 ? NT2 : 1][KS];
#pragma unroll
        for (int s = 0; s < KS; ++s) {
#pragma unroll
            for (int u = 0; u < NT1; ++u) zc[0][u][s] = W1[(u < D ? u : 0) * H + 4 * s + ql];
#pragma unroll
            for (int b = 0; b < NT2; ++b) zcc[0][b][s] = 0.0;
        }
        if constexpr (CT > 0) {
            double hc[CT > 0 ? CT : 1][KS];
#pragma unroll
            for (int i = 0; i < L - 1; ++i) {
#pragma unroll
                for (int s = 0; s < KS; ++s) {
                    double a1, a2, a3;
                    act_saved<ACT>(S[i][s], ACT == HPV_ACT_SIN ? S1[ACT == HPV_ACT_SIN ? i : 0][s] : 0.0, a1, a2, a3);
#pragma unroll
                    for (int u = 0; u < NT1; ++u) hc[u][s] = a1 * zc[i][u][s];
#pragma unroll
                    for (int b = 0; b < NT2; ++b) {
                        const double z1 = zc[i][b < NT1 ? b : 0][s];
                        if constexpr (T2Mix<NT1, NT2>::value) hc[NT1 + b][s] = a2 * t2_square<NT1, NT2>(g.t2w, b, zc[i][0][s], zc[i][NT1 > 1 ? 1 : 0][s]) + a1 * zcc[i][b][s];
                        else hc[NT1 + b][s] = a2 * z1 * z1 + a1 * zcc[i][b][s];
                    }
                }
                double zt[CT > 0 ? CT : 1][KS];
                wide_fwd_layer<H, (CT > 0 ? CT : 1), false>(lds + M::FF + i * W::FRAG, nullptr, lofs, hc, zt);
#pragma unroll
                for (int s = 0; s < KS; ++s) {
#pragma unroll
                    for (int u = 0; u < NT1; ++u) zc[i + 1][u][s] = zt[u][s];
#pragma unroll
                    for (int b = 0; b < NT2; ++b) zcc[i + 1][b][s] = zt[NT1 + b][s];
                }
            }
        }
        auto outputs_of = [&](int i, int ch, double (&hv)[KS]) {   // channel ch of layer i's outputs
#pragma unroll
            for (int s = 0; s < KS; ++s) {
                double a1, a2, a3;
                act_saved<ACT>(S[i][s], ACT == HPV_ACT_SIN ? S1[ACT == HPV_ACT_SIN ? i : 0][s] : 0.0, a1, a2, a3);
                if (ch == 0) hv[s] = S[i][s];
                else if (ch <= NT1) hv[s] = a1 * zc[i][(ch - 1) < NT1 ? (ch - 1) : 0][s];
                else {
                    const int b = ch - 1 - NT1;
                    const double z1 = zc[i][b < NT1 ? b : 0][s];
                    if constexpr (T2Mix<NT1, NT2>::value) hv[s] = a2 * t2_square<NT1, NT2>(g.t2w, b, zc[i][0][s], zc[i][NT1 > 1 ? 1 : 0][s]) + a1 * zcc[i][0][s];
                    else hv[s] = a2 * z1 * z1 + a1 * zcc[i][b < NT2 ? b : 0][s];
                }
            }
        };
        double hbar[C][KS], zbar[C][KS];
#pragma unroll
        for (int ch = 0; ch < C; ++ch) {
            double hv[KS];
            outputs_of(L - 1, ch, hv);
#pragma unroll
            for (int s = 0; s < KS; ++s) {
                dWo[s] = fma(hv[s], gb[ch], dWo[s]);
                hbar[ch][s] = gb[ch] * W1[D * H + 4 * s + ql];
            }
        }
        if (q == 0) dbo += gb[0];
#pragma unroll
        for (int i = L - 1; i >= 0; --i) {
#pragma unroll
            for (int s = 0; s < KS; ++s) {
                double a1, a2, a3;
                act_saved<ACT>(S[i][s], ACT == HPV_ACT_SIN ? S1[ACT == HPV_ACT_SIN ? i : 0][s] : 0.0, a1, a2, a3);
                double zb = hbar[0][s] * a1;
#pragma unroll
                for (int u = 0; u < NT1; ++u) {
                    zbar[1 + u][s] = hbar[1 + u][s] * a1;
                    zb += hbar[1 + u][s] * a2 * zc[i][u][s];
                }
#pragma unroll
                for (int b = 0; b < NT2; ++b) {
                    const int u = b < NT1 ? b : 0;
                    const double hb = hbar[1 + NT1 + b][s];
                    zbar[1 + NT1 + b][s] = hb * a1;
                    if constexpr (T2Mix<NT1, NT2>::value) {
                        zbar[1][s] += 2.0 * hb * a2 * g.t2w[0] * zc[i][0][s];
                        zbar[2][s] += 2.0 * hb * a2 * g.t2w[1] * zc[i][NT1 > 1 ? 1 : 0][s];
                        zb += hb * (a3 * t2_square<NT1, NT2>(g.t2w, b, zc[i][0][s], zc[i][NT1 > 1 ? 1 : 0][s]) + a2 * zcc[i][b][s]);
                    } else {
                        zbar[1 + u][s] += 2.0 * hb * a2 * zc[i][u][s];
                        zb += hb * (a3 * zc[i][u][s] * zc[i][u][s] + a2 * zcc[i][b][s]);
                    }
                }
                zbar[0][s] = zb;
                db[i][s] += zb;
            }
            if (i == 0) {
#pragma unroll
                for (int s = 0; s < KS; ++s) {
#pragma unroll
                    for (int c = 0; c < D; ++c) dW1[c][s] += x[c] * zbar[0][s];
#pragma unroll
                    for (int u = 0; u < NT1; ++u) dW1[u < D ? u : 0][s] += zbar[1 + u][s];
                }
            } else {
#pragma unroll
                for (int ch = 0; ch < C; ++ch) {
                    double hv[KS];
                    outputs_of(i - 1, ch, hv);
                    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
                    __builtin_amdgcn_wave_barrier();
                    wide_transpose_store<H>(TA, TB, q, pt, hv, zbar[ch]);
                    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
                    __builtin_amdgcn_wave_barrier();
                    wide_dw_accumulate<H>(TA, TB, lane, dW[i - 1]);
                }
#pragma unroll
                for (int ch = 0; ch < C; ++ch) wide_hbar<H>(lds + M::FR + (i - 1) * W::FRAG, lofs, zbar[ch], hbar[ch]);
            }
        }
    }
    wide_epilogue<H, L, D, WF_WAVES>(lds + M::TAB, dW, db, dW1, dWo, dbo, g.GPART + (long)blockIdx.x * g.P, g.woff, g.boff);
}

// ------------------------------------------------------------------------------------------------
// host side
// ------------------------------------------------------------------------------------------------
template <typename K>
static bool wide_set_lds(K kernel, size_t bytes) {
    if (bytes > 160 * 1024) return false;
    if (bytes <= 64 * 1024) return true;
    const hipError_t e = hipFuncSetAttribute((const void*)kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
    if (e != hipSuccess) { (void)hipGetLastError(); return false; }
    return true;
}

template <int D, int NT1, int NT2, int ACT, int L, int H>
static void run_fwd_wide(const MfmaArgs& a, int blocks, hipStream_t s) {
    constexpr size_t bytes = (size_t)WideFwdLds<H, L, D>::TOTAL * sizeof(double);
#ifdef HPV_EXPERIMENTS
    if constexpr (H <= 32 && L >= 2 && NT1 + NT2 > 0) {
        if (a.save_act == 2) { hipLaunchKernelGGL((k_fwd_wide<D, NT1, NT2, ACT, L, H, 2>), dim3(blocks), dim3(WF_BLOCK), bytes, s, a); return; }
    }
#endif
    hipLaunchKernelGGL((k_fwd_wide<D, NT1, NT2, ACT, L, H, 1>), dim3(blocks), dim3(WF_BLOCK), bytes, s, a);
}
template <int D, int NT1, int NT2, int ACT, int L, int H>
static void run_bwd_wide(const MfmaArgs& a, int blocks, hipStream_t s) {
    constexpr size_t bytes = (size_t)WideBwdLds<H, L, D>::TOTAL * sizeof(double);
    hipLaunchKernelGGL((k_bwd_wide<D, NT1, NT2, ACT, L, H>), dim3(blocks), dim3(WF_BLOCK), bytes, s, a);
}

template <int D, int NT1, int NT2, int ACT, int L, int H>
static void run_bwd_wide_rc(const MfmaArgs& a, int blocks, hipStream_t s) {
    constexpr size_t bytes = (size_t)WideBwdRcLds<H, L, D>::TOTAL * sizeof(double);
    hipLaunchKernelGGL((k_bwd_wide_rc<D, NT1, NT2, ACT, L, H>), dim3(blocks), dim3(WF_BLOCK), bytes, s, a);
}

template <int D, int NT1, int NT2, int ACT, int L, int H>
static bool pick_wide(HpvMfma* m) {
    constexpr size_t fb = (size_t)WideFwdLds<H, L, D>::TOTAL * sizeof(double), bb = (size_t)WideBwdLds<H, L, D>::TOTAL * sizeof(double);
    if (!wide_set_lds(k_fwd_wide<D, NT1, NT2, ACT, L, H, 1>, fb) ||
        !wide_set_lds(k_bwd_wide<D, NT1, NT2, ACT, L, H>, bb))
        return false;
    m->fwd = run_fwd_wide<D, NT1, NT2, ACT, L, H>;
    m->bwd = run_bwd_wide<D, NT1, NT2, ACT, L, H>;
    m->store_s_only = false;
    // s-only store + tangent recompute in the reverse kernel (where tangents exist, H <= 32, forward fragments fit beside the
    // reverse ones): OPT-IN, HPV_WIDE_RC=1.  Measured slower than the full store on every case tried (round 4, us per iteration,
    // recompute / full store: config-4 grid H = 24 125.1 / 112.9, H = 32 192.9 / 149.2; 1-D 16 elements [1,32,32,32,32,1]
    // 60.2 / 48.6): the forward kernel's store shrinks to a third, but the reverse kernel -- the compiler-scheduled recompute at
    // one wave per SIMD, 380 B of scratch at H = 32 -- loses more than the forward kernel gains.
#ifdef HPV_EXPERIMENTS      // (libhpvpinn_testhooks.so: width 24 only, csrc/build.sh)
    if constexpr (H <= 32 && L >= 2 && NT1 + NT2 > 0) {
        constexpr size_t rb = (size_t)WideBwdRcLds<H, L, D>::TOTAL * sizeof(double);
        const char* e = getenv("HPV_WIDE_RC");
        if ((e && e[0] == '1') && wide_set_lds(k_fwd_wide<D, NT1, NT2, ACT, L, H, 2>, fb) && wide_set_lds(k_bwd_wide_rc<D, NT1, NT2, ACT, L, H>, rb)) {
            m->bwd = run_bwd_wide_rc<D, NT1, NT2, ACT, L, H>;
            m->store_s_only = true;
        }
    }
#endif
    m->bwd_fused = nullptr;
    int of = 1, ob = 1;
    (void)hipOccupancyMaxActiveBlocksPerMultiprocessor(&of, k_fwd_wide<D, NT1, NT2, ACT, L, H, 1>, WF_BLOCK, fb);
    (void)hipOccupancyMaxActiveBlocksPerMultiprocessor(&ob, k_bwd_wide<D, NT1, NT2, ACT, L, H>, WF_BLOCK, bb);
    m->occ_fwd = of > 0 ? of : 1;
    m->occ_bwd = ob > 0 ? ob : 1;
    const char* an = ACT == HPV_ACT_SIN ? "sin" : "tanh";
    snprintf(m->vfwd, sizeof m->vfwd, "k_fwd_wide<D=%d,NT1=%d,NT2=%d,%s,L=%d,H=%d>", D, NT1, NT2, an, L, H);
#ifdef HPV_EXPERIMENTS
    const char* bn = m->store_s_only ? "k_bwd_wide_rc" : "k_bwd_wide";
#else
    const char* bn = "k_bwd_wide";
#endif
    snprintf(m->vbwd, sizeof m->vbwd, "%s<D=%d,NT1=%d,NT2=%d,%s,L=%d,H=%d>", bn, D, NT1, NT2, an, L, H);
    return true;
}
template <int D, int NT1, int NT2, int ACT, int H>
static bool pick_wide_L(HpvMfma* m, int L) {
    switch (L) {
        case 1: return pick_wide<D, NT1, NT2, ACT, 1, H>(m);
        case 2: return pick_wide<D, NT1, NT2, ACT, 2, H>(m);
        case 3: return pick_wide<D, NT1, NT2, ACT, 3, H>(m);
        case 4: return pick_wide<D, NT1, NT2, ACT, 4, H>(m);
        // (deeper networks at the narrower widths only: the weight fragments of five layers of 40 and more do not fit the LDS)
        case 5: if constexpr (H <= 32) return pick_wide<D, NT1, NT2, ACT, 5, H>(m); else return false;
        case 6: if constexpr (H <= 32) return pick_wide<D, NT1, NT2, ACT, 6, H>(m); else return false;
        default: return false;
    }
}
template <int H, int D>
static bool pick_wide_key(HpvMfma* m, int key, int act, int L) {
    if constexpr (D == 1) {
        if (act != HPV_ACT_SIN) return false;
        if (key == 111) return pick_wide_L<1, 1, 1, HPV_ACT_SIN, H>(m, L);
        if (key == 110) return pick_wide_L<1, 1, 0, HPV_ACT_SIN, H>(m, L);
        if (key == 100) return pick_wide_L<1, 0, 0, HPV_ACT_SIN, H>(m, L);
        return false;
    } else {
        if (act == HPV_ACT_SIN) return false;
        if (key == 222) return pick_wide_L<2, 2, 2, HPV_ACT_TANH, H>(m, L);
        if (key == 220) return pick_wide_L<2, 2, 0, HPV_ACT_TANH, H>(m, L);
        if (key == 200) return pick_wide_L<2, 0, 0, HPV_ACT_TANH, H>(m, L);
        if (key == 221) return pick_wide_L<2, 2, 1, HPV_ACT_TANH, H>(m, L);
        return false;
    }
}

// One translation unit per hidden width AND input dimension (csrc/build.sh compiles this file once per entry of WIDE_WIDTHS x {1, 2}
// with -DHPV_WIDE_H=<H> -DHPV_WIDE_D=<D>: 3 or 4 channel sets x up to 6 depths x 3 kernels each, in parallel -- the 64-wide unit
// alone took 94 s as one file); kernels_mfma.hip dispatches on H and the activation (hpv_wide_pick).
#if !defined(HPV_WIDE_H) || !defined(HPV_WIDE_D)
#error "compile with -DHPV_WIDE_H=<hidden width> -DHPV_WIDE_D=<1|2> (csrc/build.sh)"
#endif
#define WIDE_CAT4_(a, b, c, d) a##b##c##d
#define WIDE_CAT4(a, b, c, d) WIDE_CAT4_(a, b, c, d)
bool WIDE_CAT4(hpv_wide_pick_, HPV_WIDE_H, _d, HPV_WIDE_D)(HpvMfma* m, int key, int act, int L) { return pick_wide_key<HPV_WIDE_H, HPV_WIDE_D>(m, key, act, L); }
