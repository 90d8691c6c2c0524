// MFMA fast path: Taylor-mode MLP forward / reverse for 20-wide hidden layers on gfx950 using
// v_mfma_f64_16x16x4_f64 and v_mfma_f64_4x4x4_4b_f64.
//
// Formulation (everything "transposed" so layer outputs feed the next MFMA straight from registers):
//   one wavefront owns a tile of 16 quadrature points.  For a hidden->hidden layer
//       Z^T[out][pt] = W^T[out][in] * H^T[in][pt]          (M = out = 20 = 16 + 4, K = in = 20 = 5 k-steps, N = 16 points)
//   16x16x4:  A (lane l: A[m = l&15][k = l>>4]) = W^T fragments (LDS, lane-major),
//             B (lane l: B[k = l>>4][n = l&15]) = H^T -> lane holds neuron 4s+(l>>4), point l&15 for k-step s,
//             D (lane l, reg r: row (l>>4)+4r, col l&15) -> neuron 4r+(l>>4), point l&15:
//             D register r is exactly the B operand of k-step r of the next layer: no LDS, no shuffles.
//   4x4x4_4b (four independent 4x4x4 blocks b = (l&15)>>2; A_b[i = l&3][k = l>>4], B_b[k = l>>4][j = l&3],
//             D_b[i = l>>4][j = l&3]; layout probed on the device, scripts/mfma_4x4_probe.hip): the remaining output
//             neurons 16..19 with the four blocks = the four groups of 4 points; B is the SAME register as for the
//             16x16x4 tile and D lands on the lane's fifth value (neuron 16+(l>>4), point l&15) -- a second 16-row
//             tile would be 3/4 padding, and fp64 MFMA / VALU share one datapath on this chip, so padding is pure loss.
//   Each lane therefore carries 5 useful values per channel per layer (neurons 4s+q, s=0..4, q = l>>4)
//   for point pt = l&15, and all activation math runs on full 64-lane VALU instructions.
//   The reverse pass uses the same chaining for hbar_in^T = W * zbar^T; only the weight gradient
//   dW[in][out] = sum_pt h_in[pt][in] zbar[pt][out] contracts over points, which needs the operands in the
//   other orientation -> one per-wave LDS transpose per channel and layer; dW = one 16x16 tile (16x16x4) + the
//   4x16 and 16x4 strips and the 4x4 corner on 4x4x4_4b.
//
// First layer (d -> 20) and linear head (20 -> 1) are VALU work (K = 1..2 and M = 1 are no MFMA shapes).
#include <cstdio>
#include <cstdlib>
#include <type_traits>

#include "hpv_mfma_dev.h"

// ------------------------------------------------------------------------------------------------
// forward
// ------------------------------------------------------------------------------------------------
// SAVE: the activation store for the reverse pass is a compile-time choice -- a run-time flag put a scalar branch
// around every store, which cut each layer's activation math into one basic block per value (no interleaving of
// the five independent tanh chains of a lane).
template <int D, int NT1, int NT2, int ACT, int L, bool SAVE = true>
__global__ void __launch_bounds__(MF_BLOCK, 2) k_fwd_mfma(MfmaArgs g) {
    constexpr int BLK = MF_BLOCK;
    constexpr int C = 1 + NT1 + NT2;
    constexpr int NS = SlotCount<ACT, NT1, NT2>::value;
    constexpr int SA1 = 1;                                   // slot of A1 (sin only)
    constexpr int SZC = 1 + (ACT == HPV_ACT_SIN ? 1 : 0);    // first ZC slot
    constexpr int SZCC = SZC + NT1;
    const int lane = threadIdx.x & 63;
    const int q = lane >> 4, pt = lane & 15;
    const long wave = ((long)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const long nwaves = ((long)gridDim.x * blockDim.x) >> 6;
    const long tile_end = g.ntiles;
    const double* __restrict__ th = g.theta;

    // per-lane weight fragments
    double w1[D][MF_KS], b1[MF_KS], wo[MF_KS];
#pragma unroll
    for (int s = 0; s < MF_KS; ++s) {
        const int j = 4 * s + q;
#pragma unroll
        for (int c = 0; c < D; ++c) w1[c][s] = th[g.woff[0] + c * MF_H + j];
        b1[s] = th[g.boff[0] + j];
        wo[s] = th[g.woff[L] + j];
    }
    // the coordinates of the NEXT tile are fetched while this one computes (2 doubles per lane): otherwise every
    // tile starts with a full HBM round trip in front of a ~10 us dependent chain.  The first tile's request is issued
    // here, before the weight staging, so that it overlaps that round trip.
    double xn[D];
    {
        long p0 = wave * 16 + pt;
        p0 = p0 < g.N ? p0 : g.N - 1;
#pragma unroll
        for (int c = 0; c < D; ++c) xn[c] = g.X[(long)c * g.N + p0];
    }
    // A-operand fragments W^T[out = 16t+pt][in = 4s+q] and bias fragments of the hidden->hidden layers live in
    // LDS, lane-major (conflict-free ds_read_b64), shared by the block's waves: frees ~60 VGPRs per wave
    extern __shared__ __attribute__((aligned(16))) double fl[];
    // Output neurons 0..15 go through one 16-row MFMA tile; neurons 16..19 (the would-be second tile, 3/4
    // padding) through v_mfma_f64_4x4x4_4b with the A operand WR = W[in = 4s+q][16+a] (see the layer loop).
    // fp64 MFMA and fp64 VALU share one execution resource on gfx950 (measured: no additive throughput), so a
    // padded tile -- or the same products on the VALU -- would be a net loss of FP64 issue slots.
    double* WT = fl;                                   // [(L-1)][MF_KS][64]
    double* BH = fl + (L > 1 ? L - 1 : 0) * MF_KS * 64;       // [(L-1)][MF_KS][64]
    double* WR = BH + (L > 1 ? L - 1 : 0) * MF_KS * 64;       // [(L-1)][MF_KS][4 (q)][4 (a)]
    {   // all global reads of the staging are issued before the first LDS store (one round trip instead of three); the layer
        // index is a compile-time constant, so that the kernarg offsets are scalar loads and not a dependent vector load
        constexpr int LH_ = L > 1 ? L - 1 : 1, ITW = (MF_KS * 64 + BLK - 1) / BLK;
        static_assert(MF_KS * 16 <= BLK, "one remainder fragment per thread");
        double vw[LH_][ITW], vb[LH_][ITW], vr[LH_];
#pragma unroll
        for (int i_ = 1; i_ < L; ++i_) {
            const int wo_ = g.woff[i_], bo_ = g.boff[i_];
#pragma unroll
            for (int it = 0; it < ITW; ++it) {
                const int f = it * BLK + threadIdx.x, fc = f < MF_KS * 64 ? f : 0;
                const int ln = fc & 63, s_ = fc >> 6;
                vw[i_ - 1][it] = th[wo_ + (4 * s_ + (ln >> 4)) * MF_H + (ln & 15)];
                vb[i_ - 1][it] = th[bo_ + 4 * s_ + (ln >> 4)];
            }
            const int fr = threadIdx.x < MF_KS * 16 ? threadIdx.x : 0;
            vr[i_ - 1] = th[wo_ + (4 * (fr >> 4) + ((fr >> 2) & 3)) * MF_H + 16 + (fr & 3)];
        }
#pragma unroll
        for (int i_ = 1; i_ < L; ++i_) {
#pragma unroll
            for (int it = 0; it < ITW; ++it) {
                const int f = it * BLK + threadIdx.x;
                if (f < MF_KS * 64) { WT[(i_ - 1) * MF_KS * 64 + f] = vw[i_ - 1][it]; BH[(i_ - 1) * MF_KS * 64 + f] = vb[i_ - 1][it]; }
            }
            if (threadIdx.x < MF_KS * 16) WR[(i_ - 1) * MF_KS * 16 + threadIdx.x] = vr[i_ - 1];
        }
    }
    __syncthreads();
    const double bo = th[g.boff[L]];

    for (long tile = wave; tile < tile_end; tile += nwaves) {
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
        double h[C][MF_KS];
        double* sv = g.ACTS + (tile * L) * (long)(NS * MF_KS * 64) + lane;
        int lofs = lane;                       // opaque per iteration: keeps the LDS fragment reads inside the
        asm volatile("" : "+v"(lofs));         // loop instead of being hoisted into ~60 loop-invariant VGPRs

        // ---- layer 1 (VALU): z = b + x W, z_c = W[c,:], z_cc = 0 ----
        // (sin: the activation loops exist in a branch-free and a guarded copy, chosen by ONE wave-uniform test per layer -- a
        //  fallback branch per value would cut the five independent chains of a lane into separate basic blocks)
        {
            double z1[MF_KS];
#pragma unroll
            for (int s = 0; s < MF_KS; ++s) {
                double z = b1[s];
#pragma unroll
                for (int c = 0; c < D; ++c) z += x[c] * w1[c][s];
                z1[s] = z;
            }
            auto act1 = [&](auto fast) {
#pragma unroll
                for (int s = 0; s < MF_KS; ++s) {
                    double a, a1, a2;
                    act_fwd<ACT, decltype(fast)::value>(z1[s], a, a1, a2);
                    h[0][s] = a;
                    if constexpr (SAVE) {
                        sv[(0 * MF_KS + s) * 64] = a;
                        if constexpr (ACT == HPV_ACT_SIN) sv[(SA1 * MF_KS + s) * 64] = a1;
                    }
#pragma unroll
                    for (int t = 0; t < NT1; ++t) h[1 + t][s] = a1 * w1[t < D ? t : 0][s];   // T1 = coordinates 0..NT1-1
#pragma unroll
                    for (int b = 0; b < NT2; ++b) {
                        const double zc = w1[b < D ? b : 0][s];   // T2 = coordinates 0..NT2-1
                        if constexpr (T2Mix<NT1, NT2>::value) h[1 + NT1 + b][s] = a2 * t2_square<NT1, NT2>(g.t2w, b, w1[0][s], w1[D > 1 ? 1 : 0][s]);
                        else h[1 + NT1 + b][s] = a2 * zc * zc;
                    }
                }
            };
            if (act_wave_needs_safe<ACT>(z1)) act1(std::false_type{}); else act1(std::true_type{});
        }
        // ---- hidden -> hidden layers (MFMA) ----
#pragma unroll
        for (int i = 1; i < L; ++i) {
            v4d acc[C];
            const double* bhl = BH + (i - 1) * MF_KS * 64 + lofs;
            double z16[C];
#pragma unroll
            for (int ch = 0; ch < C; ++ch)
                acc[ch] = (ch == 0) ? v4d{bhl[0], bhl[64], bhl[128], bhl[192]} : v4d{0.0, 0.0, 0.0, 0.0};
#pragma unroll
            for (int s = 0; s < MF_KS; ++s)
#pragma unroll
                for (int ch = 0; ch < C; ++ch)
                    acc[ch] = __builtin_amdgcn_mfma_f64_16x16x4f64(WT[((i - 1) * MF_KS + s) * 64 + lofs], h[ch][s], acc[ch], 0, 0, 0);
            // neurons 16..19: Z^T[16+i'][pt] = sum_in W[in][16+i'] h[in][pt] on v_mfma_f64_4x4x4_4b (four 4x4x4 blocks = the
            // four groups of 4 points; no padding).  A[i'][k] = W[4s+k][16+i'] (lane: k = q, i' = lane&3, same for every
            // block), B_b[k][j] = h[4s+k][pt = 4b+j] = this lane's own h[ch][s]; D lane (q,pt) = neuron 16+q at point pt,
            // i.e. exactly the lane's fifth value.
            {
                const double* wrl = WR + (i - 1) * MF_KS * 16 + (lofs >> 4) * 4 + (lofs & 3);
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
            double* svl = sv + (long)i * (NS * MF_KS * 64);
            double zv[MF_KS];
#pragma unroll
            for (int s = 0; s < MF_KS; ++s) zv[s] = s < 4 ? acc[0][s & 3] : z16[0];
            auto acti = [&](auto fast) {
#pragma unroll
                for (int s = 0; s < MF_KS; ++s) {
                    double a, a1, a2;
                    act_fwd<ACT, decltype(fast)::value>(zv[s], a, a1, a2);
                    h[0][s] = a;
                    if constexpr (SAVE) {
                        svl[(0 * MF_KS + s) * 64] = a;
                        if constexpr (ACT == HPV_ACT_SIN) svl[(SA1 * MF_KS + s) * 64] = a1;
                    }
                    double zc[NT1 > 0 ? NT1 : 1];
#pragma unroll
                    for (int u = 0; u < NT1; ++u) {
                        zc[u] = s < 4 ? acc[1 + u][s & 3] : z16[1 + u];
                        if constexpr (SAVE) svl[((SZC + u) * MF_KS + s) * 64] = zc[u];
                        h[1 + u][s] = a1 * zc[u];
                    }
#pragma unroll
                    for (int b = 0; b < NT2; ++b) {
                        const double zcc = s < 4 ? acc[1 + NT1 + b][s & 3] : z16[1 + NT1 + b];
                        const double z1 = zc[b < NT1 ? b : 0];
                        if constexpr (SAVE) svl[((SZCC + b) * MF_KS + s) * 64] = zcc;
                        if constexpr (T2Mix<NT1, NT2>::value) h[1 + NT1 + b][s] = a2 * t2_square<NT1, NT2>(g.t2w, b, zc[0], zc[NT1 > 1 ? 1 : 0]) + a1 * zcc;
                        else h[1 + NT1 + b][s] = a2 * z1 * z1 + a1 * zcc;
                    }
                }
            };
            if (act_wave_needs_safe<ACT>(zv)) acti(std::false_type{}); else acti(std::true_type{});
        }
        // ---- linear head (VALU + 2 cross-lane adds over the 4 neuron groups) ----
#pragma unroll
        for (int ch = 0; ch < C; ++ch) {
            double v = 0.0;
#pragma unroll
            for (int s = 0; s < MF_KS; ++s) v += h[ch][s] * wo[s];
            v += __shfl_xor(v, 16, 64);
            v += __shfl_xor(v, 32, 64);
            if (ch == 0) v += bo;
            if (q == 0 && valid) g.OUT[(long)ch * g.N + p] = v;
            if (ch == 0 && g.data_off >= 0 && tile * 16 >= g.data_off) {
                // lossb = w mean((u_d - u)^2) (P1:98, P2:122, P3:184): adjoint + per-tile partial sum, no extra launch
                double dd = 0.0;
                if (q == 0 && valid) {
                    dd = g.ud[p - g.data_off] - v;
                    if (g.data_write_gbar) g.gbar0[p] = g.data_scale * dd;
                }
                double sq = dd * dd;
                sq += __shfl_xor(sq, 1, 64);
                sq += __shfl_xor(sq, 2, 64);
                sq += __shfl_xor(sq, 4, 64);
                sq += __shfl_xor(sq, 8, 64);
                if (lane == 0) g.data_part[tile - g.data_off / 16] = sq;
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------
// reverse
// ------------------------------------------------------------------------------------------------
// PQX > 0: ELEMENT-BLOCK mode with the projection fused in.  Workgroup b owns element b: it first projects the
// element (residual, element loss, adjoint of the integrated channels -> GBAR; project_element_wg), then runs the
// reverse pass over the element's PQX*PQY/16 tiles (+ one of the boundary/data tiles).  The per-element
// projection thus rides on all CUs inside the reverse kernel instead of being a separate latency-bound launch on
// a fraction of them, and -- unlike fusing it behind the forward pass -- costs no extra tile imbalance
// (25 tiles over 4 waves = the same 7-tile makespan as the round-robin assignment).
// WAVES = wavefronts per workgroup: 4 (one per SIMD), or 8 in element-block mode -- two waves per SIMD (<= 256 registers
// each; also chosen for 4-wave blocks whenever channels x layers <= 10, where that costs no spills) that cover each
// other's LDS-transpose / MFMA-result latencies, with the same 7-tile makespan per SIMD (25 tiles
// over 8 waves = 4,3,3,3 | 3,3,3,3) and the element's projection spread over twice the threads.
template <int D, int NT1, int NT2, int ACT, int L, int PQX = 0, int PQY = 0, int PNTX = 0, int PNTY = 0, int WAVES = MF_WAVES>
__global__ void __launch_bounds__(WAVES * 64, (WAVES == 8 || (1 + NT1 + NT2) * L <= 10) ? 2 : 1) k_bwd_mfma(MfmaArgs g) {
    constexpr int C = 1 + NT1 + NT2;
    constexpr int NS = SlotCount<ACT, NT1, NT2>::value;
    constexpr int SZC = 1 + (ACT == HPV_ACT_SIN ? 1 : 0);
    constexpr int SZCC = SZC + NT1;
    constexpr int LH = L > 1 ? L - 1 : 1;
    extern __shared__ __attribute__((aligned(16))) double lds[];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int q = lane >> 4, pt = lane & 15;
    const long wave = ((long)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const long nwaves = ((long)gridDim.x * blockDim.x) >> 6;
    const double* __restrict__ th = g.theta;
    // LDS map: [region A: per-wave transpose tiles during the tile loop, per-wave gradient rows in the
    //           epilogue] [region B: A-operand fragments of W for hbar_in^T = W zbar^T, lane-major]
    constexpr int REGION_A = WAVES * C * 2 * MF_TRB * MF_LD;
    double* TAB = lds + wv * (C * 2 * MF_TRB * MF_LD);   // per-wave transpose tiles, one (h_in, zbar) pair per channel
    const int regA = REGION_A > WAVES * g.P ? REGION_A : WAVES * g.P;
    double* WN = lds + regA;                      // [(L-1)][MF_KS][64]  A fragments W[in = pt][out = 4s+q], rows 0..15
    double* WRB = WN + (L > 1 ? L - 1 : 0) * MF_KS * 64;   // [(L-1)][MF_KS][4 (q)][4 (a)]  W[in = 16+a][out = 4s+q]

    // per-lane weight fragments
    // first-layer and head weights of this lane's neurons: lane-major in LDS, re-read per tile (not 30 resident VGPRs)
    double* W1O = WRB + (L > 1 ? L - 1 : 0) * MF_KS * 16;    // [(D+1)][MF_KS][64]
    {   // all global reads of the weight staging are issued before the first LDS store (one round trip instead of three);
        // compile-time layer index: scalar kernarg offsets instead of a dependent vector load per lane
        constexpr int BT = WAVES * 64;
        constexpr int N1 = (D + 1) * MF_KS * 64, IT1 = (N1 + BT - 1) / BT;
        constexpr int LH_ = L > 1 ? L - 1 : 1, ITW = (MF_KS * 64 + BT - 1) / BT;
        static_assert(MF_KS * 16 <= BT, "one remainder fragment per thread");
        double v1[IT1], vw[LH_][ITW], vr[LH_];
        const int w0o = g.woff[0], wLo = g.woff[L];
#pragma unroll
        for (int it = 0; it < IT1; ++it) {
            const int f = it * BT + threadIdx.x, fc = f < N1 ? f : 0;
            const int ln = fc & 63, s_ = (fc >> 6) % MF_KS, c_ = fc / (64 * MF_KS);
            const int j = 4 * s_ + (ln >> 4);
            v1[it] = th[(c_ < D ? w0o + c_ * MF_H : wLo) + j];
        }
        // A operand of hbar_in^T = W zbar^T : W[in = 16t+pt][out = 4s+q], kept in LDS (not registers) so that
        // two waves per SIMD fit; every wave of the block reads the same lane-major fragments, conflict-free
#pragma unroll
        for (int i_ = 1; i_ < L; ++i_) {
            const int wo_ = g.woff[i_];
#pragma unroll
            for (int it = 0; it < ITW; ++it) {
                const int f = it * BT + threadIdx.x, fc = f < MF_KS * 64 ? f : 0;
                const int ln = fc & 63, s_ = fc >> 6;
                vw[i_ - 1][it] = th[wo_ + (ln & 15) * MF_H + 4 * s_ + (ln >> 4)];
            }
            const int fr = threadIdx.x < MF_KS * 16 ? threadIdx.x : 0;
            vr[i_ - 1] = th[wo_ + (16 + (fr & 3)) * MF_H + 4 * (fr >> 4) + ((fr >> 2) & 3)];
        }
#pragma unroll
        for (int it = 0; it < IT1; ++it) { const int f = it * BT + threadIdx.x; if (f < N1) W1O[f] = v1[it]; }
#pragma unroll
        for (int i_ = 1; i_ < L; ++i_) {
#pragma unroll
            for (int it = 0; it < ITW; ++it) {
                const int f = it * BT + threadIdx.x;
                if (f < MF_KS * 64) WN[(i_ - 1) * MF_KS * 64 + f] = vw[i_ - 1][it];
            }
            if (threadIdx.x < MF_KS * 16) WRB[(i_ - 1) * MF_KS * 16 + threadIdx.x] = vr[i_ - 1];
        }
    }
    __syncthreads();
    if constexpr (PQX > 0) {
        // region A is free until the tile loop: use it as the projection's scratch
        // (with proj_split > 1 the workgroups sharing an element all project it: identical values, benign duplicate stores)
        project_element_wg<PQX, PQY, PNTX, PNTY, (WAVES * 64)>(g.pa, (long)blockIdx.x / g.proj_split, lds);
        __threadfence_block();
        __syncthreads();
    }
    // gradient accumulators (per wave, over all its tiles)
    // dW of a hidden->hidden layer = one 16x16 MFMA tile (in, out < 16) + two 4x16 strips on v_mfma_f64_4x4x4_4b
    // (4 blocks of 4x4x4, no padding: a 16x16x4 tile there would be 3/4 zeros) + the 4x4 corner (one more 4x4x4_4b)
    v4d dWacc[LH];
    double dS10[LH], dS01[LH];   // lane (q,pt): dW[in = 16+q][out = pt]  and  dW[in = pt][out = 16+q]
#pragma unroll
    for (int i = 0; i < LH; ++i) { dWacc[i] = v4d{0.0, 0.0, 0.0, 0.0}; dS10[i] = 0.0; dS01[i] = 0.0; }
    // corner dW[16+i'][16+j]: one 4x4x4_4b MFMA per channel whose four blocks are the four groups of 4 points; lane
    // (q, 4b+j) holds the partial sum of block b for (in = 16+q, out = 16+j), the blocks are summed in the epilogue
    double accC[LH];
#pragma unroll
    for (int i = 0; i < LH; ++i) accC[i] = 0.0;
    double db[L][MF_KS], dW1[D][MF_KS], dWo[MF_KS], dbo = 0.0;
#pragma unroll
    for (int s = 0; s < MF_KS; ++s) {
        dWo[s] = 0.0;
#pragma unroll
        for (int i = 0; i < L; ++i) db[i][s] = 0.0;
#pragma unroll
        for (int c = 0; c < D; ++c) dW1[c][s] = 0.0;
    }

    // Saved slots of one hidden layer for this lane (5 neurons x (s, [cos], z_c.., z_cc..)).  Each layer's slots
    // are loaded ONCE per tile and one layer AHEAD of their first use (the loads of layer i-1 are issued before
    // the activation-backward of layer i), so the HBM/MALL latency hides behind a full layer of MFMA work
    // even at one wave per SIMD.
    struct Slots {
        double a[MF_KS], a1s[MF_KS];
        double zc[NT1 > 0 ? NT1 : 1][MF_KS];
        double zcc[NT2 > 0 ? NT2 : 1][MF_KS];
    };
    int lofs = lane;
    auto load_slots = [&](const double* svl, bool first_layer, Slots& S) {
#pragma unroll
        for (int s = 0; s < MF_KS; ++s) {
            S.a[s] = svl[(0 * MF_KS + s) * 64];
            if constexpr (ACT == HPV_ACT_SIN) S.a1s[s] = svl[(1 * MF_KS + s) * 64]; else S.a1s[s] = 0.0;
#pragma unroll
            for (int u = 0; u < NT1; ++u) S.zc[u][s] = first_layer ? W1O[((u < D ? u : 0) * MF_KS + s) * 64 + lofs] : svl[((SZC + u) * MF_KS + s) * 64];
#pragma unroll
            for (int b = 0; b < NT2; ++b) S.zcc[b][s] = first_layer ? 0.0 : svl[((SZCC + b) * MF_KS + s) * 64];
        }
    };
    auto outputs_of = [&](const Slots& S, int ch, double (&hv)[MF_KS]) {   // channel ch of the layer's outputs
#pragma unroll
        for (int s = 0; s < MF_KS; ++s) {
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

    // (requesting the next tile's inputs ahead -- at the loop top, or late, during the first layer -- was measured twice:
    //  no gain, 100 more registers)
    auto load_tile_inputs = [&](long tile, double (&x)[D], double (&gb)[C], Slots& S) {
        const long p = tile * 16 + pt;
        const bool valid = p < g.N;
#pragma unroll
        for (int c = 0; c < D; ++c) x[c] = valid ? g.X[(long)c * g.N + p] : 0.0;
#pragma unroll
        for (int ch = 0; ch < C; ++ch) gb[ch] = valid ? g.GBAR[(long)ch * g.N + p] : 0.0;
        load_slots(g.ACTS + (tile * L + (L - 1)) * (long)(NS * MF_KS * 64) + lane, L == 1, S);
    };
    double x[D], gb[C];
    Slots cur;
    // tile sequence of this wave: round-robin over the batch, or (element-block mode) the tiles of the
    // workgroup's element followed by at most one boundary/data tile
    constexpr long TPE = PQX > 0 ? (PQX * PQY) / 16 : 1;
    // element-block mode: workgroup b owns part (b % split) of element b / split -- small shards (multi-GPU) spread
    // an element over up to 8 workgroups so that all CUs work, each projecting the element for itself
    const int split = PQX > 0 ? g.proj_split : 1, part = PQX > 0 ? (int)(blockIdx.x % split) : 0;
    const long tl0 = (part * TPE) / split, n_own = ((part + 1) * TPE) / split - tl0;
    const long ebase = ((long)blockIdx.x / split) * TPE + tl0;
    const long dtile = g.proj_n_elem * TPE + blockIdx.x;          // the data/pad tile this workgroup adopts
    auto tile_of = [&](long k) -> long {                            // k-th tile of this wave, -1 when exhausted
        if constexpr (PQX > 0) {
            const long lt = wv + k * WAVES;                      // local index among TPE (+1) tiles
            if (lt < n_own) return ebase + lt;
            if (lt == n_own && dtile < g.ntiles) return dtile;
            return -1;
        } else {
            const long t = wave + k * nwaves;
            return t < g.ntiles ? t : -1;
        }
    };
    for (long kt = 0;; ++kt) {
        const long tile = tile_of(kt);
        if (tile < 0) break;
        lofs = lane;
        asm volatile("" : "+v"(lofs));   // opaque per tile: the LDS weight reads stay inside the loop
        load_tile_inputs(tile, x, gb, cur);
        const double* sv = g.ACTS + (tile * L) * (long)(NS * MF_KS * 64) + lane;

        double hbar[C][MF_KS], zbar[C][MF_KS];
        // ---- linear head ----
#pragma unroll
        for (int ch = 0; ch < C; ++ch) {
            double hv[MF_KS];
            outputs_of(cur, ch, hv);
#pragma unroll
            for (int s = 0; s < MF_KS; ++s) {
                dWo[s] = fma(hv[s], gb[ch], dWo[s]);
                hbar[ch][s] = gb[ch] * W1O[(D * MF_KS + s) * 64 + lofs];
            }
        }
        if (q == 0) dbo += gb[0];

        // ---- hidden layers, last to first ----
#pragma unroll
        for (int i = L - 1; i >= 0; --i) {
            Slots prev;
            if (i > 0) load_slots(sv + (long)(i - 1) * (NS * MF_KS * 64), i == 1, prev);   // one layer ahead
#pragma unroll
            for (int s = 0; s < MF_KS; ++s) {
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
                for (int s = 0; s < MF_KS; ++s) {
#pragma unroll
                    for (int c = 0; c < D; ++c) dW1[c][s] += x[c] * zbar[0][s];
#pragma unroll
                    for (int u = 0; u < NT1; ++u) dW1[u < D ? u : 0][s] += zbar[1 + u][s];
                }
            } else {
                // weight gradient: contraction over the 16 points of the tile (and over channels).  All channels'
                // tiles are written first (one wave-level sync), then fragments are read channel by channel, so the
                // LDS reads of channel ch+1 overlap the MFMAs of channel ch.
                __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
                __builtin_amdgcn_wave_barrier();
#pragma unroll
                for (int ch = 0; ch < C; ++ch) {
                    double hv[MF_KS];
                    outputs_of(prev, ch, hv);
                    double* TA = TAB + (2 * ch) * (MF_TRB * MF_LD);
                    double* TB = TA + MF_TRB * MF_LD;
#pragma unroll
                    for (int s = 0; s < MF_KS; ++s) {
                        TA[(4 * s + q) * MF_LD + pt] = hv[s];
                        TB[(4 * s + q) * MF_LD + pt] = zbar[ch][s];
                    }
                }
                __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
                __builtin_amdgcn_wave_barrier();
#pragma unroll
                for (int ch = 0; ch < C; ++ch) {
                    const double* TA = TAB + (2 * ch) * (MF_TRB * MF_LD);
                    const double* TB = TA + MF_TRB * MF_LD;
                    // operand fragments, k-step kk = points 4kk..4kk+3 (k index = q):
                    //   aF/bF: rows in/out = pt of the transposed tiles (16x16 tile; B operands of the strips)
                    //   aS/bS: rows 16 + (lane & 3) (A operands of the 4x4x4 strips, same for all four blocks)
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
                        // D_b[i'][j] = sum_k A[i'][k] B_b[k][j]; D lane l <-> (i' = l>>4, 4b+j = l&15)
                        dS10[i - 1] = __builtin_amdgcn_mfma_f64_4x4x4f64(aS[kk], bF[kk], dS10[i - 1], 0, 0, 0);   // h[.,16+i'] x zbar[.,out]
                        dS01[i - 1] = __builtin_amdgcn_mfma_f64_4x4x4f64(bS[kk], aF[kk], dS01[i - 1], 0, 0, 0);   // zbar[.,16+i'] x h[.,in]
                    }
                    // corner: A_b[i'][k] = h[pt = 4b+k][16+i'], B_b[k][j] = zbar[pt = 4b+k][16+j]  (k = q, b = (lane&15)>>2)
                    accC[i - 1] = __builtin_amdgcn_mfma_f64_4x4x4f64(TA[(16 + (lane & 3)) * MF_LD + (pt & 12) + q],
                                                                   TB[(16 + (lane & 3)) * MF_LD + (pt & 12) + q], accC[i - 1], 0, 0, 0);
                }
                // hbar_in^T = W zbar^T
#pragma unroll
                for (int ch = 0; ch < C; ++ch) {
                    v4d acc = v4d{0.0, 0.0, 0.0, 0.0};
                    double h4 = 0.0;   // inputs 16..19 on the 4x4x4 MFMA: A[i'][k] = W[16+i'][4s+k], B = this lane's zbar, D lane = (16+q, pt)
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
                cur = prev;
            }
        }
    }

    // ---- epilogue: per-wave partials -> LDS -> one row per block ----
    __syncthreads();
    double* WP = lds + (long)wv * g.P;   // region A is free now
    for (int idx = lane; idx < g.P; idx += 64) WP[idx] = 0.0;
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
    // hidden->hidden weight gradients: complete sums over this wave's points, D layouts (see the accumulators)
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
        t += __shfl_xor(t, 4, 64);
        t += __shfl_xor(t, 8, 64);
        if (pt < 4) WP[g.woff[i] + (16 + q) * MF_H + 16 + pt] = t;
    }
    // per-lane partials: reduce over the 16 point lanes of each neuron group
#pragma unroll
    for (int s = 0; s < MF_KS; ++s) {
        const int j = 4 * s + q;
        double v[L + D + 1];
#pragma unroll
        for (int i = 0; i < L; ++i) v[i] = db[i][s];
#pragma unroll
        for (int c = 0; c < D; ++c) v[L + c] = dW1[c][s];
        v[L + D] = dWo[s];
#pragma unroll
        for (int k = 0; k < L + D + 1; ++k) {
            double t = v[k];
            t += __shfl_xor(t, 1, 64);
            t += __shfl_xor(t, 2, 64);
            t += __shfl_xor(t, 4, 64);
            t += __shfl_xor(t, 8, 64);
            v[k] = t;
        }
        if (pt == 0) {
#pragma unroll
            for (int i = 0; i < L; ++i) WP[g.boff[i] + j] = v[i];
#pragma unroll
            for (int c = 0; c < D; ++c) WP[g.woff[0] + c * MF_H + j] = v[L + c];
            WP[g.woff[L] + j] = v[L + D];
        }
    }
    {
        double t = dbo;
        t += __shfl_xor(t, 1, 64);
        t += __shfl_xor(t, 2, 64);
        t += __shfl_xor(t, 4, 64);
        t += __shfl_xor(t, 8, 64);
        if (lane == 0) WP[g.boff[L]] = t;
    }
    __syncthreads();
    const double* W0 = lds;
    double* row = g.GPART + (long)blockIdx.x * g.P;
    for (int idx = threadIdx.x; idx < g.P; idx += blockDim.x) {
        double acc = 0.0;
#pragma unroll
        for (int w = 0; w < WAVES; ++w) acc += W0[(long)w * g.P + idx];
        row[idx] = acc;
    }
}

// ------------------------------------------------------------------------------------------------
// host side
// ------------------------------------------------------------------------------------------------
static size_t fwd_lds_bytes(int L) { return (size_t)(L > 1 ? L - 1 : 0) * (2 * MF_KS * 64 + MF_KS * 16) * sizeof(double); }
static size_t bwd_lds_bytes(int P, int L, int C, int waves = MF_WAVES) {
    size_t regA = (size_t)waves * C * 2 * MF_TRB * MF_LD;
    if ((size_t)waves * P > regA) regA = (size_t)waves * P;
    return (regA + (size_t)(L > 1 ? L - 1 : 0) * (MF_KS * 64 + MF_KS * 16) + 3 * MF_KS * 64) * sizeof(double);
}

template <int D, int NT1, int NT2, int ACT, int L>
static void run_fwd(const MfmaArgs& a, int blocks, hipStream_t s) {
    if (a.save_act)
        hipLaunchKernelGGL((k_fwd_mfma<D, NT1, NT2, ACT, L>), dim3(blocks), dim3(MF_BLOCK), fwd_lds_bytes(L), s, a);
    else
        hipLaunchKernelGGL((k_fwd_mfma<D, NT1, NT2, ACT, L, false>), dim3(blocks), dim3(MF_BLOCK), fwd_lds_bytes(L), s, a);
}
static size_t bwd_lds_bytes(int P, int L, int C, int waves);
template <int D, int NT1, int NT2, int ACT, int L, int QX, int QY, int NTX, int NTY>
static void run_bwd_fused(const MfmaArgs& a, int blocks, hipStream_t s) {
    // wavefronts per element block (a 4-hidden-layer net needs > 256 registers per wave; four channels: their transpose tiles for eight
    // waves are 174 KB of LDS)
    constexpr int FW = (L <= 3 && NT2 == 0) ? 8 : 4;
    size_t lds = bwd_lds_bytes(a.P, L, 1 + NT1 + NT2, FW);
    static bool attr_set = false;
    if (!attr_set) {
        (void)hipFuncSetAttribute((const void*)k_bwd_mfma<D, NT1, NT2, ACT, L, QX, QY, NTX, NTY, FW>,
                                  hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        attr_set = true;
    }
    hipLaunchKernelGGL((k_bwd_mfma<D, NT1, NT2, ACT, L, QX, QY, NTX, NTY, FW>), dim3(blocks), dim3(FW * 64), lds, s, a);
}

template <int D, int NT1, int NT2, int ACT, int L>
static void run_bwd(const MfmaArgs& a, int blocks, hipStream_t s) {
    size_t lds = bwd_lds_bytes(a.P, L, 1 + NT1 + NT2);
    hipLaunchKernelGGL((k_bwd_mfma<D, NT1, NT2, ACT, L>), dim3(blocks), dim3(MF_BLOCK), lds, s, a);
}

template <int D, int NT1, int NT2, int ACT, int L>
static bool pick(HpvMfma* m) {
    m->fwd = run_fwd<D, NT1, NT2, ACT, L>;
    m->bwd = run_bwd<D, NT1, NT2, ACT, L>;
    // BASELINE config 4 (Poisson-2D var_form 1).  (Round 6 measured it for the four-channel forms on that element shape, which have no
    // whole-iteration instantiation with three hidden layers there: 95.3 against 95.5 us on the separate launches -- four waves per block
    // instead of eight, their transpose tiles would be 174 KB: not instantiated)
    if constexpr (D == 2 && NT1 == 2 && NT2 == 0 && ACT == HPV_ACT_TANH)
        m->bwd_fused = run_bwd_fused<D, NT1, NT2, ACT, L, 20, 20, 10, 10>;
    const char* an = ACT == HPV_ACT_SIN ? "sin" : "tanh";
    snprintf(m->vfwd, sizeof m->vfwd, "k_fwd_mfma<D=%d,NT1=%d,NT2=%d,%s,L=%d,H=20>", D, NT1, NT2, an, L);
    snprintf(m->vbwd, sizeof m->vbwd, "k_bwd_mfma<D=%d,NT1=%d,NT2=%d,%s,L=%d,H=20>", D, NT1, NT2, an, L);
    if (m->bwd_fused) snprintf(m->vbwd_fused, sizeof m->vbwd_fused, "k_bwd_mfma<D=%d,NT1=%d,NT2=%d,%s,L=%d,H=20,proj=20x20/10x10,waves=%d>", D, NT1, NT2, an, L, (L <= 3 && NT2 == 0) ? 8 : 4);
    size_t lds = bwd_lds_bytes(m->nd.P, L, 1 + NT1 + NT2);
    int of = 1, ob = 1;
    (void)hipOccupancyMaxActiveBlocksPerMultiprocessor(&of, k_fwd_mfma<D, NT1, NT2, ACT, L>, MF_BLOCK, fwd_lds_bytes(L));
    (void)hipOccupancyMaxActiveBlocksPerMultiprocessor(&ob, k_bwd_mfma<D, NT1, NT2, ACT, L>, MF_BLOCK, lds);
    m->occ_fwd = of > 0 ? of : 1;
    m->occ_bwd = ob > 0 ? ob : 1;
    return true;
}
template <int D, int NT1, int NT2, int ACT>
static bool pick_L(HpvMfma* m, int L) {
    switch (L) {
        case 1: return pick<D, NT1, NT2, ACT, 1>(m);
        case 2: return pick<D, NT1, NT2, ACT, 2>(m);
        case 3: return pick<D, NT1, NT2, ACT, 3>(m);
        case 4: return pick<D, NT1, NT2, ACT, 4>(m);
        case 5: return pick<D, NT1, NT2, ACT, 5>(m);
        case 6: return pick<D, NT1, NT2, ACT, 6>(m);
        default: return false;
    }
}

bool hpv_wide_pick(HpvMfma* m, int H, int key, int act, int L) {
    switch (H) {
#define HPV_WIDE_CASE(Hw) case Hw: return act == HPV_ACT_SIN ? hpv_wide_pick_##Hw##_d1(m, key, act, L) : hpv_wide_pick_##Hw##_d2(m, key, act, L);
        HPV_WIDE_WIDTHS(HPV_WIDE_CASE)
#undef HPV_WIDE_CASE
        default: return false;
    }
}

HpvMfma* hpv_mfma_create(const NetDesc& nd, long N, std::string* why, bool need_store) {
    auto no = [&](const char* msg) -> HpvMfma* { if (why) *why = msg; return nullptr; };
    const int L = nd.nl - 1;
    const int H = nd.width[1];
    if (L < 1 || L > 6 || (L > 4 && H > 32)) return no("1..6 hidden layers are covered at widths <= 32, 1..4 at the wider ones");
    for (int l = 1; l <= L; ++l)
        if (nd.width[l] != H) return no("all hidden layers must have the same width (the Python classes zero-pad to one)");
    for (int u = 0; u < nd.nT1; ++u) if (nd.t1dim[u] != u) return no("tangent channels must be coordinates 0..nT1-1");
    for (int b = 0; b < nd.nT2; ++b) if (nd.t2idx[b] != b) return no("second tangents must be coordinates 0..nT2-1");
    HpvMfma* m = new HpvMfma();
    m->nd = nd; m->N = N; m->L = L;
    m->ntiles = (N + 15) / 16;
    bool ok = false;
    const int key = nd.d * 100 + nd.nT1 * 10 + nd.nT2;
    m->H = H; m->ks = H / 4;
    if (H != MF_H) {
        // any other width: the width-generic kernels (kernels_wide.hip), forward + activation store -> projection -> reverse
        ok = H % 4 == 0 && hpv_wide_pick(m, H, key, nd.act, L);
        m->ns = (nd.act == HPV_ACT_SIN ? 2 : 1) + nd.nT1 + nd.nT2;
        if (!ok) { delete m; return no("hidden width not instantiated (kernels_wide.hip: 24, 32, 40, 48, 64; 20: kernels_mfma.hip) or too deep for its LDS"); }
    } else if (nd.act == HPV_ACT_SIN) {
        // Poisson-1D channel sets (P1:82-91)
        if (key == 111) ok = pick_L<1, 1, 1, HPV_ACT_SIN>(m, L);
        else if (key == 110) ok = pick_L<1, 1, 0, HPV_ACT_SIN>(m, L);
        else if (key == 100) ok = pick_L<1, 0, 0, HPV_ACT_SIN>(m, L);
        m->ns = 2 + nd.nT1 + nd.nT2;
    } else {
        // Poisson-2D (P2:93-115) and AdvDiff (P3:161-174) channel sets
        if (key == 222) ok = pick_L<2, 2, 2, HPV_ACT_TANH>(m, L);
        else if (key == 220) ok = pick_L<2, 2, 0, HPV_ACT_TANH>(m, L);
        else if (key == 200) ok = pick_L<2, 0, 0, HPV_ACT_TANH>(m, L);
        else if (key == 221) ok = pick_L<2, 2, 1, HPV_ACT_TANH>(m, L);
        m->ns = 1 + nd.nT1 + nd.nT2;
    }
    if (!ok) { delete m; return no("channel set / activation combination not instantiated"); }
    if (need_store || H != MF_H) {   // forward-only users (predict) run with save_act = 0 and need no activation store -- the width-generic
        // forward kernel wants ONE tile's block to alias its stores onto (kernels_wide.hip, k_fwd_wide)
        size_t bytes = (size_t)(need_store ? m->ntiles : 1) * L * m->ns * m->ks * 64 * sizeof(double);
        if (hipMalloc((void**)&m->ACTS, bytes) != hipSuccess) { delete m; return no("hipMalloc of the activation store failed"); }
        (void)hipMemset(m->ACTS, 0, bytes);
    }
    if (need_store && H == MF_H && nd.d == 2 && nd.nT1 == 2 && nd.nT2 <= 1 && nd.act == HPV_ACT_TANH) {
        // elements the exchange buffers of the split whole-iteration kernels are sized for (the largest grid this batch can hold)
        m->xsync_elems = N / 144 + 1;       // (the smallest element shape of the whole-iteration kernels: 12x12 points)
        // tagged-exchange granules: 2 words per exchanged double (tall elements: <= CUs x 25 doubles; SPLIT mode: 800 per element)
        m->xg_words = (size_t)2 * 800 * (size_t)std::min<long>(m->xsync_elems, 512);
        if (hipMalloc((void**)&m->xg, m->xg_words * sizeof(unsigned long long)) == hipSuccess &&
            hipMalloc((void**)&m->xiter, sizeof(unsigned int)) == hipSuccess) {
            (void)hipMemset(m->xg, 0, m->xg_words * sizeof(unsigned long long));
            (void)hipMemset(m->xiter, 0, sizeof(unsigned int));
        } else {
            (void)hipGetLastError();
            if (m->xg) { (void)hipFree(m->xg); m->xg = nullptr; }
            m->xiter = nullptr;
        }
        m->xdebug_skip = hpv_test_hook_split_skip();      // (hpv_api.hip: HPV_DEBUG_SPLIT_SKIP in libhpvpinn_testhooks.so, the constant 0 in the product)
    }
    int dev = 0, cus = 256;
    (void)hipGetDevice(&dev);
    (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
    m->n_cus = cus;
    const long tiles_per_block = MF_WAVES;
    long want = (m->ntiles + tiles_per_block - 1) / tiles_per_block;
    // persistent-style grids: exactly the number of blocks that are resident at once (the wave loops
    // over its tiles), so weight fragments are loaded and gradient rows written once per resident wave
    m->fwd_blocks = (int)std::min<long>(want, (long)cus * std::max(1, m->occ_fwd));
    m->bwd_blocks = (int)std::min<long>(want, (long)cus * std::max(1, m->occ_bwd));
    {
        const char* e = getenv("HPV_FUSE");
        m->fuse_bwd = !(e && e[0] == 'n');
        m->iter_fused_ok = !(e && (e[0] == 'n' || e[0] == 'b'));
        m->iter_fused_force = e && e[0] == 'i';
        m->iter_split_ok = !(e && e[0] == 's');
        m->prefer_elem = e && e[0] == 'e';
        m->multi_off = e && e[0] == '1';
        m->multi_force = e && e[0] == 'm';
    }
    MfmaArgs& a = m->base;
    a = MfmaArgs{};
    a.N = N; a.ntiles = m->ntiles; a.ACTS = m->ACTS; a.P = nd.P;
    for (int l = 0; l < nd.nl; ++l) { a.woff[l] = nd.woff[l]; a.boff[l] = nd.boff[l]; }
    for (int i = 0; i < 2; ++i) { a.t1dim[i] = nd.t1dim[i]; a.t2idx[i] = nd.t2idx[i]; a.t2w[i] = nd.t2w[i]; }
    return m;
}

void hpv_mfma_destroy(HpvMfma* m) {
    if (!m) return;
    if (m->ACTS) (void)hipFree(m->ACTS);
    if (m->xg) (void)hipFree(m->xg);
    if (m->xiter) (void)hipFree(m->xiter);
    delete m;
}

int hpv_mfma_grad_rows(HpvMfma* m) { return m->bwd_blocks; }
const char* hpv_mfma_variant(HpvMfma* m, int which) {
    if (!m) return "";
    return which == 0 ? m->variant : (which == 1 ? m->vfwd : (which == 2 ? m->vbwd : m->vbwd_fused));
}
unsigned int* hpv_mfma_xiter(HpvMfma* m) { return m ? m->xiter : nullptr; }
bool hpv_mfma_prefers_elem(HpvMfma* m) { return m && m->prefer_elem; }
double* hpv_mfma_activation_store(HpvMfma* m) { return m ? m->ACTS : nullptr; }
size_t hpv_mfma_activation_store_doubles(HpvMfma* m) { return m && m->ACTS ? (size_t)m->ntiles * m->L * m->ns * m->ks * 64 : 0; }   // (the timing builds park their stamps there)
// Workgroups per element of the fused reverse kernel: one when the shard has an element for every CU, more for the
// small shards of a multi-GPU run (each workgroup walks 1/split of the element's 25 tiles).
static int fused_split(HpvMfma* m, long n_elem) {
    int split = 1;
    while (split < 8 && n_elem * split * 2 <= m->n_cus) split *= 2;
    return split;
}
// rows the caller must allocate: the element-block mode writes one row per workgroup
int hpv_mfma_max_rows(HpvMfma* m, long n_elem, long n_data_tiles) {
    int r = hpv_mfma_grad_rows(m);
    // kernels_elem.hip: one row per element plus one per 8 .. 16 boundary / data tiles no element wave has a free slot for
    if (n_elem + n_data_tiles / 8 + 2 > r && n_elem <= 65536) r = (int)(n_elem + n_data_tiles / 8 + 2);
    const long fused_rows = n_elem * fused_split(m, n_elem);
    if (m->bwd_fused && fused_rows > r && fused_rows <= 65536) r = (int)fused_rows;
    if (n_elem > r && n_elem <= 65536) r = (int)n_elem;      // the whole-iteration kernel writes one row per element
    // SPLIT mode of the whole-iteration kernel: 2 - 8 workgroups per element while they fit the chip (three-channel sets have the fused
    // reverse kernel's count above; the four-channel sets of the general forms do not)
    if (n_elem * 2 <= m->n_cus) { long sp = 1; while (sp < 8 && n_elem * sp * 2 <= m->n_cus) sp *= 2; if (n_elem * sp > r) r = (int)(n_elem * sp); }
    // ... and, on a grid larger than the chip with a ragged last round, up to 8 rows per element of the tail (at most half a round of elements)
    if (n_elem > m->n_cus && n_elem + m->n_cus > r && n_elem <= 65536) r = (int)(n_elem + m->n_cus);
    if (n_elem * 64 > r && n_elem * 64 <= m->n_cus) r = (int)(n_elem * 64);   // tall-element kernel: up to 64 workgroups per element
    // kernels_tile.hip: one row per element plus one per 6..8 boundary/data tiles that the elements' free waves do not take
    if (m->ntiles <= 8192 && n_elem + m->ntiles / 6 + 1 > r) r = (int)(n_elem + m->ntiles / 6 + 1);
    m->max_rows = r;
    return r;
}

void hpv_mfma_forward(HpvMfma* m, const double* theta, const double* X, double* OUT, int save_act, hipStream_t s,
                      const MfmaDataTerm* dt) {
    MfmaArgs a = m->base;
    a.theta = theta; a.X = X; a.OUT = OUT; a.save_act = (save_act && m->store_s_only) ? 2 : save_act;
    a.data_off = -1;
    if (dt && dt->n_data > 0) {
        a.data_off = dt->data_off; a.ud = dt->ud; a.gbar0 = dt->gbar0; a.data_part = dt->data_part;
        a.data_scale = dt->scale; a.data_write_gbar = dt->write_gbar;
    }
    m->fwd(a, m->fwd_blocks, s);
}

void hpv_mfma_backward(HpvMfma* m, const double* theta, const double* X, const double* GBAR, double* GPART, int* rows,
                       hipStream_t s) {
    MfmaArgs a = m->base;
    a.theta = theta; a.X = X; a.GBAR = GBAR; a.GPART = GPART;
    m->bwd(a, m->bwd_blocks, s);
    if (rows) *rows = hpv_mfma_grad_rows(m);
}

// Whole training pass of a shard of elements of any instantiated shape in one launch (kernels_elem.hip).  Returns false when the
// shape / channel set / width / layout is not covered; the caller then runs the separate kernels.
bool hpv_mfma_iter_elem(HpvMfma* m, const double* theta, const double* X, double* GPART, int* rows, hipStream_t s,
                        const MfmaDataTerm* dt, const ProjArgs& pa, long n_elem) {
    const ProjDesc& pd = pa.pd;
    const NetDesc& nd = m->nd;
    if (!m->iter_fused_ok || pd.edge || pd.nact || n_elem <= 0 || m->L < 2 || m->L > 3 || nd.d != 2 || nd.act != HPV_ACT_TANH) return false;
    const int key = nd.d * 100 + nd.nT1 * 10 + nd.nT2;
    // Where this structure is the DEFAULT (measured, profiles/r04_element_shapes.md: 16x16-element grids, us per iteration, this
    // kernel with 4 / 8 waves against the separate launches): few channel-layers or small elements -- [2,20,20,1] var_form 1
    // 52.6 / 46.1 vs 54.1, one channel (var_form 2) 45.0 / 40.4 vs 60.9, 12x12 points 39.6 / 43.5 vs 56.0.  With three or more
    // channels through three hidden layers the compiler-scheduled tile bodies at one or two waves per SIMD lose to the
    // hand-scheduled two-kernel path (16x16 points, var_form 1: 80.6 / 75.2 vs 69.4; five channels 118 / 132 vs 92.6), and so
    // do wider layers (H = 32: 164 vs 110, register spills): those run it only on request (HPV_FUSE=e).
    const int C_ = 1 + nd.nT1 + nd.nT2, tpe_ = (pd.qx * pd.qy + 15) / 16;
    const bool light = C_ == 1 || C_ * m->L <= 6;
    if (!m->prefer_elem && !(m->H <= 24 && (light || tpe_ <= 9))) return false;
    // many small elements: one workgroup per element pays its launch-once phases per element, the separate launches amortise them
    // (1 024 elements of 12x12 points: 138 against 99 us)
    if (!m->prefer_elem && tpe_ <= 9 && !light && n_elem > 3L * m->n_cus) return false;
    // wavefronts per workgroup (kernels_elem.hip): two per SIMD for the light channel sets, where 256 registers per wave suffice
    int waves = (m->H <= 24 && tpe_ >= 8 && light) ? 8 : 4;
    const int nq = pd.qx * pd.qy, tpe = (nq + 15) / 16, tpw = (tpe + waves - 1) / waves, slots = waves * tpw, nfree = slots - tpe;
    // batch layout [element points | pad to 16 | data points]
    const long npad = ((long)n_elem * nq + 15) / 16 * 16;
    const bool has_data = dt && dt->n_data > 0;
    if (has_data ? (dt->data_off != npad || m->N != npad + dt->n_data) : (m->N != npad && m->N != n_elem * nq)) return false;
    const long n_dt = has_data ? (dt->n_data + 15) / 16 : 0;
    const long left = n_dt - n_elem * nfree;
    const long blocks = n_elem + (left > 0 ? (left + slots - 1) / slots : 0);
    if (blocks > hpv_mfma_grad_rows(m) && blocks > m->max_rows) return false;
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
    a.elem_waves = waves;
    bool ok = false, known = false;
#define HPV_ELEM_TRY(A_, B_, C_, D_)                                                           \
    if (!known && pd.qx == A_ && pd.qy == B_ && pd.ntx >= 1 && pd.ntx <= C_ && pd.nty >= 1 && pd.nty <= D_) {                 \
        known = true;                                                                          \
        ok = hpv_elem_launch_##A_##B_##_##C_##_##D_(m->H, key, m->L, a, (int)blocks, s);         \
    }
    HPV_ELEM_SHAPES(HPV_ELEM_TRY)
#undef HPV_ELEM_TRY
    if (!ok) return false;
    m->last_split = false;
    snprintf(m->variant, sizeof m->variant, "k_iter_elem<D=2,NT1=%d,NT2=%d,tanh,L=%d,H=%d,%dx%d/%dx%d,waves=%d,tiles/wave=%d>", nd.nT1, nd.nT2,
             m->L, m->H, pd.qx, pd.qy, pd.ntx, pd.nty, waves, tpw);
    if (rows) *rows = (int)blocks;
    return true;
}

// Reverse pass with the per-element projection fused in front (element-block mode).  Returns false when not
// applicable; the caller then launches projection and reverse pass separately.
bool hpv_mfma_backward_fused(HpvMfma* m, const double* theta, const double* X, const double* GBAR, double* GPART, int* rows,
                             hipStream_t s, const ProjArgs& pa, long n_elem) {
    const ProjDesc& pd = pa.pd;
    if (!m->bwd_fused || !m->fuse_bwd || pd.edge || pd.nact || n_elem <= 0) return false;
    if (!(pd.qx == 20 && pd.qy == 20 && pd.ntx >= 1 && pd.ntx <= 10 && pd.nty >= 1 && pd.nty <= 10)) return false;   // (counts: run-time values)
    // (element-block mode = whole elements in rounds of one workgroup per CU: on a ragged grid larger than the chip the three separate
    //  launches are faster -- 289 elements: 114 against 87.5 us, profiles/r05_multi_element.md)
    if (n_elem > m->n_cus) {
        const long rounds = (n_elem + m->n_cus - 1) / m->n_cus;
        if (n_elem * 100 < rounds * m->n_cus * 80) return false;
    }
    const long tpe = (20 * 20) / 16;
    const long rest = m->ntiles - n_elem * tpe;                 // pad + data tiles: at most one per workgroup
    const int split = fused_split(m, n_elem);
    const long blocks = n_elem * split;
    if (rest < 0 || rest > blocks) return false;
    if (blocks > hpv_mfma_grad_rows(m) && blocks > m->max_rows) return false;
    MfmaArgs a = m->base;
    a.theta = theta; a.X = X; a.GBAR = GBAR; a.GPART = GPART;
    a.proj_n_elem = n_elem;
    a.proj_split = split;
    a.pa = pa;
    m->bwd_fused(a, (int)blocks, s);
    if (rows) *rows = (int)blocks;
    return true;
}

