// Element-resident whole-iteration kernel for Poisson-2D var_form 1 and every other two-term "one-hot" form, [2,20,...,20,1] tanh
// networks (two or three hidden layers).  Written for the BASELINE config-4 shape -- 20x20-point / 10x10-test elements, which the
// description below uses -- and since round 4 a template over the element shape (FZ_SHAPES: also 16x16 / 8x8 and 12x12 / 6x6 points /
// largest test-function counts; the run's own counts are run-time values):
//
//   ONE workgroup (4 wavefronts, one per SIMD, up to 512 registers each) owns ONE element and runs the whole
//   iteration for it without touching HBM in between:
//     phase F  Taylor-mode forward of the element's 25 16-point tiles (7,6,6,6 per wave; one wave also takes one of the
//              boundary/data tiles).  Only s = tanh(z) of every hidden layer is kept -- 15 doubles per lane and tile,
//              IN REGISTERS (hand-placed in the AGPR half of the unified register file; one tile per wave in LDS) -- and the two
//              integrated channels u_x, u_y go to LDS.
//     phase P  projection of the element from LDS (sum-factorised, P2:98-105), residual R = U - F, element loss
//              (P2:117-120), adjoint of u_x, u_y back into the same LDS array.
//     phase R  reverse pass of the same tiles: the tangent pre-activations z_x, z_y are RECOMPUTED from s on the MFMA
//              pipe (2 channels x (L-1) layer products), then the hand-derived reverse pass of kernels_mfma.hip.
//   The activation store of the two-kernel path (1 120 B/point written by the forward and read by the reverse kernel,
//   248 MB of HBM traffic per iteration at config 4) does not exist here: per iteration the kernel reads the
//   coordinates twice (L2-resident), F, the parameters, and writes R, the element losses and one gradient row per
//   workgroup.
//
// Lane layout, MFMA formulation and the 16 + 4 split of a 20-wide layer are those of kernels_mfma.hip.
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <type_traits>

#include "hpv_mfma_dev.h"
#include "hpv_fused_dev.h"

#define FZ_WAVES 4
#define FZ_BLOCK (FZ_WAVES * 64)
#define FZ_PRE_PER_THREAD 4     // deferred-update prologue: parameters per thread (networks of up to 1 024 parameters)
#define FZ_C 3             // u, u_x, u_y
// The element shape (QX x QY quadrature points, NTX x NTY test functions) is a template parameter of the kernel (round 4; it was
// written for 20x20 / 10x10 alone); FZ_SHAPE_CONSTS names the derived constants inside a template body.  Needed of a shape: whole
// 16-point tiles (QX QY % 16 == 0) and a tile count of 0 or 1 mod 4 (the parking plan below gives waves 2, 3 one LDS slot each).
#define FZ_SHAPE_CONSTS                                                                                              \
    static constexpr int FZ_QX = QX_, FZ_QY = QY_, FZ_NTX = NTX_, FZ_NTY = NTY_, FZ_NQ = QX_ * QY_, FZ_NR = NTX_ * NTY_; \
    static constexpr int FZ_TPE = FZ_NQ / 16;                                                                        \
    static constexpr int FZ_MAXT = FZ_TPE / 4 + 1;      /* tiles per wave: ceil(TPE / 4), or TPE / 4 + one boundary/data tile */ \
    static_assert(FZ_NQ % 16 == 0 && FZ_TPE % 4 <= 1, "element shape not covered by the whole-iteration kernel");
// instantiated shapes (host dispatch, build guard of csrc/build.sh): X(QX, QY, NTX, NTY)
#define FZ_SHAPES(X) X(20, 20, 10, 10) X(16, 16, 8, 8) X(12, 12, 6, 6)

// Build with -DHPV_FZ_TIMING to make the kernel record the duration of its phases (staging, forward, barrier wait,
// projection, reverse, barrier wait, epilogue) per wave into MfmaArgs::OUT; scripts/fz_timing.py prints them.
#ifdef HPV_FZ_TIMING
#define FZ_STAMP(I) fz_t[I] = clock64()
// segments of the reverse tile body, accumulated over the wave's tiles: [0] fetch + tangent recompute, [1] head,
// [2 + 2 (L-1-i)] layer i: zbar + transposes + hbar chain, [3 + 2 (L-1-i)] layer i: dW products (layer 0: dW1)
#define FZ_SEG(I) do { const long long t_ = clock64(); fz_seg[I] += t_ - fz_last; fz_last = t_; } while (0)
#else
#define FZ_STAMP(I)
#define FZ_SEG(I)
#endif

// C_ = network channels (3: u, u_x, u_y; 4: + the mixed second tangent, NT2 = 1); TRG_ = channels whose transposes of the weight-
// gradient products are in LDS at a time (C_ = 3: all three, ONE pass -- the headline plan; C_ = 4: two passes of two, 22 KB less)
// NPART_ (the tight plan, FzPlan below): the first NPART_ saved doubles of one MORE tile of every wave live here instead of in the stash
// (the stash is 2 NPART_ registers shorter), and the epilogue's gradient rows overlay the parking area (dead by then) instead of the
// transpose region, which then only has to hold ONE channel's tiles.
template <int L, int QX_, int QY_, int NTX_, int NTY_, bool MULTI = false, int C_ = FZ_C, int TRG_ = FZ_C, int NPART_ = 0>
struct FzLds {
    FZ_SHAPE_CONSTS
    static constexpr int LH = L > 1 ? L - 1 : 0;
    static constexpr int WT = 0;                           // [LH][5][64]  forward A fragments  W[4s+q][out = pt]
    static constexpr int BH = WT + LH * MF_KS * 64;        // [LH][5][64]  bias fragments       b[4s+q]
    static constexpr int WR = BH + LH * MF_KS * 64;        // [LH][5][16]  W[4s+q][16+a]
    static constexpr int WN = WR + LH * MF_KS * 16;        // [LH][5][64]  reverse A fragments  W[in = pt][out = 4s+q]
    static constexpr int WRB = WN + LH * MF_KS * 64;       // [LH][5][16]  W[16+a][4s+q]
    static constexpr int W1O = WRB + LH * MF_KS * 16;      // [4][5][64]   W1[0][4s+q], W1[1][4s+q], Wo[4s+q], b1[4s+q]
    static constexpr int CH = W1O + 4 * MF_KS * 64;        // [2][400]     u_x, u_y of the element -> their adjoints
    static constexpr int PK = CH + 2 * FZ_NQ;              // [6][L*5][64] parked s: slot w = tile 0 of wave w, slots 4, 5 = tile 1 of waves 0, 1
    static constexpr int XS = PK;                          // GS (slots 0..3 of PK are free then): [x | y | u_d][400 + 16] coordinates of the element and of the data tile
    static constexpr int XLD = FZ_NQ + 16;
    // (four channels: the compiler needs ~60 registers more -- one more tile of every wave is parked here: slots 4..7 = tile 1 of wave
    //  w - 4, slots 8, 9 = the quarter tiles' s (QT) or tile 2 of waves 0, 1 (whole tiles: the two waves that may own FZ_MAXT tiles))
    static constexpr int PKS = C_ > FZ_C ? 10 : 6;
    static constexpr int PP = PK + PKS * L * MF_KS * 64;   // [4][NPART_][64] the tight plan's part of a stash tile
    static constexpr int PZ = PP + FZ_WAVES * NPART_ * 64; // [4][LH*5][32] QT: the quarter tiles' tangent pre-activations of the layers >= 2 (tangent lanes, compact)
    static constexpr int TR = PZ + (NPART_ > 0 ? 0 : FZ_WAVES * LH * MF_KS * 32);   // (the tight plan runs whole tiles: no quarter tiles' array)   // phase P: projection scratch | phase R: per-wave transpose tiles | epilogue rows
    static constexpr int TR_WAVE = TRG_ * 2 * MF_TRB * MF_LD;
    // projection scratch inside the TR region.  MULTI (several elements per workgroup): the tables live BEHIND the region instead -- the
    // reverse phase's transposes would overwrite them and every element would have to stage them again
    static constexpr int TRSZ = FZ_WAVES * TR_WAVE;        // (>= FZ_WAVES * P: the epilogue rows; asserted in total())
    static constexpr int NTABS = 2 * FZ_NTX * FZ_QX + 2 * FZ_NTY * FZ_QY;
    static constexpr int AX = MULTI ? TR + TRSZ : TR;      // [2][NTX][QX] w_x phi^(dx_t)
    static constexpr int BY = AX + 2 * FZ_NTX * FZ_QX;     // [2][NTY][QY] w_y phi^(dy_t)
    static constexpr int T = MULTI ? TR : BY + 2 * FZ_NTY * FZ_QY;      // [2][QY][NTX]
    static constexpr int UP = T + 2 * FZ_QY * FZ_NTX;      // [2][NR]      per-term partial of U
    static constexpr int U = UP + 2 * FZ_NR;               // [NR]
    static constexpr int S = U + FZ_NR;                    // [2][NTY][QX]
    static constexpr int RED = S + 2 * FZ_NTY * FZ_QX;     // [16]
    static_assert(RED + 16 - TR <= FZ_WAVES * TR_WAVE, "projection scratch fits the transpose region");
    static_assert(3 * XLD <= 4 * L * MF_KS * 64, "the staged coordinates fit the parking slots the GS plan leaves free");
    static constexpr int EPI = NPART_ > 0 ? PK : TR;       // the epilogue's rows [4][P]
    static constexpr int total(int P) { return NPART_ > 0 ? TR + TRSZ : TR + (TRSZ > FZ_WAVES * P ? TRSZ : FZ_WAVES * P) + (MULTI ? NTABS : 0); }
};
// The tight plan (round 6): four channels with three hidden layers on 20x20 points.  Seven tiles on a wave, four of them in the
// stash, leave the compiler 136 registers where it needs ~150; a third parked tile per wave is 14 LDS slots, 179 KB.  So: the
// weight-gradient transposes go through LDS one channel at a time (TRG = 1: 21 KB instead of 43), the epilogue's rows overlay the
// parking area, the quarter tiles' array is not needed (whole tiles), and the room that frees holds NPART of the 15 saved doubles of
// one stash tile: the stash begins at a160 (the compiler's high-water mark in that instantiation: a149).
template <int L, int QX_, int QY_, int NT2>
struct FzPlan {
    static constexpr bool TIGHT = NT2 > 0 && L == 3 && QX_ * QY_ / 16 > 16;
    static constexpr int TRG = TIGHT ? 1 : (NT2 > 0 ? 2 : FZ_C);
    static constexpr int NPART = TIGHT ? 12 : 0;
};

// SPLIT: an element is shared by g.proj_split (2, 4 or 8) workgroups -- the shards of a multi-GPU run are too small to fill the
// chip with one workgroup per element.  Workgroup (e, part) walks the tiles [25 part / S, 25 (part + 1) / S) of element e,
// publishes their u_x, u_y as tagged granules (hpv_fused_dev.h, xg_*), gathers the whole element's from its partners, then EVERY partner
// projects the whole element for itself (identical values, benign duplicate stores of R and loss_e) and reverses its own tiles.
//
// QT (one workgroup per element, the full grids): the 25 tiles of an element do not divide by four waves -- 7 + 6 + 6 + 6, and three
// waves wait a whole tile (12 % of the launch) for the fourth.  With QT every wave takes six tiles and a QUARTER of the 25th:
// its four points travel as ONE packed operand whose 16 point slots are {value, d/dx, d/dy} x 4 points + 4 points of the
// workgroup's boundary / data tile (value only) -- the layer products, the hbar chain, the tangent recompute and the dW
// products then run ONCE for the packed operand instead of once per channel (a third of a full tile's MFMA work, same weight
// fragments), the channels of a point meet through DPP row shifts in the element-wise steps, and the boundary / data points
// ride in the slots that would be idle, so no wave owns a seventh tile any more.
//
// GS (round 4, opt-in: measured slower, see launch_iter_fused_L): what the reverse pass needs of a whole tile -- s of every hidden layer AND the tangent pre-activations z_x, z_y of the
// layers >= 2 (35 doubles per lane at L = 3) -- travels through device memory instead of being parked in AGPRs / LDS and recomputed:
// the forward pass stores it (16-byte lane-contiguous stores into the handle's activation store, [tile][pair][lane][2]), the
// reverse pass requests tile k + 1's while it works on tile k.  The kernel is bound by the fp64 datapath and leaves 8 TB/s of HBM
// (and the 256 MB memory-side cache, which holds the whole 110 MB) idle; the recompute was 1 600 of a reverse tile's 7 650 datapath
// cycles plus the AGPR shuffles around it.  Every workgroup reads back only what its own waves wrote (same CU, same L2): no fences.
typedef double v2d __attribute__((ext_vector_type(2)));
//
// MULTI (round 5): grids with more elements than CUs.  One workgroup per element pays the launch-once phases -- weight staging, the
// epilogue's cross-wave reduction and gradient row, the dispatch of a fresh workgroup -- per ELEMENT (~11 k of an element's 132 k
// cycles at 20x20 points, of 90 k at 16x16).  Here gridDim.x = CUs workgroups walk the elements b, b + gridDim.x, ..: the weight
// fragments stay in LDS (only the projection tables, which share the transpose region, are restaged per element), and the 45
// per-lane gradient accumulators of an element are ADDED INTO a per-wave scratch block in device memory (g.ACTS: load-add-store of
// lane-private, coalesced 512-byte rows; L2-resident) at the end of its reverse phase instead of being reduced across lanes and
// waves -- they are zeroed again before the next element's reverse phase, so nothing lives across a forward phase (carrying them in
// registers took the compiler to a151 of the 106 AGPRs the stash leaves: profiles/r04_notes.md 13).  The epilogue runs once per workgroup.
//
// GEN / NT2 (round 6): every other variational form of the 2-D drivers on the same structure.  GEN: the two integrated arrays
// in LDS are no longer the channels u_x, u_y themselves (term t <-> channel 1 + t, "one-hot") but COMBINATIONS of the channels
// with the TermDesc weights, G_t = sum_c (a0 + eps a1)[t][c] ch_c, and the trainable epsilon (P3:163, 171) gets its gradient:
//   two terms (AdvDiff var_form 1, P3:169-174):  slot 0 = G_0, slot 1 = G_1;  d eps from the terms that carry eps as a factor
//   one term  (AdvDiff var_form 0, P3:161-167; Poisson-2D var_form 0, P2:91-96):  slot 0 = G_0, slot 1 = E = dG_0 / d eps
//   (sum_c a1[c] ch_c), which meets the adjoint of G_0 at the end of the projection phase: d eps = sum_p Gbar_0[p] E[p]
// NT2 = 1: a fourth channel, the mixed second tangent w0 u_xx + w1 u_yy (NetDesc::t2w) -- forward, tangent recompute and reverse
// pass of kernels_tall.hip's algebra; its transposes of the weight-gradient products go through LDS in two passes of two channels.
template <int L, bool SPLIT, bool QT, bool GS, int QX_, int QY_, int NTX_, int NTY_, bool MULTI = false, int NT2 = 0, bool GEN = false>
__global__ void __launch_bounds__(FZ_BLOCK, 1) k_iter_fused(MfmaArgs g) {
    static_assert(!(SPLIT && QT), "the quarter-tile scheme is for whole elements");
    static_assert(!(MULTI && (SPLIT || GS)), "the element loop is for whole elements on the register stash");
    static_assert(!((NT2 > 0 || GEN) && (GS || MULTI)), "the general forms run on the register stash, one workgroup (or a split) per element");
    static_assert(NT2 == 0 || GEN, "a second-tangent channel is always integrated through the general term weights");
    FZ_SHAPE_CONSTS
    static_assert(!(NT2 > 0 && QT && FZ_TPE % FZ_WAVES != 0), "four channels leave the packed quarter tile no slot for the data points: data-quarter plan only");
    constexpr int C = FZ_C + NT2;                  // channels of the network: u, u_x, u_y[, w0 u_xx + w1 u_yy]
    using PLAN = FzPlan<L, QX_, QY_, NT2>;
    constexpr int TRG = PLAN::TRG;                 // channels per transpose pass of the weight-gradient products
    constexpr int NPART = PLAN::NPART;             // saved doubles of the stash's last tile that live in LDS (the tight plan)
    static_assert(C % TRG == 0, "whole transpose passes");
    static_assert(NPART == 0 || !QT, "the tight plan: whole tiles");
    using M = FzLds<L, QX_, QY_, NTX_, NTY_, MULTI, C, TRG, NPART>;
    static_assert(!GEN || (M::AX == M::TR && 4 <= M::TR_WAVE), "the d-epsilon partials live in wave 0's part of the transpose region (it reads them back itself)");
    constexpr int LH = L > 1 ? L - 1 : 1;
    constexpr int NSV = L * MF_KS;                 // saved doubles per lane and tile
    extern __shared__ __attribute__((aligned(16))) double lds[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);   // wave-uniform (SGPR): tile counts and branches on it stay scalar
    const int q = lane >> 4, pt = lane & 15;
    const int split = SPLIT ? g.proj_split : 1;
    // SPLIT: the partners of an element are the workgroups b, b + n_elem, b + 2 n_elem, ..: workgroups are dealt to the XCDs round-robin
    // (observed: XCC id == blockIdx % 8), so with an element count that 8 divides -- the shards of a multi-GPU run -- the partners share
    // an XCD and their exchange is served by one L2 (1.56 against 2.39 us per exchange, profiles/r04_xchg_probe.txt).  A speed choice
    // only: nothing depends on the placement.
    // (SPLIT: the launch may own only the LAST g.proj_n_elem elements of the shard -- the ragged tail of a grid larger than the chip,
    //  behind a one-workgroup-per-element launch of the full rounds: g.elem0 = its first element, el = the element inside this launch)
    const long el = SPLIT ? (long)(blockIdx.x % (unsigned)g.proj_n_elem) : (long)blockIdx.x;
    long e = SPLIT ? el + g.elem0 : el;      // (MULTI: advances by gridDim.x per trip)
    const int part = SPLIT ? (int)(blockIdx.x / (unsigned)g.proj_n_elem) : 0;
    const double* __restrict__ th = g.theta;
    const ProjArgs& pa = g.pa;
#ifdef HPV_FZ_TIMING
    long long fz_t[8];
    const long long fz_start = clock64(), fz_wall = wall_clock64();
    long long fz_seg[2 + 2 * L], fz_last = clock64();      // (kernel scope: the epilogue behind the element loop reads them)
    int fz_n_own = 0;
#endif

    // ---- stage every weight fragment and the projection tables ----
    [[maybe_unused]] constexpr int TNX = FZ_NTX, TNY = FZ_NTY, TQX = FZ_QX, TQY = FZ_QY;
    const int rnx = pa.pd.ntx, rny = pa.pd.nty, rnr = rnx * rny;      // the run's test functions per direction (<= NTX, NTY)
    static_assert(FZ_NTX * FZ_QX == FZ_NTY * FZ_QY, "table staging walks both tables with one index");
    double bo = 0.0;
    // (`th`: the parameters -- device memory, or the LDS copy the deferred-update prologue below has formed)
    auto stage_all = [&](auto th) {
        // layer index as a compile-time constant (kernarg offsets become scalar loads instead of a dependent vector load per
        // lane), every global read issued before the first LDS store (one memory round trip)
        constexpr int N1 = 4 * MF_KS * 64, IT1 = (N1 + FZ_BLOCK - 1) / FZ_BLOCK;
        constexpr int ITW = (MF_KS * 64 + FZ_BLOCK - 1) / FZ_BLOCK;
        double vwt[L > 1 ? L - 1 : 1][ITW], vbh[L > 1 ? L - 1 : 1][ITW], vwn[L > 1 ? L - 1 : 1][ITW];
        double vwr[L > 1 ? L - 1 : 1], vwrb[L > 1 ? L - 1 : 1], v1[IT1];
#pragma unroll
        for (int i_ = 1; i_ < L; ++i_) {
            const int wo = g.woff[i_], bo_ = g.boff[i_];
#pragma unroll
            for (int it = 0; it < ITW; ++it) {
                const int f = it * FZ_BLOCK + tid, fc = f < MF_KS * 64 ? f : 0;
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
            const int f = it * FZ_BLOCK + tid, fc = f < N1 ? f : 0;
            const int ln = fc & 63, s_ = (fc >> 6) % MF_KS, c_ = fc / (64 * MF_KS);
            const int j = 4 * s_ + (ln >> 4);
            v1[it] = th[(c_ < 2 ? w0o + c_ * MF_H : (c_ == 2 ? wLo : b0o)) + j];
        }
        constexpr int NTAB = 2 * TNX * TQX, ITT = (NTAB + FZ_BLOCK - 1) / FZ_BLOCK;
        const int dx0 = pa.pd.t[0].dx, dx1 = pa.pd.t[1].dx, dy0 = pa.pd.t[0].dy, dy1 = pa.pd.t[1].dy;
        double vax[ITT], vby[ITT];
#pragma unroll
        for (int it = 0; it < ITT; ++it) {
            const int f = it * FZ_BLOCK + tid, fc = f < NTAB ? f : 0;
            const int tt_ = fc / (TNX * TQX), ti_ = fc % (TNX * TQX);
            // (fewer test functions than the instantiation's NTX x NTY: the tables of the missing ones are zero -- their residuals are
            //  exactly 0 and leave the sums alone; R, F and the means below use the run's own counts rnx, rny)
            const int rr_ = ti_ / TQX, ii_ = ti_ % TQX;
            // (GEN, a form of ONE term: the second term's tables are zero, like its coefficient pc1)
            const bool ton = !GEN || tt_ == 0 || pa.pd.nterms > 1;
            vax[it] = (rr_ < rnx && ton) ? pa.wtx[((long)(tt_ ? dx1 : dx0) * rnx + rr_) * TQX + ii_] : 0.0;
            vby[it] = (rr_ < rny && ton) ? pa.wty[((long)(tt_ ? dy1 : dy0) * rny + rr_) * TQY + ii_] : 0.0;
        }
#pragma unroll
        for (int i_ = 1; i_ < L; ++i_) {
#pragma unroll
            for (int it = 0; it < ITW; ++it) {
                const int f = it * FZ_BLOCK + tid;
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
        for (int it = 0; it < IT1; ++it) {
            const int f = it * FZ_BLOCK + tid;
            if (f < N1) lds[M::W1O + f] = v1[it];
        }
#pragma unroll
        for (int it = 0; it < ITT; ++it) {
            const int f = it * FZ_BLOCK + tid;
            if (f < NTAB) { lds[M::AX + f] = vax[it]; lds[M::BY + f] = vby[it]; }
        }
        bo = th[g.boff[L]];
    };
    [[maybe_unused]] int pre_failed = -1;      // the deferred-update prologue's verdict (identical in every workgroup); -1: no prologue
    if constexpr (!GS && !MULTI) {
        if (g.pre_g) {
            // The multi-GPU iteration in two launches (round 5): the previous iteration's update has not been applied -- its
            // all-reduced gradient sits in g.pre_g.  Every workgroup forms the updated parameters for itself (the parking area is
            // free until the forward phase) and stages its fragments from there; nothing is written: k_finalize, behind this launch,
            // stores parameters, moments and beta powers with the same arithmetic (hpv_adam_one).  A failed exchange on any rank
            // (pad slot of the reduced buffer) or here (sticky flag): no update -- and workgroup 0 latches the flag for k_finalize.
            double* TH = lds + M::PK;
            const AdamArgs& ad = g.pre_ad;
            const double flag = g.pre_g[g.pre_Ptot + 3];
            const int xe = *ad.xerr;
            const bool failed = !(flag == 0.0) || xe;
            pre_failed = failed ? 1 : 0;
            const double b1p = ad.state[0], b2p = ad.state[1];
            // every operand requested before the first is used: ONE memory round trip (the host declines networks of more than
            // FZ_PRE_PER_THREAD * FZ_BLOCK parameters)
            double t0[FZ_PRE_PER_THREAD], g0[FZ_PRE_PER_THREAD], m0[FZ_PRE_PER_THREAD], v0[FZ_PRE_PER_THREAD];
#pragma unroll
            for (int k = 0; k < FZ_PRE_PER_THREAD; ++k) {
                const int idx = tid + k * FZ_BLOCK;
                const bool in = idx < g.P;
                t0[k] = in ? ad.theta[idx] : 0.0; g0[k] = in ? g.pre_g[idx] : 0.0; m0[k] = in ? ad.m[idx] : 0.0; v0[k] = in ? ad.v[idx] : 1.0;
            }
#pragma unroll
            for (int k = 0; k < FZ_PRE_PER_THREAD; ++k) {
                const int idx = tid + k * FZ_BLOCK;
                double m1, v1, t1;
                hpv_adam_one(ad.lr, ad.b1, ad.b2, ad.eps, b1p, b2p, g0[k], m0[k], v0[k], t0[k], m1, v1, t1);
                if (idx < g.P) TH[idx] = failed ? t0[k] : t1;
            }
            // (k_finalize decides the DEFERRED update by this launch's verdict, xerr[1], which every rank forms from the same reduced
            //  buffer -- not by the sticky flag, which a barrier of THIS launch may set on one rank only: the replicas would part by one update)
            if (blockIdx.x == 0 && tid == 0) {
                ad.xerr[1] = failed ? 1 : 0;
                if (failed && !xe) __hip_atomic_store(ad.xerr, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
            __syncthreads();
            stage_all((const double*)TH);
        } else {
            stage_all(th);
        }
    } else {
        stage_all(th);
    }
    // SPLIT: has a barrier of an EARLIER launch of this handle failed?  A plain load (kernel boundaries make those stores
    // visible) requested behind the staging loads and consumed at the barrier
    int xsticky = 0;
    unsigned xtag = 0;             // this launch's exchange tag (hpv_fused_dev.h, xg_*)
    // (behind a deferred-update prologue the verdict formed there is used instead: workgroup 0 may be storing to *g.xerr in THIS
    //  launch -- a partner that re-read the flag could see it while another does not, and wait for a partner that stays away
    //  (advisor, round 5); the prologue's verdict comes from the pad slot and the pre-launch flag, the same in every workgroup)
    if constexpr (SPLIT) { xsticky = pre_failed >= 0 ? pre_failed : *g.xerr; xtag = *g.xiter + 1u; }
    // the element's projection constants, requested now so that no global latency sits inside phase P
    double pc0 = pa.coef[e], pc1 = (!GEN || pa.pd.nterms > 1) ? pa.coef[pa.coef_stride + e] : 0.0;
    // GEN: the trainable coefficient and the term weights.  slot 0 of the LDS channel array holds G_0 = sum_c wa0(c) ch_c; slot 1
    // holds G_1 (two terms) or E = dG_0 / d eps (one term); the channels' adjoints are gb_c = wa0(c) Gbar_0 + wb1(c) Gbar_1.
    // (formed where they are used from kernel arguments -- scalar registers -- and eps: nothing lives across the phases)
    // (wave-uniform values made so explicitly -- v_readfirstlane of both halves -- and formed ONCE: they live in scalar registers.  Formed at
    //  their uses they were hoisted as VECTOR values parked in a102..a111 across the phases: the general quarter tile on 20x20 points did
    //  not fit its stash (a111 of 106); now a99, the other general instantiations a91 / a147 instead of a99 / a159)
    [[maybe_unused]] auto uni = [](double x) -> double {
        return __hiloint2double(__builtin_amdgcn_readfirstlane(__double2hiint(x)), __builtin_amdgcn_readfirstlane(__double2loint(x)));
    };
    [[maybe_unused]] const double geps = (GEN && pa.eps_ptr) ? uni(pa.eps_ptr[0]) : 0.0;
    [[maybe_unused]] const bool gtwo = GEN && pa.pd.nterms > 1;
    [[maybe_unused]] double WA0[C], WA1[C], WB1[C];
    if constexpr (GEN) {
#pragma unroll
        for (int c = 1; c < C; ++c) {
            const double t1 = uni(fma(geps, pa.pd.t[1].a1[c], pa.pd.t[1].a0[c]));
            WA0[c] = uni(fma(geps, pa.pd.t[0].a1[c], pa.pd.t[0].a0[c]));
            WA1[c] = uni(gtwo ? t1 : pa.pd.t[0].a1[c]);
            WB1[c] = uni(gtwo ? t1 : 0.0);
        }
    }
    [[maybe_unused]] auto wa0 = [&](int c) -> double { return WA0[c]; };
    [[maybe_unused]] auto wa1 = [&](int c) -> double { return WA1[c]; };
    [[maybe_unused]] auto wb1 = [&](int c) -> double { return WB1[c]; };
    [[maybe_unused]] const double gm0 = uni((GEN && pa.pd.t[0].eps_mult) ? geps : 1.0), gm1 = uni((gtwo && pa.pd.t[1].eps_mult) ? geps : 1.0);   // the factor eps of a term (P3:171)
    [[maybe_unused]] const double ge0 = (GEN && pa.pd.t[0].eps_mult) ? 1.0 : 0.0, ge1 = (gtwo && pa.pd.t[1].eps_mult) ? 1.0 : 0.0;
    const int ro_k = tid / FZ_NTX, ro_r = tid % FZ_NTX;                 // residual (k, r) of thread tid < NR, and whether the run has it
    const bool ro_on = tid < FZ_NR && ro_k < rny && ro_r < rnx;
    long ro_idx = e * rnr + ro_k * rnx + ro_r;
    double pF = (pa.F && ro_on) ? pa.F[ro_idx] : 0.0;
    // ---- tile list of this wave: element tiles wv, wv+4, .., and possibly one boundary/data tile `dtile` ----
    const int lg = SPLIT ? __builtin_ctz(split) : 0;                             // split is 2, 4 or 8
    const int tbase = SPLIT ? (part * FZ_TPE) >> lg : 0;                         // this workgroup's tile range of the element
    const int tend = SPLIT ? ((part + 1) * FZ_TPE) >> lg : FZ_TPE;
    // DQ: a shape whose tiles divide by four (16x16 points: 4 + 4 + 4 + 4) has no tile to cut in quarters, but one wave owns the
    // workgroup's boundary / data tile as a fifth -- the same 25 % imbalance.  Its "quarter-tile" instantiation packs ONLY the data
    // points (slot c = 3 of every wave's operand, four points each); the element-point slots are switched off (no channel
    // written, zero adjoint), so every wave owns TPE / 4 whole tiles + a packed operand.
    constexpr bool DQ = QT && FZ_TPE % FZ_WAVES == 0;
    const int n_el = QT ? (DQ ? FZ_TPE / FZ_WAVES : (FZ_TPE - 1) / FZ_WAVES)
                        : SPLIT ? (tend - tbase - wv + FZ_WAVES - 1 > 0 ? (tend - tbase - wv + FZ_WAVES - 1) / FZ_WAVES : 0)
                                : (FZ_TPE - wv + FZ_WAVES - 1) / FZ_WAVES;
    static_assert(!QT || (FZ_TPE - 1) % FZ_WAVES == 0 || DQ, "QT: 24 whole tiles over four waves + one tile in quarters");
    // The boundary/data tiles behind the elements go one per workgroup to the first wave with the fewest element tiles
    // (25 tiles over 4 waves: wave 1).  SPLIT: only to workgroups in which that wave has a free slot compared with its
    // neighbours (tile count not a multiple of 4) -- otherwise the adopted tile is a whole extra forward + reverse that the
    // element's partners wait for -- as long as there are enough of those.
    // (SPLIT: the boundary / data tiles begin behind ALL elements of the shard, g.data_tile0 -- not behind this launch's)
    long dtile = (SPLIT ? g.data_tile0 : g.proj_n_elem * FZ_TPE) + blockIdx.x;
    if constexpr (SPLIT) {
        int n_free = 0, before = 0;
        bool mine = false;
        for (int p = 0; p < split; ++p) {
            const bool fr = (((((p + 1) * FZ_TPE) >> lg) - ((p * FZ_TPE) >> lg)) % FZ_WAVES) != 0;
            n_free += fr;
            before += (fr && p < part);
            mine = mine || (fr && p == part);
        }
        if (g.ntiles - g.data_tile0 <= g.proj_n_elem * n_free)
            dtile = mine ? g.data_tile0 + el * n_free + before : g.ntiles;
    }
    if constexpr (GS) {
        // the coordinates of the element's points and of the workgroup's boundary / data tile (+ its targets) to LDS: the tile loops
        // then hold no global load besides the saved values' -- a wait for a coordinate would be a wait for every store / request
        // issued before it (in-order vmcnt)
        for (int i = tid; i < FZ_NQ; i += FZ_BLOCK) {
            lds[M::XS + i] = g.X[e * FZ_NQ + i];
            lds[M::XS + M::XLD + i] = g.X[g.N + e * FZ_NQ + i];
        }
        if (tid < 16) {
            const long pd_ = dtile * 16 + tid;
            const bool vd = dtile < g.ntiles && pd_ < g.N;
            lds[M::XS + FZ_NQ + tid] = vd ? g.X[pd_] : 0.0;
            lds[M::XS + M::XLD + FZ_NQ + tid] = vd ? g.X[g.N + pd_] : 0.0;
            lds[M::XS + 2 * M::XLD + tid] = vd ? g.ud[pd_ - g.data_off] : 0.0;
        }
    }
    FZ_STAMP(0);
    __syncthreads();
    FZ_STAMP(1);
    // gradient accumulators of this wave (zeroed at the head of every reverse phase; MULTI: spilled into g.ACTS between elements)
    constexpr int LHA = L > 1 ? L - 1 : 1;
    v4d dWacc[LHA];
    double dS10[LHA], dS01[LHA], accC[LHA];
    double db[L][MF_KS], dW1[2][MF_KS], dWo[MF_KS], dbo = 0.0;
    double gdat = 0.0;           // adjoint of u at the data tile's point (boundary term, P2:122)
    [[maybe_unused]] double gdat_q = 0.0;
    [[maybe_unused]] bool m_first = true;      // MULTI: this is the workgroup's first element (its spill is a plain store)
    // MULTI: the wave's spill block, [slot][64 lanes] (slot order: acc_spill below)
    constexpr int NACC = LHA * 7 + (L + 3) * MF_KS + 1;
    [[maybe_unused]] double* ASP = MULTI ? g.ACTS + ((long)blockIdx.x * FZ_WAVES + wv) * (NACC * 64) + lane : nullptr;
#pragma unroll 1
    for (;;) {       // (one trip unless MULTI)
    // MULTI: the thread index is laundered per trip and SHADOWS the kernel-scope one inside the loop -- every per-thread LDS / global
    // address of the phases below is then formed inside the trip.  Hoisted out of the element loop they were ~30 loop-invariant
    // values parked in AGPRs above the hand-managed base (a136 of 106) across the stash's live range.
    int tid_trip_ = threadIdx.x;
    if constexpr (MULTI) asm volatile("" : "+v"(tid_trip_));
    const int tid = tid_trip_, lane = tid & 63, q = lane >> 4, pt = lane & 15;
    const bool has_d = !QT && (wv == (tend - tbase) % FZ_WAVES) && dtile < g.ntiles;     // (QT: the data points ride in the quarter tile)
    const int n_own = n_el + (has_d ? 1 : 0);
    auto tile_of = [&](int k) -> long { return k < n_el ? e * FZ_TPE + tbase + wv + (long)k * FZ_WAVES : dtile; };

    // s = tanh(z) of every hidden layer: tile 0's go to LDS (what is left of it), the tiles 1..6 of this wave to the top
    // AGPRs a[ABASE + (k-1) * 2 NSV ..] (see acc_put)
    constexpr int NREG = FZ_MAXT - 2 - (NT2 > 0 ? 1 : 0);  // tiles whose s live in AGPRs; the first one (waves 0, 1 -- NT2: every wave -- two) of a wave is parked in LDS
    constexpr int ABASE = 256 - NREG * 2 * NSV + 2 * NPART;    // (tight plan: the stash's FIRST place is NPART doubles short -- they are in LDS)
    constexpr int AFULL = ABASE - 2 * NPART;                   // place K of the stash begins at AFULL + K * 2 NSV (place 0: its doubles NPART.. only)
    if constexpr (!GS) asm volatile("" ::: "a255");       // the kernel owns all 256 AGPRs
    // (waves 0, 1 may own FZ_MAXT tiles, waves 2, 3 one less -- QT: every wave FZ_MAXT - 1 whole ones; QT: slots 4, 5 -- NT2: 8, 9 -- of the
    //  parking area hold the quarter tiles' s)
    const int n_lds = GS ? 0 : NT2 > 0 ? (QT ? 2 : (wv <= 1 ? 3 : 2)) : QT ? 1 : (wv <= 1 ? 2 : 1);
    // GS: pairs per tile and lane -- {z_x, z_y}[layer >= 2][k-step], then s two by two; a tile's block of the activation store
    constexpr int NZP = (L > 1 ? L - 1 : 0) * MF_KS, NSP = (NSV + 1) / 2, NP = NZP + NSP;
    constexpr long GS_STRIDE = (long)L * 3 * MF_KS * 64;         // doubles per tile of the activation store (3 slots: kernels_mfma.hip)
    static_assert(NP * 128 <= GS_STRIDE, "a tile's pairs fit its block of the activation store");
    auto gs_ptr = [&](long tile) -> v2d* { return reinterpret_cast<v2d*>(g.ACTS + tile * GS_STRIDE) + lane; };
    double* PKw = lds + M::PK + wv * (NSV * 64) + lane;
    double* PKw2 = lds + M::PK + (4 + (NT2 > 0 ? wv : (wv & 1))) * (NSV * 64) + lane;
    gdat = 0.0;

    // SPLIT: this workgroup takes no part in the exchange (an earlier launch of the handle failed -- sticky flag -- or the test
    // knob keeps partner 1 of element 0 away): it publishes nothing and leaves at the hand-off point
    const bool xstay = SPLIT && (xsticky || HPV_XDEBUG_SKIP(g, xtag, e, part));
    // =============================================================================================
    // phase F: forward
    // =============================================================================================
    // Two tiles per trip: their instruction streams are independent, so the scheduler fills the MFMA -> tanh -> MFMA
    // dependency bubbles of one tile with the other's work (a single wave per SIMD has nothing else to issue); the forward
    // working set is small enough to hold twice.
    // QT: the packed quarter tile of this wave.  Slot c = pt >> 2 of point j = pt & 3: c = 0 value, c = 1 d/dx, c = 2 d/dy of element
    // point 384 + 4 wv + j; c = 3 value of point 4 wv + j of the workgroup's boundary / data tile.
    [[maybe_unused]] const int qcs = pt >> 2, qj = pt & 3;
    [[maybe_unused]] const bool q_tan = qcs == 1 || qcs == 2;
    [[maybe_unused]] const long q_pdat = dtile * 16 + 4 * wv + qj;
    [[maybe_unused]] const bool q_vdat = QT && qcs == 3 && dtile < g.ntiles && q_pdat < g.N;
    [[maybe_unused]] const long q_p = qcs == 3 ? (q_vdat ? q_pdat : 0) : (e * FZ_TPE + (FZ_TPE - 1)) * 16 + 4 * wv + qj;
    [[maybe_unused]] const int q_lp = (FZ_TPE - 1) * 16 + 4 * wv + qj;               // the element point inside the element
    [[maybe_unused]] double* PKQ = lds + M::PK + (NT2 > 0 ? 8 : 4) * (NSV * 64) + wv * (NSV * 32);    // compact: the 32 value / data lanes only
    [[maybe_unused]] const int q_ci = q * 8 + (qcs == 3 ? 4 : 0) + qj;                // ... at this index (tangent slots: their point's)
    [[maybe_unused]] double* PKZ = lds + M::PZ + wv * ((L > 1 ? L - 1 : 1) * MF_KS * 32);   // tangent pre-activations, the 32 tangent lanes only
    [[maybe_unused]] const int q_cz = q * 8 + (qcs == 2 ? 4 : 0) + qj;
    gdat_q = 0.0;
    [[maybe_unused]] double qx0 = 0.0, qx1 = 0.0, qud = 0.0;
    [[maybe_unused]] const int q_xs = qcs == 3 ? FZ_NQ + 4 * wv + qj : q_lp;      // GS: the slot's point in the staged coordinates
    if constexpr (QT && GS) {
        qx0 = lds[M::XS + q_xs]; qx1 = lds[M::XS + M::XLD + q_xs];
        qud = q_vdat ? lds[M::XS + 2 * M::XLD + 4 * wv + qj] : 0.0;
    } else if constexpr (QT) {              // (requested before the whole tiles: consumed after them)
        qx0 = g.X[q_p]; qx1 = g.X[g.N + q_p];
        qud = q_vdat ? g.ud[q_pdat - g.data_off] : 0.0;
    }
    auto load_x = [&](int k, double (&x)[2], bool& valid, long& p) {
        const long tile = tile_of(k < n_own ? k : 0);
        p = tile * 16 + pt;
        valid = (p < g.N) && (k < n_own);
        const long pc = p < g.N ? p : g.N - 1;
        x[0] = g.X[pc]; x[1] = g.X[g.N + pc];
    };
    auto stash = [&](int k, const double (&sv)[NSV]) {
        if constexpr (GS) {
            v2d* zs = gs_ptr(tile_of(k)) + NZP * 64;
#pragma unroll
            for (int j = 0; j < NSV / 2; ++j) zs[j * 64] = v2d{sv[2 * j], sv[2 * j + 1]};
            if constexpr (NSV & 1) reinterpret_cast<double*>(zs + (NSV / 2) * 64 - lane)[lane] = sv[NSV - 1];   // (odd count: the last one alone, 8-byte lanes)
        } else if (k < n_lds) {     // wave-uniform
            double* pk = k == 0 ? PKw : PKw2;
            // (NT2: the slot is formed HERE -- a third parking pointer kept alive across the phases sent the register allocator to
            //  scratch memory, 420 - 540 scratch accesses per instantiation and 148 instead of 58 us per iteration: profiles/r06_notes.md)
            if constexpr (NT2 > 0) pk = lds + M::PK + (k == 0 ? wv : (k == 1 ? 4 + wv : 8 + (wv & 1))) * (NSV * 64) + lane;
#pragma unroll
            for (int j = 0; j < NSV; ++j) pk[j * 64] = sv[j];
        } else {
            switch (k - n_lds) {
#define FZ_STASH(K) case K: if constexpr (K < NREG) acc_put_all<AFULL + K * 2 * NSV, NSV>(sv); break;
                case 0:
                    if constexpr (NREG < 1) {
                    } else if constexpr (NPART > 0) {
                        double* pp = lds + M::PP + wv * (NPART * 64) + lane;
#pragma unroll
                        for (int j = 0; j < NPART; ++j) pp[j * 64] = sv[j];
                        acc_put_from<ABASE, NSV, NPART>(sv);
                    } else {
                        acc_put_all<ABASE, NSV>(sv);
                    }
                    break;
                FZ_STASH(1) FZ_STASH(2) FZ_STASH(3) FZ_STASH(4)
#undef FZ_STASH
            }
        }
    };
    // one trip over NT (2, or 1 for the odd tile out) tiles starting at the wave's k0-th tile; coordinates were requested a trip ahead
    double xn[2][2];
    bool vn[2];
    long pn[2];
    // (WQ: the wave's packed quarter tile rides along with this trip -- one more independent instruction stream for the bubbles)
    auto fwd_trip = [&](int k0, auto NT_, auto WQ_) {
        constexpr int NT = decltype(NT_)::value;
        constexpr bool WQ = decltype(WQ_)::value;
        double xx[NT][2];
        bool valid[NT];
        long pp[NT];
        if constexpr (GS) {
#pragma unroll
            for (int t = 0; t < NT; ++t) {
                const int k = k0 + t;
                const int lp = k < n_el ? (tbase + wv + k * FZ_WAVES) * 16 + pt : FZ_NQ + pt;
                pp[t] = tile_of(k) * 16 + pt;
                valid[t] = pp[t] < g.N;
                xx[t][0] = lds[M::XS + lp]; xx[t][1] = lds[M::XS + M::XLD + lp];       // (points beyond the batch were staged as 0)
            }
        } else {
#pragma unroll
            for (int t = 0; t < NT; ++t) {
                valid[t] = vn[t]; pp[t] = pn[t];
                xx[t][0] = valid[t] ? xn[t][0] : 0.0; xx[t][1] = valid[t] ? xn[t][1] : 0.0;
            }
            load_x(k0 + NT, xn[0], vn[0], pn[0]);       // the next trip's coordinates travel while this one computes
            load_x(k0 + NT + 1, xn[1], vn[1], pn[1]);
        }
        int lofs = lane;
        asm volatile("" : "+v"(lofs));       // opaque: the LDS fragment reads stay inside the loop
        double h[NT][C][MF_KS], sv[NT][NSV];
        [[maybe_unused]] double QH[MF_KS], QA[NSV];      // WQ: the packed quarter tile's layer input and its s
        // layer 1 (VALU)
#pragma unroll
        for (int s = 0; s < MF_KS; ++s) {
            const double w0 = lds[M::W1O + (0 * MF_KS + s) * 64 + lofs], w1 = lds[M::W1O + (1 * MF_KS + s) * 64 + lofs];
            const double b1v = lds[M::W1O + (3 * MF_KS + s) * 64 + lofs];
#pragma unroll
            for (int t = 0; t < NT; ++t) {
                const double z = b1v + xx[t][0] * w0 + xx[t][1] * w1;
                double a, a1, a2;
                act_fwd<HPV_ACT_TANH>(z, a, a1, a2);
                sv[t][s] = a;
                h[t][0][s] = a; h[t][1][s] = a1 * w0; h[t][2][s] = a1 * w1;
                if constexpr (NT2 > 0) h[t][3][s] = a2 * fma(g.t2w[0], w0 * w0, g.t2w[1] * (w1 * w1));      // (z_cc = 0 in front of the first layer)
            }
            if constexpr (WQ) {      // every slot evaluates its own point (the tangent slots share the element point of slot 0)
                const double z = b1v + qx0 * w0 + qx1 * w1;
                double a, a1, a2;
                act_fwd<HPV_ACT_TANH>(z, a, a1, a2);
                QA[s] = a;
                QH[s] = q_tan ? a1 * (qcs == 1 ? w0 : w1) : a;
            }
        }
#pragma unroll
        for (int i = 1; i < L; ++i) {
            double z[NT][C][MF_KS];
#pragma unroll
            for (int t = 0; t < NT; ++t) {
                fz_layer<true>(lds + M::WT + (i - 1) * MF_KS * 64, lds + M::WR + (i - 1) * MF_KS * 16, lds + M::BH + (i - 1) * MF_KS * 64, lofs, h[t][0], z[t][0]);
                fz_layer<false>(lds + M::WT + (i - 1) * MF_KS * 64, lds + M::WR + (i - 1) * MF_KS * 16, nullptr, lofs, h[t][1], z[t][1]);
                fz_layer<false>(lds + M::WT + (i - 1) * MF_KS * 64, lds + M::WR + (i - 1) * MF_KS * 16, nullptr, lofs, h[t][2], z[t][2]);
                if constexpr (NT2 > 0)
                    fz_layer<false>(lds + M::WT + (i - 1) * MF_KS * 64, lds + M::WR + (i - 1) * MF_KS * 16, nullptr, lofs, h[t][3], z[t][3]);
            }
            if constexpr (GS) {
#pragma unroll
                for (int t = 0; t < NT; ++t) {
                    v2d* zs = gs_ptr(tile_of(k0 + t)) + (i - 1) * MF_KS * 64;
#pragma unroll
                    for (int s = 0; s < MF_KS; ++s) zs[s * 64] = v2d{z[t][1][s], z[t][2][s]};
                }
            }
            [[maybe_unused]] double QZ[MF_KS];
            if constexpr (WQ)
                fz_layer_m(lds + M::WT + (i - 1) * MF_KS * 64, lds + M::WR + (i - 1) * MF_KS * 16, lds + M::BH + (i - 1) * MF_KS * 64, lofs,
                           q_tan ? 0.0 : 1.0, QH, QZ);
#pragma unroll
            for (int t = 0; t < NT; ++t)
#pragma unroll
                for (int s = 0; s < MF_KS; ++s) {
                    double a, a1, a2;
                    act_fwd<HPV_ACT_TANH>(z[t][0][s], a, a1, a2);
                    sv[t][i * MF_KS + s] = a;
                    h[t][0][s] = a; h[t][1][s] = a1 * z[t][1][s]; h[t][2][s] = a1 * z[t][2][s];
                    if constexpr (NT2 > 0)      // h_cc = s'' (w0 z_x^2 + w1 z_y^2) + s' z_cc
                        h[t][3][s] = a2 * fma(g.t2w[0], z[t][1][s] * z[t][1][s], g.t2w[1] * (z[t][2][s] * z[t][2][s])) + a1 * z[t][3][s];
                }
            if constexpr (WQ) {
#pragma unroll
                for (int s = 0; s < MF_KS; ++s) {
                    double a, a1, a2;
                    act_fwd<HPV_ACT_TANH>(QZ[s], a, a1, a2);         // (tangent slots: of a tangent pre-activation, not used)
                    const double a4 = dpp_move<0x114>(a), a8 = dpp_move<0x118>(a);      // row_shr:4 / :8 = the value slot of my point
                    const double ab = qcs == 1 ? a4 : (qcs == 2 ? a8 : a);
                    QA[i * MF_KS + s] = ab;
                    QH[s] = q_tan ? (1.0 - ab * ab) * QZ[s] : ab;
                }
                if (q_tan) {     // (the reverse pass reads the tangent pre-activations back instead of recomputing them)
#pragma unroll
                    for (int s = 0; s < MF_KS; ++s) PKZ[((i - 1) * MF_KS + s) * 32 + q_cz] = QZ[s];
                }
            }
        }
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            const int k = k0 + t;
            // linear head
            double o[C];
#pragma unroll
            for (int ch = 0; ch < C; ++ch) {
                double v = 0.0;
#pragma unroll
                for (int s = 0; s < MF_KS; ++s) v += h[t][ch][s] * lds[M::W1O + (2 * MF_KS + s) * 64 + lofs];
                v = xrow_sum16(v);
                v = xrow_sum32(v);
                o[ch] = v;
            }
            o[0] += bo;
            if (k < n_el) {
                if (q == 0) {
                    const int lp = (tbase + wv + k * FZ_WAVES) * 16 + pt;     // point index inside the element
                    // the two integrated arrays: the channels u_x, u_y themselves, or (GEN) the terms' combinations of the channels
                    double g0 = o[1], g1 = o[2];
                    if constexpr (GEN) {
                        g0 = 0.0; g1 = 0.0;
#pragma unroll
                        for (int c = 1; c < C; ++c) { g0 = fma(wa0(c), o[c], g0); g1 = fma(wa1(c), o[c], g1); }
                    }
                    if constexpr (SPLIT) {   // tagged granules, fire and forget: the partners poll the granules themselves
                        if (!xstay) {
                            xg_publish(g.xg + (el * (2 * FZ_NQ) + lp) * 2, g0, xtag);
                            xg_publish(g.xg + (el * (2 * FZ_NQ) + FZ_NQ + lp) * 2, g1, xtag);
                        }
                    } else {
                        lds[M::CH + lp] = g0;
                        lds[M::CH + FZ_NQ + lp] = g1;
                    }
                }
            } else {
                // lossb = w mean((u_d - u)^2) (P2:122,127): adjoint of u kept in a register, per-tile partial sum to memory
                double udv;
                if constexpr (GS) udv = lds[M::XS + 2 * M::XLD + pt]; else udv = valid[t] ? g.ud[pp[t] - g.data_off] : 0.0;
                const double dd = valid[t] ? udv - o[0] : 0.0;
                gdat = g.data_scale * dd;
                const double sq = row_sum16(dd * dd);
                if (lane == 0) g.data_part[pp[t] / 16 - g.data_off / 16] = sq;
            }
            stash(k, sv[t]);
        }
        if constexpr (WQ) {
            double v = 0.0;
#pragma unroll
            for (int s = 0; s < MF_KS; ++s) v += QH[s] * lds[M::W1O + (2 * MF_KS + s) * 64 + lofs];
            v = xrow_sum16(v);
            v = xrow_sum32(v);
            if constexpr (GEN && !DQ) {
                // slot 1 (d/dx) writes array 0, slot 2 (d/dy) array 1 -- each needs the OTHER tangent slot of its point as well
                const double vo = dpp_move<0x104>(v), vm = dpp_move<0x114>(v);       // row_shl:4 = lane + 4, row_shr:4 = lane - 4
                const double vx = qcs == 1 ? v : vm, vy = qcs == 1 ? vo : v;
                const double gq = qcs == 1 ? fma(wa0(1), vx, wa0(2) * vy) : fma(wa1(1), vx, wa1(2) * vy);
                if (q == 0 && q_tan) lds[M::CH + (qcs - 1) * FZ_NQ + q_lp] = gq;
            } else {
                if (!DQ && q == 0 && q_tan) lds[M::CH + (qcs - 1) * FZ_NQ + q_lp] = v;
            }
            const double dd = q_vdat ? qud - (v + bo) : 0.0;
            gdat_q = g.data_scale * dd;
            const double sq = row_sum16(q == 0 ? dd * dd : 0.0);
            if (lane == 0) lds[M::RED + 8 + wv] = sq;
            if (!q_tan) {
#pragma unroll
                for (int j = 0; j < NSV; ++j) PKQ[j * 32 + q_ci] = QA[j];
            }
        }
    };
    if constexpr (!GS) {
        load_x(0, xn[0], vn[0], pn[0]);
        load_x(1, xn[1], vn[1], pn[1]);
    }
    int k0 = 0;
    if constexpr (QT) {      // six whole tiles: two trips of two, then the last two with the quarter tile beside them
        static_assert((FZ_TPE - (DQ ? 0 : 1)) / FZ_WAVES >= 2 && ((FZ_TPE - (DQ ? 0 : 1)) / FZ_WAVES) % 2 == 0, "trip plan of the QT instantiation: pairs of whole tiles");
#pragma unroll 1
        for (; k0 + 3 < n_own; k0 += 2) fwd_trip(k0, std::integral_constant<int, 2>{}, std::false_type{});
        fwd_trip(k0, std::integral_constant<int, 2>{}, std::true_type{});
    } else {
#pragma unroll 1
        for (; k0 + 1 < n_own; k0 += 2) fwd_trip(k0, std::integral_constant<int, 2>{}, std::false_type{});
        if (k0 < n_own) fwd_trip(k0, std::integral_constant<int, 1>{}, std::false_type{});
    }
    FZ_STAMP(2);
    if constexpr (SPLIT) {
        // the element's u_x, u_y from all partners (its own included), straight into the LDS channel array: the granules are
        // polled until every one carries this launch's tag -- the data's arrival is its own notification (no counter barrier:
        // that chain of store-acknowledge, fetch-add, poll and reload cost 8.7 k cycles)
        constexpr int NITG = (2 * FZ_NQ * 2 + FZ_BLOCK - 1) / FZ_BLOCK;
        bool ok = true;
        if (!xstay) ok = xg_gather<NITG, FZ_BLOCK>(g.xg + el * (2 * FZ_NQ) * 2, 2 * FZ_NQ * 2, xtag, (unsigned*)(lds + M::CH), tid);
        const int timed_out = __syncthreads_or(ok ? 0 : 1);
        if (timed_out || xstay) {
            // nothing of this iteration has been written: the kernels that follow skip the update (kernels_generic.hip), the host
            // reports -7, clears the flag and goes on (hpv_api.hip, sync_check)
            if (timed_out && tid == 0) __hip_atomic_store(g.xerr, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            return;      // (the launch counter is advanced by the kernel that FOLLOWS this launch: k_finalize, hpv_fused_dev.h)
        }
    }
    // (GS: an LDS-only barrier -- the stores of the saved values need no acknowledgement here: the wave that wrote them reads them back)
    if constexpr (GS && !SPLIT) pj_lds_barrier(); else __syncthreads();
    FZ_STAMP(3);
    // GS: the first reverse tile's pairs travel during the projection phase
    [[maybe_unused]] v2d BA[GS ? NP : 1], BB[GS ? NP : 1];
    [[maybe_unused]] auto request = [&](int k, v2d (&B)[GS ? NP : 1]) {
        if constexpr (GS) {
            const v2d* zs = gs_ptr(tile_of(k));
            constexpr int NF = NZP + NSV / 2;      // full pairs
#pragma unroll
            for (int j = 0; j < NF; ++j) B[j] = zs[j * 64];
            // (an odd s count: the last value alone -- a 16-byte load whose upper half nobody reads makes the compiler wait for the
            //  whole request as soon as it reuses that register)
            if constexpr (NSV & 1) B[NF] = v2d{reinterpret_cast<const double*>(zs + NF * 64 - lane)[lane], 0.0};
        }
    };
    if constexpr (GS) { if (n_own > 0) request(0, BA); }
    if constexpr (QT) {      // lossb partial of the workgroup's boundary / data tile: the four waves' quarters (P2:122,127)
        if (tid == 0 && dtile < g.ntiles)
            g.data_part[dtile - g.data_off / 16] = (lds[M::RED + 8] + lds[M::RED + 9]) + (lds[M::RED + 10] + lds[M::RED + 11]);
    }

    // =============================================================================================
    // phase P: projection of the element from LDS (two one-hot terms: term t integrates channel 1 + t)
    // =============================================================================================
    {
        const double* G = lds + M::CH;
        // T_t[j][r] = sum_i AX_t[r][i] G_t[j][i]
        for (int o = tid; o < 2 * FZ_QY * FZ_NTX; o += FZ_BLOCK) {
            const int t = o / (FZ_QY * FZ_NTX), j = (o / FZ_NTX) % FZ_QY, r = o % FZ_NTX;
            const double* ax = lds + M::AX + (t * FZ_NTX + r) * FZ_QX;
            const double* gr = G + t * FZ_NQ + j * FZ_QX;
            double acc = 0.0;
#pragma unroll
            for (int i = 0; i < FZ_QX; ++i) acc = fma(ax[i], gr[i], acc);
            lds[M::T + o] = acc;
        }
        __syncthreads();
        // per-term partial of U[k][r] = c_t sum_j BY_t[k][j] T_t[j][r]
        if (tid < 2 * FZ_NR) {
            const int t = tid / FZ_NR, o = tid % FZ_NR, kk = o / FZ_NTX, r = o % FZ_NTX;
            const double* by = lds + M::BY + (t * FZ_NTY + kk) * FZ_QY;
            const double* tt = lds + M::T + t * FZ_QY * FZ_NTX + r;
            double acc = 0.0;
#pragma unroll
            for (int j = 0; j < FZ_QY; ++j) acc = fma(by[j], tt[j * FZ_NTX], acc);
            lds[M::UP + tid] = (t == 0 ? pc0 : pc1) * acc;
        }
        __syncthreads();
        double sq = 0.0;
        const double sc = 2.0 / (double)rnr;
        [[maybe_unused]] double deps = 0.0;      // GEN: this thread's share of d loss_e / d eps
        if (tid < FZ_NR) {
            double u;
            if constexpr (GEN) {
                // (UP holds the terms WITHOUT their factor eps: U = m_0 UP_0 + m_1 UP_1 - F, and a term that carries eps as a factor
                //  contributes (2/NR) U UP_t to d loss_e / d eps -- P3:171)
                const double u0 = lds[M::UP + tid], u1 = lds[M::UP + FZ_NR + tid];
                u = fma(gm0, u0, gm1 * u1) - pF;
                deps = sc * u * fma(ge0, u0, ge1 * u1);
            } else {
                u = (lds[M::UP + tid] + lds[M::UP + FZ_NR + tid]) - pF;
            }
            lds[M::U + tid] = u;
            if (ro_on) pa.R[ro_idx] = u;
            sq = u * u;
        }
        if (wv < 2) {
            sq = pj_wave_sum_dpp(sq);
            if (lane == 0) lds[M::RED + wv] = sq;
        }
        __syncthreads();
        if (tid == 0) pa.loss_e[e] = (lds[M::RED] + lds[M::RED + 1]) / (double)rnr;
        // adjoint: S_t[k][i] = (2/NR) c_t sum_r AX_t[r][i] U[k][r];  Gbar_t[j][i] = sum_k BY_t[k][j] S_t[k][i]
        for (int o = tid; o < 2 * FZ_NTY * FZ_QX; o += FZ_BLOCK) {
            const int t = o / (FZ_NTY * FZ_QX), kk = (o / FZ_QX) % FZ_NTY, i = o % FZ_QX;
            const double* ax = lds + M::AX + t * FZ_NTX * FZ_QX + i;
            const double* ur = lds + M::U + kk * FZ_NTX;
            double acc = 0.0;
#pragma unroll
            for (int r = 0; r < FZ_NTX; ++r) acc = fma(ax[r * FZ_QX], ur[r], acc);
            if constexpr (GEN) lds[M::S + o] = acc * sc * (t == 0 ? pc0 * gm0 : pc1 * gm1);
            else lds[M::S + o] = acc * sc * (t == 0 ? pc0 : pc1);
        }
        __syncthreads();
        // (GEN, one term: only array 0 gets an adjoint; array 1 holds E = dG_0 / d eps, which meets it here: d eps += Gbar_0 E)
        const int n_adj = (GEN && !gtwo) ? FZ_NQ : 2 * FZ_NQ;
        for (int o = tid; o < n_adj; o += FZ_BLOCK) {
            const int t = o / FZ_NQ, j = (o / FZ_QX) % FZ_QY, i = o % FZ_QX;
            const double* by = lds + M::BY + t * FZ_NTY * FZ_QY + j;
            const double* sr = lds + M::S + t * FZ_NTY * FZ_QX + i;
            double acc = 0.0;
#pragma unroll
            for (int kk = 0; kk < FZ_NTY; ++kk) acc = fma(by[kk * FZ_QY], sr[kk * FZ_QX], acc);
            if constexpr (GEN) { if (!gtwo) deps = fma(acc, lds[M::CH + FZ_NQ + o], deps); }
            lds[M::CH + o] = acc;
        }
        if constexpr (GEN) {
            if (pa.pd.has_eps) {      // (kernel-uniform)
                // (into the head of the x-tables: their last reader was the S step, two barriers back; wave 0's own transpose tiles later)
                deps = pj_wave_sum_dpp(deps);
                if (lane == 0) lds[M::AX + wv] = deps;
            }
        }
        __syncthreads();
        if constexpr (GEN) {
            // (the partials sit in wave 0's own part of the transpose region -- static_assert at the top: it reads them before it writes there)
            if (pa.pd.has_eps && tid == 0) pa.deps_e[e] = (lds[M::AX] + lds[M::AX + 1]) + (lds[M::AX + 2] + lds[M::AX + 3]);
        }
    }

    // =============================================================================================
    // phase R: reverse pass (tangent pre-activations recomputed from s)
    // =============================================================================================
    FZ_STAMP(4);
    double* TAB = lds + M::TR + wv * M::TR_WAVE;
#pragma unroll
    for (int i = 0; i < LH; ++i) { dWacc[i] = v4d{0.0, 0.0, 0.0, 0.0}; dS10[i] = 0.0; dS01[i] = 0.0; accC[i] = 0.0; }
    dbo = 0.0;
#pragma unroll
    for (int s = 0; s < MF_KS; ++s) {
        dWo[s] = 0.0; dW1[0][s] = 0.0; dW1[1][s] = 0.0;
#pragma unroll
        for (int i = 0; i < L; ++i) db[i][s] = 0.0;
    }

#ifdef HPV_FZ_TIMING
    fz_last = clock64();
    fz_n_own = n_own;
    for (int i = 0; i < 2 + 2 * L; ++i) fz_seg[i] = 0;
#endif
    // one reverse tile; GS: `B` = the tile's pairs, requested a tile ahead
    auto rev_tile = [&](int k, [[maybe_unused]] const v2d (&B)[GS ? NP : 1]) {
#ifdef HPV_FZ_TIMING
        fz_last = clock64();
#endif
        double x0, x1;
        if constexpr (GS) {
            const int lp = k < n_el ? (tbase + wv + k * FZ_WAVES) * 16 + pt : FZ_NQ + pt;
            x0 = lds[M::XS + lp]; x1 = lds[M::XS + M::XLD + lp];
        } else {
            const long tile = tile_of(k);
            const long p = tile * 16 + pt;
            const bool valid = p < g.N;
            x0 = valid ? g.X[p] : 0.0; x1 = valid ? g.X[g.N + p] : 0.0;
        }
        int lofs = lane;
        asm volatile("" : "+v"(lofs));
        double sv[NSV];
        if constexpr (GS) {
#pragma unroll
            for (int j = 0; j < NSV; ++j) sv[j] = B[NZP + j / 2][j & 1];
        } else if (k < n_lds) {
            const double* pk = k == 0 ? PKw : PKw2;
            // (NT2: the slot is formed HERE -- a third parking pointer kept alive across the phases sent the register allocator to
            //  scratch memory, 420 - 540 scratch accesses per instantiation and 148 instead of 58 us per iteration: profiles/r06_notes.md)
            if constexpr (NT2 > 0) pk = lds + M::PK + (k == 0 ? wv : (k == 1 ? 4 + wv : 8 + (wv & 1))) * (NSV * 64) + lane;
#pragma unroll
            for (int j = 0; j < NSV; ++j) sv[j] = pk[j * 64];
        } else {
            switch (k - n_lds) {
#define FZ_FETCH(K) case K: if constexpr (K < NREG) acc_get_all<AFULL + K * 2 * NSV, NSV>(sv); break;
                case 0:
                    if constexpr (NREG < 1) {
                    } else if constexpr (NPART > 0) {
                        const double* pp = lds + M::PP + wv * (NPART * 64) + lane;
#pragma unroll
                        for (int j = 0; j < NPART; ++j) sv[j] = pp[j * 64];
                        acc_get_from<ABASE, NSV, NPART>(sv);
                    } else {
                        acc_get_all<ABASE, NSV>(sv);
                    }
                    break;
                FZ_FETCH(1) FZ_FETCH(2) FZ_FETCH(3) FZ_FETCH(4)
#undef FZ_FETCH
                default:
#pragma unroll
                    for (int j = 0; j < NSV; ++j) sv[j] = 0.0;
            }
        }
        double gb[C];
        if (k < n_el) {
            const int lp = (tbase + wv + k * FZ_WAVES) * 16 + pt;
            gb[0] = 0.0; gb[1] = lds[M::CH + lp]; gb[2] = lds[M::CH + FZ_NQ + lp];
            if constexpr (GEN) {      // the channels' adjoints from the integrated arrays' (one term: array 1 carries none)
                const double g0 = gb[1], g1 = gb[2];
#pragma unroll
                for (int c = 1; c < C; ++c) gb[c] = fma(wa0(c), g0, wb1(c) * g1);
            }
        } else {
            gb[0] = gdat; gb[1] = 0.0; gb[2] = 0.0;
            if constexpr (NT2 > 0) gb[3] = 0.0;
        }
        // tangent pre-activations of every hidden layer: layer 0 has z_c = W1[c,:]; layer i: z_c = (sigma'(z_{i-1}) z_c,{i-1}) W_i
        // (GS: read back; otherwise recomputed from s on the matrix pipe)
        double zc[L][2][MF_KS];
        [[maybe_unused]] double zq[L][MF_KS];       // NT2: second-order tangent pre-activations z_cc of the mixed channel (layer 0: zero)
        // s'' z_c^2 as the mixed second tangent sees it
        [[maybe_unused]] auto sq2 = [&](int i, int s) -> double { return fma(g.t2w[0], zc[i][0][s] * zc[i][0][s], g.t2w[1] * (zc[i][1][s] * zc[i][1][s])); };
#define ZC(I, CC, SS) zc[(I)][(CC)][(SS)]
#pragma unroll
        for (int s = 0; s < MF_KS; ++s) {
            zc[0][0][s] = lds[M::W1O + (0 * MF_KS + s) * 64 + lofs];
            zc[0][1][s] = lds[M::W1O + (1 * MF_KS + s) * 64 + lofs];
        }
#pragma unroll
        for (int i = 1; i < L; ++i) {
            if constexpr (GS) {
#pragma unroll
                for (int s = 0; s < MF_KS; ++s) { zc[i][0][s] = B[(i - 1) * MF_KS + s][0]; zc[i][1][s] = B[(i - 1) * MF_KS + s][1]; }
            } else {
                double hx[MF_KS], hy[MF_KS];
                [[maybe_unused]] double hq[MF_KS];
#pragma unroll
                for (int s = 0; s < MF_KS; ++s) {
                    const double a = sv[(i - 1) * MF_KS + s], a1 = 1.0 - a * a;
                    hx[s] = a1 * ZC(i - 1, 0, s); hy[s] = a1 * ZC(i - 1, 1, s);
                    if constexpr (NT2 > 0) hq[s] = (-2.0 * a * a1) * sq2(i - 1, s) + (i > 1 ? a1 * zq[i > 1 ? i - 1 : 0][s] : 0.0);
                }
                fz_layer<false>(lds + M::WT + (i - 1) * MF_KS * 64, lds + M::WR + (i - 1) * MF_KS * 16, nullptr, lofs, hx, zc[i][0]);
                fz_layer<false>(lds + M::WT + (i - 1) * MF_KS * 64, lds + M::WR + (i - 1) * MF_KS * 16, nullptr, lofs, hy, zc[i][1]);
                if constexpr (NT2 > 0) fz_layer<false>(lds + M::WT + (i - 1) * MF_KS * 64, lds + M::WR + (i - 1) * MF_KS * 16, nullptr, lofs, hq, zq[i]);
            }
        }

        FZ_SEG(0);
        double hbar[C][MF_KS], zbar[C][MF_KS];
        // channel 3 (NT2) of layer i's outputs: s'' (w0 z_x^2 + w1 z_y^2) + s' z_cc
        [[maybe_unused]] auto hv3 = [&](int i, int s) -> double {
            const double a = sv[i * MF_KS + s], a1 = 1.0 - a * a;
            return (-2.0 * a * a1) * sq2(i, s) + (i > 0 ? a1 * zq[i][s] : 0.0);
        };
        // ---- linear head ----
#pragma unroll
        for (int s = 0; s < MF_KS; ++s) {
            const double a = sv[(L - 1) * MF_KS + s], a1 = 1.0 - a * a;
            const double wo = lds[M::W1O + (2 * MF_KS + s) * 64 + lofs];
            if (k >= n_el) dWo[s] = fma(a, gb[0], dWo[s]);       // (wave-uniform; element tiles have no adjoint of the value channel)
            dWo[s] = fma(a1 * zc[L - 1][0][s], gb[1], dWo[s]);
            dWo[s] = fma(a1 * zc[L - 1][1][s], gb[2], dWo[s]);
            hbar[0][s] = gb[0] * wo; hbar[1][s] = gb[1] * wo; hbar[2][s] = gb[2] * wo;
            if constexpr (NT2 > 0) { dWo[s] = fma(hv3(L - 1, s), gb[3], dWo[s]); hbar[3][s] = gb[3] * wo; }
        }
        if (q == 0) dbo += gb[0];
        FZ_SEG(1);

        // ---- hidden layers, last to first ----
#pragma unroll
        for (int i = L - 1; i >= 0; --i) {
#pragma unroll
            for (int s = 0; s < MF_KS; ++s) {
                const double a = sv[i * MF_KS + s];
                const double a1 = 1.0 - a * a, a2 = -2.0 * a * a1;
                zbar[1][s] = hbar[1][s] * a1;
                zbar[2][s] = hbar[2][s] * a1;
                double zb = hbar[0][s] * a1 + a2 * (hbar[1][s] * ZC(i, 0, s) + hbar[2][s] * ZC(i, 1, s));
                if constexpr (NT2 > 0) {
                    // the mixed second tangent rides on both first tangents (hand-derived third-order reverse pass)
                    const double hb = hbar[3][s], a3 = -2.0 * a1 * (1.0 - 3.0 * a * a);
                    zbar[3][s] = hb * a1;
                    zbar[1][s] = fma(2.0 * hb * a2 * g.t2w[0], ZC(i, 0, s), zbar[1][s]);
                    zbar[2][s] = fma(2.0 * hb * a2 * g.t2w[1], ZC(i, 1, s), zbar[2][s]);
                    zb = fma(hb, a3 * sq2(i, s) + (i > 0 ? a2 * zq[i][s] : 0.0), zb);
                }
                zbar[0][s] = zb;
                db[i][s] += zb;
            }
            if (i == 0) {
#pragma unroll
                for (int s = 0; s < MF_KS; ++s) {
                    dW1[0][s] += x0 * zbar[0][s] + zbar[1][s];
                    dW1[1][s] += x1 * zbar[0][s] + zbar[2][s];
                }
            } else {
                // weight gradient dW_i[in][out] = sum_pt sum_ch h_{i-1,ch}[pt][in] zbar_ch[pt][out]: the operands are needed
                // point-major -> per-wave LDS transpose tiles, all channels written first (one wave-level sync)
                if constexpr (NT2 == 0) {      // (three channels: ONE pass -- the headline instantiations' text, kept as it was)
                pj_wave_sync();
#pragma unroll
                for (int ch = 0; ch < FZ_C; ++ch) {
                    double* TA = TAB + (2 * ch) * (MF_TRB * MF_LD);
                    double* TB = TA + MF_TRB * MF_LD;
#pragma unroll
                    for (int s = 0; s < MF_KS; ++s) {
                        const double a = sv[(i - 1) * MF_KS + s], a1 = 1.0 - a * a;
                        const double hv = ch == 0 ? a : a1 * ZC(i - 1, ch - 1, s);
                        TA[(4 * s + q) * MF_LD + pt] = hv;
                        TB[(4 * s + q) * MF_LD + pt] = zbar[ch][s];
                    }
                }
                // hbar_{i-1}^T = W_i zbar^T  (independent of the transposes: issued while the LDS writes above land)
#pragma unroll
                for (int ch = 0; ch < FZ_C; ++ch) {
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
                FZ_SEG(2 + 2 * (L - 1 - i));
                pj_wave_sync();
#pragma unroll
                for (int ch = 0; ch < FZ_C; ++ch) {
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
                FZ_SEG(3 + 2 * (L - 1 - i));
                } else {
                pj_wave_sync();
                // (C > TRG, the four-channel forms: the transposes go through LDS in passes of TRG channels; `put` / `mul` = one pass)
                auto tr_put = [&](int c0) {
#pragma unroll
                    for (int cc = 0; cc < TRG; ++cc) {
                        const int ch = c0 + cc;
                        double* TA = TAB + (2 * cc) * (MF_TRB * MF_LD);
                        double* TB = TA + MF_TRB * MF_LD;
#pragma unroll
                        for (int s = 0; s < MF_KS; ++s) {
                            const double a = sv[(i - 1) * MF_KS + s], a1 = 1.0 - a * a;
                            double hv;
                            if constexpr (NT2 > 0) hv = ch == 0 ? a : (ch == 3 ? hv3(i - 1, s) : a1 * ZC(i - 1, ch == 3 ? 0 : ch - 1, s));
                            else hv = ch == 0 ? a : a1 * ZC(i - 1, ch - 1, s);
                            TA[(4 * s + q) * MF_LD + pt] = hv;
                            TB[(4 * s + q) * MF_LD + pt] = zbar[ch][s];
                        }
                    }
                };
                tr_put(0);
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
                FZ_SEG(2 + 2 * (L - 1 - i));
                pj_wave_sync();
                auto tr_mul = [&]() {
#pragma unroll
                    for (int ch = 0; ch < TRG; ++ch) {
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
                };
                tr_mul();
#pragma unroll
                for (int c0 = TRG; c0 < C; c0 += TRG) {      // the further passes: channels c0 .. c0 + TRG - 1 through the same tiles
                    pj_wave_sync();
                    tr_put(c0);
                    pj_wave_sync();
                    tr_mul();
                }
                FZ_SEG(3 + 2 * (L - 1 - i));
                }
            }
        }
        FZ_SEG(1 + 2 * L);
    };
#undef ZC
    if constexpr (GS) {
        // two tiles per trip of the loop (ping-pong buffers: no register copies), tile k + 1's pairs in flight during tile k
#pragma unroll 1
        for (int k = 0; k < n_own; k += 2) {
            // (requests are unconditional -- past the wave's last tile they repeat it: a branch around a request would make the
            //  compiler's wait for the OTHER buffer a wait for everything in flight)
            request(k + 1 < n_own ? k + 1 : n_own - 1, BB);
            rev_tile(k, BA);
            __builtin_amdgcn_sched_barrier(0);       // (nothing of tile k + 1 -- it would wait for the request above -- moves up into tile k)
            if (k + 1 < n_own) {
                request(k + 2 < n_own ? k + 2 : n_own - 1, BA);
                rev_tile(k + 1, BB);
            }
        }
    } else {
#ifdef HPV_FZ_REV2
        constexpr bool REV2 = FZ_TPE <= 16;       // (where the stash leaves the compiler the registers for it: a125 against a95)
#else
        constexpr bool REV2 = false;
#endif
        if constexpr (REV2) {   // experiment: two tiles per trip, the scheduler may overlap tile k + 1's recompute with tile k's tail
            int k = 0;
#pragma unroll 1
            for (; k + 1 < n_own; k += 2) { rev_tile(k, BA); rev_tile(k + 1, BA); }
            if (k < n_own) rev_tile(k, BA);
        } else {
#pragma unroll 1
            for (int k = 0; k < n_own; ++k) rev_tile(k, BA);
        }
    }

    if constexpr (QT) {
        // ---- the packed quarter tile, reverse: the whole-tile steps above for ONE packed operand (every slot's adjoint at once) ----
        int lofs = lane;
        asm volatile("" : "+v"(lofs));
        double X0, X1;                                         // (consumed at the very end: first-layer weight gradient)
        if constexpr (GS) { X0 = lds[M::XS + q_xs]; X1 = lds[M::XS + M::XLD + q_xs]; } else { X0 = g.X[q_p]; X1 = g.X[g.N + q_p]; }
        double AAq[NSV];
#pragma unroll
        for (int j = 0; j < NSV; ++j) AAq[j] = PKQ[j * 32 + q_ci];
        // (branch-free on purpose: per-lane selects of loaded values and 0 / 1 masks -- a conditional LDS read or a conditional
        //  expression with work in it becomes an exec-masked block of its own and cuts the schedule into pieces)
        const double mval = q_tan ? 0.0 : 1.0;                        // value-like slots (element value, data point)
        // adjoint of the slot's output: d/dx, d/dy slots from the projection, the data slot from the boundary term, value slot none
        double gch = lds[M::CH + (q_tan ? (qcs - 1) * FZ_NQ : 0) + q_lp];
        if constexpr (GEN && !DQ) {      // the tangent slots' adjoints from BOTH integrated arrays' (slot 1 = d/dx, slot 2 = d/dy)
            const double g0 = lds[M::CH + q_lp], g1 = lds[M::CH + FZ_NQ + q_lp];
            gch = qcs == 1 ? fma(wa0(1), g0, wb1(1) * g1) : fma(wa0(2), g0, wb1(2) * g1);
        }
        const double GB = q_tan ? (DQ ? 0.0 : gch) : (qcs == 3 ? gdat_q : 0.0);
        // packed layer inputs H_i from s and the tangent pre-activations (tangent slots; 0 elsewhere) the forward pass left in LDS
        double Hq[L][MF_KS], ZCq[L][MF_KS];
#pragma unroll
        for (int s = 0; s < MF_KS; ++s) {
            const double w0 = lds[M::W1O + (0 * MF_KS + s) * 64 + lofs], w1 = lds[M::W1O + (1 * MF_KS + s) * 64 + lofs];
            const double a = AAq[s];
            ZCq[0][s] = qcs == 1 ? w0 : (qcs == 2 ? w1 : 0.0);
            const double ht = (1.0 - a * a) * ZCq[0][s];
            Hq[0][s] = q_tan ? ht : a;
        }
#pragma unroll
        for (int i = 1; i < L; ++i) {
#pragma unroll
            for (int s = 0; s < MF_KS; ++s) {
                const double a = AAq[i * MF_KS + s];
                const double zr = PKZ[((i - 1) * MF_KS + s) * 32 + q_cz];      // (value-like lanes read a neighbour's, unused)
                const double zc = q_tan ? zr : 0.0;
                const double ht = (1.0 - a * a) * zc;
                ZCq[i][s] = zc;
                Hq[i][s] = q_tan ? ht : a;
            }
        }
        double HB[MF_KS], ZB[MF_KS];
        // linear head
#pragma unroll
        for (int s = 0; s < MF_KS; ++s) {
            const double wo = lds[M::W1O + (2 * MF_KS + s) * 64 + lofs];
            dWo[s] = fma(Hq[L - 1][s], GB, dWo[s]);
            HB[s] = GB * wo;
        }
        dbo += (q == 0 && qcs == 3) ? GB : 0.0;
        // hidden layers, last to first
#pragma unroll
        for (int i = L - 1; i >= 0; --i) {
#pragma unroll
            for (int s = 0; s < MF_KS; ++s) {
                const double a = AAq[i * MF_KS + s];
                const double a1 = 1.0 - a * a, a2 = -2.0 * a * a1;
                const double t = HB[s] * ZCq[i][s];                                     // tangent slots: hbar_c z_c
                const double t4 = dpp_move<0x104>(t), t8 = dpp_move<0x108>(t);        // row_shl:4 / :8 = my point's d/dx, d/dy slots
                const double zb = fma(a2 * mval, t4 + t8, HB[s] * a1);
                ZB[s] = zb;
                db[i][s] = fma(mval, zb, db[i][s]);
            }
            if (i == 0) {
                const double c0 = q_tan ? (qcs == 1 ? 1.0 : 0.0) : X0, c1 = q_tan ? (qcs == 2 ? 1.0 : 0.0) : X1;
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
    if constexpr (!MULTI) {
        break;
    } else {
        // every accumulator in one fixed order = the slot order of the spill block.  MODE 0: spill (first element: store; later:
        // load-add-store), MODE 1: add the spilled sums back into the registers.  The earlier sums travel through the TOP 2 NACC
        // AGPRs by hand (acc_load_all: all 45 loads in flight, one round trip; the stash is dead here and the build guard keeps the
        // compiler below min(ABASE, LBASE)) -- through compiler registers they were six serialized round trips per element
        // (measured: 185 against 166 us on 32x32 elements of 16x16 points) or 90 registers parked inside the stash.
        constexpr int LBASE = 256 - 2 * NACC;
        auto acc_walk = [&](auto mode_, bool first) {
            constexpr int MODE = decltype(mode_)::value;
            // (the block's base address is laundered here: hoisted out of the element loop the slot addresses were 45 64-bit values
            //  parked in a106..a195 across the forward and reverse phases)
            double* asp = ASP;
            asm volatile("" : "+v"(asp));
            double t[NACC];
#pragma unroll
            for (int jj = 0; jj < NACC; ++jj) t[jj] = 0.0;
            if (!first) {
                acc_load_all<LBASE, NACC>(asp);
                acc_load_wait();
                acc_get_all<LBASE, NACC>(t);
            }
            int j = 0;
            auto one = [&](double v) -> double {
                const double r = v + t[j];
                if constexpr (MODE == 0) asp[j * 64] = r;
                ++j;
                return r;
            };
#pragma unroll
            for (int i = 0; i < LH; ++i) {
#pragma unroll
                for (int r = 0; r < 4; ++r) dWacc[i][r] = one(dWacc[i][r]);
                dS10[i] = one(dS10[i]); dS01[i] = one(dS01[i]); accC[i] = one(accC[i]);
            }
#pragma unroll
            for (int s = 0; s < MF_KS; ++s) {
#pragma unroll
                for (int i = 0; i < L; ++i) db[i][s] = one(db[i][s]);
                dW1[0][s] = one(dW1[0][s]); dW1[1][s] = one(dW1[1][s]); dWo[s] = one(dWo[s]);
            }
            dbo = one(dbo);
        };
        const long e_next = e + gridDim.x;
        if (e_next >= g.proj_n_elem) {
            // the last element of this workgroup: the earlier elements' sums come back into the registers, then the epilogue
            if (!m_first) acc_walk(std::integral_constant<int, 1>{}, false);
            break;
        }
        acc_walk(std::integral_constant<int, 0>{}, m_first);
        m_first = false;
        __syncthreads();          // every wave has left the reverse phase: the transpose region and the channel array are free
        e = e_next;
        pc0 = pa.coef[e]; pc1 = pa.coef[pa.coef_stride + e];
        ro_idx = e * rnr + ro_k * rnx + ro_r;
        pF = (pa.F && ro_on) ? pa.F[ro_idx] : 0.0;
        dtile = g.ntiles;         // the boundary / data tiles rode with the first element
    }
    }   // element loop
    FZ_STAMP(5);
    // ---- epilogue: per-wave partials -> LDS -> one gradient row per workgroup ----
    __syncthreads();
    FZ_STAMP(6);
    double* WP = lds + M::EPI + (long)wv * g.P;     // every one of the P entries is written below
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
    const double* W0 = lds + M::EPI;
    double* row = g.GPART + (long)blockIdx.x * g.P;
    for (int idx = tid; idx < g.P; idx += FZ_BLOCK) {
        double acc = 0.0;
#pragma unroll
        for (int w = 0; w < FZ_WAVES; ++w) acc += W0[(long)w * g.P + idx];
        row[idx] = acc;      // (streaming / write-through stores measured: no gain, profiles/r06_notes.md 12)
    }
#ifdef HPV_FZ_TIMING
    if (lane == 0 && pa.GBAR) {   // phase durations in shader cycles: [block][wave][8], into the (otherwise unused) adjoint buffer
        FZ_STAMP(7);
        double* o = pa.GBAR + ((long)blockIdx.x * 4 + wv) * 10;
        for (int i = 1; i < 7; ++i) o[i] = (double)(fz_t[i + 1] - fz_t[i]);
        o[0] = (double)(fz_t[1] - fz_start);                   // staging + first barrier
        const long long fz_end = wall_clock64();
        o[7] = (double)(fz_end - fz_wall) * 0.01;              // whole wave, microseconds (100 MHz constant clock)
        o[8] = (double)(fz_wall & 0xffffffffffll) * 0.01;      // absolute start / end, microseconds: launch skew across workgroups
        o[9] = (double)(fz_end & 0xffffffffffll) * 0.01;
        if (!SPLIT && g.OUT) {     // reverse-body segments of this wave (sums over its tiles) + its tile count, into the channel buffer
            double* os = g.OUT + ((long)blockIdx.x * 4 + wv) * 12;
            for (int i = 0; i < 2 + 2 * L; ++i) os[i] = (double)fz_seg[i];
            os[2 + 2 * L] = (double)fz_n_own;
        }
    }
#endif
}

// ================================================================================================================
// Small elements: 10x10 quadrature points, 5x5 test functions (BASELINE config 3, the 2-D reference defaults).
// An element is 7 tiles (the last one partial), so the latency of ONE tile's forward + reverse is what an iteration
// costs: ONE workgroup of EIGHT wavefronts per element (two per SIMD, 256 registers each), ONE tile per wave -- waves
// 0..6 the element's tiles, wave 7 one boundary/data tile -- and nothing is recomputed or parked: s and the tangent
// pre-activations of the wave's single tile simply stay in registers across the projection barrier.  The separate
// path costs four launches here (forward 7.5 + projection 7.1 + reverse 11.5 + finalize 4.8 us at config 3, each with its own
// launch + weight-staging prologue for one tile of work per wave); this kernel + finalize are two.
// ================================================================================================================
#define SM_WAVES 8
#define SM_BLOCK (SM_WAVES * 64)
#define SM_QX 10
#define SM_QY 10
#define SM_NTX 5
#define SM_NTY 5
#define SM_NQ (SM_QX * SM_QY)
#define SM_NR (SM_NTX * SM_NTY)
#define SM_TPE ((SM_NQ + 15) / 16)

template <int L>
struct SmLds {
    static constexpr int LH = L > 1 ? L - 1 : 0;
    static constexpr int WT = 0;                           // weight fragments: as FzLds
    static constexpr int BH = WT + LH * MF_KS * 64;
    static constexpr int WR = BH + LH * MF_KS * 64;
    static constexpr int WN = WR + LH * MF_KS * 16;
    static constexpr int WRB = WN + LH * MF_KS * 64;
    static constexpr int W1O = WRB + LH * MF_KS * 16;
    static constexpr int CH = W1O + 4 * MF_KS * 64;        // [2][NQ] u_x, u_y -> adjoints
    static constexpr int AX = CH + 2 * SM_NQ;              // [2][NTX][QX]
    static constexpr int BY = AX + 2 * SM_NTX * SM_QX;     // [2][NTY][QY]
    static constexpr int T = BY + 2 * SM_NTY * SM_QY;      // [2][QY][NTX]
    static constexpr int UP = T + 2 * SM_QY * SM_NTX;      // [2][NR]
    static constexpr int U = UP + 2 * SM_NR;               // [NR]
    static constexpr int S = U + SM_NR;                    // [2][NTY][QX]
    static constexpr int RED = S + 2 * SM_NTY * SM_QX;     // [16]
    static constexpr int TR = RED + 16;                    // per-wave transpose pair (ONE channel at a time) | epilogue rows
    static constexpr int TR_WAVE = 2 * MF_TRB * MF_LD;
    static constexpr int total(int P) { return TR + (SM_WAVES * TR_WAVE > SM_WAVES * P ? SM_WAVES * TR_WAVE : SM_WAVES * P); }
};

template <int L>
__global__ void __launch_bounds__(SM_BLOCK, 1) k_iter_small(MfmaArgs g) {
    using M = SmLds<L>;
    extern __shared__ __attribute__((aligned(16))) double lds[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int q = lane >> 4, pt = lane & 15;
    const long e = blockIdx.x;
    const double* __restrict__ th = g.theta;
    const ProjArgs& pa = g.pa;
#ifdef HPV_FZ_TIMING
    long long fz_t[8];
    FZ_STAMP(0);
    const long long fz_wall = wall_clock64();
#endif

    // ---- stage weight fragments and projection tables ----
    [[maybe_unused]] constexpr int TNX = SM_NTX, TNY = SM_NTY, TQX = SM_QX, TQY = SM_QY;
    const int rnx = pa.pd.ntx, rny = pa.pd.nty, rnr = rnx * rny;      // the run's test functions per direction (<= NTX, NTY: see k_iter_fused)
    static_assert(SM_NTX * SM_QX == SM_NTY * SM_QY, "table staging walks both tables with one index");
    {
        // layer index as a compile-time constant (kernarg offsets become scalar loads instead of a dependent vector load per
        // lane), every global read issued before the first LDS store (one memory round trip)
        constexpr int N1 = 4 * MF_KS * 64, IT1 = (N1 + SM_BLOCK - 1) / SM_BLOCK;
        constexpr int ITW = (MF_KS * 64 + SM_BLOCK - 1) / SM_BLOCK;
        double vwt[L > 1 ? L - 1 : 1][ITW], vbh[L > 1 ? L - 1 : 1][ITW], vwn[L > 1 ? L - 1 : 1][ITW];
        double vwr[L > 1 ? L - 1 : 1], vwrb[L > 1 ? L - 1 : 1], v1[IT1];
#pragma unroll
        for (int i_ = 1; i_ < L; ++i_) {
            const int wo = g.woff[i_], bo_ = g.boff[i_];
#pragma unroll
            for (int it = 0; it < ITW; ++it) {
                const int f = it * SM_BLOCK + tid, fc = f < MF_KS * 64 ? f : 0;
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
            const int f = it * SM_BLOCK + tid, fc = f < N1 ? f : 0;
            const int ln = fc & 63, s_ = (fc >> 6) % MF_KS, c_ = fc / (64 * MF_KS);
            const int j = 4 * s_ + (ln >> 4);
            v1[it] = th[(c_ < 2 ? w0o + c_ * MF_H : (c_ == 2 ? wLo : b0o)) + j];
        }
        constexpr int NTAB = 2 * TNX * TQX, ITT = (NTAB + SM_BLOCK - 1) / SM_BLOCK;
        const int dx0 = pa.pd.t[0].dx, dx1 = pa.pd.t[1].dx, dy0 = pa.pd.t[0].dy, dy1 = pa.pd.t[1].dy;
        double vax[ITT], vby[ITT];
#pragma unroll
        for (int it = 0; it < ITT; ++it) {
            const int f = it * SM_BLOCK + tid, fc = f < NTAB ? f : 0;
            const int tt_ = fc / (TNX * TQX), ti_ = fc % (TNX * TQX);
            // (fewer test functions than the instantiation's NTX x NTY: the tables of the missing ones are zero -- their residuals are
            //  exactly 0 and leave the sums alone; R, F and the means below use the run's own counts rnx, rny)
            const int rr_ = ti_ / TQX, ii_ = ti_ % TQX;
            vax[it] = rr_ < rnx ? pa.wtx[((long)(tt_ ? dx1 : dx0) * rnx + rr_) * TQX + ii_] : 0.0;
            vby[it] = rr_ < rny ? pa.wty[((long)(tt_ ? dy1 : dy0) * rny + rr_) * TQY + ii_] : 0.0;
        }
#pragma unroll
        for (int i_ = 1; i_ < L; ++i_) {
#pragma unroll
            for (int it = 0; it < ITW; ++it) {
                const int f = it * SM_BLOCK + tid;
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
        for (int it = 0; it < IT1; ++it) {
            const int f = it * SM_BLOCK + tid;
            if (f < N1) lds[M::W1O + f] = v1[it];
        }
#pragma unroll
        for (int it = 0; it < ITT; ++it) {
            const int f = it * SM_BLOCK + tid;
            if (f < NTAB) { lds[M::AX + f] = vax[it]; lds[M::BY + f] = vby[it]; }
        }
    }
    const double bo = th[g.boff[L]];
    const double pc0 = pa.coef[e], pc1 = pa.coef[pa.coef_stride + e];
    const int ro_k = tid / SM_NTX, ro_r = tid % SM_NTX;
    const bool ro_on = tid < SM_NR && ro_k < rny && ro_r < rnx;
    const long ro_idx = e * rnr + ro_k * rnx + ro_r;
    const double pF = (pa.F && ro_on) ? pa.F[ro_idx] : 0.0;

    // ---- this wave's tile: waves 0..6 element tile wv, wave 7 the boundary/data tile blockIdx.x (if there is one) ----
    const bool is_el = wv < SM_TPE;
    const long n_dt = g.data_off >= 0 ? g.ntiles - g.data_off / 16 : 0;
    const bool is_dt = !is_el && (long)blockIdx.x < n_dt;
    const bool active = is_el || is_dt;                    // wave-uniform
    const int lp = wv * 16 + pt;                           // point inside the element (element tiles)
    long p = is_el ? e * SM_NQ + lp : g.data_off + (long)blockIdx.x * 16 + pt;
    const bool valid = active && (is_el ? lp < SM_NQ : p < g.N);
    p = valid ? p : 0;
    const double x0 = valid ? g.X[p] : 0.0, x1 = valid ? g.X[g.N + p] : 0.0;
    __syncthreads();
    FZ_STAMP(1);

    // =============================================================================================
    // phase F: forward of the wave's tile; s and the tangent pre-activations stay in registers
    // =============================================================================================
    double sv[L][MF_KS], zc[L][2][MF_KS];
    double gdat = 0.0;
    if (active) {
        double h[FZ_C][MF_KS];
#pragma unroll
        for (int s = 0; s < MF_KS; ++s) {
            const double w0 = lds[M::W1O + (0 * MF_KS + s) * 64 + lane], w1 = lds[M::W1O + (1 * MF_KS + s) * 64 + lane];
            const double z = lds[M::W1O + (3 * MF_KS + s) * 64 + lane] + x0 * w0 + x1 * w1;
            double a, a1, a2;
            act_fwd<HPV_ACT_TANH>(z, a, a1, a2);
            sv[0][s] = a; zc[0][0][s] = w0; zc[0][1][s] = w1;
            h[0][s] = a; h[1][s] = a1 * w0; h[2][s] = a1 * w1;
        }
#pragma unroll
        for (int i = 1; i < L; ++i) {
            double z0[MF_KS];
            fz_layer<true>(lds + M::WT + (i - 1) * MF_KS * 64, lds + M::WR + (i - 1) * MF_KS * 16, lds + M::BH + (i - 1) * MF_KS * 64, lane, h[0], z0);
            fz_layer<false>(lds + M::WT + (i - 1) * MF_KS * 64, lds + M::WR + (i - 1) * MF_KS * 16, nullptr, lane, h[1], zc[i][0]);
            fz_layer<false>(lds + M::WT + (i - 1) * MF_KS * 64, lds + M::WR + (i - 1) * MF_KS * 16, nullptr, lane, h[2], zc[i][1]);
#pragma unroll
            for (int s = 0; s < MF_KS; ++s) {
                double a, a1, a2;
                act_fwd<HPV_ACT_TANH>(z0[s], a, a1, a2);
                sv[i][s] = a;
                h[0][s] = a; h[1][s] = a1 * zc[i][0][s]; h[2][s] = a1 * zc[i][1][s];
            }
        }
        double o[FZ_C];
#pragma unroll
        for (int ch = 0; ch < FZ_C; ++ch) {
            double v = 0.0;
#pragma unroll
            for (int s = 0; s < MF_KS; ++s) v += h[ch][s] * lds[M::W1O + (2 * MF_KS + s) * 64 + lane];
            v = xrow_sum16(v);
            v = xrow_sum32(v);
            o[ch] = v;
        }
        o[0] += bo;
        if (is_el) {
            if (q == 0 && valid) {
                lds[M::CH + lp] = o[1];
                lds[M::CH + SM_NQ + lp] = o[2];
            }
        } else {
            const double dd = valid ? g.ud[p - g.data_off] - o[0] : 0.0;
            gdat = g.data_scale * dd;
            const double sq = row_sum16(dd * dd);
            if (lane == 0) g.data_part[blockIdx.x] = sq;
        }
    }
    FZ_STAMP(2);
    __syncthreads();
    FZ_STAMP(3);

    // =============================================================================================
    // phase P: projection of the element from LDS (two one-hot terms)
    // =============================================================================================
    {
        const double* G = lds + M::CH;
        for (int o = tid; o < 2 * SM_QY * SM_NTX; o += SM_BLOCK) {
            const int t = o / (SM_QY * SM_NTX), j = (o / SM_NTX) % SM_QY, r = o % SM_NTX;
            const double* ax = lds + M::AX + (t * SM_NTX + r) * SM_QX;
            const double* gr = G + t * SM_NQ + j * SM_QX;
            double acc = 0.0;
#pragma unroll
            for (int i = 0; i < SM_QX; ++i) acc = fma(ax[i], gr[i], acc);
            lds[M::T + o] = acc;
        }
        __syncthreads();
        if (tid < 2 * SM_NR) {
            const int t = tid / SM_NR, o = tid % SM_NR, kk = o / SM_NTX, r = o % SM_NTX;
            const double* by = lds + M::BY + (t * SM_NTY + kk) * SM_QY;
            const double* tt = lds + M::T + t * SM_QY * SM_NTX + r;
            double acc = 0.0;
#pragma unroll
            for (int j = 0; j < SM_QY; ++j) acc = fma(by[j], tt[j * SM_NTX], acc);
            lds[M::UP + tid] = (t == 0 ? pc0 : pc1) * acc;
        }
        __syncthreads();
        double sq = 0.0;
        if (tid < SM_NR) {
            const double u = (lds[M::UP + tid] + lds[M::UP + SM_NR + tid]) - pF;
            lds[M::U + tid] = u;
            if (ro_on) pa.R[ro_idx] = u;
            sq = u * u;
        }
        if (wv == 0) {
            sq = pj_wave_sum_dpp(sq);
            if (lane == 0) lds[M::RED] = sq;
        }
        __syncthreads();
        if (tid == 0) pa.loss_e[e] = lds[M::RED] / (double)rnr;
        const double sc = 2.0 / (double)rnr;
        for (int o = tid; o < 2 * SM_NTY * SM_QX; o += SM_BLOCK) {
            const int t = o / (SM_NTY * SM_QX), kk = (o / SM_QX) % SM_NTY, i = o % SM_QX;
            const double* ax = lds + M::AX + t * SM_NTX * SM_QX + i;
            const double* ur = lds + M::U + kk * SM_NTX;
            double acc = 0.0;
#pragma unroll
            for (int r = 0; r < SM_NTX; ++r) acc = fma(ax[r * SM_QX], ur[r], acc);
            lds[M::S + o] = acc * sc * (t == 0 ? pc0 : pc1);
        }
        __syncthreads();
        for (int o = tid; o < 2 * SM_NQ; o += SM_BLOCK) {
            const int t = o / SM_NQ, j = (o / SM_QX) % SM_QY, i = o % SM_QX;
            const double* by = lds + M::BY + t * SM_NTY * SM_QY + j;
            const double* sr = lds + M::S + t * SM_NTY * SM_QX + i;
            double acc = 0.0;
#pragma unroll
            for (int kk = 0; kk < SM_NTY; ++kk) acc = fma(by[kk * SM_QY], sr[kk * SM_QX], acc);
            lds[M::CH + o] = acc;
        }
        __syncthreads();
    }

    // =============================================================================================
    // phase R: reverse pass of the wave's tile; every gradient piece goes to the wave's LDS row as soon as it exists
    // =============================================================================================
    // (the rows alias the transpose region: a row is only written after the last transpose read of the tile)
    FZ_STAMP(4);
    double* TA = lds + M::TR + wv * M::TR_WAVE;
    double* TB = TA + MF_TRB * MF_LD;
    constexpr int LH = L > 1 ? L - 1 : 1;
    v4d dWacc[LH];
    double dS10[LH], dS01[LH], accC[LH];
    double dbv[L][MF_KS], dW1v[2][MF_KS], dWov[MF_KS], dbo = 0.0;
#pragma unroll
    for (int i = 0; i < LH; ++i) { dWacc[i] = v4d{0.0, 0.0, 0.0, 0.0}; dS10[i] = 0.0; dS01[i] = 0.0; accC[i] = 0.0; }
#pragma unroll
    for (int s = 0; s < MF_KS; ++s) {
        dWov[s] = 0.0; dW1v[0][s] = 0.0; dW1v[1][s] = 0.0;
#pragma unroll
        for (int i = 0; i < L; ++i) dbv[i][s] = 0.0;
    }
    if (active) {
        double gb[FZ_C];
        if (is_el) { gb[0] = 0.0; gb[1] = valid ? lds[M::CH + lp] : 0.0; gb[2] = valid ? lds[M::CH + SM_NQ + lp] : 0.0; }
        else { gb[0] = gdat; gb[1] = 0.0; gb[2] = 0.0; }
        double hbar[FZ_C][MF_KS], zbar[FZ_C][MF_KS];
#pragma unroll
        for (int s = 0; s < MF_KS; ++s) {
            const double a = sv[L - 1][s], a1 = 1.0 - a * a;
            const double wo = lds[M::W1O + (2 * MF_KS + s) * 64 + lane];
            dWov[s] = a * gb[0] + a1 * zc[L - 1][0][s] * gb[1] + a1 * zc[L - 1][1][s] * gb[2];
            hbar[0][s] = gb[0] * wo; hbar[1][s] = gb[1] * wo; hbar[2][s] = gb[2] * wo;
        }
        if (q == 0) dbo = gb[0];
#pragma unroll
        for (int i = L - 1; i >= 0; --i) {
#pragma unroll
            for (int s = 0; s < MF_KS; ++s) {
                const double a = sv[i][s];
                const double a1 = 1.0 - a * a, a2 = -2.0 * a * a1;
                zbar[1][s] = hbar[1][s] * a1;
                zbar[2][s] = hbar[2][s] * a1;
                const double zb = hbar[0][s] * a1 + a2 * (hbar[1][s] * zc[i][0][s] + hbar[2][s] * zc[i][1][s]);
                zbar[0][s] = zb;
                dbv[i][s] = zb;
            }
            if (i == 0) {
#pragma unroll
                for (int s = 0; s < MF_KS; ++s) {
                    dW1v[0][s] = x0 * zbar[0][s] + zbar[1][s];
                    dW1v[1][s] = x1 * zbar[0][s] + zbar[2][s];
                }
            } else {
                // h_{i-1}^T = W_i zbar^T first (needs only registers), then dW_i channel by channel through ONE transpose pair
                double hnext[FZ_C][MF_KS];
#pragma unroll
                for (int ch = 0; ch < FZ_C; ++ch) {
                    v4d acc = v4d{0.0, 0.0, 0.0, 0.0};
                    double h4 = 0.0;
                    const double* wrl = lds + M::WRB + (i - 1) * MF_KS * 16 + q * 4 + (lane & 3);
#pragma unroll
                    for (int s = 0; s < MF_KS; ++s) {
                        acc = __builtin_amdgcn_mfma_f64_16x16x4f64(lds[M::WN + ((i - 1) * MF_KS + s) * 64 + lane], zbar[ch][s], acc, 0, 0, 0);
                        h4 = __builtin_amdgcn_mfma_f64_4x4x4f64(wrl[s * 16], zbar[ch][s], h4, 0, 0, 0);
                    }
#pragma unroll
                    for (int s = 0; s < 4; ++s) hnext[ch][s] = acc[s];
                    hnext[ch][4] = h4;
                }
#pragma unroll
                for (int ch = 0; ch < FZ_C; ++ch) {
                    pj_wave_sync();
#pragma unroll
                    for (int s = 0; s < MF_KS; ++s) {
                        const double a = sv[i - 1][s], a1 = 1.0 - a * a;
                        TA[(4 * s + q) * MF_LD + pt] = ch == 0 ? a : a1 * zc[i - 1][ch - 1][s];
                        TB[(4 * s + q) * MF_LD + pt] = zbar[ch][s];
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
#pragma unroll
                for (int ch = 0; ch < FZ_C; ++ch)
#pragma unroll
                    for (int s = 0; s < MF_KS; ++s) hbar[ch][s] = hnext[ch][s];
            }
        }
    }

    // ---- epilogue: each wave's gradient row -> LDS -> one row per workgroup ----
    FZ_STAMP(5);
    __syncthreads();
    FZ_STAMP(6);
    double* WP = lds + M::TR + (long)wv * g.P;
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
        for (int i = 0; i < L; ++i) v[i] = dbv[i][s];
        v[L] = dW1v[0][s]; v[L + 1] = dW1v[1][s]; v[L + 2] = dWov[s];
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
    for (int idx = tid; idx < g.P; idx += SM_BLOCK) {
        double acc = 0.0;
#pragma unroll
        for (int w = 0; w < SM_WAVES; ++w) acc += W0[(long)w * g.P + idx];
        row[idx] = acc;
    }
#ifdef HPV_FZ_TIMING
    if (lane == 0 && pa.GBAR) {   // [block][wave][8]: staging, forward, wait, projection, reverse, wait, epilogue, total
        FZ_STAMP(7);
        double* o = pa.GBAR + ((long)blockIdx.x * SM_WAVES + wv) * 10;
        for (int i = 0; i < 7; ++i) o[i] = (double)(fz_t[i + 1] - fz_t[i]);
        const long long fz_end = wall_clock64();
        o[7] = (double)(fz_end - fz_wall) * 0.01;
        o[8] = (double)(fz_wall & 0xffffffffffll) * 0.01;
        o[9] = (double)(fz_end & 0xffffffffffll) * 0.01;
    }
#endif
}

// ------------------------------------------------------------------------------------------------
// host side
// ------------------------------------------------------------------------------------------------
template <int L, bool SPLIT, bool QT, bool GS, int QX_, int QY_, int NTX_, int NTY_, bool MULTI = false, int NT2 = 0, bool GEN = false>
static void launch_iter_fused(const MfmaArgs& a, int blocks, hipStream_t s) {
    using LDS = FzLds<L, QX_, QY_, NTX_, NTY_, MULTI, FZ_C + NT2, FzPlan<L, QX_, QY_, NT2>::TRG, FzPlan<L, QX_, QY_, NT2>::NPART>;
    static_assert(LDS::EPI == LDS::TR || FZ_WAVES * (2 * MF_H + MF_H + (L - 1) * (MF_H * MF_H + MF_H) + MF_H + 1) <= LDS::PP - LDS::PK, "the epilogue's rows fit the parking area");
    const size_t bytes = (size_t)LDS::total(a.P) * sizeof(double);
    static_assert(LDS::total(2 * MF_H + MF_H + (L - 1) * (MF_H * MF_H + MF_H) + MF_H + 1) * sizeof(double) <= 160 * 1024, "LDS");
    static bool attr_set = false;
    if (!attr_set) {
        (void)hipFuncSetAttribute((const void*)k_iter_fused<L, SPLIT, QT, GS, QX_, QY_, NTX_, NTY_, MULTI, NT2, GEN>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
        attr_set = true;
    }
    hipLaunchKernelGGL((k_iter_fused<L, SPLIT, QT, GS, QX_, QY_, NTX_, NTY_, MULTI, NT2, GEN>), dim3(blocks), dim3(FZ_BLOCK), bytes, s, a);
}
// The general forms (GEN: term weights, trainable epsilon; NT2 = 1: the mixed second tangent) are instantiated in a translation unit
// of their own (kernels_fused_gen.hip = this file with HPV_FZ_GEN_TU): plan 0 = SPLIT, 1 = whole tiles, 2 = quarter tiles.
// false: that (shape, depth, plan, channel set) is not instantiated -- or compiled out by the build guard (csrc/build.sh).
bool hpv_fused_launch_gen(const ProjDesc& pd, int L, int plan, int nt2, const MfmaArgs& a, int blocks, hipStream_t s);
const char* hpv_fused_gen_build_state();
#ifdef HPV_FZ_GEN_TU
template <int QX_, int QY_, int NTX_, int NTY_>
static bool launch_iter_fused_gen_shape(int L, int plan, int nt2, const MfmaArgs& a, int blocks, hipStream_t s) {
    constexpr int TPE = QX_ * QY_ / 16;
    constexpr bool HAS_QT = (TPE % 4) <= 1 && TPE >= 8;
    constexpr bool HAS_DQ = HAS_QT && TPE % 4 == 0;            // four channels: the data-quarter plan only
    // (20x20 points keep four tiles per wave in the stash -- ABASE = 256 - 4 x 2 x 5 L: room for the fourth channel's registers with two
    //  hidden layers, a77 of 176, not with three, a159 of 136: three layers run the tight plan (FzPlan), whole tiles on whole elements only)
#ifdef HPV_FZ_GEN_NO_TIGHT
    constexpr bool HAS_TIGHT = false;
#else
    constexpr bool HAS_TIGHT = true;
#endif
    const bool has_nt2 = QX_ != 20 || L == 2 || (HAS_TIGHT && plan <= 1);
    if (L != 2 && L != 3) return false;
#define FZ_GG(L_, SPLIT_, QT_, NT2_) launch_iter_fused<L_, SPLIT_, QT_, false, QX_, QY_, NTX_, NTY_, false, NT2_, true>(a, blocks, s)
    if (nt2 == 0) {
        if (plan == 0) { if (L == 2) FZ_GG(2, true, false, 0); else FZ_GG(3, true, false, 0); }
        else if (plan == 1) { if (L == 2) FZ_GG(2, false, false, 0); else FZ_GG(3, false, false, 0); }
        else if (plan == 2) {
#ifdef HPV_FZ_GEN_NO_QT
            return false;
#else
            if constexpr (HAS_QT) { if (L == 2) FZ_GG(2, false, true, 0); else FZ_GG(3, false, true, 0); } else return false;
#endif
        } else return false;
        return true;
    }
#ifdef HPV_FZ_GEN_NO_NT2
    return false;
#else
    if (has_nt2) {
        if (plan == 0) { if (L == 2) FZ_GG(2, true, false, 1); else if constexpr (QX_ != 20 || HAS_TIGHT) FZ_GG(3, true, false, 1); }
        else if (plan == 1) { if (L == 2) FZ_GG(2, false, false, 1); else if constexpr (QX_ != 20 || HAS_TIGHT) FZ_GG(3, false, false, 1); }
        else if (plan == 2) {
#ifdef HPV_FZ_GEN_NO_QT
            return false;
#else
            if constexpr (HAS_DQ) { if (L == 2) FZ_GG(2, false, true, 1); else FZ_GG(3, false, true, 1); } else return false;
#endif
        } else return false;
        return true;
    } else return false;
#endif
#undef FZ_GG
}
bool hpv_fused_launch_gen(const ProjDesc& pd, int L, int plan, int nt2, const MfmaArgs& a, int blocks, hipStream_t s) {
#ifdef HPV_FZ_GEN_TRIPPED
    return false;
#else
#define FZ_TRY(A_, B_, C_, D_) \
    if (pd.qx == A_ && pd.qy == B_ && pd.ntx <= C_ && pd.nty <= D_) return launch_iter_fused_gen_shape<A_, B_, C_, D_>(L, plan, nt2, a, blocks, s);
    FZ_SHAPES(FZ_TRY)
#undef FZ_TRY
    return false;
#endif
}
const char* hpv_fused_gen_build_state() {
#if defined(HPV_FZ_GEN_TRIPPED)
    return "absent";
#elif defined(HPV_FZ_GEN_NO_NT2) && defined(HPV_FZ_GEN_NO_QT)
    return "three-channel-whole-tiles-only";
#elif defined(HPV_FZ_GEN_NO_NT2)
    return "three-channel-only";
#elif defined(HPV_FZ_GEN_NO_QT) && defined(HPV_FZ_GEN_NO_TIGHT)
    return "no-quarter-tile,no-tight-plan";
#elif defined(HPV_FZ_GEN_NO_QT)
    return "no-quarter-tile";
#elif defined(HPV_FZ_GEN_NO_TIGHT)
    return "no-tight-plan";
#else
    return "ok";
#endif
}
#else    // ---- everything below: the main translation unit ----
// One element shape: plan 0 = SPLIT, 1 = whole tiles, 2 = quarter tiles (shapes with 1 mod 4 tiles).  false: not instantiated.
// (GS: the saved values travel through the activation store instead of AGPRs / LDS + recompute; HPV_FUSED_GSTASH=1 opts in; built
//  for the 20x20 / 10x10 shape only)
// plan 3 / 4 = plan 1 / 2 walking several elements per workgroup (MULTI: grids larger than the chip)
template <int QX_, int QY_, int NTX_, int NTY_>
static bool launch_iter_fused_shape(int L, int plan, bool gs, const MfmaArgs& a, int blocks, hipStream_t s) {
#ifdef HPV_EXPERIMENTS      // GS (measured slower: 67.8 against 60.6 us, profiles/r04_notes.md 6) is instantiated in libhpvpinn_testhooks.so only
    constexpr bool HAS_GS = QX_ == 20;
#else
    constexpr bool HAS_GS = false;
#endif
    constexpr bool HAS_QT = ((QX_ * QY_ / 16) % 4) <= 1 && QX_ * QY_ / 16 >= 8;     // (0 mod 4: the data-quarter plan)
#define FZ_GO(L_, SPLIT_, QT_, GS_) launch_iter_fused<L_, SPLIT_, QT_, GS_, QX_, QY_, NTX_, NTY_>(a, blocks, s)
    if (L != 2 && L != 3) return false;
    if (gs) {
        if constexpr (HAS_GS) {
            if (plan == 0) { if (L == 2) FZ_GO(2, true, false, true); else FZ_GO(3, true, false, true); }
            else if (plan == 1) { if (L == 2) FZ_GO(2, false, false, true); else FZ_GO(3, false, false, true); }
            else { if (L == 2) FZ_GO(2, false, true, true); else FZ_GO(3, false, true, true); }
            return true;
        } else return false;
    }
#define FZ_GOM(L_, QT_) launch_iter_fused<L_, false, QT_, false, QX_, QY_, NTX_, NTY_, true>(a, blocks, s)
    if (plan == 0) { if (L == 2) FZ_GO(2, true, false, false); else FZ_GO(3, true, false, false); }
    else if (plan == 1) { if (L == 2) FZ_GO(2, false, false, false); else FZ_GO(3, false, false, false); }
    // (MULTI with three hidden layers on 20x20 points is not instantiated: inside the element loop the compiler parks the reverse
    //  loop's accumulators in a104..a117 -- inside the stash, base a106, while later tiles' values are still there; fz_multi_built)
    else if (plan == 3) { if (L == 2) FZ_GOM(2, false); else if constexpr (QX_ != 20) FZ_GOM(3, false); else return false; }
    else if (plan == 4) {
        if constexpr (HAS_QT) { if (L == 2) FZ_GOM(2, true); else if constexpr (QX_ != 20) FZ_GOM(3, true); else return false; }
        else return false;
    }
    else {
        if constexpr (HAS_QT) { if (L == 2) FZ_GO(2, false, true, false); else FZ_GO(3, false, true, false); }
        else return false;
    }
#undef FZ_GOM
#undef FZ_GO
    return true;
}
static bool launch_iter_fused_any(const ProjDesc& pd, int L, int plan, bool gs, const MfmaArgs& a, int blocks, hipStream_t s) {
#define FZ_TRY(A_, B_, C_, D_) \
    if (pd.qx == A_ && pd.qy == B_ && pd.ntx <= C_ && pd.nty <= D_) return launch_iter_fused_shape<A_, B_, C_, D_>(L, plan, gs, a, blocks, s);
    FZ_SHAPES(FZ_TRY)
#undef FZ_TRY
    return false;
}
static bool fused_shape_ok(const ProjDesc& pd) {
#define FZ_TRY(A_, B_, C_, D_) if (pd.qx == A_ && pd.qy == B_ && pd.ntx >= 1 && pd.ntx <= C_ && pd.nty >= 1 && pd.nty <= D_) return true;
    FZ_SHAPES(FZ_TRY)
#undef FZ_TRY
    return false;
}

template <int L>
static void launch_iter_small(const MfmaArgs& a, int blocks, hipStream_t s) {
    const size_t bytes = (size_t)SmLds<L>::total(a.P) * sizeof(double);
    static bool attr_set = false;
    if (!attr_set) {
        (void)hipFuncSetAttribute((const void*)k_iter_small<L>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
        attr_set = true;
    }
    hipLaunchKernelGGL((k_iter_small<L>), dim3(blocks), dim3(SM_BLOCK), bytes, s, a);
}

// Whole training pass (forward, projection, reverse) of a shard of 20x20 / 10x10 elements in one launch.  Returns false
// when the shape / variational form / shard is not covered; the caller then runs the separate kernels.
// (libhpvpinn_testhooks.so: HPV_TRACE_DISPATCH=1 names the line at which the whole-iteration kernel declines a pass)
#ifdef HPV_EXPERIMENTS
#define fz_no(ID) (getenv("HPV_TRACE_DISPATCH") ? (fprintf(stderr, "hpv_mfma_iter_fused: declined at check %d (line %d)\n", ID, __LINE__), false) : false)
#else
#define fz_no(ID) false
#endif
bool hpv_mfma_iter_fused(HpvMfma* m, const double* theta, const double* X, double* GPART, int* rows, hipStream_t s,
                         const MfmaDataTerm* dt, const ProjArgs& pa, long n_elem, const MfmaPendingAdam* pre) {
    const ProjDesc& pd = pa.pd;
    const NetDesc& nd = m->nd;
    if (!m->iter_fused_ok) return fz_no(1);
    if (m->H != MF_H) return fz_no(2);      // written for 20-wide layers (other widths: kernels_wide.hip)
    if (!(nd.d == 2 && nd.nT1 == 2 && nd.nT2 <= 1 && nd.act == HPV_ACT_TANH) || m->L < 2 || m->L > 3) return fz_no(3);
    const bool small = pd.qx == SM_QX && pd.qy == SM_QY && pd.ntx >= 1 && pd.ntx <= SM_NTX && pd.nty >= 1 && pd.nty <= SM_NTY;
    if (!fused_shape_ok(pd) && !small) return fz_no(4);
    const int NQ = pd.qx * pd.qy, TPE = NQ / 16;              // points and 16-point tiles of an element
    const bool has_qt = TPE % 4 <= 1 && TPE >= 8, q20 = pd.qx == 20 && pd.qy == 20;
    const bool base_shape = q20 && pd.ntx == 10 && pd.nty == 10;           // BASELINE config 4 itself
#ifdef HPV_FZ_NO_EXTRA_SHAPES     // csrc/build.sh: the AGPR guard tripped in an instantiation of a shape other than 20x20 / 10x10
    if (!q20 && !small) return fz_no(5);
#endif
    if (pd.edge || pd.nact || pd.nterms < 1 || pd.nterms > 2) return fz_no(6);
    // one-hot (Poisson-2D var_form 1, the headline instantiations): term t integrates exactly channel 1 + t with weight 1, no epsilon.
    // Every other form of these channel sets (round 6): the general instantiations (k_iter_fused<.., NT2, GEN>)
    bool onehot = !pd.has_eps && pd.nterms == 2 && nd.nT2 == 0;
    for (int t = 0; t < 2 && onehot; ++t)
        for (int ch = 0; ch < HPV_MAXC; ++ch)
            if (pd.t[t].a0[ch] != (ch == 1 + t ? 1.0 : 0.0) || pd.t[t].a1[ch] != 0.0 || pd.t[t].eps_mult) onehot = false;
    const bool gen = !onehot;
    const int C = 3 + nd.nT2;
    if (gen) {
        if (small) return fz_no(7);                            // (10x10 points: kernels_tile.hip)
        for (int t = 0; t < pd.nterms; ++t) {
            if (pd.t[t].a0[0] != 0.0 || pd.t[t].a1[0] != 0.0) return fz_no(8);       // the value channel is not integrated here
            for (int ch = C; ch < HPV_MAXC; ++ch) if (pd.t[t].a0[ch] != 0.0 || pd.t[t].a1[ch] != 0.0) return fz_no(9);
            // two terms fill both LDS arrays: none is left for dG / d eps of a term whose WEIGHTS depend on epsilon
            if (pd.nterms == 2) for (int ch = 0; ch < HPV_MAXC; ++ch) if (pd.t[t].a1[ch] != 0.0) return fz_no(10);
        }
        if (pd.has_eps && !pa.eps_ptr) return fz_no(11);
        if (pre && pd.has_eps) return fz_no(12);                // the deferred-update prologue forms the network parameters only
    }
    if (n_elem <= 0) return fz_no(13);
    // the tight plan (four channels, three hidden layers, 20x20 points: whole tiles 7 + 7 + 6 + 6) gains 12 % on one round of elements, 5 % on
    // two (552 elements: 196.3 against 207.1 us) and nothing from six on (1 600: 516.0 against 513.8 -- the separate launches amortise
    // to 82 us per 256 elements there): profiles/r06_tight_plan.txt
    if (gen && nd.nT2 == 1 && q20 && m->L == 3 && n_elem > 5L * m->n_cus && !m->iter_fused_force) return fz_no(33);
#ifdef HPV_AGPR_GUARD_TRIPPED     // csrc/build.sh: the compiler's registers reached the hand-managed AGPR range of k_iter_fused
    if (!small) return fz_no(14);
#endif
    // shapes other than the headline one: one workgroup per element pays the launch-once phases (staging, projection, epilogue:
    // ~7 us) per element -- on grids of many small elements the separate launches amortise them better (scripts/elem_bench.py:
    // 1 024 elements of 12x12 points 106 against 99 us, of 16x16 points 158 against 163; 256 elements 30.7 / 48.9 against 49.3 / 60.5)
    // Grids larger than the chip (round 5; hpv_fused_grid_plan, hpv_mfma.h).  One workgroup per element runs ceil(n / CUs) rounds of
    // whole elements and pays staging, epilogue and dispatch per element; the MULTI instantiation (gridDim = CUs workgroups walk the
    // elements) pays them once per workgroup.  Both only on full rounds -- a static deal wastes the unfilled part of the last one.
    // HPV_FUSE=1: never the loop; HPV_FUSE=m: the loop on every grid larger than the chip (tests); HPV_FUSE=i: one workgroup per element.
#ifdef HPV_FZ_NO_MULTI            // csrc/build.sh: the AGPR guard tripped in an instantiation of the element loop
    constexpr bool multi_built = false;
#else
    constexpr bool multi_built = true;
#endif
    // (plan 3, round 6: the full rounds with one workgroup per element and, in a SECOND launch, the ragged tail's elements shared by
    //  2 - 8 workgroups each (SPLIT) -- needs the exchange machinery of the split mode)
    const bool tail_ok = m->xerr && m->xg && m->xiter && m->iter_split_ok && !pre;
    int gplan = small ? 1 : hpv_fused_grid_plan(pd.qx, m->L, n_elem, m->n_cus, !gen && multi_built && m->base.ACTS != nullptr, m->multi_off, m->multi_force && !gen, m->iter_fused_force, tail_ok);
    if (gplan == 0) return fz_no(15);
    const bool multi = gplan == 2;
    long n_tail = 0;
    int tsplit = 1;
    if (gplan == 3) {
        n_tail = n_elem % m->n_cus;
        while (tsplit < 8 && n_tail * tsplit * 2 <= m->n_cus) tsplit *= 2;
        const long data_tiles = m->ntiles - n_elem * TPE;
        if (tsplit < 2 || n_tail > m->xsync_elems || (size_t)n_tail * 2 * NQ * 2 > m->xg_words || data_tiles > n_tail * tsplit) {
            // (the tail cannot run in split mode: whole rounds as before)
            const long rounds = (n_elem + m->n_cus - 1) / m->n_cus;
            if (n_elem * 100 < rounds * m->n_cus * 80) return fz_no(16);
            n_tail = 0; gplan = 1;
        }
    }
    const long n_main = n_elem - n_tail;
    if (pre && (small || multi || nd.P > FZ_PRE_PER_THREAD * FZ_BLOCK)) return fz_no(17);       // the deferred-update prologue exists in the one-workgroup-per-element / SPLIT instantiations
    if (small) {
        // thousands of small elements: one workgroup per element pays staging / projection / epilogue per element, the separate
        // launches stream (scripts/grid_sweep.py: 1 024 elements 80.8 against 77.5 us, 4 096 elements 292 against 273)
        if (n_elem > hpv_elem_resident_max(2, SM_QX, m->n_cus) && !m->iter_fused_force) return fz_no(18);
        // batch layout [element points | pad to 16 | data points]; at most one boundary/data tile per workgroup
        const long npad = (n_elem * SM_NQ + 15) / 16 * 16;
        const bool has_data = dt && dt->n_data > 0;
        if (has_data ? dt->data_off != npad : (m->N != npad && m->N != n_elem * SM_NQ)) return fz_no(19);
        if (has_data && m->ntiles - npad / 16 > n_elem) return fz_no(20);
        if (n_elem > hpv_mfma_grad_rows(m) && n_elem > m->max_rows) return fz_no(21);
        MfmaArgs a = m->base;
        a.theta = theta; a.X = X; a.GPART = GPART;
        a.OUT = const_cast<double*>(pa.OUT);   // (only written by the -DHPV_FZ_TIMING build)
        a.data_off = -1;
        if (has_data) {
            a.data_off = dt->data_off; a.ud = dt->ud; a.gbar0 = dt->gbar0; a.data_part = dt->data_part;
            a.data_scale = dt->scale; a.data_write_gbar = dt->write_gbar;
        }
        a.proj_n_elem = n_elem;
        a.proj_split = 1;
        a.pa = pa;
        m->last_split = false;
        if (pd.ntx == SM_NTX && pd.nty == SM_NTY) snprintf(m->variant, sizeof m->variant, "k_iter_small<L=%d>", m->L);
        else snprintf(m->variant, sizeof m->variant, "k_iter_small<L=%d,10x10/%dx%d>", m->L, pd.ntx, pd.nty);
        if (m->L == 2) launch_iter_small<2>(a, (int)n_elem, s); else launch_iter_small<3>(a, (int)n_elem, s);
        if (rows) *rows = (int)n_elem;
        return true;
    }
    // small shards (the multi-GPU runs of config 4): an element is shared by 2 / 4 / 8 workgroups so that every CU works; the
    // partners meet at a barrier in device memory, which needs all of them resident: at most one workgroup per CU
    int split = 1;
    if (n_elem * 2 <= m->n_cus && !m->iter_fused_force) {
        if (!m->xerr || !m->xg || !m->xiter || !m->iter_split_ok) return fz_no(22);
        while (split < 8 && n_elem * split * 2 <= m->n_cus) split *= 2;
        if (n_elem * split > m->n_cus || n_elem > m->xsync_elems || (size_t)n_elem * 2 * NQ * 2 > m->xg_words) return fz_no(23);
    }
    const long blocks = multi ? (long)m->n_cus : n_main * split;
    const long rest = m->ntiles - n_elem * TPE;                  // pad + boundary/data tiles: at most one per workgroup
    if (rest < 0 || (n_tail == 0 && rest > blocks)) return fz_no(24);
    const long rows_all = blocks + n_tail * tsplit;
    if (rows_all > hpv_mfma_grad_rows(m) && rows_all > m->max_rows) return fz_no(25);
    MfmaArgs a = m->base;
    a.theta = theta; a.X = X; a.GPART = GPART;
    a.OUT = const_cast<double*>(pa.OUT);   // SPLIT: the partners' channel exchange; otherwise only written by the -DHPV_FZ_TIMING build
    a.data_off = -1;
    if (dt && dt->n_data > 0) {
        a.data_off = dt->data_off; a.ud = dt->ud; a.gbar0 = dt->gbar0; a.data_part = dt->data_part;
        a.data_scale = dt->scale; a.data_write_gbar = dt->write_gbar;
    } else if (rest > 0) {
        return fz_no(26);    // tiles behind the elements but no data term: not a layout this kernel knows
    }
    a.proj_n_elem = n_main;
    a.proj_split = split;
    a.elem0 = 0;
    a.data_tile0 = n_elem * TPE;
    // (a ragged tail: the boundary / data tiles ride in the tail's launch -- this one sees a batch that ends behind its elements)
    if (n_tail > 0) a.ntiles = n_main * TPE;
    if (pre) { a.pre_g = pre->g; a.pre_Ptot = pre->Ptot; a.pre_ad = pre->ad; }
    a.xerr = m->xerr;
    a.xdebug_skip = m->xdebug_skip;
    a.xg = m->xg;
    a.xiter = m->xiter;
    a.pa = pa;
    // GS is opt-in (HPV_FUSED_GSTASH=1): measured 67.8 against 60.6 us at config 4 -- the reverse phase does shrink (73.1 k -> 60.9 k
    // cycles) but the forward phase pays for its stores (42.9 k -> 49.4 k: the four waves' bursts share one 64 B/clk path), and with
    // 220 MB of extra traffic per iteration the chip clocks 10 % lower (1.94 against 2.16 GHz); profiles/r04_notes.md
#ifdef HPV_EXPERIMENTS
    const char* ge = getenv("HPV_FUSED_GSTASH");
    const bool gs = q20 && a.ACTS != nullptr && ge && ge[0] == '1';
#else
    constexpr bool gs = false;
#endif
    // the prologue is paid per WORKGROUP: worth it only where every workgroup is resident at once (one round); on larger grids the caller's
    // k_adam launch in front of the pass is cheaper (4 096 elements: +12.3 us against +4.5)
    if (pre && (gs || blocks > (long)m->n_cus)) return fz_no(27);
    // MULTI spills 45 doubles per lane into the activation store: [workgroup][wave][slot][64] -- it must hold that (it is sized for
    // the separate launches' slots of every tile: far larger on any grid that takes this branch)
    if (multi && (size_t)blocks * FZ_WAVES * 64 * 48 > hpv_mfma_activation_store_doubles(m)) return fz_no(28);
    int plan = 2;
    if (split > 1) plan = 0;
#ifdef HPV_AGPR_GUARD_TRIPPED_QT                    // csrc/build.sh: the compiler's registers reached the stash of the QT instantiation
    else if (!gs) plan = 1;
#endif
    else if (!has_qt || getenv("HPV_NO_QUARTER_TILE")) plan = 1;      // (A/B switch: whole tiles only, read per launch / capture)
    if (gen && plan == 2 && nd.nT2 == 1 && TPE % 4 != 0) plan = 1;    // four channels: the packed quarter has room for the data points only
    if (multi) plan += 2;                                              // plans 3 / 4: several elements per workgroup
    if (gen) {
        if (gs || multi) return fz_no(29);
        if (!hpv_fused_launch_gen(pd, m->L, plan, nd.nT2, a, (int)blocks, s)) {
            // (a quarter-tile instantiation the build guard compiled out: whole tiles)
            if (plan != 2 || !hpv_fused_launch_gen(pd, m->L, plan = 1, nd.nT2, a, (int)blocks, s)) return fz_no(30);
        }
    } else
    if (!launch_iter_fused_any(pd, m->L, plan, gs, a, (int)blocks, s)) return fz_no(31);
    if (n_tail > 0) {
        // the ragged tail: elements n_main .. n_elem - 1, tsplit workgroups each, gradient rows behind the first launch's
        MfmaArgs b = a;
        b.ntiles = m->ntiles;
        b.proj_n_elem = n_tail;
        b.proj_split = tsplit;
        b.elem0 = n_main;
        b.GPART = GPART + blocks * (long)nd.P;
        const bool ok = gen ? hpv_fused_launch_gen(pd, m->L, 0, nd.nT2, b, (int)(n_tail * tsplit), s)
                            : launch_iter_fused_any(pd, m->L, 0, false, b, (int)(n_tail * tsplit), s);
        if (!ok) return fz_no(32);      // (cannot happen for an instantiated shape: plan 0 exists wherever plans 1 / 2 do)
    }
    m->last_split = split > 1 || n_tail > 0;
    if (split > 1 || n_tail > 0) m->split_used = true;
    char shp[64] = "";
    if (!base_shape || gen) snprintf(shp, sizeof shp, ",%dx%d/%dx%d%s", pd.qx, pd.qy, pd.ntx, pd.nty, gen ? (nd.nT2 ? ",NT2=1,GEN" : ",GEN") : "");
    if (split > 1) snprintf(m->variant, sizeof m->variant, "k_iter_fused<L=%d,SPLIT=true,QT=false,GS=%s%s> split=%d", m->L, gs ? "true" : "false", shp, split);
    else snprintf(m->variant, sizeof m->variant, "k_iter_fused<L=%d,SPLIT=false,QT=%s,GS=%s%s>%s", m->L, (plan == 2 || plan == 4) ? "true" : "false",
                  gs ? "true" : "false", shp, multi ? " elements-per-workgroup>1" : "");
    if (n_tail > 0) {
        const size_t l = strlen(m->variant);
        snprintf(m->variant + l, sizeof m->variant - l, " + SPLIT=true split=%d on the last %ld elements", tsplit, n_tail);
    }
    m->pre_used = pre != nullptr;
    if (rows) *rows = (int)rows_all;
    return true;
}

const char* hpv_fused_build_state() {
#if defined(HPV_AGPR_GUARD_TRIPPED)
    return "absent";
#elif defined(HPV_AGPR_GUARD_TRIPPED_QT)
    return "no-quarter-tile";
#else
    return "ok";
#endif
}
bool hpv_mfma_sync_failed_possible(HpvMfma* m) { return m && m->last_split; }
void hpv_mfma_set_err_flag(HpvMfma* m, int* dev_flag) { if (m) m->xerr = dev_flag; }
bool hpv_mfma_split_used(HpvMfma* m) { return m && m->split_used; }
void hpv_mfma_set_split_ok(HpvMfma* m, bool on) {
    if (!m) return;
    const char* e = getenv("HPV_FUSE");
    m->iter_split_ok = on && !(e && e[0] == 's');
}

// the element loop (MULTI instantiations) is in this build (csrc/build.sh compiles them out when their AGPR guard trips); read by
// hpv_rule_advice so that its plan is the dispatch's
bool hpv_fused_loop_built() {
#ifdef HPV_FZ_NO_MULTI
    return false;
#else
    return true;
#endif
}
#endif   // HPV_FZ_GEN_TU
