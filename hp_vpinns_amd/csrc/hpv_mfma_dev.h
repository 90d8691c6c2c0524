// Device-side definitions shared by the MFMA kernels (kernels_mfma.hip: forward / reverse; kernels_fused.hip: the
// element-resident whole-iteration kernel): argument block, lane layout constants, activation helpers.
#pragma once
#include "hpv_mfma.h"
#include "hpv_math.h"
#include "hpv_project_wg.h"

typedef double v4d __attribute__((ext_vector_type(4)));

#define MF_H 20
#define MF_KS 5        // k-steps of 4 over the 20 inputs
#define MF_LD 17       // padded leading dimension of the LDS transpose tiles
#define MF_TRB 20      // rows of a transpose tile of k_bwd_mfma (20 neurons; every fragment row is in range)
#define MF_BLOCK 256
#define MF_WAVES (MF_BLOCK / 64)

struct MfmaArgs {
    const double* theta;
    const double* X;      // [d][N]
    double* OUT;          // [C][N]
    const double* GBAR;   // [C][N]
    double* ACTS;         // [tile][layer][slot][5][64]
    double* GPART;        // [block][P]
    long N;
    long ntiles;
    int save_act;
    int woff[HPV_MAX_LAYERS];
    int boff[HPV_MAX_LAYERS];
    int t1dim[2];
    int t2idx[2];
    double t2w[2];        // NT1 == 2 && NT2 == 1: weights of the mixed second tangent (NetDesc::t2w; {1, 0} = d2/dc0^2)
    int P;
    // boundary/data term folded into the forward kernel (tiles at and beyond data_off; -1: none)
    long data_off;
    const double* ud;     // [n_data] target values
    double* gbar0;        // adjoint row of the value channel (written when data_write_gbar)
    double* data_part;    // [data tiles] partial sums of (u_d - u)^2
    double data_scale;    // -2 w / n_data
    int data_write_gbar;
    // element-block mode of the reverse kernel (projection fused in): blocks own the elements [0, proj_n_elem)
    long proj_n_elem;
    int proj_split;       // workgroups per element in the reverse kernel's element-block mode (1, 2, 4 or 8)
    long elem0;           // k_iter_fused<SPLIT>: first element of this launch inside the shard (the ragged tail behind the full rounds; else 0)
    long data_tile0;      // k_iter_fused<SPLIT>: first boundary / data tile of the batch (= shard elements x tiles per element)
    ProjArgs pa;
    // split whole-iteration kernels: the handle's sticky failure flag, test knob, exchange buffers
    int* xerr;            // sticky failure flag of the handle (hpv_ctx::d_xerr): set when an exchange times out; see hpv_fused_dev.h
    int xdebug_skip;      // -DHPV_TEST_HOOKS builds only (libhpvpinn_testhooks.so; always 0 in the product): HPV_DEBUG_SPLIT_SKIP=k >= 1 --
                          // partner 1 of element 0 stays away from the exchange from the k-th launch on
    // tagged exchange (hpv_fused_dev.h, xg_*): the payload travels as 8-byte granules {32 bits of data | 32-bit launch tag}, so that
    // its arrival is its own notification -- no counter, no store-acknowledge / fetch-add / poll chain
    unsigned long long* xg;      // granule buffer
    unsigned int* xiter;         // launches so far: the tag of a launch is *xiter + 1; workgroup 0 advances it at its end
    // single-workgroup grids (the reference's own 1-element 1-D default, BASELINE config 1): the whole-iteration tile kernel
    // finishes the iteration itself -- packed buffer, TF1 Adam, loss history -- instead of a dependent k_finalize launch
    int elem_waves;       // k_iter_elem: wavefronts per workgroup (4 or 8; kernels_elem.hip)
    int tall_qt;          // k_iter_tall: the quarter-tile plan (every workgroup's tile count is 0 or 1 mod 4, see kernels_tall.hip)
    int persist_iters;    // k_iter_tile<.., PERSIST>: whole iterations per launch (one-workgroup grids)
    int fin_mode;         // 0: k_finalize follows; 1: packed buffer only; 2: packed buffer + Adam update
    AdamArgs fin_ad;
    double* fin_RB;       // [grad (P) | d eps | lossv | w*lossb | msq | pad]
    double fin_lossb_weight;
    int fin_n_data, fin_n_data_part, fin_has_eps, fin_ncopies;
    // k_iter_fused, the multi-GPU iteration in two launches (round 5): the PREVIOUS iteration's all-reduced packed buffer whose TF1-Adam
    // update (pre_ad) has not been applied yet -- the kernel computes with the updated parameters (formed in its prologue, every
    // workgroup for itself, nothing written), k_finalize behind it stores them.  nullptr: parameters as they are.
    const double* pre_g;
    int pre_Ptot;
    AdamArgs pre_ad;
};

// Fault injection for the exchange-timeout tests: compiled into libhpvpinn_testhooks.so only (csrc/build.sh); in the product
// library the condition is the constant false and the kernels carry no such branch.
#ifdef HPV_TEST_HOOKS
#define HPV_XDEBUG_SKIP(g, xtag, e, part) ((g).xdebug_skip && (xtag) >= (unsigned)(g).xdebug_skip && (e) == 0 && (part) == 1)
#else
#define HPV_XDEBUG_SKIP(g, xtag, e, part) false
#endif

struct HpvMfma {
    NetDesc nd;
    long N, ntiles;
    int L;
    int H = MF_H;      // uniform hidden width (20: kernels_mfma.hip and the whole-iteration kernels; other widths: kernels_wide.hip)
    bool store_s_only = false;   // kernels_wide.hip: the forward kernel stores s (and cos) only, the reverse kernel recomputes the tangents
    int ks = MF_KS;    // values per lane, channel and layer = H / 4 (the activation store is [tile][layer][slot][ks][64])
    int ns;            // saved slots per layer
    double* ACTS = nullptr;
    int fwd_blocks, bwd_blocks;
    MfmaArgs base;
    void (*fwd)(const MfmaArgs&, int, hipStream_t) = nullptr;
    void (*bwd)(const MfmaArgs&, int, hipStream_t) = nullptr;
    void (*bwd_fused)(const MfmaArgs&, int, hipStream_t) = nullptr; // projection + reverse, element-block mode
    int occ_fwd = 1, occ_bwd = 1;   // resident 256-thread blocks per CU
    int n_cus = 256;                // compute units of the device
    bool pre_used = false;          // the last k_iter_fused launch carried a deferred TF1-Adam update in its prologue
    bool multi_off = false, multi_force = false;   // HPV_FUSE=1 / m at creation: k_iter_fused's element loop never / on every grid larger than the chip
    int max_rows = 0;               // gradient rows the caller allocated (>= every launch mode's row count)
    // A/B switches read at creation (HPV_FUSE): default = the element-resident whole-iteration kernel where it applies,
    // 'b' = forward + (projection fused into the reverse kernel), 'n' = forward, projection, reverse as separate launches
    // 'i' = the whole-iteration kernel also for shards too small to fill the chip with one workgroup per element (tests)
    bool fuse_bwd = true, iter_fused_ok = true, iter_fused_force = false;
    // 's' keeps small shards on the forward + split reverse kernels (the whole-iteration kernel's split mode off)
    bool iter_split_ok = true;
    // 'e' tries the generic element-resident kernel (kernels_elem.hip) BEFORE the hand-tuned whole-iteration kernels (A/B runs, tests)
    bool prefer_elem = false;
    int* xerr = nullptr;                   // NOT owned: the handle's sticky failure flag (hpv_mfma_set_err_flag)
    int xdebug_skip = 0;
    unsigned long long* xg = nullptr;      // tagged-exchange granules (owned)
    unsigned int* xiter = nullptr;
    size_t xg_words = 0;
    long xsync_elems = 0;
    bool split_used = false;               // a split launch happened: hpv_step then reads the timeout flag back
    bool last_split = false;               // the most recent whole-iteration launch was a split one (hpv_pass_structure)
    char variant[160] = "";                // whole-iteration kernel instantiation most recently launched through this object (hpv_kernel_variant)
    char vfwd[96] = "", vbwd[96] = "", vbwd_fused[128] = "";   // the separate forward / reverse kernels of this object
};

// FAST (sin only): the caller has checked |z| <= HPV_SINCOS_MAX for the whole wave (act_wave_needs_safe below)
template <int ACT, bool FAST = false>
__device__ __forceinline__ void act_fwd(double z, double& a, double& a1, double& a2) {
    if constexpr (ACT == HPV_ACT_TANH) {
        a = hpv_tanh(z);
        a1 = 1.0 - a * a;
        a2 = -2.0 * a * a1;
    } else {
        if constexpr (FAST) hpv_sincos_fast(z, &a, &a1); else hpv_sincos(z, &a, &a1);
        a2 = -a;
    }
}
// wave-uniform: some lane holds a pre-activation the branch-free sincos does not cover (or a NaN)
template <int ACT, int NV>
__device__ __forceinline__ bool act_wave_needs_safe(const double (&z)[NV]) {
    if constexpr (ACT != HPV_ACT_SIN) return false;
    bool big = false;
#pragma unroll
    for (int s = 0; s < NV; ++s) big = big || !(fabs(z[s]) <= HPV_SINCOS_MAX);
    return __builtin_amdgcn_ballot_w64(big) != 0ull;
}
template <int ACT>
__device__ __forceinline__ void act_saved(double a, double a1s, double& a1, double& a2, double& a3) {
    if constexpr (ACT == HPV_ACT_TANH) {
        a1 = 1.0 - a * a;
        a2 = -2.0 * a * a1;
        a3 = -2.0 * a1 * (1.0 - 3.0 * a * a);
    } else {
        a1 = a1s;
        a2 = -a;
        a3 = -a1s;
    }
}

// z_c^2 as the second-tangent channel b sees it.  NT1 == 2 && NT2 == 1: the MIXED second tangent (NetDesc::t2w) rides on both first
// tangents, w0 z_c0^2 + w1 z_c1^2; every other channel set: channel b rides on first tangent b.
template <int NT1, int NT2>
struct T2Mix { static constexpr bool value = NT1 == 2 && NT2 == 1; };
template <int NT1, int NT2>
__device__ __forceinline__ double t2_square(const double (&w)[2], int b, double zc0, double zc1) {
    if constexpr (T2Mix<NT1, NT2>::value) return fma(w[0], zc0 * zc0, w[1] * (zc1 * zc1));
    else { const double z = b == 0 ? zc0 : zc1; return z * z; }
}

template <int ACT, int NT1, int NT2>
struct SlotCount {
    static constexpr int value = 1 + (ACT == HPV_ACT_SIN ? 1 : 0) + NT1 + NT2;
};


// sum over the 16 lanes of a DPP row (= the 16 points of a neuron group), result in every lane: quad butterflies,
// then row_half_mirror and row_mirror -- VALU only (a __shfl_xor tree is two ds_bpermute per step and double)
template <int CTRL>
__device__ __forceinline__ double dpp_move(double v) {
    // (bound_ctrl: a lane whose source lies outside its row reads 0 -- the row shifts rely on it -- and no `old` operand has to
    //  be zeroed first)
    const int lo = __builtin_amdgcn_mov_dpp(__double2loint(v), CTRL, 0xf, 0xf, true);
    const int hi = __builtin_amdgcn_mov_dpp(__double2hiint(v), CTRL, 0xf, 0xf, true);
    return __hiloint2double(hi, lo);
}
// Sums ACROSS the 16-lane rows (the MFMA operand layouts keep one k-group per row): v[l] + v[l ^ 16] and v[l] + v[l ^ 32] in
// every lane through gfx950's permlane swaps (v_permlane16_swap exchanges the odd rows of one register with the even rows of
// another, v_permlane32_swap the upper half of one with the lower half of the other: with both registers = v, the two results
// are {v_even, v_even} and {v_odd, v_odd} per row pair / half pair, and their sum is the exchange sum) -- two VALU instructions
// per 64-bit value and step instead of a ds_bpermute pair with its LDS round trip in the middle of a dependency chain.
// Values identical to `v += __shfl_xor(v, 16, 64)` / `(v, 32, 64)`.
__device__ __forceinline__ double xrow_sum16(double v) {
    const auto lo = __builtin_amdgcn_permlane16_swap(__double2loint(v), __double2loint(v), false, false);
    const auto hi = __builtin_amdgcn_permlane16_swap(__double2hiint(v), __double2hiint(v), false, false);
    return __hiloint2double(hi[0], lo[0]) + __hiloint2double(hi[1], lo[1]);
}
__device__ __forceinline__ double xrow_sum32(double v) {
    const auto lo = __builtin_amdgcn_permlane32_swap(__double2loint(v), __double2loint(v), false, false);
    const auto hi = __builtin_amdgcn_permlane32_swap(__double2hiint(v), __double2hiint(v), false, false);
    return __hiloint2double(hi[0], lo[0]) + __hiloint2double(hi[1], lo[1]);
}
// t[l] + t[l ^ 4] + t[l ^ 8] + t[l ^ 12] for the lanes of the first quad of every row (the others get another association of
// the same four values): two row rotations instead of two ds_bpermute pairs
__device__ __forceinline__ double quad4_sum(double t) {
    t += dpp_move<0x124>(t);   // row_ror:4
    t += dpp_move<0x128>(t);   // row_ror:8
    return t;
}
__device__ __forceinline__ double row_sum16(double v) {
    v += dpp_move<0xB1>(v);    // quad_perm [1,0,3,2]
    v += dpp_move<0x4E>(v);    // quad_perm [2,3,0,1]
    v += dpp_move<0x141>(v);   // row_half_mirror
    v += dpp_move<0x140>(v);   // row_mirror
    return v;
}


// kernels_elem.hip: the generic element-resident whole-iteration kernel, one translation unit per element shape
// (= ELEM_SHAPES of csrc/build.sh); false: that (H, channel set, depth) is not instantiated or its LDS does not fit
#define HPV_ELEM_SHAPES(X) X(16, 16, 8, 8) X(20, 20, 10, 10) X(12, 12, 6, 6)
#define HPV_ELEM_DECL(qx, qy, ntx, nty) bool hpv_elem_launch_##qx##qy##_##ntx##_##nty(int H, int key, int L, const MfmaArgs& a, int blocks, hipStream_t s);
HPV_ELEM_SHAPES(HPV_ELEM_DECL)
#undef HPV_ELEM_DECL
