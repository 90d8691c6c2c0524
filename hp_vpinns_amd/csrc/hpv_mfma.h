// MFMA (v_mfma_f64_16x16x4_f64) fast path for the BASELINE-shaped networks; see kernels_mfma.hip.
#pragma once
#include <string>

#include "hpv_internal.h"

struct HpvMfma;
// kernels_wide.hip: forward / reverse kernels for the hidden widths kernels_mfma.hip (H = 20) does not take; false = not instantiated
bool hpv_wide_pick(HpvMfma* m, int H, int key, int act, int L);
#define HPV_WIDE_WIDTHS(X) X(24) X(32) X(40) X(48) X(64)      // = WIDE_WIDTHS of csrc/build.sh, = WIDE_WIDTHS of hp_vpinns_amd/init.py
#define HPV_WIDE_DECL(Hw) bool hpv_wide_pick_##Hw##_d1(HpvMfma* m, int key, int act, int L); bool hpv_wide_pick_##Hw##_d2(HpvMfma* m, int key, int act, int L);
HPV_WIDE_WIDTHS(HPV_WIDE_DECL)
#undef HPV_WIDE_DECL

// Largest shard (elements) the element-resident whole-iteration kernel of a rule takes before the separate launches amortise the
// per-element launch-once phases better; ONE definition for the launch functions (kernels_fused.hip, kernels_tile.hip) and for
// hpv_rule_advice (the Python classes pad a smaller rule onto an instantiated one only while the kernel would take the shard).
// q = points per direction of the instantiated rule, dim = 1 | 2.  Measured: scripts/elem_bench.py, scripts/grid_sweep.py,
// scripts/rule1d_sweep.py (profiles/).
inline long hpv_elem_resident_max(int dim, int q, int n_cus) {
    if (dim == 1) return 65536;                     // k_iter_tile 80 / 60: the grid's own limit (rows of the gradient buffer)
    if (q == 10) return 4L * n_cus - 1;             // k_iter_small / k_iter_tile 10x10
    if (q == 12) return 3L * n_cus;                 // k_iter_fused, fewer than 16 tiles per element
    if (q == 16) return 6L * n_cus;
    return 1L << 40;                                // 20x20: every size
}
// How the whole-iteration kernel k_iter_fused takes a shard of n_elem elements of its 2-D rules (q = 12, 16, 20 points per direction):
// 0 not at all (the separate launches), 1 one workgroup per element, 2 the element loop (gridDim = CUs workgroups walk the elements).
// ONE definition for the dispatch (kernels_fused.hip) and for hpv_rule_advice (a smaller rule is padded onto an instantiated one only
// while the kernel would take the shard).  Numbers behind it: profiles/r05_multi_element.md.
bool hpv_fused_loop_built();      // kernels_fused.hip: the element loop (MULTI) survived the build guard
// 3 (round 6): the full rounds with one workgroup per element + the ragged tail (n_elem % n_cus elements, at most half a round) in a
// second launch in SPLIT mode, 2 - 8 workgroups per element -- 1 600 elements of the config-4 shape: 6 rounds + a 64-element tail
// instead of 7 rounds.  20x20-point elements from two full rounds on (see below).
inline int hpv_fused_grid_plan(int q, int L, long n_elem, int n_cus, bool loop_built, bool loop_off = false, bool loop_force = false, bool one_force = false,
                               bool tail_ok = true) {
    if (n_elem <= n_cus) return 1;
    const long rounds = (n_elem + n_cus - 1) / n_cus;
    const bool built = loop_built && !(q == 20 && L == 3);                   // (three hidden layers on 20x20 points: the loop does not fit the registers)
    const bool pays = rounds >= ((q == 16 && L == 3) ? 4 : 6) && n_elem * 100 >= rounds * n_cus * 95;
    if (built && !loop_off && (pays || loop_force)) return 2;
    if (one_force) return 1;
    if (q != 20 && n_elem > hpv_elem_resident_max(2, q, n_cus)) return 0;
    const long tail = n_elem % n_cus;
    // (measured, scripts/ragged_bench.py: 20x20 points -- 1 600 elements 365 against 385 (7 rounds) / 415 us (separate), 1 296 elements 303 /
    //  330 / 350; with ONE full round in front of the tail the separate launches win, 289 elements 90 against 85 us; 16x16 points: the
    //  separate launches win on every ragged grid, 552 elements 112 against 93 us)
    if (tail_ok && tail > 0 && tail * 2 <= n_cus && q == 20 && n_elem >= 2L * n_cus) return 3;
    return n_elem * 100 >= rounds * n_cus * 80 ? 1 : 0;
}

// Up to this many elements a 1-D rule smaller than 80 points is worth padding onto the 80 / 60 instantiation: while every element
// has a CU to itself the padded kernel runs 27-29 us per iteration whatever the rule (against 29-42 us on the separate launches);
// with two workgroups per CU it takes 51 us and loses to the rule as it is below 56 points (32-47 us), beyond that it scales with
// the padded work: 1 432 us against 191 us at 16 384 elements of 10 points (profiles/r05_rule1d_sweep.md)
inline long hpv_rule1d_pad_max(int q, int n_cus) { return (q >= 56 ? 2L : 1L) * n_cus; }

// Fault injection of the exchange-timeout tests (HPV_DEBUG_SPLIT_SKIP): read in hpv_api.hip, which is compiled once per library --
// the product's copy returns the constant 0 and does not contain the name (this file's object is shared by both libraries).
int hpv_test_hook_split_skip();

// Returns nullptr (and a reason) when the network shape is not covered by the fast path.
HpvMfma* hpv_mfma_create(const NetDesc& nd, long N, std::string* why, bool need_store = true);
void hpv_mfma_destroy(HpvMfma* m);
int hpv_mfma_grad_rows(HpvMfma* m);
double* hpv_mfma_activation_store(HpvMfma* m);
size_t hpv_mfma_activation_store_doubles(HpvMfma* m);
// Boundary/data term evaluated inside the forward kernel for the data tiles of a merged batch.
struct MfmaDataTerm {
    long data_off;        // first data point (multiple of 16)
    int n_data;
    const double* ud;
    double* gbar0;
    double* data_part;    // one partial per data tile
    double scale;         // -2 w / n_data
    int write_gbar;
};
void hpv_mfma_forward(HpvMfma* m, const double* theta, const double* X, double* OUT, int save_act, hipStream_t s,
                      const MfmaDataTerm* dt = nullptr);
void hpv_mfma_backward(HpvMfma* m, const double* theta, const double* X, const double* GBAR, double* GPART, int* rows,
                       hipStream_t s);
struct ProjArgs;
bool hpv_mfma_backward_fused(HpvMfma* m, const double* theta, const double* X, const double* GBAR, double* GPART, int* rows,
                             hipStream_t s, const ProjArgs& pa, long n_elem);
// Element-resident whole-iteration kernel (kernels_fused.hip): forward, projection and reverse pass of the shard in ONE
// launch, no activation store.  Returns false when not applicable (shape, variational form, small shard).
// pre (optional): a deferred TF1-Adam update the kernel is to compute WITH (MfmaArgs::pre_g) -- only k_iter_fused takes it; when the
// function returns false nothing has been launched and the caller applies the update itself (k_adam) before it goes on.
struct MfmaPendingAdam { AdamArgs ad; const double* g; int Ptot; };
bool hpv_mfma_iter_fused(HpvMfma* m, const double* theta, const double* X, double* GPART, int* rows, hipStream_t s,
                         const MfmaDataTerm* dt, const ProjArgs& pa, long n_elem, const MfmaPendingAdam* pre = nullptr);
// The same for small elements of any channel set (kernels_tile.hip): one tile per wave, the tile's saved state in registers.
// fin (optional): everything the finalize step needs; when the grid is ONE workgroup the kernel runs it itself and *fin_done
// is set (the caller then skips k_finalize).
struct MfmaFinalize {
    AdamArgs ad;          // ad.theta == nullptr: packed buffer only
    double* RB;
    double lossb_weight;
    int n_data, n_data_part, has_eps, ncopies;
    int n_iters = 1;              // iterations the caller wants back to back (a one-workgroup grid may run them in ONE persistent launch)
    int* iters_done = nullptr;    // out: iterations this launch performs (1, or n_iters on the persistent path)
};
bool hpv_mfma_iter_tile(HpvMfma* m, const double* theta, const double* X, double* GPART, int* rows, hipStream_t s,
                        const MfmaDataTerm* dt, const ProjArgs& pa, long n_elem, const MfmaFinalize* fin = nullptr,
                        bool* fin_done = nullptr);
// The same for ANY instantiated tensor-product element shape and 2-D channel set (kernels_elem.hip): several tiles per wave, s of
// every tile in registers, tangents recomputed in the reverse phase.
bool hpv_mfma_iter_elem(HpvMfma* m, const double* theta, const double* X, double* GPART, int* rows, hipStream_t s,
                        const MfmaDataTerm* dt, const ProjArgs& pa, long n_elem);
// Split whole-iteration kernels (SPLIT mode of k_iter_fused, k_iter_tall): the handle's sticky failure flag (device int, owned
// by the caller) that a timed-out exchange sets; without one those modes are not used.  hpv_mfma_split_used: such a launch
// happened since creation.
void hpv_mfma_set_err_flag(HpvMfma* m, int* dev_flag);
// false: never choose a launch structure in which workgroups of one element exchange partial results (SPLIT mode, k_iter_tall);
// true: allow them again unless HPV_FUSE=s forbids them
void hpv_mfma_set_split_ok(HpvMfma* m, bool on);
bool hpv_mfma_split_used(HpvMfma* m);
// Tall elements (80x80 points, 5x5 test functions: BASELINE config 5) split over `split` workgroups each (kernels_tall.hip);
// hpv_mfma_tall_split: workgroups per element (0 = not applicable), loss_e / deps_e then hold n_elem * split entries.
struct ProjDesc;
int hpv_mfma_tall_split(HpvMfma* m, const ProjDesc& pd, long n_elem);
bool hpv_mfma_iter_tall(HpvMfma* m, const double* theta, const double* X, double* GPART, int* rows, hipStream_t s,
                        const MfmaDataTerm* dt, const ProjArgs& pa, long n_elem);
// Names of the kernel instantiations (hpv_kernel_variant): which = 0 the whole-iteration kernel most recently launched, 1 the
// separate forward kernel, 2 the separate reverse kernel, 3 the reverse kernel with the projection fused in
const char* hpv_mfma_variant(HpvMfma* m, int which);
bool hpv_mfma_prefers_elem(HpvMfma* m);             // HPV_FUSE=e: the generic element-resident kernel before the hand-tuned ones
unsigned int* hpv_mfma_xiter(HpvMfma* m);         // launch counter of the tagged exchange (advanced by k_finalize behind a shared-element launch)
// "ok" | "no-quarter-tile" (AGPR guard tripped in the QT instantiation) | "absent" (guard tripped: kernel compiled out)
const char* hpv_fused_build_state();
// the general forms of k_iter_fused (kernels_fused_gen.hip): "ok" | "no-quarter-tile" | "three-channel-only" | "three-channel-whole-tiles-only" | "absent"
const char* hpv_fused_gen_build_state();
const char* hpv_tall_build_state();
bool hpv_mfma_sync_failed_possible(HpvMfma* m);   // the last whole-iteration launch ran in SPLIT mode
int hpv_mfma_max_rows(HpvMfma* m, long n_elem, long n_data_tiles = 0);
