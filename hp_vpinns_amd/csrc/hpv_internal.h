// Internal descriptors shared by the host orchestration (hpv_api.hip, hpv_exchange.hip, hpv_bench.hip) and the kernels.
// Everything here is plain-old-data passed to kernels by value (kernarg segment).
#pragma once
#include <hip/hip_runtime.h>

#include "../../include/hpvpinn.h"

#define HPV_MAXC 5    // value + up to 2 first tangents + up to 2 second tangents
#define HPV_MAXH 64   // widest layer the generic kernels handle
#define HPV_MAXT 2    // integrand terms per variational form

// Network + Taylor-channel layout.  Channel order everywhere: [value | d/dc for c in T1 | d2/dc2 for c in T2].
struct NetDesc {
    int d;                       // input dimension (1 or 2)
    int nl;                      // number of affine layers = n_layers - 1
    int width[HPV_MAX_LAYERS];   // layer widths, width[0] = d, width[nl] = 1
    int woff[HPV_MAX_LAYERS];    // offset of W_l (row-major [in][out]) in theta
    int boff[HPV_MAX_LAYERS];    // offset of b_l in theta
    int act;                     // HPV_ACT_*
    int nT1, nT2;                // number of first / second tangent channels
    int t1dim[2];                // input coordinate of each first tangent
    int t2idx[2];                // index INTO the T1 list of each second tangent
    // MIXED second tangent (round 6; nT1 == 2 && nT2 == 1 only): the one second-order channel is  t2w[0] d2/dc0^2 + t2w[1] d2/dc1^2
    // -- second-order Taylor channels propagate LINEARLY (z_cc -> s'' z_c^2 + s' z_cc -> W), so a weighted sum of them is itself one
    // channel: Poisson-2D var_form 0 integrates u_xx + u_yy (P2:91) and needs 4 channels instead of 5 in the forward, the tangent
    // recompute and the reverse pass.  {1, 0} = the plain second tangent of coordinate t2idx[0] = 0 (AdvDiff var_form 0, P3:163).
    double t2w[2];
    int C;                       // 1 + nT1 + nT2
    int nslot;                   // saved slots per hidden layer: A, A1, ZC[nT1], ZCC[nT2]
    long actoff[HPV_MAX_LAYERS]; // hidden layer l block starts at actoff[l] * N doubles in ACT
    int P;                       // number of network parameters (without epsilon)
};

// One integrand term: U[k][r] += mult * coef[e] * sum_q WTX[dx][r][i] WTY[dy][k][j] G[q],
// G = sum_ch (a0[ch] + eps * a1[ch]) OUT[ch][q],  mult = eps if eps_mult else 1.
struct TermDesc {
    int dx, dy;      // derivative order (0,1,2) of the test-function table per direction
    int eps_mult;    // the term carries the trainable epsilon as a factor (P3:171)
    double a0[HPV_MAXC];
    double a1[HPV_MAXC];
};

struct ProjDesc {
    int nterms;
    TermDesc t[HPV_MAXT];
    int qx, qy, ntx, nty;
    int C;
    int has_eps;     // theta carries a trailing trainable epsilon
    int edge;        // Poisson-1D var_form 3 boundary term (P1:90)
    // p-refinement of the 1-D driver (P1:66-67, 268-281: F_ext_total[e] may be shorter in some elements): number of ACTIVE test
    // functions per owned element (device pointer, nullptr = all ntx); the residual rows beyond it are zero and the element's
    // mean runs over the active ones.  Only the general projections (k_project, project_element_wg) honour it; the
    // specialised 2-D kernels are bypassed when it is set.
    const int* nact;
};

static inline size_t hpv_proj_lds_bytes(const ProjDesc& pd) {
    size_t nq = (size_t)pd.qx * pd.qy, nr = (size_t)pd.ntx * pd.nty;
    return (nq + (size_t)pd.qy * pd.ntx + nr + (size_t)pd.nterms * nr + (size_t)pd.nterms * pd.nty * pd.qx + 256) *
           sizeof(double);
}

struct AdamArgs {
    double *theta, *m, *v, *state;   // theta == nullptr: no update
    double lr, b1, b2, eps;
    // loss history: every training iteration appends {lossv, w*lossb, mean sq, epsilon} of ITS forward pass (= the loss
    // and the trainable coefficient after the previous update) at hist[4 * (*hist_idx)++]; entries beyond hist_cap are dropped
    double* hist;
    int* hist_idx;
    int hist_cap;
    // sticky failure flag of the handle (a SPLIT-mode element barrier timed out, kernels_fused.hip): while it is set -- or the
    // pad slot of the all-reduced buffer says some rank's is -- no update is applied (theta, m, v, beta powers, history untouched)
    int* xerr;
    // number of updates applied through this handle since it was created (thread 0 of the kernel that applies one adds 1);
    // the host reads it after a failed run to learn how many of the requested iterations took place (hpv_updates_applied)
    unsigned long long* n_upd;
};
#define HPV_HIST_CAP 4096

#ifdef __HIPCC__
// One TF1-Adam update of one parameter (P1:103-104; eps OUTSIDE the bias correction).  ONE definition for every place an update
// is applied (k_finalize, k_adam, k_p2p_exchange, the one-workgroup tail of k_iter_tile) or merely FORMED (the prologue of
// k_iter_fused computes with the updated parameter of the deferred update, k_finalize behind it stores it): all of them must
// produce the same bits, in every translation unit, under every compiler release -- so the operation sequence is pinned:
// floating-point contraction is OFF inside (each *, +, -, /, sqrt rounds on its own; no fma is formed here or there by the
// optimiser).  Advisor, round 5: under the default -ffp-contract the two compilations were free to fuse a*b + c differently.
__device__ __forceinline__ double hpv_adam_lr_t(double lr, double b1p, double b2p) {
#pragma clang fp contract(off)
    return lr * sqrt(1.0 - b2p) / (1.0 - b1p);
}
__device__ __forceinline__ void hpv_adam_one_lr(double lr_t, double b1, double b2, double eps, double g, double m0, double v0,
                                                double t0, double& m1, double& v1, double& t1) {
#pragma clang fp contract(off)
    const double gm = (1.0 - b1) * g, gv = (1.0 - b2) * g;
    m1 = b1 * m0 + gm;
    v1 = b2 * v0 + gv * g;
    t1 = t0 - lr_t * m1 / (sqrt(v1) + eps);
}
__device__ __forceinline__ void hpv_adam_one(double lr, double b1, double b2, double eps, double b1p, double b2p, double g, double m0,
                                             double v0, double t0, double& m1, double& v1, double& t1) {
    hpv_adam_one_lr(hpv_adam_lr_t(lr, b1p, b2p), b1, b2, eps, g, m0, v0, t0, m1, v1, t1);
}
#endif

// One-shot exchange of the packed buffer between the ranks of one node (multi-GPU path without a collective library
// in the iteration): every rank owns a mailbox [2 parities][world][n] doubles + arrival counters [2][world], mapped
// into every peer through hipIpc; see k_p2p_exchange (kernels_generic.hip).
#define HPV_P2P_MAX 8
struct P2PArgs {
    double* inbox[HPV_P2P_MAX];                 // inbox[r] = rank r's mailbox as mapped here (own rank: the local pointer)
    unsigned long long* flag[HPV_P2P_MAX];      // flag[r]  = rank r's arrival counters
    unsigned long long* counter;                // exchanges done so far (local)
    int* err;                                   // set to 1 when a peer did not arrive in time (sticky; the exchange then is a no-op)
    int world, rank, n;
    unsigned long long timeout_ticks;           // wait budget in s_memrealtime ticks (100 MHz)
};
#define HPV_P2P_POISON 0xFFFFFFFFFFFFFFFFULL     // arrival-counter value a rank publishes after giving up

// ---- kernel launchers (kernels_generic.hip) ----
void launch_mlp_fwd_generic(const NetDesc& nd, const double* theta, const double* X, double* ACT, double* OUT, long N,
                            int save_act, hipStream_t s);
void launch_mlp_bwd_generic(const NetDesc& nd, const double* theta, const double* X, const double* ACT,
                            const double* GBAR, double* GPART, int rows, long N, hipStream_t s);
int mlp_bwd_generic_rows(long N);
void launch_project(const ProjDesc& pd, const double* OUT, double* GBAR, double* R, const double* F, const double* coef,
                    long coef_stride, const double* wtx, const double* wty, const double* eps_ptr, double* loss_e,
                    double* deps_e, long N, long n_elem, int do_adjoint, const double* edge_u, const double* edge_dphi,
                    const double* edge_coef, double* edge_gbar, hipStream_t s);
void launch_data_loss(const double* U, const double* Ud, double* GBAR, double scale_grad, double* part, int n,
                      hipStream_t s);
void launch_finalize(const double* GPART_v, int rows_v, const double* GPART_b, int rows_b, const double* GPART_e,
                     int rows_e, const double* loss_e, long n_elem, const double* deps_e, const double* data_part,
                     int n_data_part, double lossb_weight, int n_data, int P, int has_eps, double* RB, int write_grad,
                     const AdamArgs* fused_adam, hipStream_t s, const int* xerr = nullptr, unsigned int* xiter_bump = nullptr,
                     int pending_adam = 0);
void launch_adam(const AdamArgs& ad, const double* RB, int P, int Ptot, hipStream_t s);
void launch_p2p_exchange(const P2PArgs& pp, double* RB, const AdamArgs* adam_or_null, int P, int Ptot, hipStream_t s);
int adam_state_doubles(int P);
void launch_debug_act(int act, const double* x, int n, double* a, double* a1, double* ref, hipStream_t s);
void launch_gll_rule(int Q, double* x, double* w, hipStream_t s);
void launch_test_tables(int ntest, int q, const double* xi, double* tab, hipStream_t s);
bool launch_project_tp(const ProjDesc& pd, const double* OUT, double* GBAR, double* R, const double* F, const double* coef,
                       long coef_stride, const double* wtx, const double* wty, const double* eps_ptr, double* loss_e,
                       double* deps_e, long N, long n_elem, int do_adjoint, hipStream_t s);
int pinn_residual_parts(int n);
void launch_pinn_residual(const double* OUT, const double* f, double* GBAR, double* part, long N, int n, long n_total,
                          int write_gbar, hipStream_t s);
bool launch_project_wg(const ProjDesc& pd, const double* OUT, double* GBAR, double* R, const double* F, const double* coef,
                       long coef_stride, const double* wtx, const double* wty, const double* eps_ptr, double* loss_e,
                       double* deps_e, long N, long n_elem, int do_adjoint, const double* edge_u, const double* edge_dphi,
                       const double* edge_coef, double* edge_gbar, hipStream_t s, double* upart = nullptr);
int project_row_split(const ProjDesc& pd, long n_elem, int backend_generic);   // workgroups per element of the row-split projection
