/* hpvpinn.h -- C-ABI of libhpvpinn.so, the MI355X (gfx950) hp-VPINN training path.
 *
 * The reference (ehsankharazmi/hp-VPINNs) has no FFI / plugin interface: its boundary is the
 * Python `class VPINN` of each driver script.  This header is the plain-C surface a binding of
 * that class needs -- opaque handle, plain pointers and sizes, status codes; no torch types.
 * Every entry point cites the reference lines whose work it replaces
 * (P1 = main/Poisson-1D/hp-VPINN-Poisson-1D.py, P2 = main/Poisson-2D/hp-VPINN-Poisson-2D.py,
 *  P3 = main/AdvDiff-Identification/hp-VPINN-AdvDiff-Identification.py).
 *
 * Conventions
 *   - all floating point data is IEEE double (the reference is tf.float64 throughout);
 *   - the caller owns every host buffer; the library copies on set_* and owns all device memory;
 *   - return 0 = OK, negative = error (message via hpv_last_error); nothing is thread-safe per
 *     handle, independent handles may live on different threads / processes (one per GPU);
 *   - parameter packing: theta = [W0, b0, W1, b1, ..., (epsilon)], W_l row-major [in, out]
 *     (row-vector convention H @ W + b of P1:128-138), epsilon only for HPV_PDE_ADVDIFF;
 *   - element index e = ex * ney + ey (P2:69-70 loop order; F_ext_total[ex, ey] at P2:414);
 *     quadrature index inside an element q = j * qx + i, x fastest (P2:362-365);
 *   - residual entries of one element are [k (y / t index)][r (x index)] (P2:94-96).
 */
#ifndef HPVPINN_H
#define HPVPINN_H

#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

#define HPV_MAX_LAYERS 16

enum { HPV_PDE_POISSON1D = 0, HPV_PDE_POISSON2D = 1, HPV_PDE_ADVDIFF = 2 };
enum { HPV_ACT_TANH = 0, HPV_ACT_SIN = 1 };
enum { HPV_BACKEND_AUTO = 0, HPV_BACKEND_GENERIC = 1, HPV_BACKEND_MFMA = 2 };
enum { HPV_SCHEME_VPINN = 0, HPV_SCHEME_PINN = 1 };

typedef struct hpv_ctx* hpv_handle;

typedef struct hpv_config {
    int pde;                      /* HPV_PDE_*                                                        */
    int var_form;                 /* P1:82-91 (1,2,3) / P2:93-115 (0,1,2) / P3:161-174 (0,1)           */
    int act;                      /* HPV_ACT_SIN for P1:134, HPV_ACT_TANH for P2:165, P3:226          */
    int n_layers;                 /* len(layers), e.g. 5 for [2,20,20,20,1]                           */
    int layers[HPV_MAX_LAYERS];
    double lr, beta1, beta2, eps; /* tf.train.AdamOptimizer(LR) defaults: 1e-3, 0.9, 0.999, 1e-8     */
    double lossb_weight;          /* P1:100 (1), P2:127 (10), P3:184 (10)                             */
    double V;                     /* advection speed, P3:43                                           */
    int device;                   /* HIP device ordinal                                               */
    int backend;                  /* HPV_BACKEND_*                                                    */
    int scheme;                   /* HPV_SCHEME_VPINN (default) or HPV_SCHEME_PINN (P2:126-129)       */
} hpv_config;

/* Construction = the graph-build part of VPINN.__init__ (P1:31-107, P2:28-136, P3:60-197). */
int hpv_create(hpv_handle* out, const hpv_config* cfg);
void hpv_destroy(hpv_handle h);
const char* hpv_last_error(hpv_handle h); /* h may be NULL: last create() error */

/* Run every kernel of this handle on an existing HIP stream (hipStream_t passed as void*), e.g.
 * torch's current stream so that a torch.distributed collective orders after the kernels. */
int hpv_set_stream(hpv_handle h, void* hip_stream);

/* 1-D reference quadrature rule per direction: nodes xi in [-1,1] and weights (P1:312-316,
 * P2:355-360, P3:395-400).  qy = 1 (and yi = wy = NULL) for the 1-D problem. */
int hpv_set_quadrature(hpv_handle h, const double* xi, const double* wx, int qx,
                       const double* yi, const double* wy, int qy);

/* Test-function tables phi, phi', phi'' at the reference nodes, each [ntest][q] row-major
 * (values of VPINN.Test_fcn / dTest_fcn, P1:157-183).  nty = 1, tables NULL in 1-D.
 * edge_dphi = phi'_k(-1), phi'_k(+1) as [ntx][2], needed only by Poisson-1D var_form 3 (P1:79,90). */
int hpv_set_tables(hpv_handle h, const double* phix, const double* dphix, const double* d2phix, int ntx,
                   const double* phiy, const double* dphiy, const double* d2phiy, int nty,
                   const double* edge_dphi);

/* Element grids (P1:264-273, P2:370-376, P3:404-410) and the slice [e_begin, e_end) of flattened
 * elements this handle (this GPU) owns -- the data-parallel shard.  The affine map of the
 * reference nodes into each element (P1:69, P2:75-76, P3:120-121) is evaluated here. */
int hpv_set_elements(hpv_handle h, const double* gridx, int nex, const double* gridy, int ney,
                     int e_begin, int e_end);

/* Right-hand sides F_ext_total for ALL nex*ney elements, C-order [nex][ney][nty][ntx]
 * (P1:294, P2:414); pass NULL for a zero right-hand side (P3:180). */
int hpv_set_rhs(hpv_handle h, const double* F, size_t n);

/* p-refinement of the 1-D driver: element e projects onto its first n_active[e] <= ntest test functions only and its loss
 * is the mean over those (P1:66-67: Ntest_element = len(F_ext_total[e]); the driver builds F_ext_total from the per-element
 * list N_testfcn_total, P1:268-281).  n = nex (all elements of the grid); rows of F beyond n_active[e] are ignored and the
 * residuals returned for them are zero.  NULL restores "all ntest in every element".  1-D only: the 2-D and AdvDiff drivers
 * reshape F_ext_total into a dense array (P2:414, P3:411), which forbids ragged counts in the reference too. */
int hpv_set_active_tests(hpv_handle h, const int* n_active, int n);

/* Boundary / data points of lossb (P1:98, P2:122, P3:184): X is [n][dim] row-major, u is [n].
 * The term is weighted by cfg.lossb_weight.  n = 0 disables it (ranks other than 0). */
int hpv_set_data(hpv_handle h, const double* X, const double* u, int n);

/* Collocation points of the strong-form PINN branch (`scheme == 'PINNs'`, P2:124-129,
 * 187-194): lossp = mean((u_xx + u_yy - f)^2) over X_f [n][2] replaces the variational term.  Poisson-2D only. */
int hpv_set_collocation(hpv_handle h, const double* X, const double* f, int n);
/* One rank's shard of the collocation set (multi-GPU PINN branch): n points here, n_total over all ranks -- lossp is the
 * mean over all n_total points (P2:124), the partial sums are made global by the iteration's all-reduce. */
int hpv_set_collocation_shard(hpv_handle h, const double* X, const double* f, int n, long n_total);

size_t hpv_num_params(hpv_handle h);
int hpv_set_params(hpv_handle h, const double* theta, size_t n); /* also resets Adam state (P1:107) */
int hpv_get_params(hpv_handle h, double* theta, size_t n);

/* loss3 = {loss, lossb (as the reference reports it), lossv} at the current parameters, and
 * d loss / d theta (grad may be NULL).  Replaces the forward/backward graph of P1:64-104. */
int hpv_loss_and_grad(hpv_handle h, double* loss3, double* grad_or_null);

/* n_iters Adam iterations = n_iters x `sess.run(train_op_Adam)` (P1:208, P2:242, P3:309), all
 * device-resident; loss3_after (may be NULL) is the loss evaluated AFTER the last update, which
 * is what the reference records (P1:211, P2:243, P3:315). */
int hpv_step(hpv_handle h, int n_iters, double* loss3_after);

/* n_iters Adam iterations with the loss after EVERY update recorded -- what VPINN.train of the 2-D script does with a
 * second forward pass per iteration (P2:243-244).  The loss after update k is the value the forward pass of
 * iteration k+1 computes anyway; every training iteration appends it to a device-side history, so only the loss after
 * the last update costs a forward pass.  loss3_hist is [n_iters][3] = {loss, lossb, lossv} after update 1..n_iters;
 * eps_hist (may be NULL) is [n_iters], the trainable diffusion coefficient after each update (P3:318, 0 for the Poisson
 * problems).  hpv_history_reset / hpv_history_read expose the history to callers that drive the iterations themselves
 * (multi-GPU: entry i = loss / epsilon seen by the i-th forward pass since the reset, global after the all-reduce). */
int hpv_step_record(hpv_handle h, int n_iters, double* loss3_hist, double* eps_hist);
int hpv_history_reset(hpv_handle h);
int hpv_history_read(hpv_handle h, int n, double* loss3_hist, double* eps_hist);

/* Pieces of hpv_step for the multi-GPU path (one process per GPU): enqueue forward + backward
 * into the packed device buffer [grad (P) | lossv | lossb | pad] (all partial sums of this
 * shard), let the caller all-reduce that buffer over RCCL, then apply the TF1 Adam update.
 * hpv_reduce_buffer returns the device pointer and length (in doubles) of that buffer. */
int hpv_forward_backward(hpv_handle h);
int hpv_reduce_buffer(hpv_handle h, void** dev_ptr, size_t* n_doubles);
int hpv_apply_adam(hpv_handle h);
/* Multi-GPU default: the all-reduce issued by the library itself.  One process per GPU; the handle owns an RCCL
 * communicator and every training iteration of hpv_step / hpv_step_record is forward+backward -> ONE
 * ncclAllReduce(sum) of the packed buffer [grad (P) | d eps | lossv | w*lossb | msq | pad] over xGMI -> TF1 Adam, all
 * enqueued on the handle's stream and captured into its iteration graphs.  After hpv_rccl_connect,
 * hpv_step / hpv_step_record / hpv_loss_and_grad are collective calls: every rank issues the same sequence.
 *   hpv_rccl_available: local check that every rank runs (and agrees on) BEFORE anyone enters the collective calls below;
 *   hpv_rccl_unique_id: rank 0 creates the 128-byte ncclUniqueId; the caller distributes it to all ranks;
 *   hpv_rccl_connect  : ncclCommInitRank (collective);
 *   hpv_rccl_selftest : known-answer all-reduce (collective); out[i] must equal W(W+1)/2 + W*1e-3*i;
 *   hpv_exchange_in_use: 0 none (single GPU or caller-driven pieces above), 1 RCCL, 2 peer-mapped mailboxes (below). */
int hpv_rccl_available(void);   /* 1 when librccl and its entry points can be loaded in this process (local, not collective) */
int hpv_rccl_unique_id(hpv_handle h, void* id128);
int hpv_rccl_connect(hpv_handle h, int world, int rank, const void* id128);
int hpv_rccl_selftest(hpv_handle h, double* out, size_t n);
int hpv_rccl_disconnect(hpv_handle h);
/* The caller stopped waiting for a hpv_rccl_connect / hpv_rccl_selftest that blocks on a helper thread (wall-clock bound of the
 * communicator set-up).  The ONLY entry point that may be called while another thread is inside the library on the same
 * handle: the abandoned call returns -6 and never touches the handle again (a communicator that comes up late is destroyed).
 * After an abandoned connect the handle stays usable, unconnected; after an abandoned self-test its stream may sit behind a
 * collective that never completes -- do not use that handle again (and do not destroy it while the call is still inside). */
int hpv_rccl_abandon(hpv_handle h);
int hpv_exchange_in_use(hpv_handle h);
/* Evidence for a reader of the benchmark line (the ONE collective of the path, lossv = sum_e loss_e of P2:120):
 *   hpv_rccl_info          : world size and rank as the COMMUNICATOR reports them (ncclCommCount / ncclCommUserRank);
 *                            0 / -1 when no communicator is connected.  Local call.
 *   hpv_rccl_time_allreduce: the all-reduce of the packed buffer ALONE -- `reps` eager calls on a scratch buffer of the same
 *                            size, one hipEvent pair around them, microseconds per call on this rank.  Collective. */
int hpv_rccl_info(hpv_handle h, int* world, int* rank);
int hpv_rccl_time_allreduce(hpv_handle h, int reps, double* avg_us);
/* Opt-in alternative (HPV_EXCHANGE=p2p in the Python classes): in-library exchange over peer-mapped mailboxes instead of
 * a collective-library call every iteration: each rank owns a mailbox that every peer maps through hipIpc; one kernel per
 * iteration writes the buffer into all mailboxes over xGMI, waits for the peers' contributions (bounded), sums them in
 * rank order -- bitwise identical on all ranks -- and applies the Adam update.  After hpv_p2p_connect, hpv_step /
 * hpv_step_record / hpv_loss_and_grad are collective calls: every rank must issue the same sequence.
 *   hpv_p2p_export  : allocate the mailbox, return its two 64-byte IPC handles (handles128);
 *   hpv_p2p_connect : handles = [world][128], the exports of all ranks in rank order (all-gathered by the caller);
 *   hpv_p2p_selftest: known-answer exchange (collective); out[i] must equal W(W+1)/2 + W*1e-3*i, *timed_out 0. */
int hpv_p2p_export(hpv_handle h, int world, int rank, void* handles128);
int hpv_p2p_connect(hpv_handle h, const void* handles);
int hpv_p2p_selftest(hpv_handle h, double* out, size_t n, int* timed_out);
int hpv_p2p_disconnect(hpv_handle h);
int hpv_eval_loss(hpv_handle h);            /* forward only -> same packed buffer slots [P], [P+1] */
int hpv_read_loss(hpv_handle h, double* loss3); /* sync + copy {loss, lossb, lossv} from the buffer   */
int hpv_sync(hpv_handle h);

/* u at arbitrary points: VPINN.predict (P1:197-199, P2:255-257).  X is [n][dim] row-major. */
int hpv_predict(hpv_handle h, const double* X, int n, double* u_out);

/* Checkpoint / resume (the reference never saves weights): the packed state
 * [theta | Adam m | Adam v | beta1^t | beta2^t], 3*num_params+2 doubles; a resumed run continues bit-exactly. */
int hpv_get_state(hpv_handle h, double* buf, size_t n);
int hpv_set_state(hpv_handle h, const double* buf, size_t n);

/* Driver-side RHS assembly on the device: F_ext[e][k][r] = J_e sum_q w phi_r phi_k f(x_q)
 * (P1:277-291, P2:386-411) for the owned elements from f at the handle's quadrature points
 * (element-major, q = j*qx+i, n = n_owned*qx*qy); F_out is [n_owned][nty][ntx]. */
int hpv_assemble_rhs(hpv_handle h, const double* f_quad, size_t n, double* F_out, size_t n_out);

/* Driver-side table generation on the device.  hpv_gll_rule: the Q-point Gauss-Lobatto-
 * Legendre nodes and weights the drivers take from GaussLobattoJacobiWeights(Q, 0, 0) (Q:47-61; P1:312, P2:355,
 * P3:395), Newton on the three-term recurrence.  hpv_test_tables: phi_n = P_{n+1} - P_{n-1}, phi'_n, phi''_n for
 * n = 1..ntest at the nodes xi (VPINN.Test_fcn / dTest_fcn, P1:157-183) as [3][ntest][q], the layout
 * hpv_set_tables takes per direction. */
int hpv_gll_rule(hpv_handle h, int q, double* xi, double* w);
int hpv_test_tables(hpv_handle h, int ntest, const double* xi, int q, double* tab);

/* Introspection for tests / benchmarks. */
int hpv_get_residuals(hpv_handle h, double* R, size_t n);   /* R of the owned elements [ne][nty][ntx] */
int hpv_backend_in_use(hpv_handle h);                       /* HPV_BACKEND_GENERIC or HPV_BACKEND_MFMA */
/* Launch structure of the most recent reverse-mode pass over the quadrature batch (tests assert that the kernel they mean to
 * exercise is the one that ran): 0 separate forward / projection / reverse launches, 1 forward + projection-fused reverse,
 * 2 element-resident whole-iteration kernel (Poisson-2D var_form 1 on 20x20-, 16x16-, 12x12- or 10x10-point elements with up to
 * 10x10 / 8x8 / 6x6 / 5x5 test functions), 3 the same in SPLIT mode (small shards),
 * 4 whole-iteration tile kernel (small elements of the other channel sets), 5 whole-iteration kernel for few tall elements
 * (80x80 points: many workgroups per element exchange partial residual sums), 6 the generic element-resident whole-iteration
 * kernel (any instantiated tensor-product element shape, e.g. 16x16 / 8x8; several tiles per wave); -1 before the first such pass. */
int hpv_pass_structure(hpv_handle h);
/* The kernel INSTANTIATION(s) of that pass by name, e.g. "k_iter_fused<L=3,SPLIT=false,QT=true>" (quarter-tile plan) vs
 * "k_iter_fused<L=3,SPLIT=false,QT=false>" (7/6/6/6 whole tiles: HPV_NO_QUARTER_TILE=1, or the build's AGPR guard tripped),
 * "k_iter_tall<..,QT=true> split=32", "k_iter_tile<..>", "k_fwd_mfma<..> + k_project_tp<20x20/10x10> + k_bwd_mfma<..>",
 * "k_mlp_fwd_generic + k_project<..> + k_mlp_bwd_generic" -- so that a green test run says which code it exercised. */
int hpv_kernel_variant(hpv_handle h, char* buf, size_t n);
/* How this library was built: "k_iter_fused=ok|no-quarter-tile|absent;k_iter_tall=ok|no-quarter-tile|absent;test_hooks=0|1"
 * (csrc/build.sh compiles a whole-iteration kernel out when the compiler's registers reach its hand-managed AGPR range;
 * test_hooks=1 only in libhpvpinn_testhooks.so, the build that carries the fault-injection knobs of the tests). */
const char* hpv_build_info(void);
/* 1 when hpv_step / hpv_step_record replay captured iteration hipGraphs, 0 when they launch eagerly (HPV_NO_GRAPH=1, a foreign
 * stream, or a collective that refused stream capture -- hpv_step then drops to eager launches instead of failing). */
int hpv_graphs_in_use(hpv_handle h);
/* Number of parameter updates applied through this handle since hpv_create (synchronises).  After hpv_step returned -7 (an
 * in-kernel exchange between the workgroups of one element timed out) the difference to the value before the call says how many
 * of the requested iterations took place. */
int hpv_updates_applied(hpv_handle h, long long* n);
/* on = 0: from now on only launch structures without an in-kernel exchange (no SPLIT mode, no k_iter_tall: what HPV_FUSE=s
 * selects at creation); on = 1 allows them again.  Drops captured iteration graphs.  hpv_step / hpv_step_record call it
 * themselves when such an exchange timed out (-7 inside), and finish the requested iterations on the remaining structures:
 * the run continues instead of ending on a shared GPU (connected ranks all do so in the same call: the failure travels with
 * the all-reduced buffer).  They return -7 instead when HPV_EXCHANGE_FALLBACK=0 is set.
 * hpv_shared_element_kernels: the current setting (0 after such a fallback). */
int hpv_shared_element_kernels(hpv_handle h);
int hpv_set_shared_element_kernels(hpv_handle h, int on);
/* N_quad is a free hyper-parameter (P1:237, P2:282, P3:47); the element-resident kernels are instantiated for a few rules and
 * take a SMALLER rule padded with zero-weight points (exact: the tables are w * phi).  Whether that pays depends on the shard and
 * on the device, so the library decides: for a shard of n_elem_shard elements with q points per direction and ntx x nty test
 * functions (dim = 1: nty ignored) on `device` (its CU count; 256 when no device can be queried), *q_dev = the rule to hand to
 * hpv_set_quadrature (== q: leave the rule alone) and *nt_dev = the test-function count to hand to hpv_set_tables in 1-D (the
 * 80-point kernel takes 60 functions and per-element counts; == ntx otherwise).  exact_counts != 0: only an instantiation with
 * exactly these counts (forms whose kernel has no run-time counts).  n_hidden = the network's hidden layers (0: unknown -> three,
 * the reference's depth): the plan is evaluated exactly as the dispatch will evaluate it, and a rule is padded only where ONE
 * WORKGROUP PER ELEMENT will run -- not where the element loop or the separate launches take the grid (on many rounds the padded
 * points cost more than the structure saves).  No handle needed; the same limits gate the launch functions. */
int hpv_rule_advice(int device, int dim, int q, int ntx, int nty, long n_elem_shard, int exact_counts, int n_hidden, int* q_dev, int* nt_dev);
/* How the whole-iteration kernel takes a shard of n_elem_shard elements of one of its 2-D rules (q = 12, 16, 20 points per
 * direction; N_el_x, N_el_y are free: P2:282-283, P3:44-45) under a network of n_hidden hidden layers on `device`: 0 = not at all
 * (the separate launches), 1 = one workgroup per element, 2 = the element loop (CUs workgroups walk the elements), 3 = the full
 * rounds with one workgroup per element + the ragged tail (n mod CUs elements) shared by 2 - 8 workgroups each in a second launch.
 * The dispatch's own function (csrc/hpv_mfma.h, hpv_fused_grid_plan) with this build's instantiations; < 0: bad arguments. */
int hpv_grid_plan(int device, int q, int n_hidden, long n_elem_shard);
/* The network value and its input-derivative channels at the owned quadrature points, [C][n_owned*qx*qy]
 * (channel order: u, then d/dx, d/dy (d/dt), then the second derivatives the variational form integrates) --
 * what net_u / net_du / net_dxu / net_dyu / net_dtu return (P1:140-148, P2:171-185, P3:232-245).  One forward launch. */
int hpv_eval_channels(hpv_handle h, double* out, size_t n);
/* Test hook: the device activation s(x), s'(x) (cfg.act) and the ROCm math library's s(x) for the same
 * inputs, so tests can bound the error of the hand-written fp64 tanh (replaces tf.tanh, P2:165). */
int hpv_debug_activation(hpv_handle h, const double* x, int n, double* a, double* a1, double* ref);
/* Average device time (ms) per launch of kernel class `which` since the last reset, measured with
 * hipEvents on the handle's stream when timing is enabled; which: 0 mlp_fwd, 1 project, 2 mlp_bwd. */
int hpv_enable_timing(hpv_handle h, int on);
int hpv_kernel_time_ms(hpv_handle h, int which, double* avg_ms, long* launches);
/* Average duration (ms) of the whole-iteration kernel -- forward, projection and reverse pass of the shard in ONE launch (the work of
 * P2:98-132 on the quadrature batch) -- over `reps` back-to-back launches bracketed by ONE hipEvent pair on the handle's stream: the
 * per-launch timers above carry ~3.5 us of event overhead per pair, this one amortises it (it still contains the gaps between the
 * launches).  Parameters, moments and the packed buffer of the last pass are left as they are (no finalize, no update).  -4 when the
 * handle's iteration is not one such launch (separate launches, an in-kernel exchange between workgroups, strong-form branch). */
int hpv_time_iteration_kernel(hpv_handle h, int reps, double* avg_ms);

/* Stand-alone launch of the per-element projection (residual + adjoint) kernel on synthetic
 * integrand channels already resident on the device -- the HBM-roofline measurement of
 * SURVEY.md section 8(d).  n_elem elements of the handle's (qx,qy,ntx,nty) shape; returns the
 * average kernel time over `reps` launches. */
int hpv_bench_projection(hpv_handle h, long n_elem, int reps, double* avg_ms, double* bytes_per_launch);
/* Same with the adjoint half switchable: do_adjoint = 0 is the residual-only launch whose algorithmic bytes are
 * SURVEY.md 8(d)'s 8 (C_u N + 2 N_R) (read the integrated channels and F, write R). */
int hpv_bench_residual(hpv_handle h, long n_elem, int reps, int do_adjoint, double* avg_ms, double* bytes_per_launch);
/* One such launch on the same seeded synthetic data, condensed to checksums {sum R, sum R^2, sum loss_e, sum |gbar|, sum gbar^2,
 * sum_e (e mod 97) loss_e}: lets tests compare the kernel plans that serve large batches (streaming / column-in-registers)
 * with each other and with the general projection kernel on the SAME data. */
int hpv_bench_residual_checksums(hpv_handle h, long n_elem, int do_adjoint, double* sums6);

#ifdef __cplusplus
}
#endif
#endif /* HPVPINN_H */
