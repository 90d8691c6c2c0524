// Shared by the host-side translation units of libhpvpinn.so (hpv_api.hip: handle, set-up, passes, training loop;
// hpv_exchange.hip: the multi-GPU exchanges; hpv_bench.hip: timing / benchmark / debug hooks): the handle itself and the
// few helpers all of them use.
#pragma once
#include <dlfcn.h>
#include <unistd.h>
#include <rccl/rccl.h>   // types and prototypes only: the library is dlopen'ed on first multi-GPU use (no link-time dependency)

#include <algorithm>
#include <atomic>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <string>
#include <vector>

#include "hpv_internal.h"
#include "hpv_mfma.h"
#include "hpv_project_wg.h"



struct Batch {
    long N = 0;
    NetDesc nd{};
    double* X = nullptr;     // [d][N]
    double* ACT = nullptr;   // saved slots of every hidden layer
    double* OUT = nullptr;   // [C][N]
    double* GBAR = nullptr;  // [C][N]
    double* GPART = nullptr; // [rows][P] partial parameter gradients
    int rows = 0;
    size_t act_doubles = 0;
};

struct TimerClass {
    std::vector<hipEvent_t> ev;  // start/stop pairs
    size_t used = 0;
    double total_ms = 0.0;
    long launches = 0;
};


struct hpv_ctx {
    hpv_config cfg{};
    std::string err;
    hipStream_t stream = nullptr;
    bool own_stream = false;
    int dim = 1;
    int P = 0, Ptot = 0, has_eps = 0;
    NetDesc nd_var{}, nd_val{};
    ProjDesc pd{};
    int backend = HPV_BACKEND_GENERIC;
    // quadrature / tables (host copies + device weighted tables)
    std::vector<double> xi, wx, yi, wy;
    int qx = 0, qy = 1, ntx = 0, nty = 1;
    double *d_wtx = nullptr, *d_wty = nullptr, *d_edge_dphi = nullptr;
    bool have_quad = false, have_tables = false, have_elems = false, have_params = false;
    // elements
    int nex = 0, ney = 1, e_begin = 0, e_end = 0;
    long n_elem = 0;
    double *d_coef = nullptr, *d_edge_coef = nullptr, *d_F = nullptr, *d_R = nullptr, *d_loss_e = nullptr,
           *d_deps_e = nullptr;
    std::vector<double> F_all;
    bool have_F = false;
    std::vector<int> nact_all;     // active test functions per element of the whole grid (empty: all); see hpv_set_active_tests
    int* d_nact = nullptr;         // ... of the owned elements
    Batch var, data, edge, pred;
    // host copies of the point sets; the device batches are (re)assembled lazily (assemble_batches)
    std::vector<double> Xq_host;   // [dim][Nq] quadrature points of the owned elements
    std::vector<double> Xd_host;   // [n_data][dim] boundary / data points
    long Nq = 0;
    bool batch_dirty = true;
    bool merged = false;           // MFMA path: data points ride as extra tiles of the quadrature batch
    long data_off = 0;             // first data point inside the merged batch
    double* d_udata = nullptr;
    double* d_data_part = nullptr;
    int n_data = 0;
    // parameters / optimizer
    double *d_theta = nullptr, *d_m = nullptr, *d_v = nullptr, *d_state = nullptr, *d_RB = nullptr;
    double* d_hist = nullptr;   // [HPV_HIST_CAP][4] loss / epsilon history (AdamArgs)
    int* d_hist_idx = nullptr;
    unsigned long long* d_nupd = nullptr;   // updates applied so far (AdamArgs::n_upd)
    bool shared_elem_ok = true; // hpv_set_shared_element_kernels
    long long nupd_host = 0;    // updates ENQUEUED so far (equals *d_nupd once the stream is idle unless a run failed)
    int n_fallbacks = 0;        // runs finished on the barrier-free structures after an exchange timeout (after_exchange_timeout)
    int* d_xerr = nullptr;      // sticky failure flag: a SPLIT-mode element barrier timed out (kernels_fused.hip); see sync_check
    // mfma path (one object per batch: quadrature points, boundary/data points, element edges)
    HpvMfma* mfma = nullptr;
    HpvMfma* mfma_data = nullptr;
    HpvMfma* mfma_edge = nullptr;
    HpvMfma* mfma_pred = nullptr;
    // hpv_eval_channels reports the REFERENCE's channel list (nd_eval); where the training pass runs on fewer channels (Poisson-2D
    // var_form 0: u_xx + u_yy as one mixed second tangent, NetDesc::t2w) a forward-only object + output buffer are created on demand
    NetDesc nd_eval{};
    bool eval_differs = false;
    HpvMfma* mfma_eval = nullptr;
    double* d_eval_out = nullptr;
    long eval_N = 0;
    // strong-form PINN branch (scheme == PINNs): collocation batch: u, u_x, u_y and the Laplacian as one mixed second tangent (four channels)
    NetDesc nd_pinn{};
    Batch colloc;
    HpvMfma* mfma_colloc = nullptr;
    double *d_fcol = nullptr, *d_col_part = nullptr;
    int n_col = 0;
    long n_col_total = 0;      // collocation points of ALL shards (the mean of P2:124 runs over them)
    double* d_jac = nullptr;   // |J_e| of the owned elements (RHS assembly, hpv_assemble_rhs)
    // in-library exchange of the packed buffer between the ranks of a node (hpv_p2p_*)
    P2PArgs pp{};
    bool p2p_on = false;
    // one-workgroup grids (config 1): hpv_step asks for `persist_want` iterations in one launch; the tile kernel says how many it ran
    int persist_want = 1, persist_done = 1;
    bool persist_probed = false;
    bool persist_seen = false;   // the most recent training pass ended inside the tile kernel (in-kernel finalize): persistent launches possible
    int pass_structure = -1;   // see hpv_pass_structure
    char variant[320] = "";    // see hpv_kernel_variant
    double* d_inbox = nullptr;
    unsigned long long* d_flag = nullptr;
    unsigned long long* d_p2p_counter = nullptr;
    int* d_p2p_err = nullptr;
    void* p2p_maps[2 * HPV_P2P_MAX] = {};
    // in-library RCCL all-reduce of the packed buffer (hpv_rccl_*): the multi-GPU default
    ncclComm_t rccl_comm = nullptr;
    bool rccl_on = false;
    int rccl_world = 1, rccl_rank = 0;
    // hpv_rccl_abandon (the ONE entry point that may be called from another thread while a call is inside the library): the
    // caller has given up waiting for a blocking hpv_rccl_connect / hpv_rccl_selftest that runs on a helper thread.  The
    // abandoned call then never touches the handle again: a communicator that comes up late is destroyed, rccl_on stays false
    std::atomic<int> rccl_abandoned{0};
    double* d_upart = nullptr; // partial residual sums of the row-split projection (few tall elements)
    int proj_split = 1;        // workgroups per element there; loss_e / deps_e hold n_elem * proj_split entries
    long n_red_alloc = 0;      // entries loss_e / deps_e were allocated with (>= every launch structure's count)
    long n_loss_entries = 0;   // entries the most recent pass wrote (what the finalize kernel sums)
    // timing
    bool timing = false;
    TimerClass timers[3];
    // whole-iteration hipGraph (forward + projection + backward || boundary branch -> finalize -> Adam)
    hipStream_t stream2 = nullptr;       // side stream: the boundary/data branch runs beside the main branch
    hipEvent_t ev_fork = nullptr, ev_join = nullptr;
    bool side_active = false;            // true only while capturing
    bool use_graph = true;
    // the multi-GPU iteration in two launches (in-library RCCL exchange): inside a sequence of training iterations the TF1-Adam
    // update of iteration i is DEFERRED -- k_iter_fused of iteration i + 1 computes with the updated parameters (prologue), k_finalize
    // behind it stores them -- and the last one of the sequence is applied by k_adam (flush_adam): nothing is pending at an API boundary
    bool defer_ok = true;        // HPV_NO_DEFERRED_ADAM=1 turns it off (A/B, tests)
    bool defer_adam = false;     // a sequence is being enqueued / captured
    bool adam_pending = false;   // RB holds a reduced gradient whose update has not been applied
    hipGraphExec_t g_stepK = nullptr;    // HPV_GRAPH_ITERS iterations per replay (fewer inter-graph gaps)
    hipGraphExec_t g_rem[8] = {};        // g_rem[r]: r iterations (the remainder of a call), captured at first use
};

namespace hpvd {
extern std::string g_create_error;          // last hpv_create() error (hpv_last_error(NULL))
int fail(hpv_ctx* h, int code, const char* fmt, ...);
int upload(hpv_ctx* h, double* dst, const double* src, size_t n);
void drop_graph(hpv_ctx* h);                 // captured iteration graphs (hpv_api.hip)
void p2p_release(hpv_ctx* h);                // hpv_exchange.hip
void rccl_release(hpv_ctx* h);
int p2p_check(hpv_ctx* h);
// one ncclAllReduce(sum, double) of `n` doubles in place on stream `s` (test-hooks builds can make it fail on demand)
ncclResult_t rccl_allreduce(hpv_ctx* h, void* buf, size_t n, hipStream_t s);
const char* rccl_error_string(ncclResult_t r);
}  // namespace hpvd

#define HIPCHK(h, call)                                                                            \
    do {                                                                                           \
        hipError_t e_ = (call);                                                                    \
        if (e_ != hipSuccess) return hpvd::fail(h, -2, "%s failed: %s (%s:%d)", #call, hipGetErrorString(e_), __FILE__, __LINE__); \
    } while (0)

namespace hpvd {
template <typename T>
int dalloc(hpv_ctx* h, T** p, size_t n) {
    if (*p) { (void)hipFree(*p); *p = nullptr; }
    if (n == 0) return 0;
    HIPCHK(h, hipMalloc((void**)p, n * sizeof(T)));
    return 0;
}
}  // namespace hpvd
