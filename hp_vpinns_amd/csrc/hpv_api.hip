// C-ABI of libhpvpinn.so (include/hpvpinn.h): host orchestration of one hp-VPINN training
// handle = one GPU's shard of elements + a replica of the network parameters.  (The multi-GPU exchanges live in hpv_exchange.hip,
// the timing / benchmark / debug hooks in hpv_bench.hip; hpv_ctx.h holds the handle and the helpers they share.)
#include "hpv_ctx.h"

using namespace hpvd;

namespace hpvd {
std::string g_create_error;

int fail(hpv_ctx* h, int code, const char* fmt, ...) {
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    if (h) h->err = buf; else g_create_error = buf;
    return code;
}

int upload(hpv_ctx* h, double* dst, const double* src, size_t n) {
    if (n == 0) return 0;
    HIPCHK(h, hipMemcpyAsync(dst, src, n * sizeof(double), hipMemcpyHostToDevice, h->stream));
    HIPCHK(h, hipStreamSynchronize(h->stream));
    return 0;
}

void drop_graph(hpv_ctx* h) {
    if (h->g_stepK) { (void)hipGraphExecDestroy(h->g_stepK); h->g_stepK = nullptr; }
    for (hipGraphExec_t& g : h->g_rem) if (g) { (void)hipGraphExecDestroy(g); g = nullptr; }
}
}  // namespace hpvd

static int sync_check(hpv_ctx* h);

int hpv_test_hook_split_skip() {
#ifdef HPV_TEST_HOOKS   // libhpvpinn_testhooks.so only: the product library does not read the variable
    const char* dbg = getenv("HPV_DEBUG_SPLIT_SKIP");
    return dbg ? std::max(0, atoi(dbg)) : 0;
#else
    return 0;
#endif
}



namespace {

void free_batch(Batch& b) {
    if (b.X) (void)hipFree(b.X);
    if (b.ACT) (void)hipFree(b.ACT);
    if (b.OUT) (void)hipFree(b.OUT);
    if (b.GBAR) (void)hipFree(b.GBAR);
    if (b.GPART) (void)hipFree(b.GPART);
    b = Batch{};
}

// Build the network descriptor for a given tangent-channel selection.
NetDesc make_netdesc(const hpv_config& c, int nT1, const int* t1dim, int nT2, const int* t2idx) {
    NetDesc nd{};
    nd.d = c.layers[0];
    nd.nl = c.n_layers - 1;
    int off = 0;
    for (int l = 0; l < c.n_layers; ++l) nd.width[l] = c.layers[l];
    for (int l = 0; l < nd.nl; ++l) {
        nd.woff[l] = off; off += c.layers[l] * c.layers[l + 1];
        nd.boff[l] = off; off += c.layers[l + 1];
    }
    nd.P = off;
    nd.act = c.act;
    nd.nT1 = nT1; nd.nT2 = nT2;
    for (int i = 0; i < nT1; ++i) nd.t1dim[i] = t1dim[i];
    for (int i = 0; i < nT2; ++i) nd.t2idx[i] = t2idx[i];
    nd.t2w[0] = 1.0; nd.t2w[1] = 0.0;
    nd.C = 1 + nT1 + nT2;
    nd.nslot = 2 + nT1 + nT2;
    long a = 0;
    for (int l = 0; l < nd.nl - 1; ++l) { nd.actoff[l] = a; a += (long)nd.nslot * c.layers[l + 1]; }
    nd.actoff[nd.nl - 1] = a;  // total slots*width (per point)
    return nd;
}

int alloc_batch(hpv_ctx* h, Batch& b, const NetDesc& nd, long N, bool need_bwd) {
    free_batch(b);
    b.N = N; b.nd = nd;
    if (N == 0) return 0;
    int rc;
    if ((rc = dalloc(h, &b.X, (size_t)nd.d * N))) return rc;
    if ((rc = dalloc(h, &b.OUT, (size_t)nd.C * N))) return rc;
    if (need_bwd) {
        b.act_doubles = (size_t)nd.actoff[nd.nl - 1] * N;
        if ((rc = dalloc(h, &b.ACT, b.act_doubles))) return rc;
        if ((rc = dalloc(h, &b.GBAR, (size_t)nd.C * N))) return rc;
        HIPCHK(h, hipMemsetAsync(b.GBAR, 0, (size_t)nd.C * N * sizeof(double), h->stream));
        b.rows = mlp_bwd_generic_rows(N);
        if ((rc = dalloc(h, &b.GPART, (size_t)b.rows * nd.P))) return rc;
    }
    return 0;
}

// [n][d] row-major host points -> [d][n] device
int upload_points(hpv_ctx* h, Batch& b, const double* X, long n, int d) {
    std::vector<double> t((size_t)n * d);
    for (long p = 0; p < n; ++p)
        for (int c = 0; c < d; ++c) t[(size_t)c * n + p] = X[(size_t)p * d + c];
    return upload(h, b.X, t.data(), t.size());
}

void tstart(hpv_ctx* h, int which) {
    if (!h->timing) return;
    TimerClass& t = h->timers[which];
    if (t.used + 2 > t.ev.size()) {
        size_t old = t.ev.size();
        t.ev.resize(old + 512);
        for (size_t i = old; i < t.ev.size(); ++i) (void)hipEventCreate(&t.ev[i]);
    }
    (void)hipEventRecord(t.ev[t.used], h->stream);
}
void tstop(hpv_ctx* h, int which) {
    if (!h->timing) return;
    TimerClass& t = h->timers[which];
    (void)hipEventRecord(t.ev[t.used + 1], h->stream);
    t.used += 2;
    t.launches += 1;
    if (t.used >= 4096) {  // flush
        (void)hipStreamSynchronize(h->stream);
        for (size_t i = 0; i < t.used; i += 2) { float ms = 0; (void)hipEventElapsedTime(&ms, t.ev[i], t.ev[i + 1]); t.total_ms += ms; }
        t.used = 0;
    }
}

// Build the device point batches from the host copies.  MFMA path: ONE batch = quadrature points (padded
// to a 16-point tile) followed by the boundary/data points, so a single forward and a single reverse
// launch serve both loss terms (the data points just carry zero adjoints on the tangent channels).
// Generic path: separate batches.
int assemble_batches(hpv_ctx* h) {
    int rc;
    const long N = h->Nq;
    const int d = h->dim, nd = h->n_data;
    if (h->mfma) { hpv_mfma_destroy(h->mfma); h->mfma = nullptr; }
    if (h->mfma_data) { hpv_mfma_destroy(h->mfma_data); h->mfma_data = nullptr; }
    h->backend = HPV_BACKEND_GENERIC;
    h->merged = false;
    if (h->cfg.backend != HPV_BACKEND_GENERIC && N > 0) {
        const long Npad = (N + 15) / 16 * 16, Ntot = Npad + nd;
        std::string why;
        h->mfma = hpv_mfma_create(h->nd_var, Ntot, &why);
        if (h->mfma) {
            h->backend = HPV_BACKEND_MFMA;
            h->merged = true;
            h->data_off = Npad;
            hpv_mfma_set_err_flag(h->mfma, h->d_xerr);
            hpv_mfma_set_split_ok(h->mfma, h->shared_elem_ok);
            if ((rc = alloc_batch(h, h->var, h->nd_var, Ntot, true))) return rc;
            if (h->var.ACT) { (void)hipFree(h->var.ACT); h->var.ACT = nullptr; }   // the MFMA path has its own store
            const int rows = hpv_mfma_max_rows(h->mfma, h->n_elem, (nd + 15) / 16);
            if ((rc = dalloc(h, &h->var.GPART, (size_t)rows * h->P))) return rc;
            h->var.rows = hpv_mfma_grad_rows(h->mfma);
            std::vector<double> X((size_t)d * Ntot, 0.0);
            for (int c = 0; c < d; ++c) {
                for (long p = 0; p < N; ++p) X[(size_t)c * Ntot + p] = h->Xq_host[(size_t)c * N + p];
                for (int p = 0; p < nd; ++p) X[(size_t)c * Ntot + Npad + p] = h->Xd_host[(size_t)p * d + c];
            }
            if ((rc = upload(h, h->var.X, X.data(), X.size()))) return rc;
            HIPCHK(h, hipMemsetAsync(h->var.GBAR, 0, (size_t)h->nd_var.C * Ntot * sizeof(double), h->stream));
            HIPCHK(h, hipMemsetAsync(h->var.OUT, 0, (size_t)h->nd_var.C * Ntot * sizeof(double), h->stream));
            free_batch(h->data);
            const size_t nparts = (size_t)std::max(64, (nd + 15) / 16);
            if ((rc = dalloc(h, &h->d_data_part, nparts))) return rc;
            HIPCHK(h, hipMemsetAsync(h->d_data_part, 0, nparts * sizeof(double), h->stream));
        } else if (h->cfg.backend == HPV_BACKEND_MFMA) {
            return fail(h, -4, "MFMA backend requested but not available for this shape: %s", why.c_str());
        } else {
            // no silent cliff: the generic kernels are one to two orders of magnitude slower than the MFMA path
            static std::atomic<int> warned{0};
            if (!warned.exchange(1))
                fprintf(stderr, "libhpvpinn: WARNING -- this network / channel set is not covered by the MFMA kernels (%s); running the "
                                "generic kernels, which are far slower (hpv_backend_in_use reports HPV_BACKEND_GENERIC)\n", why.c_str());
        }
    }
    if (!h->merged) {
        if ((rc = alloc_batch(h, h->var, h->nd_var, N, true))) return rc;
        if (N > 0 && (rc = upload(h, h->var.X, h->Xq_host.data(), h->Xq_host.size()))) return rc;
        if ((rc = alloc_batch(h, h->data, h->nd_val, nd, true))) return rc;
        if (nd > 0 && (rc = upload_points(h, h->data, h->Xd_host.data(), nd, d))) return rc;
    }
    HIPCHK(h, hipStreamSynchronize(h->stream));
    h->batch_dirty = false;
    return 0;
}

int check_ready(hpv_ctx* h) {
    if (!h->have_quad) return fail(h, -3, "hpv_set_quadrature has not been called");
    if (!h->have_tables) return fail(h, -3, "hpv_set_tables has not been called");
    if (!h->have_elems) return fail(h, -3, "hpv_set_elements has not been called");
    if (!h->have_params) return fail(h, -3, "hpv_set_params has not been called");
    if (h->have_F && !h->d_F && h->n_elem > 0) return fail(h, -3, "internal: F not sliced");
    if (h->batch_dirty) return assemble_batches(h);
    return 0;
}

void run_fwd(hpv_ctx* h, Batch& b, HpvMfma* m, int save_act) {
    if (b.N == 0) return;
    if (m) hpv_mfma_forward(m, h->d_theta, b.X, b.OUT, save_act, h->stream);
    else launch_mlp_fwd_generic(b.nd, h->d_theta, b.X, b.ACT, b.OUT, b.N, save_act, h->stream);
}
void run_bwd(hpv_ctx* h, Batch& b, HpvMfma* m) {
    if (b.N == 0) return;
    if (m) hpv_mfma_backward(m, h->d_theta, b.X, b.GBAR, b.GPART, &b.rows, h->stream);
    else launch_mlp_bwd_generic(b.nd, h->d_theta, b.X, b.ACT, b.GBAR, b.GPART, b.rows, b.N, h->stream);
}

// The small value-only batches (boundary/data points, element edges) ride the MFMA path too once the
// quadrature batch does; their generic activation stores are dropped.
int ensure_small_mfma(hpv_ctx* h, Batch& b, HpvMfma** m) {
    if (*m || h->backend != HPV_BACKEND_MFMA || b.N == 0) return 0;
    std::string why;
    *m = hpv_mfma_create(b.nd, b.N, &why);
    if (!*m) return 0;  // stays on the generic kernels
    if (b.ACT) { (void)hipFree(b.ACT); b.ACT = nullptr; }
    int rows = hpv_mfma_grad_rows(*m);
    if (rows > b.rows) { int rc = dalloc(h, &b.GPART, (size_t)rows * h->P); if (rc) return rc; }
    b.rows = rows;
    return 0;
}

AdamArgs adam_args(hpv_ctx* h) {
    return AdamArgs{h->d_theta, h->d_m, h->d_v, h->d_state, h->cfg.lr, h->cfg.beta1, h->cfg.beta2, h->cfg.eps,
                    h->d_hist, h->d_hist_idx, HPV_HIST_CAP, h->d_xerr, h->d_nupd};
}

int enqueue_pinn_pass(hpv_ctx* h, bool backward, bool fuse_adam);

// the boundary / data term as the merged kernels see it, and the projection arguments of the quadrature batch (one definition for
// enqueue_pass and the stand-alone timing of the iteration kernel)
static MfmaDataTerm pass_data_term(hpv_ctx* h, bool backward) {
    return MfmaDataTerm{h->data_off, h->merged ? h->n_data : 0, h->d_udata, h->var.GBAR, h->d_data_part,
                        h->n_data > 0 ? -2.0 * h->cfg.lossb_weight / (double)h->n_data : 0.0, backward ? 1 : 0};
}
static ProjArgs pass_proj_args(hpv_ctx* h, bool backward) {
    const double* eps_ptr = h->has_eps ? h->d_theta + h->P : nullptr;
    return ProjArgs{h->pd, h->var.OUT, h->var.GBAR, h->d_R, h->d_F, h->d_coef, h->n_elem, h->d_wtx, h->d_wty, eps_ptr,
                    h->d_loss_e, h->d_deps_e, h->var.N, backward ? 1 : 0, nullptr, nullptr, nullptr, nullptr};
}

// ---- one pass over both loss terms ------------------------------------------------------------------------------------------
// What a pass did (hpv_pass_structure / hpv_kernel_variant report structure and names): filled while the pass walks its plan.
struct PassPlan {
    long n_loss;             // loss_e / deps_e entries the pass wrote
    bool fin_done = false;   // the whole-iteration tile kernel ran the finalize step itself
    bool xch_used = false;   // a shared-element kernel (SPLIT mode, k_iter_tall) ran: k_finalize advances the exchange's launch counter
    bool pend_taken = false; // the deferred TF1-Adam update rode in k_iter_fused's prologue: k_finalize stores it
};

// (1) The variational term as ONE launch -- the element-resident whole-iteration kernels, tried in the order of their speed on the
// shapes they take (each declines what it does not cover): k_iter_fused (two-term / general forms on 12x12, 16x16, 20x20 points;
// takes a deferred update `pend` into its prologue), k_iter_tile (one tile per wave: 1-D rules, 10x10 points; a one-workgroup grid
// finishes the iteration itself), k_iter_elem (any instantiated shape / channel set; first with HPV_FUSE=e), k_iter_tall (80x80
// points).  Timed as the reverse-pass class.  false: nothing was launched.
static bool pass_whole_iteration(hpv_ctx* h, const MfmaDataTerm& dt, const ProjArgs& pa, bool fuse_adam, bool& pend, PassPlan& pl) {
    HpvMfma* m = h->mfma;
    Batch& v = h->var;
    int structure = -1;
    tstart(h, 2);
    auto fused = [&](const MfmaPendingAdam* pre) {
        return hpv_mfma_iter_fused(m, h->d_theta, v.X, v.GPART, &v.rows, h->stream, &dt, pa, h->n_elem, pre);
    };
    auto elem = [&] { return hpv_mfma_iter_elem(m, h->d_theta, v.X, v.GPART, &v.rows, h->stream, &dt, pa, h->n_elem); };
    if (pend) {      // the deferred update rides in k_iter_fused's prologue -- or is applied here, before anything else is launched
        const MfmaPendingAdam pre{adam_args(h), h->d_RB, h->Ptot};
        if (fused(&pre)) { pl.pend_taken = true; structure = hpv_mfma_sync_failed_possible(m) ? 3 : 2; }
        else launch_adam(adam_args(h), h->d_RB, h->P, h->Ptot, h->stream);
        pend = false;
    }
    if (structure < 0 && hpv_mfma_prefers_elem(m) && elem()) structure = 6;      // HPV_FUSE=e (A/B runs): the generic element-resident kernel first
    if (structure < 0 && fused(nullptr)) structure = hpv_mfma_sync_failed_possible(m) ? 3 : 2;
    if (structure < 0) {
        MfmaFinalize fin{adam_args(h), h->d_RB, h->cfg.lossb_weight, h->n_data, (h->n_data + 15) / 16, h->has_eps, adam_state_doubles(h->P) / 2};
        if (!fuse_adam) fin.ad.theta = nullptr;
        fin.n_iters = fuse_adam ? h->persist_want : 1;
        fin.iters_done = &h->persist_done;
        if (hpv_mfma_iter_tile(m, h->d_theta, v.X, v.GPART, &v.rows, h->stream, &dt, pa, h->n_elem, h->merged ? &fin : nullptr, &pl.fin_done)) structure = 4;
    }
    if (structure < 0 && elem()) structure = 6;
    if (structure < 0) {
        const int ts = hpv_mfma_tall_split(m, h->pd, h->n_elem);
        if (ts > 1 && h->n_elem * ts <= h->n_red_alloc &&
            hpv_mfma_iter_tall(m, h->d_theta, v.X, v.GPART, &v.rows, h->stream, &dt, pa, h->n_elem)) { structure = 5; pl.n_loss = h->n_elem * ts; }
    }
    if (structure < 0) return false;     // (nothing was launched; the caller re-records the start event)
    tstop(h, 2);
    h->pass_structure = structure;
    snprintf(h->variant, sizeof h->variant, "%s", hpv_mfma_variant(m, 0));
    pl.xch_used = hpv_mfma_sync_failed_possible(m);
    return true;
}

// (2) / (3) The variational term as separate launches: forward (+ edge batch) -> projection -> reverse; on the MFMA path the
// projection rides inside the reverse kernel where that applies (element-block mode), otherwise it is the fastest projection
// kernel that takes the shape.
static void pass_separate(hpv_ctx* h, const MfmaDataTerm& dt, const ProjArgs& pa, bool backward, bool use_mfma) {
    Batch& v = h->var;
    const double* eps_ptr = pa.eps_ptr;
    tstart(h, 0);
    if (use_mfma) hpv_mfma_forward(h->mfma, h->d_theta, v.X, v.OUT, backward ? 1 : 0, h->stream, &dt);
    else run_fwd(h, v, nullptr, backward ? 1 : 0);
    tstop(h, 0);
    if (h->pd.edge) run_fwd(h, h->edge, h->mfma_edge, backward ? 1 : 0);
    bool bfused = false;
    if (backward && use_mfma) {
        tstart(h, 2);   // timed as the reverse-pass class (the projection is ~1 % of its flops)
        bfused = hpv_mfma_backward_fused(h->mfma, h->d_theta, v.X, v.GBAR, v.GPART, &v.rows, h->stream, pa, h->n_elem);
        if (bfused) tstop(h, 2);
    }
    if (backward) h->pass_structure = bfused ? 1 : 0;
    if (bfused) { snprintf(h->variant, sizeof h->variant, "%s + %s", hpv_mfma_variant(h->mfma, 1), hpv_mfma_variant(h->mfma, 3)); }
    else {
        tstart(h, 1);
        // (the specialised kernels need GBAR's unused channels pre-zeroed: true for every batch, see alloc_batch)
        const bool special = h->cfg.backend != HPV_BACKEND_GENERIC;
        auto wg = [&](double* upart) {
            return launch_project_wg(h->pd, v.OUT, v.GBAR, h->d_R, h->d_F, h->d_coef, h->n_elem, h->d_wtx, h->d_wty, eps_ptr, h->d_loss_e, h->d_deps_e,
                                     v.N, h->n_elem, backward ? 1 : 0, h->edge.OUT, h->d_edge_dphi, h->d_edge_coef, h->edge.GBAR, h->stream, upart);
        };
        // few elements (at most two per CU) of a 2-D shape: one workgroup per element before "a lane owns a line"
        bool small_grid = h->dim == 2 && h->n_elem <= 512 && h->proj_split == 1 && !h->pd.nact;
#ifdef HPV_EXPERIMENTS
        if (getenv("HPV_PJ_WG_SMALL")) small_grid = false;      // (A/B: "a lane owns a line" on small grids too)
#endif
        const char* pname = "k_project";
        if (special && small_grid && wg(nullptr)) pname = "k_project_wg";
        else if (special && launch_project_tp(h->pd, v.OUT, v.GBAR, h->d_R, h->d_F, h->d_coef, h->n_elem, h->d_wtx, h->d_wty, eps_ptr, h->d_loss_e,
                                              h->d_deps_e, v.N, h->n_elem, backward ? 1 : 0, h->stream)) pname = "k_project_tp";
        else if (special && wg(h->d_upart)) pname = h->proj_split > 1 ? "k_project_rows" : "k_project_wg";
        else launch_project(h->pd, v.OUT, v.GBAR, h->d_R, h->d_F, h->d_coef, h->n_elem, h->d_wtx, h->d_wty, eps_ptr, h->d_loss_e, h->d_deps_e, v.N,
                            h->n_elem, backward ? 1 : 0, h->edge.OUT, h->d_edge_dphi, h->d_edge_coef, h->edge.GBAR, h->stream);
        tstop(h, 1);
        if (backward) {
            snprintf(h->variant, sizeof h->variant, "%s + %s<%dx%d/%dx%d> + %s", use_mfma ? hpv_mfma_variant(h->mfma, 1) : "k_mlp_fwd_generic", pname,
                     h->pd.qx, h->pd.qy, h->pd.ntx, h->pd.nty, use_mfma ? hpv_mfma_variant(h->mfma, 2) : "k_mlp_bwd_generic");
            tstart(h, 2);
            if (use_mfma) hpv_mfma_backward(h->mfma, h->d_theta, v.X, v.GBAR, v.GPART, &v.rows, h->stream);
            else run_bwd(h, v, nullptr);
            tstop(h, 2);
        }
    }
    if (backward && h->pd.edge) run_bwd(h, h->edge, h->mfma_edge);
}

// backward: also the reverse pass and the gradient reduction; fuse_adam: the finalize kernel applies the TF1 Adam update itself
// (single-GPU training step); pend: RB holds the previous iteration's reduced gradient, its update not applied yet (multi-GPU
// sequence, see hpv_ctx::defer_adam): k_iter_fused takes it into its prologue and k_finalize stores it; any other structure gets a
// k_adam launch in front.
int enqueue_pass(hpv_ctx* h, bool backward, bool fuse_adam = false, bool pend = false) {
    // (an error exit BEFORE the pending update has been taken over by a launch applies it first, best effort: it was counted when
    //  its iteration was enqueued, and dropping it would leave the device one update behind the host's count -- advisor, round 5)
    auto apply_pend = [&] { if (pend) { launch_adam(adam_args(h), h->d_RB, h->P, h->Ptot, h->stream); pend = false; } };
    if (h->cfg.scheme == HPV_SCHEME_PINN || !backward) apply_pend();
    if (h->cfg.scheme == HPV_SCHEME_PINN) return enqueue_pinn_pass(h, backward, fuse_adam);
    int rc = check_ready(h);
    if (rc) { apply_pend(); return rc; }
    const bool use_mfma = h->mfma && h->backend == HPV_BACKEND_MFMA;
    // the deferred update can only ride in a whole-iteration kernel that is the ONLY reader of the parameters in this pass (boundary
    // points merged into it, no edge batch); otherwise it is applied here, before anything of this pass is launched or forked
    if (!(use_mfma && h->var.N > 0 && (h->merged || h->n_data == 0) && !h->pd.edge && !hpv_mfma_prefers_elem(h->mfma))) apply_pend();
    if (!h->side_active) {  // allocations are not allowed inside a stream capture
        if ((rc = ensure_small_mfma(h, h->data, &h->mfma_data))) { apply_pend(); return rc; }
        if (h->pd.edge && (rc = ensure_small_mfma(h, h->edge, &h->mfma_edge))) { apply_pend(); return rc; }
    }
    hipStream_t smain = h->stream;
    const bool fork = h->side_active && !h->merged && h->n_data > 0;
    if (fork) {
        (void)hipEventRecord(h->ev_fork, smain);
        (void)hipStreamWaitEvent(h->stream2, h->ev_fork, 0);
    }
    PassPlan pl{(long)h->n_elem * h->proj_split};
    // --- variational term on this shard's quadrature batch ---
    if (h->var.N > 0) {
        const MfmaDataTerm dt = pass_data_term(h, backward);
        const ProjArgs pa = pass_proj_args(h, backward);
        if (!(backward && use_mfma && pass_whole_iteration(h, dt, pa, fuse_adam, pend, pl))) pass_separate(h, dt, pa, backward, use_mfma);
    }
    // --- boundary / data term: inside the quadrature batch (one partial per 16-point data tile), or its own small batch ---
    int ndp = 0;
    if (h->n_data > 0 && h->merged) {
        ndp = (h->n_data + 15) / 16;
    } else if (h->n_data > 0) {
        if (fork) h->stream = h->stream2;   // the launch helpers read h->stream
        run_fwd(h, h->data, h->mfma_data, backward ? 1 : 0);
        ndp = (h->n_data + 255) / 256; if (ndp > 64) ndp = 64;
        launch_data_loss(h->data.OUT, h->d_udata, backward ? h->data.GBAR : nullptr,
                         -2.0 * h->cfg.lossb_weight / (double)h->n_data, h->d_data_part, h->n_data, h->stream);
        if (backward) run_bwd(h, h->data, h->mfma_data);
        h->stream = smain;
    }
    if (fork) {
        (void)hipEventRecord(h->ev_join, h->stream2);
        (void)hipStreamWaitEvent(smain, h->ev_join, 0);
    }
    const AdamArgs ad = adam_args(h);
    if (backward && fuse_adam) h->persist_seen = pl.fin_done;
    if (!pl.fin_done)
        launch_finalize(backward && h->var.N > 0 ? h->var.GPART : nullptr, h->var.rows,
                        backward && h->n_data > 0 && !h->merged ? h->data.GPART : nullptr, h->data.rows,
                        backward && h->pd.edge && h->edge.N > 0 ? h->edge.GPART : nullptr, h->edge.rows, h->d_loss_e,
                        pl.n_loss, h->d_deps_e, h->d_data_part, ndp, h->cfg.lossb_weight, h->n_data, h->P, h->has_eps, h->d_RB,
                        backward ? 1 : 0, (backward && (fuse_adam || pl.pend_taken)) ? &ad : nullptr, h->stream, h->d_xerr,
                        pl.xch_used ? hpv_mfma_xiter(h->mfma) : nullptr, pl.pend_taken ? 1 : 0);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return fail(h, -2, "kernel launch failed: %s", hipGetErrorString(e));
    return 0;
}

// scheme == 'PINNs' (P2:128-129): loss = w*lossb + lossp, lossp = mean((u_xx+u_yy-f)^2) at the collocation points
int enqueue_pinn_pass(hpv_ctx* h, bool backward, bool fuse_adam) {
    if (!h->have_params) return fail(h, -3, "hpv_set_params has not been called");
    if (h->n_col <= 0) return fail(h, -3, "hpv_set_collocation has not been called");
    int rc;
    if (h->batch_dirty) {   // only the data batch matters here
        if ((rc = alloc_batch(h, h->data, h->nd_val, h->n_data, true))) return rc;
        if (h->n_data > 0 && (rc = upload_points(h, h->data, h->Xd_host.data(), h->n_data, h->dim))) return rc;
        if (h->mfma_data) { hpv_mfma_destroy(h->mfma_data); h->mfma_data = nullptr; }
        h->merged = false;
        h->batch_dirty = false;
    }
    if (!h->side_active) {
        if (h->cfg.backend != HPV_BACKEND_GENERIC && !h->mfma_colloc) {
            std::string why;
            h->mfma_colloc = hpv_mfma_create(h->colloc.nd, h->colloc.N, &why);
            if (h->mfma_colloc) {
                h->backend = HPV_BACKEND_MFMA;
                if (h->colloc.ACT) { (void)hipFree(h->colloc.ACT); h->colloc.ACT = nullptr; }
                int rows = hpv_mfma_grad_rows(h->mfma_colloc);
                if (rows > h->colloc.rows && (rc = dalloc(h, &h->colloc.GPART, (size_t)rows * h->P))) return rc;
                h->colloc.rows = rows;
            } else if (h->cfg.backend == HPV_BACKEND_MFMA) return fail(h, -4, "MFMA backend not available: %s", why.c_str());
        }
        if ((rc = ensure_small_mfma(h, h->data, &h->mfma_data))) return rc;
    }
    run_fwd(h, h->colloc, h->mfma_colloc, backward ? 1 : 0);
    launch_pinn_residual(h->colloc.OUT, h->d_fcol, h->colloc.GBAR, h->d_col_part, h->colloc.N, h->n_col, h->n_col_total,
                         backward ? 1 : 0, h->stream);
    if (backward) run_bwd(h, h->colloc, h->mfma_colloc);
    int ndp = 0;
    if (h->n_data > 0) {
        run_fwd(h, h->data, h->mfma_data, backward ? 1 : 0);
        ndp = (h->n_data + 255) / 256; if (ndp > 64) ndp = 64;
        launch_data_loss(h->data.OUT, h->d_udata, backward ? h->data.GBAR : nullptr,
                         -2.0 * h->cfg.lossb_weight / (double)h->n_data, h->d_data_part, h->n_data, h->stream);
        if (backward) run_bwd(h, h->data, h->mfma_data);
    }
    const AdamArgs ad = adam_args(h);
    launch_finalize(backward ? h->colloc.GPART : nullptr, h->colloc.rows, backward && h->n_data > 0 ? h->data.GPART : nullptr,
                    h->data.rows, nullptr, 0, h->d_col_part, pinn_residual_parts(h->n_col), nullptr, h->d_data_part, ndp,
                    h->cfg.lossb_weight, h->n_data, h->P, 0, h->d_RB, backward ? 1 : 0, (backward && fuse_adam) ? &ad : nullptr,
                    h->stream);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return fail(h, -2, "kernel launch failed: %s", hipGetErrorString(e));
    return 0;
}

// A pass whose packed buffer is made global: with the in-library exchange connected the reduced buffer (and, for a
// training iteration, the Adam update) follows the finalize kernel on the same stream; otherwise the plain pass.
int enqueue_pass_x(hpv_ctx* h, bool backward, bool fuse_adam) {
    if (h->rccl_on) {
        // ONE collective per iteration: ncclAllReduce(sum) of [grad | d eps | lossv | w*lossb | msq | pad] over
        // xGMI, on the handle's stream (captured into the iteration graphs like the kernels), then the identical TF1 Adam
        // update on every rank
        const bool pend = h->adam_pending;
        h->adam_pending = false;                  // (taken into this pass's kernels, or applied in front of them)
        int rc = enqueue_pass(h, backward, false, pend);
        if (rc) return rc;
        ncclResult_t r = rccl_allreduce(h, h->d_RB, (size_t)h->Ptot + 4, h->stream);
        if (r != ncclSuccess) return fail(h, -6, "ncclAllReduce failed: %s", rccl_error_string(r));
        if (backward && fuse_adam) {
            // inside a sequence of iterations the update is deferred into the NEXT iteration's kernels (two launches + one collective
            // per iteration); the sequence's last one -- and every stand-alone iteration -- is applied here
            if (h->defer_adam && h->defer_ok) h->adam_pending = true;
            else launch_adam(adam_args(h), h->d_RB, h->P, h->Ptot, h->stream);
        }
        hipError_t e = hipGetLastError();
        if (e != hipSuccess) return fail(h, -2, "adam launch failed: %s", hipGetErrorString(e));
        return 0;
    }
    if (!h->p2p_on) return enqueue_pass(h, backward, fuse_adam);
    int rc = enqueue_pass(h, backward, false);
    if (rc) return rc;
    const AdamArgs ad = adam_args(h);
    launch_p2p_exchange(h->pp, h->d_RB, (backward && fuse_adam) ? &ad : nullptr, h->P, h->Ptot, h->stream);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return fail(h, -2, "exchange launch failed: %s", hipGetErrorString(e));
    return 0;
}

// end of a sequence of training iterations: the deferred update, if any, is applied (k_adam)
void flush_adam(hpv_ctx* h) {
    if (h->adam_pending) launch_adam(adam_args(h), h->d_RB, h->P, h->Ptot, h->stream);
    h->adam_pending = false;
    h->defer_adam = false;
}

#define HPV_GRAPH_ITERS 8

// Capture `iters` whole training iterations (incl. the Adam updates) into an executable graph.
int build_step_graph(hpv_ctx* h, int iters, hipGraphExec_t* out) {
    int rc;
    // one direct pass first: lazily created objects (small-batch MFMA stores) must exist before capture
    if ((rc = ensure_small_mfma(h, h->data, &h->mfma_data))) return rc;
    if (h->pd.edge && (rc = ensure_small_mfma(h, h->edge, &h->mfma_edge))) return rc;
    HIPCHK(h, hipStreamSynchronize(h->stream));
    hipGraph_t graph = nullptr;
    HIPCHK(h, hipStreamBeginCapture(h->stream, hipStreamCaptureModeThreadLocal));
    h->side_active = true;
    h->defer_adam = true;
    for (int k = 0; k < iters && !rc; ++k) rc = enqueue_pass_x(h, true, true);   // forward .. finalize (+ fused Adam / exchange)
    if (!rc) flush_adam(h); else { h->adam_pending = false; h->defer_adam = false; }
    h->side_active = false;
    hipError_t e = hipStreamEndCapture(h->stream, &graph);
    if (rc) { if (graph) (void)hipGraphDestroy(graph); return rc; }
    if (e != hipSuccess) return fail(h, -2, "hipStreamEndCapture failed: %s", hipGetErrorString(e));
    e = hipGraphInstantiate(out, graph, nullptr, nullptr, 0);
    (void)hipGraphDestroy(graph);
    if (e != hipSuccess) { *out = nullptr; return fail(h, -2, "hipGraphInstantiate failed: %s", hipGetErrorString(e)); }
    return 0;
}

}  // namespace

extern "C" {

const char* hpv_last_error(hpv_handle h) { return h ? h->err.c_str() : g_create_error.c_str(); }

int hpv_create(hpv_handle* out, const hpv_config* cfg) {
    if (!out || !cfg) return fail(nullptr, -1, "null argument");
    *out = nullptr;
    if (cfg->n_layers < 2 || cfg->n_layers > HPV_MAX_LAYERS) return fail(nullptr, -1, "n_layers out of range");
    const int dim = (cfg->pde == HPV_PDE_POISSON1D) ? 1 : 2;
    if (cfg->pde < 0 || cfg->pde > 2) return fail(nullptr, -1, "unknown pde %d", cfg->pde);
    if (cfg->layers[0] != dim) return fail(nullptr, -1, "layers[0]=%d but the problem is %d-D", cfg->layers[0], dim);
    if (cfg->layers[cfg->n_layers - 1] != 1) return fail(nullptr, -1, "the network must have one output");
    for (int l = 1; l < cfg->n_layers - 1; ++l)
        if (cfg->layers[l] < 1 || cfg->layers[l] > HPV_MAXH) return fail(nullptr, -1, "hidden width %d not in 1..%d", cfg->layers[l], HPV_MAXH);
    if (cfg->act != HPV_ACT_TANH && cfg->act != HPV_ACT_SIN) return fail(nullptr, -1, "unknown activation");
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0) return fail(nullptr, -2, "no HIP device available");
    if (cfg->device < 0 || cfg->device >= ndev) return fail(nullptr, -1, "device %d out of range (%d devices)", cfg->device, ndev);
    if (hipSetDevice(cfg->device) != hipSuccess) return fail(nullptr, -2, "hipSetDevice failed");

    hpv_ctx* h = new hpv_ctx();
    h->cfg = *cfg;
    h->dim = dim;
    if (hipStreamCreate(&h->stream) != hipSuccess) { delete h; return fail(nullptr, -2, "hipStreamCreate failed"); }
    h->own_stream = true;
    if (hipStreamCreateWithFlags(&h->stream2, hipStreamNonBlocking) != hipSuccess ||
        hipEventCreateWithFlags(&h->ev_fork, hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&h->ev_join, hipEventDisableTiming) != hipSuccess) {
        delete h; return fail(nullptr, -2, "side stream / event creation failed");
    }
    { const char* ng = getenv("HPV_NO_GRAPH"); h->use_graph = !(ng && ng[0] == '1'); }
    { const char* nd = getenv("HPV_NO_DEFERRED_ADAM"); h->defer_ok = !(nd && nd[0] == '1'); }

    // channel selection + integrand terms per (pde, var_form)
    int t1[2] = {0, 1}, t2[2] = {0, 1};
    ProjDesc& pd = h->pd;
    pd = ProjDesc{};
    int nT1 = 0, nT2 = 0;
    bool mixed = false;
    const int vf = cfg->var_form;
    auto term = [&](int dx, int dy, int eps_mult) -> TermDesc& {
        TermDesc& t = pd.t[pd.nterms++];
        t = TermDesc{}; t.dx = dx; t.dy = dy; t.eps_mult = eps_mult; return t;
    };
    if (cfg->pde == HPV_PDE_POISSON1D) {
        if (vf == 1) { nT1 = 1; nT2 = 1; term(0, 0, 0).a0[2] = 1.0; }            // P1:83-84  -J int u'' phi
        else if (vf == 2) { nT1 = 1; term(1, 0, 0).a0[1] = 1.0; }                // P1:86-87  int u' phi'
        else if (vf == 3) { term(2, 0, 0).a0[0] = 1.0; pd.edge = 1; }            // P1:89-91
        else { delete h; return fail(nullptr, -1, "Poisson-1D var_form must be 1, 2 or 3"); }
    } else if (cfg->pde == HPV_PDE_POISSON2D) {
        if (vf == 0) {                                                           // P2:91, 93-96: integrand u_xx + u_yy
            // ONE mixed second tangent (NetDesc::t2w = {1, 1}) instead of the two channels u_xx, u_yy: 4 channels through the
            // forward, the tangent recompute and the reverse pass instead of 5 (second-order channels propagate linearly)
            nT1 = 2; nT2 = 1; mixed = true; term(0, 0, 0).a0[3] = 1.0;
        }
        else if (vf == 1) { nT1 = 2; term(1, 0, 0).a0[1] = 1.0; term(0, 1, 0).a0[2] = 1.0; }            // P2:98-105
        else if (vf == 2) { term(2, 0, 0).a0[0] = 1.0; term(0, 2, 0).a0[0] = 1.0; }                     // P2:108-115
        else { delete h; return fail(nullptr, -1, "Poisson-2D var_form must be 0, 1 or 2"); }
    } else {
        h->has_eps = 1;
        if (vf == 0) {                                                           // P3:161-167
            nT1 = 2; nT2 = 1;  // channels u, u_x, u_t, u_xx
            TermDesc& t = term(0, 0, 0); t.a0[1] = cfg->V; t.a0[2] = 1.0; t.a1[3] = -1.0;
        } else if (vf == 1) {                                                    // P3:169-174
            nT1 = 2;
            TermDesc& t = term(0, 0, 0); t.a0[1] = cfg->V; t.a0[2] = 1.0;
            term(1, 0, 1).a0[1] = 1.0;
        } else { delete h; return fail(nullptr, -1, "AdvDiff var_form must be 0 or 1"); }
    }
    h->nd_var = make_netdesc(*cfg, nT1, t1, nT2, t2);
    h->nd_eval = h->nd_var;
    if (mixed) {
        h->nd_var.t2w[0] = 1.0; h->nd_var.t2w[1] = 1.0;
        h->nd_eval = make_netdesc(*cfg, 2, t1, 2, t2);      // what hpv_eval_channels reports: u, u_x, u_y, u_xx, u_yy
        h->eval_differs = true;
    }
    h->nd_val = make_netdesc(*cfg, 0, t1, 0, t2);
    if (cfg->scheme == HPV_SCHEME_PINN) {
        if (cfg->pde != HPV_PDE_POISSON2D) { delete h; return fail(nullptr, -1, "scheme PINNs is the Poisson-2D branch (P2:128-129)"); }
        // (u_xx + u_yy as ONE mixed second tangent, NetDesc::t2w: four channels instead of five through forward and reverse -- P2:187-194)
        h->nd_pinn = make_netdesc(*cfg, 2, t1, 1, t2);
        h->nd_pinn.t2w[0] = 1.0; h->nd_pinn.t2w[1] = 1.0;
    } else if (cfg->scheme != HPV_SCHEME_VPINN) { delete h; return fail(nullptr, -1, "unknown scheme %d", cfg->scheme); }
    pd.C = h->nd_var.C;
    pd.has_eps = h->has_eps;
    h->P = h->nd_var.P;
    h->Ptot = h->P + h->has_eps;

    int rc = 0;
    rc |= dalloc(h, &h->d_theta, (size_t)h->Ptot);
    rc |= dalloc(h, &h->d_m, (size_t)h->Ptot);
    rc |= dalloc(h, &h->d_v, (size_t)h->Ptot);
    rc |= dalloc(h, &h->d_state, (size_t)adam_state_doubles(h->P));
    rc |= dalloc(h, &h->d_hist, (size_t)4 * HPV_HIST_CAP);
    rc |= dalloc(h, &h->d_hist_idx, (size_t)1);
    if (!rc) (void)hipMemset(h->d_hist_idx, 0, sizeof(int));
    rc |= dalloc(h, &h->d_xerr, (size_t)2);      // [0] the sticky flag; [1] "the deferred update was suppressed" (k_iter_fused prologue -> k_finalize)
    if (!rc) (void)hipMemset(h->d_xerr, 0, 2 * sizeof(int));
    rc |= dalloc(h, &h->d_nupd, (size_t)1);
    if (!rc) (void)hipMemset(h->d_nupd, 0, sizeof(unsigned long long));
    rc |= dalloc(h, &h->d_RB, (size_t)h->Ptot + 4);
    rc |= dalloc(h, &h->d_data_part, 64);
    if (rc) { g_create_error = h->err; hpv_destroy(h); return -2; }
    (void)hipMemset(h->d_RB, 0, ((size_t)h->Ptot + 4) * sizeof(double));
    (void)hipMemset(h->d_data_part, 0, 64 * sizeof(double));
    *out = h;
    return 0;
}

void hpv_destroy(hpv_handle h) {
    if (!h) return;
    (void)hipSetDevice(h->cfg.device);
    if (h->stream) (void)hipStreamSynchronize(h->stream);
    drop_graph(h);
    if (h->stream2) { (void)hipStreamSynchronize(h->stream2); (void)hipStreamDestroy(h->stream2); }
    if (h->ev_fork) (void)hipEventDestroy(h->ev_fork);
    if (h->ev_join) (void)hipEventDestroy(h->ev_join);
    if (h->mfma) hpv_mfma_destroy(h->mfma);
    if (h->mfma_data) hpv_mfma_destroy(h->mfma_data);
    if (h->mfma_edge) hpv_mfma_destroy(h->mfma_edge);
    if (h->mfma_pred) hpv_mfma_destroy(h->mfma_pred);
    if (h->mfma_eval) hpv_mfma_destroy(h->mfma_eval);
    if (h->d_eval_out) (void)hipFree(h->d_eval_out);
    if (h->mfma_colloc) hpv_mfma_destroy(h->mfma_colloc);
    free_batch(h->colloc);
    if (h->d_fcol) (void)hipFree(h->d_fcol);
    if (h->d_col_part) (void)hipFree(h->d_col_part);
    if (h->d_jac) (void)hipFree(h->d_jac);
    if (h->d_upart) (void)hipFree(h->d_upart);
    p2p_release(h);
    rccl_release(h);
    free_batch(h->var); free_batch(h->data); free_batch(h->edge); free_batch(h->pred);
    double* ptrs[] = {h->d_wtx, h->d_wty, h->d_edge_dphi, h->d_coef, h->d_edge_coef, h->d_F, h->d_R, h->d_loss_e,
                      h->d_deps_e, h->d_udata, h->d_data_part, h->d_theta, h->d_m, h->d_v, h->d_state, h->d_RB, h->d_hist};
    for (double* p : ptrs) if (p) (void)hipFree(p);
    if (h->d_hist_idx) (void)hipFree(h->d_hist_idx);
    if (h->d_xerr) (void)hipFree(h->d_xerr);
    if (h->d_nupd) (void)hipFree(h->d_nupd);
    if (h->d_nact) (void)hipFree(h->d_nact);
    for (auto& t : h->timers) for (auto e : t.ev) (void)hipEventDestroy(e);
    if (h->own_stream && h->stream) (void)hipStreamDestroy(h->stream);
    delete h;
}

int hpv_set_stream(hpv_handle h, void* s) {
    if (!h) return -1;
    if (h->own_stream && h->stream) { (void)hipStreamSynchronize(h->stream); (void)hipStreamDestroy(h->stream); }
    drop_graph(h);
    h->stream = (hipStream_t)s;
    h->own_stream = false;
    return 0;
}

int hpv_set_quadrature(hpv_handle h, const double* xi, const double* wx, int qx, const double* yi, const double* wy, int qy) {
    if (!h) return -1;
    if (!xi || !wx || qx < 1) return fail(h, -1, "bad x quadrature");
    if (h->dim == 1) { if (qy != 1) return fail(h, -1, "qy must be 1 for the 1-D problem"); }
    else if (!yi || !wy || qy < 1) return fail(h, -1, "bad y quadrature");
    h->xi.assign(xi, xi + qx); h->wx.assign(wx, wx + qx);
    if (h->dim == 2) { h->yi.assign(yi, yi + qy); h->wy.assign(wy, wy + qy); }
    else { h->yi.assign(1, 0.0); h->wy.assign(1, 1.0); }
    h->qx = qx; h->qy = qy;
    h->pd.qx = qx; h->pd.qy = qy;
    h->have_quad = true;
    drop_graph(h);
    h->have_tables = false;  // weighted tables depend on the weights
    h->have_elems = false;
    return 0;
}

int hpv_set_tables(hpv_handle h, const double* phix, const double* dphix, const double* d2phix, int ntx,
                   const double* phiy, const double* dphiy, const double* d2phiy, int nty, const double* edge_dphi) {
    if (!h) return -1;
    if (!h->have_quad) return fail(h, -3, "call hpv_set_quadrature first");
    if (!phix || !dphix || !d2phix || ntx < 1) return fail(h, -1, "bad x tables");
    if (h->dim == 1) { if (nty != 1) return fail(h, -1, "nty must be 1 for the 1-D problem"); }
    else if (!phiy || !dphiy || !d2phiy || nty < 1) return fail(h, -1, "bad y tables");
    for (size_t i = 0; i < h->nact_all.size(); ++i)     // counts given for an earlier, larger table set
        if (h->nact_all[i] > ntx)
            return fail(h, -1, "n_active[%zu] = %d exceeds the new ntest %d (hpv_set_active_tests(NULL) drops the counts)", i, h->nact_all[i], ntx);
    const int qx = h->qx, qy = h->qy;
    std::vector<double> wtx((size_t)3 * ntx * qx), wty((size_t)3 * nty * qy);
    const double* tx[3] = {phix, dphix, d2phix};
    const double* ty[3] = {phiy, dphiy, d2phiy};
    for (int d = 0; d < 3; ++d)
        for (int r = 0; r < ntx; ++r)
            for (int i = 0; i < qx; ++i) wtx[((size_t)d * ntx + r) * qx + i] = h->wx[i] * tx[d][(size_t)r * qx + i];
    for (int d = 0; d < 3; ++d)
        for (int k = 0; k < nty; ++k)
            for (int j = 0; j < qy; ++j)
                wty[((size_t)d * nty + k) * qy + j] = (h->dim == 1) ? 1.0 : h->wy[j] * ty[d][(size_t)k * qy + j];
    int rc;
    if ((rc = dalloc(h, &h->d_wtx, wtx.size()))) return rc;
    if ((rc = dalloc(h, &h->d_wty, wty.size()))) return rc;
    if ((rc = upload(h, h->d_wtx, wtx.data(), wtx.size()))) return rc;
    if ((rc = upload(h, h->d_wty, wty.data(), wty.size()))) return rc;
    if (h->pd.edge) {
        if (!edge_dphi) return fail(h, -1, "var_form 3 needs edge_dphi");
        if ((rc = dalloc(h, &h->d_edge_dphi, (size_t)2 * ntx))) return rc;
        if ((rc = upload(h, h->d_edge_dphi, edge_dphi, (size_t)2 * ntx))) return rc;
    }
    h->ntx = ntx; h->nty = nty;
    h->pd.ntx = ntx; h->pd.nty = nty;
    if (hpv_proj_lds_bytes(h->pd) > 64 * 1024) return fail(h, -1, "element too large for the projection kernel's LDS (%zu B)", hpv_proj_lds_bytes(h->pd));
    h->have_tables = true;
    drop_graph(h);
    return 0;
}

int hpv_set_elements(hpv_handle h, const double* gridx, int nex, const double* gridy, int ney, int e_begin, int e_end) {
    if (!h) return -1;
    if (!h->have_quad || !h->have_tables) return fail(h, -3, "call hpv_set_quadrature and hpv_set_tables first");
    if (!gridx || nex < 1) return fail(h, -1, "bad x grid");
    if (h->dim == 1) { if (ney != 1) return fail(h, -1, "ney must be 1 for the 1-D problem"); }
    else if (!gridy || ney < 1) return fail(h, -1, "bad y grid");
    const int ne_tot = nex * ney;
    if (e_begin < 0 || e_end > ne_tot || e_begin > e_end) return fail(h, -1, "bad element range [%d,%d) of %d", e_begin, e_end, ne_tot);
    // per-grid inputs given earlier must fit the new grid: checked BEFORE anything of the old grid is replaced
    if (!h->nact_all.empty() && (int)h->nact_all.size() != ne_tot)
        return fail(h, -1, "n_active has %zu entries but the new grid has %d elements (hpv_set_active_tests(NULL) drops them)", h->nact_all.size(), ne_tot);
    if (h->have_F && h->F_all.size() != (size_t)ne_tot * h->ntx * h->nty)
        return fail(h, -1, "F has %zu entries but the new grid needs %zu (hpv_set_rhs(NULL) drops it)", h->F_all.size(), (size_t)ne_tot * h->ntx * h->nty);
    h->nex = nex; h->ney = ney; h->e_begin = e_begin; h->e_end = e_end;
    const long ne = e_end - e_begin;
    h->n_elem = ne;
    const int qx = h->qx, qy = h->qy, NQ = qx * qy;
    const long N = ne * NQ;
    int rc;
    h->Nq = N;
    const int nterms = h->pd.nterms;
    std::vector<double> X((size_t)h->dim * N), coef((size_t)nterms * ne), ecoef((size_t)ne), EX((size_t)2 * ne), jac((size_t)ne);
    for (long le = 0; le < ne; ++le) {
        const int e = e_begin + (int)le;
        const int ex = e / ney, ey = e % ney;
        const double gx0 = gridx[ex], gx1 = gridx[ex + 1];
        double gy0 = 0, gy1 = 0;
        if (h->dim == 2) { gy0 = gridy[ey]; gy1 = gridy[ey + 1]; }
        // affine map of the reference nodes, same expression as P1:69 / P2:75-76 / P3:120-121
        for (int j = 0; j < qy; ++j)
            for (int i = 0; i < qx; ++i) {
                const long p = le * NQ + (long)j * qx + i;
                X[p] = gx0 + (gx1 - gx0) / 2 * (h->xi[i] + 1);
                if (h->dim == 2) X[(size_t)N + p] = gy0 + (gy1 - gy0) / 2 * (h->yi[j] + 1);
            }
        const double Jx = (gx1 - gx0) / 2;
        jac[le] = (h->dim == 1) ? Jx : ((gx1 - gx0) / 2) * ((gy1 - gy0) / 2);   // P1:276 / P2:393
        if (h->cfg.pde == HPV_PDE_POISSON1D) {
            const double J = Jx;                                     // P1:71
            if (h->cfg.var_form == 1) coef[le] = -J;
            else if (h->cfg.var_form == 2) coef[le] = 1.0;
            else { coef[le] = -1 / J; ecoef[le] = 1 / J; EX[2 * le] = gx0; EX[2 * le + 1] = gx1; }
        } else if (h->cfg.pde == HPV_PDE_POISSON2D) {
            const double Jy = (gy1 - gy0) / 2;
            const double J = Jx * Jy;                                // P2:77-79
            if (h->cfg.var_form == 0) coef[le] = J;
            else if (h->cfg.var_form == 1) { coef[le] = -(J / Jx); coef[ne + le] = -(J / Jy); }
            else { coef[le] = J; coef[ne + le] = J; }
        } else {
            const double J = (gy1 - gy0) / 2 * (gx1 - gx0) / 2;      // P3:115
            if (h->cfg.var_form == 0) coef[le] = J;
            else { coef[le] = J; coef[ne + le] = J / Jx; }
        }
    }
    if ((rc = dalloc(h, &h->d_coef, coef.size()))) return rc;
    if ((rc = dalloc(h, &h->d_jac, jac.size()))) return rc;
    if (ne > 0 && (rc = upload(h, h->d_jac, jac.data(), jac.size()))) return rc;
    if ((rc = dalloc(h, &h->d_R, (size_t)ne * h->ntx * h->nty))) return rc;
    h->proj_split = project_row_split(h->pd, ne, h->cfg.backend == HPV_BACKEND_GENERIC);
    // (the tall-element kernel shares an element among up to 64 workgroups, each with its own loss / d-epsilon / partial-sum slot)
    const bool tall = h->pd.qx == 80 && h->pd.qy == 80 && h->pd.ntx == 5 && h->pd.nty == 5 && h->cfg.backend != HPV_BACKEND_GENERIC && ne <= 128;
    const size_t nred = (size_t)ne * std::max(h->proj_split, tall ? 64 : 1);
    h->n_red_alloc = (long)nred;
    if ((rc = dalloc(h, &h->d_loss_e, nred))) return rc;
    if ((rc = dalloc(h, &h->d_deps_e, nred))) return rc;
    if ((rc = dalloc(h, &h->d_upart, h->proj_split > 1 ? (size_t)ne * h->proj_split * h->ntx * h->nty : 0))) return rc;
    h->Xq_host = X;
    h->batch_dirty = true;
    if (N > 0) {
        if ((rc = upload(h, h->d_coef, coef.data(), coef.size()))) return rc;
        HIPCHK(h, hipMemsetAsync(h->d_deps_e, 0, nred * sizeof(double), h->stream));
    }
    if (h->pd.edge) {
        if ((rc = alloc_batch(h, h->edge, h->nd_val, 2 * ne, true))) return rc;
        if ((rc = dalloc(h, &h->d_edge_coef, (size_t)ne))) return rc;
        if (ne > 0) {
            if ((rc = upload(h, h->edge.X, EX.data(), EX.size()))) return rc;
            if ((rc = upload(h, h->d_edge_coef, ecoef.data(), ecoef.size()))) return rc;
        }
    }
    h->have_elems = true;
    drop_graph(h);
    // (re)slice F if it was given before the elements
    if (h->have_F) {
        std::vector<double> F = h->F_all;
        if ((rc = hpv_set_rhs(h, F.data(), F.size()))) return rc;
    } else if (h->d_F) { (void)hipFree(h->d_F); h->d_F = nullptr; }
    if (!h->nact_all.empty()) {   // ... and the active test counts
        std::vector<int> na = h->nact_all;
        if ((rc = hpv_set_active_tests(h, na.data(), (int)na.size()))) return rc;
    }
    if (h->mfma_edge) { hpv_mfma_destroy(h->mfma_edge); h->mfma_edge = nullptr; }
    return 0;
}

// p-refinement of the 1-D driver: element e uses only its first n_active[e] test functions (P1:66-67: Ntest_element =
// len(F_ext_total[e]); P1:268-281 builds F_ext_total from a per-element list N_testfcn_total).  n = elements of the whole grid.
int hpv_set_active_tests(hpv_handle h, const int* n_active, int n) {
    if (!h) return -1;
    drop_graph(h);
    if (!n_active) {
        h->nact_all.clear();
        if (h->d_nact) { (void)hipFree(h->d_nact); h->d_nact = nullptr; }
        h->pd.nact = nullptr;
        return 0;
    }
    if (h->dim != 1) return fail(h, -1, "per-element test-function counts exist in the 1-D problem only (P2:414 / P3:411 reshape F_ext_total)");
    if (!h->have_tables) return fail(h, -3, "call hpv_set_tables first");
    for (int i = 0; i < n; ++i)
        if (n_active[i] < 1 || n_active[i] > h->ntx) return fail(h, -1, "n_active[%d] = %d is outside 1..%d", i, n_active[i], h->ntx);
    if (h->have_elems) {
        if (n != h->nex * h->ney) return fail(h, -1, "n_active has %d entries, expected %d", n, h->nex * h->ney);
        if (h->d_nact) { (void)hipFree(h->d_nact); h->d_nact = nullptr; }
        h->pd.nact = nullptr;
        if (h->n_elem > 0) {
            HIPCHK(h, hipMalloc((void**)&h->d_nact, (size_t)h->n_elem * sizeof(int)));
            HIPCHK(h, hipMemcpyAsync(h->d_nact, n_active + h->e_begin, (size_t)h->n_elem * sizeof(int), hipMemcpyHostToDevice, h->stream));
            HIPCHK(h, hipStreamSynchronize(h->stream));
            h->pd.nact = h->d_nact;
        }
    }
    if (n_active != h->nact_all.data()) h->nact_all.assign(n_active, n_active + n);
    return 0;
}

int hpv_set_rhs(hpv_handle h, const double* F, size_t n) {
    if (!h) return -1;
    drop_graph(h);
    if (!F) { h->have_F = false; h->F_all.clear(); if (h->d_F) { (void)hipFree(h->d_F); h->d_F = nullptr; } return 0; }
    if (h->have_elems) {
        const size_t NR = (size_t)h->ntx * h->nty;
        if (n != (size_t)h->nex * h->ney * NR) return fail(h, -1, "F has %zu entries, expected %zu", n, (size_t)h->nex * h->ney * NR);
        int rc;
        if ((rc = dalloc(h, &h->d_F, (size_t)h->n_elem * NR))) return rc;
        if (h->n_elem > 0 && (rc = upload(h, h->d_F, F + (size_t)h->e_begin * NR, (size_t)h->n_elem * NR))) return rc;
    }
    if (F != h->F_all.data()) h->F_all.assign(F, F + n);
    h->have_F = true;
    return 0;
}

int hpv_set_data(hpv_handle h, const double* X, const double* u, int n) {
    if (!h) return -1;
    if (n < 0 || (n > 0 && (!X || !u))) return fail(h, -1, "bad data arguments");
    int rc;
    h->n_data = n;
    drop_graph(h);
    h->Xd_host.assign(X, X + (size_t)n * h->dim);
    h->batch_dirty = true;
    if ((rc = dalloc(h, &h->d_udata, (size_t)n))) return rc;
    if (n > 0 && (rc = upload(h, h->d_udata, u, (size_t)n))) return rc;
    return 0;
}

int hpv_set_collocation(hpv_handle h, const double* X, const double* f, int n) {
    return hpv_set_collocation_shard(h, X, f, n, (long)n);
}

int hpv_set_collocation_shard(hpv_handle h, const double* X, const double* f, int n, long n_total) {
    if (!h) return -1;
    if (h->cfg.scheme != HPV_SCHEME_PINN) return fail(h, -1, "collocation points belong to scheme PINNs");
    if (n < 1 || !X || !f || n_total < n) return fail(h, -1, "bad collocation arguments");
    drop_graph(h);
    int rc;
    if (h->mfma_colloc) { hpv_mfma_destroy(h->mfma_colloc); h->mfma_colloc = nullptr; }
    if ((rc = alloc_batch(h, h->colloc, h->nd_pinn, n, true))) return rc;
    if ((rc = upload_points(h, h->colloc, X, n, h->dim))) return rc;
    if ((rc = dalloc(h, &h->d_fcol, (size_t)n))) return rc;
    if ((rc = upload(h, h->d_fcol, f, (size_t)n))) return rc;
    if ((rc = dalloc(h, &h->d_col_part, 64))) return rc;
    h->n_col = n;
    h->n_col_total = n_total;
    return 0;
}

size_t hpv_num_params(hpv_handle h) { return h ? (size_t)h->Ptot : 0; }

int hpv_set_params(hpv_handle h, const double* theta, size_t n) {
    if (!h) return -1;
    if (!theta || n != (size_t)h->Ptot) return fail(h, -1, "theta has %zu entries, expected %d", n, h->Ptot);
    int rc;
    if ((rc = upload(h, h->d_theta, theta, n))) return rc;
    HIPCHK(h, hipMemsetAsync(h->d_m, 0, n * sizeof(double), h->stream));
    HIPCHK(h, hipMemsetAsync(h->d_v, 0, n * sizeof(double), h->stream));
    std::vector<double> st((size_t)adam_state_doubles(h->P));
    for (size_t i = 0; i < st.size(); i += 2) { st[i] = h->cfg.beta1; st[i + 1] = h->cfg.beta2; }   // beta^1, every copy
    if ((rc = upload(h, h->d_state, st.data(), st.size()))) return rc;
    h->have_params = true;
    return 0;
}

int hpv_get_params(hpv_handle h, double* theta, size_t n) {
    if (!h) return -1;
    if (!theta || n != (size_t)h->Ptot) return fail(h, -1, "theta buffer has %zu entries, expected %d", n, h->Ptot);
    HIPCHK(h, hipMemcpyAsync(theta, h->d_theta, n * sizeof(double), hipMemcpyDeviceToHost, h->stream));
    HIPCHK(h, hipStreamSynchronize(h->stream));
    return 0;
}

int hpv_forward_backward(hpv_handle h) { return h ? enqueue_pass(h, true) : -1; }
int hpv_eval_loss(hpv_handle h) { return h ? enqueue_pass(h, false) : -1; }

int hpv_reduce_buffer(hpv_handle h, void** dev_ptr, size_t* n_doubles) {
    if (!h || !dev_ptr || !n_doubles) return -1;
    *dev_ptr = h->d_RB;
    *n_doubles = (size_t)h->Ptot + 4;
    return 0;
}

int hpv_apply_adam(hpv_handle h) {
    if (!h) return -1;
    launch_adam(adam_args(h), h->d_RB, h->P, h->Ptot, h->stream);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return fail(h, -2, "adam launch failed: %s", hipGetErrorString(e));
    h->nupd_host += 1;
    return 0;
}

int hpv_sync(hpv_handle h) {
    if (!h) return -1;
    HIPCHK(h, hipStreamSynchronize(h->stream));
    return sync_check(h);     // (the caller-driven multi-GPU pieces -- forward_backward / apply_adam -- end their runs here)
}

int hpv_read_loss(hpv_handle h, double* loss3) {
    if (!h || !loss3) return -1;
    double t[4];
    HIPCHK(h, hipMemcpyAsync(t, h->d_RB + h->Ptot, 4 * sizeof(double), hipMemcpyDeviceToHost, h->stream));
    HIPCHK(h, hipStreamSynchronize(h->stream));
    loss3[0] = t[0] + t[1];
    loss3[1] = (h->cfg.pde == HPV_PDE_ADVDIFF) ? t[1] : t[2];  // P3:184 folds the weight into lossb
    loss3[2] = t[0];
    return 0;
}

int hpv_loss_and_grad(hpv_handle h, double* loss3, double* grad) {
    if (!h) return -1;
    int rc = enqueue_pass_x(h, grad != nullptr, false);
    if (rc) return rc;
    if (loss3 && (rc = hpv_read_loss(h, loss3))) return rc;
    if (grad) {
        HIPCHK(h, hipMemcpyAsync(grad, h->d_RB, (size_t)h->Ptot * sizeof(double), hipMemcpyDeviceToHost, h->stream));
    }
    HIPCHK(h, hipStreamSynchronize(h->stream));
    if (grad && (rc = sync_check(h))) return rc;     // a gradient from a failed SPLIT-mode barrier is not a gradient
    return p2p_check(h);
}

// n_iters training iterations enqueued on the handle's stream (graph replays where possible), no synchronisation
static int enqueue_iterations(hpv_ctx* h, int n_iters) {
    int rc;
    // one-workgroup grids whose kernel finishes the iteration itself: the remaining iterations in ONE persistent launch
    // (k_iter_tile<.., PERSIST>; no graph needed -- there is one launch)
#ifdef HPV_EXPERIMENTS
    const bool persist_on = getenv("HPV_PERSIST") && getenv("HPV_PERSIST")[0] == '1';     // opt-in (kernels_tile.hip, tile_body)
#else
    constexpr bool persist_on = false;      // (measured no faster: the persistent launch exists in libhpvpinn_testhooks.so only)
#endif
    if (persist_on && !h->persist_probed && n_iters > 1 && !h->rccl_on && !h->p2p_on && h->cfg.scheme == HPV_SCHEME_VPINN) {
        // (the first training pass of a handle tells whether its grid is one such workgroup: one eager iteration)
        if ((rc = enqueue_pass(h, true, true))) return rc;
        h->persist_probed = true;
        h->nupd_host += 1;
        n_iters -= 1;
    }
    if (persist_on && h->persist_seen && n_iters > 1 && !h->rccl_on && !h->p2p_on && !h->timing && h->cfg.scheme == HPV_SCHEME_VPINN) {
        h->persist_want = n_iters;
        h->persist_done = 1;
        rc = enqueue_pass(h, true, true);
        h->persist_want = 1;
        if (rc) return rc;
        const int did = h->persist_seen ? h->persist_done : 1;
        h->nupd_host += did;
        if (did >= n_iters) return 0;
        n_iters -= did;
    }
    if (h->use_graph && h->own_stream && !h->timing && n_iters > 0 && h->cfg.scheme == HPV_SCHEME_VPINN) {
        if ((rc = check_ready(h))) return rc;
        // n = a * HPV_GRAPH_ITERS + r: a replays of the K-iteration graph and ONE replay of an r-iteration graph (captured
        // the first time a call leaves that remainder -- callers that time a call run it once untimed before)
        static_assert(HPV_GRAPH_ITERS <= 8, "g_rem size");
        if (n_iters >= HPV_GRAPH_ITERS && !h->g_stepK && (rc = build_step_graph(h, HPV_GRAPH_ITERS, &h->g_stepK))) {
            if (!h->rccl_on) return rc;
            // a collective that refuses stream capture must not stop the run: eager launches from here on
            h->use_graph = false;
            h->err.clear();
            return enqueue_iterations(h, n_iters);
        }
        int it = 0;
        for (; it + HPV_GRAPH_ITERS <= n_iters; it += HPV_GRAPH_ITERS) HIPCHK(h, hipGraphLaunch(h->g_stepK, h->stream));
        const int rem = n_iters - it;
        if (rem > 0) {
            if (!h->g_rem[rem] && (rc = build_step_graph(h, rem, &h->g_rem[rem]))) {
                if (!h->rccl_on) { h->nupd_host += it; return rc; }
                h->use_graph = false;
                h->err.clear();
                h->nupd_host += it;           // (the iterations already launched through g_stepK: advisor, round 3)
                return enqueue_iterations(h, rem);
            }
            HIPCHK(h, hipGraphLaunch(h->g_rem[rem], h->stream));
        }
    } else {
        h->defer_adam = true;
        for (int it = 0; it < n_iters; ++it) {
            if ((rc = enqueue_pass_x(h, true, true))) { h->adam_pending = false; h->defer_adam = false; return rc; }
            h->nupd_host += 1;            // (counted per enqueued iteration: an error exit leaves the count right -- a pending update is
                                          //  either stored by the failing pass's k_finalize or applied by enqueue_pass's error exit; the
                                          //  handle is in a failed state afterwards: hpv_last_error says why)
        }
        flush_adam(h);
        return 0;
    }
    h->nupd_host += n_iters;
    return 0;
}

// after a synchronisation point: did a SPLIT-mode element barrier time out (on this rank, or -- carried by the pad slot of the
// all-reduced buffer -- on any rank)?  The kernels have left theta, m, v, the beta powers and the loss history as they were
// before the failing iteration and have ignored every iteration since (sticky flag); here the failure is reported (-7), the
// flag cleared (the exchange is stateless: tagged granules), so that the caller may go on (e.g. with HPV_FUSE=s on a shared GPU).
static int sync_check(hpv_ctx* h) {
    if (!h->d_xerr || !h->mfma) return 0;
    if (!hpv_mfma_split_used(h->mfma) && !h->rccl_on && !h->p2p_on) return 0;   // nothing can have set it
    int err = 0;
    HIPCHK(h, hipMemcpyAsync(&err, h->d_xerr, sizeof(int), hipMemcpyDeviceToHost, h->stream));
    HIPCHK(h, hipStreamSynchronize(h->stream));
    if (!err) return 0;
    HIPCHK(h, hipMemsetAsync(h->d_xerr, 0, sizeof(int), h->stream));
    HIPCHK(h, hipStreamSynchronize(h->stream));
    return fail(h, -7, "whole-iteration kernel (SPLIT mode): the workgroups sharing an element did not meet at their barrier "
                       "(timeout) on some rank; the failing iteration and every later one of this call were NOT applied -- "
                       "parameters and optimizer state are those before it.  Is the GPU shared with another process?  HPV_FUSE=s "
                       "selects the barrier-free kernels");
}

// A run of `n` enqueued iterations ended with -7.  The device counter says how many of them were applied; unless the caller
// opted out (HPV_EXCHANGE_FALLBACK=0) the handle is switched to the launch structures without an in-kernel exchange and the
// caller finishes the run on those.  Handles connected to other ranks (in-library RCCL / mailbox exchange) take the same
// decision without talking to each other: the failure flag travels with the all-reduced buffer of the failing iteration, so
// every rank's k_adam / exchange kernel skips from the same iteration on, every rank's hpv_step sees -7 in the same call with
// the same count, and every rank enqueues the same number of remaining iterations (collectives stay matched).
// Returns the number of iterations that took place, or -1: report the -7.
static int after_exchange_timeout(hpv_ctx* h, int n) {
    long long dev = 0;
    if (hpv_updates_applied(h, &dev)) return -1;
    const long long done = dev - (h->nupd_host - n);
    h->nupd_host = dev;
    const char* e = getenv("HPV_EXCHANGE_FALLBACK");
    if ((e && e[0] == '0') || !h->shared_elem_ok || done < 0 || done > n) return -1;
    if (hpv_set_shared_element_kernels(h, 0)) return -1;
    if (h->n_fallbacks++ == 0)
        fprintf(stderr, "libhpvpinn: an in-kernel exchange between the workgroups of one element timed out after %lld of %d "
                        "iterations (is the GPU shared?); continuing on the launch structures without an exchange "
                        "(HPV_EXCHANGE_FALLBACK=0: return -7 instead)\n", done, n);
    h->err.clear();
    return (int)done;
}


int hpv_step(hpv_handle h, int n_iters, double* loss3_after) {
    if (!h) return -1;
    int rc;
    for (int left = n_iters, round = 0;; ++round) {
        if ((rc = enqueue_iterations(h, left))) return rc;
        if (loss3_after) {
            if ((rc = enqueue_pass_x(h, false, false))) return rc;
            if ((rc = hpv_read_loss(h, loss3_after))) return rc;
        } else {
            HIPCHK(h, hipStreamSynchronize(h->stream));
        }
        if ((rc = sync_check(h)) != -7 || round) break;
        const int done = after_exchange_timeout(h, left);      // the rest of the run on the barrier-free structures
        if (done < 0) break;
        left -= done;
    }
    if (rc) return rc;
    return p2p_check(h);
}

// ---- loss history (the reference records the loss after every update with a second forward pass, P2:243-244; the
//      forward pass of the NEXT iteration computes exactly that value, so the device keeps it) ----
static void loss_triple(hpv_ctx* h, const double* raw, double* loss3) {
    loss3[0] = raw[0] + raw[1];
    loss3[1] = (h->cfg.pde == HPV_PDE_ADVDIFF) ? raw[1] : raw[2];  // P3:184 folds the weight into lossb
    loss3[2] = raw[0];
}

int hpv_history_reset(hpv_handle h) {
    if (!h) return -1;
    HIPCHK(h, hipMemsetAsync(h->d_hist_idx, 0, sizeof(int), h->stream));
    return 0;
}

int hpv_history_read(hpv_handle h, int n, double* loss3_hist, double* eps_hist) {
    if (!h || !loss3_hist || n < 0) return -1;
    if (n > HPV_HIST_CAP) return fail(h, -1, "history holds %d entries, %d requested", HPV_HIST_CAP, n);
    int have = 0;
    std::vector<double> raw((size_t)4 * n);
    HIPCHK(h, hipMemcpyAsync(&have, h->d_hist_idx, sizeof(int), hipMemcpyDeviceToHost, h->stream));
    if (n) HIPCHK(h, hipMemcpyAsync(raw.data(), h->d_hist, raw.size() * sizeof(double), hipMemcpyDeviceToHost, h->stream));
    HIPCHK(h, hipStreamSynchronize(h->stream));
    if (have < n) return fail(h, -3, "only %d training iterations since hpv_history_reset, %d requested", have, n);
    for (int i = 0; i < n; ++i) {
        loss_triple(h, &raw[(size_t)4 * i], loss3_hist + 3 * i);
        if (eps_hist) eps_hist[i] = raw[(size_t)4 * i + 3];
    }
    return 0;
}

int hpv_time_iteration_kernel(hpv_handle h, int reps, double* avg_ms) {
    if (!h || reps < 1 || !avg_ms) return -1;
    int rc = check_ready(h);
    if (rc) return rc;
    if (h->cfg.scheme == HPV_SCHEME_PINN || !(h->mfma && h->backend == HPV_BACKEND_MFMA) || h->var.N <= 0)
        return fail(h, -4, "no whole-iteration kernel on this handle");
    if ((rc = enqueue_pass(h, true))) return rc;        // everything a first pass allocates / selects
    HIPCHK(h, hipStreamSynchronize(h->stream));
    if ((rc = sync_check(h))) return rc;
    if (h->pass_structure != 2) return fail(h, -4, "the handle's iteration is not ONE whole-iteration launch without an in-kernel exchange");
    const MfmaDataTerm dt = pass_data_term(h, true);
    const ProjArgs pa = pass_proj_args(h, true);
    hipEvent_t e0 = nullptr, e1 = nullptr;
    HIPCHK(h, hipEventCreate(&e0));
    HIPCHK(h, hipEventCreate(&e1));
    bool ok = true;
    (void)hipEventRecord(e0, h->stream);
    for (int i = 0; i < reps && ok; ++i)
        ok = hpv_mfma_iter_fused(h->mfma, h->d_theta, h->var.X, h->var.GPART, &h->var.rows, h->stream, &dt, pa, h->n_elem);
    (void)hipEventRecord(e1, h->stream);
    hipError_t e = hipStreamSynchronize(h->stream);
    float ms = 0.0f;
    if (e == hipSuccess) e = hipEventElapsedTime(&ms, e0, e1);
    (void)hipEventDestroy(e0);
    (void)hipEventDestroy(e1);
    if (!ok) return fail(h, -4, "the whole-iteration kernel declined the launch");
    if (e != hipSuccess) return fail(h, -2, "timing the iteration kernel failed: %s", hipGetErrorString(e));
    *avg_ms = (double)ms / reps;
    return 0;
}

int hpv_step_record(hpv_handle h, int n_iters, double* loss3_hist, double* eps_hist) {
    if (!h || !loss3_hist || n_iters < 0) return -1;
    int rc;
    std::vector<double> chunk((size_t)3 * HPV_HIST_CAP), ceps(HPV_HIST_CAP);
    for (int done = 0; done < n_iters;) {
        const int c = std::min(HPV_HIST_CAP, n_iters - done);
        if ((rc = hpv_history_reset(h))) return rc;
        if ((rc = enqueue_iterations(h, c))) return rc;
        HIPCHK(h, hipStreamSynchronize(h->stream));
        int got = c;       // (a skipped iteration records nothing: after a timeout the history holds the `got` that took place)
        if ((rc = sync_check(h)) == -7 && (got = after_exchange_timeout(h, c)) >= 0) rc = 0;
        if (rc || (rc = p2p_check(h))) return rc;
        if (got && (rc = hpv_history_read(h, got, chunk.data(), ceps.data()))) return rc;
        // entry j was computed by the forward pass that preceded update done+j+1, i.e. it belongs to the state after update done+j
        for (int j = 0; j < got; ++j)
            if (done + j >= 1) {
                std::copy_n(&chunk[(size_t)3 * j], 3, loss3_hist + (size_t)3 * (done + j - 1));
                if (eps_hist) eps_hist[done + j - 1] = ceps[j];
            }
        done += got;
    }
    if (n_iters > 0) {   // the state after the last update: one forward pass
        if ((rc = enqueue_pass_x(h, false, false))) return rc;
        if ((rc = hpv_read_loss(h, loss3_hist + (size_t)3 * (n_iters - 1)))) return rc;
        if (eps_hist) {
            eps_hist[n_iters - 1] = 0.0;
            if (h->has_eps) {
                HIPCHK(h, hipMemcpyAsync(eps_hist + n_iters - 1, h->d_theta + h->P, sizeof(double), hipMemcpyDeviceToHost, h->stream));
                HIPCHK(h, hipStreamSynchronize(h->stream));
            }
        }
    }
    return 0;
}

int hpv_predict(hpv_handle h, const double* X, int n, double* u_out) {
    if (!h) return -1;
    if (!h->have_params) return fail(h, -3, "hpv_set_params has not been called");
    if (n < 0 || (n > 0 && (!X || !u_out))) return fail(h, -1, "bad predict arguments");
    if (n == 0) return 0;
    int rc;
    if (h->pred.N != n) {
        if ((rc = alloc_batch(h, h->pred, h->nd_val, n, false))) return rc;
        if (h->mfma_pred) { hpv_mfma_destroy(h->mfma_pred); h->mfma_pred = nullptr; }
        if (h->cfg.backend != HPV_BACKEND_GENERIC) h->mfma_pred = hpv_mfma_create(h->nd_val, n, nullptr, false);
    }
    if ((rc = upload_points(h, h->pred, X, n, h->dim))) return rc;
    if (h->mfma_pred) hpv_mfma_forward(h->mfma_pred, h->d_theta, h->pred.X, h->pred.OUT, 0, h->stream);
    else launch_mlp_fwd_generic(h->pred.nd, h->d_theta, h->pred.X, nullptr, h->pred.OUT, n, 0, h->stream);
    HIPCHK(h, hipGetLastError());
    HIPCHK(h, hipMemcpyAsync(u_out, h->pred.OUT, (size_t)n * sizeof(double), hipMemcpyDeviceToHost, h->stream));
    HIPCHK(h, hipStreamSynchronize(h->stream));
    return 0;
}

// [theta | m | v | beta1^t | beta2^t]: everything a bit-exact resume needs.
int hpv_get_state(hpv_handle h, double* buf, size_t n) {
    if (!h || !buf) return -1;
    const size_t P = (size_t)h->Ptot;
    if (n != 3 * P + 2) return fail(h, -1, "state buffer has %zu entries, expected %zu", n, 3 * P + 2);
    HIPCHK(h, hipMemcpyAsync(buf, h->d_theta, P * sizeof(double), hipMemcpyDeviceToHost, h->stream));
    HIPCHK(h, hipMemcpyAsync(buf + P, h->d_m, P * sizeof(double), hipMemcpyDeviceToHost, h->stream));
    HIPCHK(h, hipMemcpyAsync(buf + 2 * P, h->d_v, P * sizeof(double), hipMemcpyDeviceToHost, h->stream));
    HIPCHK(h, hipMemcpyAsync(buf + 3 * P, h->d_state, 2 * sizeof(double), hipMemcpyDeviceToHost, h->stream));
    HIPCHK(h, hipStreamSynchronize(h->stream));
    return 0;
}

int hpv_set_state(hpv_handle h, const double* buf, size_t n) {
    if (!h || !buf) return -1;
    const size_t P = (size_t)h->Ptot;
    if (n != 3 * P + 2) return fail(h, -1, "state buffer has %zu entries, expected %zu", n, 3 * P + 2);
    int rc;
    if ((rc = upload(h, h->d_theta, buf, P))) return rc;
    if ((rc = upload(h, h->d_m, buf + P, P))) return rc;
    if ((rc = upload(h, h->d_v, buf + 2 * P, P))) return rc;
    std::vector<double> st((size_t)adam_state_doubles(h->P));
    for (size_t i = 0; i < st.size(); i += 2) { st[i] = buf[3 * P]; st[i + 1] = buf[3 * P + 1]; }
    if ((rc = upload(h, h->d_state, st.data(), st.size()))) return rc;
    h->have_params = true;
    return 0;
}

// F_ext[e][k][r] = J_e sum_q w_x phi_r w_y phi_k f(x_q)  (P1:289, P2:405-407) for the owned elements, from
// the values of f at this handle's quadrature points (element-major, q = j*qx+i): the driver-side RHS
// assembly done by the projection kernel instead of the reference's Python loops.
int hpv_assemble_rhs(hpv_handle h, const double* f_quad, size_t n, double* F_out, size_t n_out) {
    if (!h || !f_quad || !F_out) return -1;
    if (!h->have_quad || !h->have_tables || !h->have_elems) return fail(h, -3, "set quadrature, tables and elements first");
    const long NQ = (long)h->qx * h->qy, NR = (long)h->ntx * h->nty, ne = h->n_elem;
    if (n != (size_t)(ne * NQ) || n_out != (size_t)(ne * NR)) return fail(h, -1, "f has %zu / F has %zu entries, expected %ld / %ld", n, n_out, ne * NQ, ne * NR);
    if (ne == 0) return 0;
    double *d_f = nullptr, *d_F = nullptr, *d_le = nullptr;
    int rc = 0;
    rc |= dalloc(h, &d_f, n); rc |= dalloc(h, &d_F, n_out); rc |= dalloc(h, &d_le, (size_t)ne);
    if (!rc) rc = upload(h, d_f, f_quad, n);
    if (!rc) {
        ProjDesc pd{};
        pd.nterms = 1; pd.t[0].dx = 0; pd.t[0].dy = 0; pd.t[0].a0[0] = 1.0;
        pd.qx = h->qx; pd.qy = h->qy; pd.ntx = h->ntx; pd.nty = h->nty; pd.C = 1;
        if (h->cfg.backend == HPV_BACKEND_GENERIC ||
            !launch_project_tp(pd, d_f, nullptr, d_F, nullptr, h->d_jac, ne, h->d_wtx, h->d_wty, nullptr, d_le, nullptr, ne * NQ, ne, 0, h->stream))
            launch_project(pd, d_f, nullptr, d_F, nullptr, h->d_jac, ne, h->d_wtx, h->d_wty, nullptr, d_le, nullptr, ne * NQ, ne, 0,
                           nullptr, nullptr, nullptr, nullptr, h->stream);
        hipError_t e = hipMemcpyAsync(F_out, d_F, n_out * sizeof(double), hipMemcpyDeviceToHost, h->stream);
        if (e == hipSuccess) e = hipStreamSynchronize(h->stream);
        if (e != hipSuccess) rc = fail(h, -2, "hpv_assemble_rhs failed: %s", hipGetErrorString(e));
    }
    double* ptrs[] = {d_f, d_F, d_le};
    for (double* p : ptrs) if (p) (void)hipFree(p);
    return rc;
}

// Table generation on the device: the Gauss-Lobatto-Legendre rule of the drivers
// (GaussLobattoJacobiWeights(Q, 0, 0): P1:312, P2:355, P3:395) and the test-function tables (Test_fcn / dTest_fcn).
int hpv_gll_rule(hpv_handle h, int q, double* xi, double* w) {
    if (!h || !xi || !w || q < 2) return -1;
    double* d = nullptr;
    int rc = dalloc(h, &d, (size_t)2 * q);
    if (rc) return rc;
    launch_gll_rule(q, d, d + q, h->stream);
    hipError_t e = hipMemcpyAsync(xi, d, (size_t)q * sizeof(double), hipMemcpyDeviceToHost, h->stream);
    if (e == hipSuccess) e = hipMemcpyAsync(w, d + q, (size_t)q * sizeof(double), hipMemcpyDeviceToHost, h->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(h->stream);
    (void)hipFree(d);
    if (e != hipSuccess) return fail(h, -2, "hpv_gll_rule failed: %s", hipGetErrorString(e));
    return 0;
}

int hpv_test_tables(hpv_handle h, int ntest, const double* xi, int q, double* tab) {
    if (!h || !xi || !tab || ntest < 1 || q < 1) return -1;
    double* d = nullptr;
    const size_t nt = (size_t)3 * ntest * q;
    int rc = dalloc(h, &d, nt + q);
    if (rc) return rc;
    rc = upload(h, d + nt, xi, (size_t)q);
    if (!rc) {
        launch_test_tables(ntest, q, d + nt, d, h->stream);
        hipError_t e = hipMemcpyAsync(tab, d, nt * sizeof(double), hipMemcpyDeviceToHost, h->stream);
        if (e == hipSuccess) e = hipStreamSynchronize(h->stream);
        if (e != hipSuccess) rc = fail(h, -2, "hpv_test_tables failed: %s", hipGetErrorString(e));
    }
    (void)hipFree(d);
    return rc;
}


int hpv_get_residuals(hpv_handle h, double* R, size_t n) {
    if (!h || !R) return -1;
    const size_t want = (size_t)h->n_elem * h->ntx * h->nty;
    if (n != want) return fail(h, -1, "R buffer has %zu entries, expected %zu", n, want);
    HIPCHK(h, hipMemcpyAsync(R, h->d_R, n * sizeof(double), hipMemcpyDeviceToHost, h->stream));
    HIPCHK(h, hipStreamSynchronize(h->stream));
    return 0;
}

int hpv_eval_channels(hpv_handle h, double* out, size_t n) {
    if (!h || !out) return -1;
    if (h->cfg.scheme != HPV_SCHEME_VPINN) return fail(h, -1, "hpv_eval_channels belongs to the variational scheme");
    int rc = check_ready(h);
    if (rc) return rc;
    const int C = h->nd_eval.C;
    if (n != (size_t)C * h->Nq) return fail(h, -1, "channel buffer has %zu entries, expected %zu", n, (size_t)C * h->Nq);
    if (h->Nq == 0) return 0;
    const double* src = h->var.OUT;
    if (h->eval_differs) {
        // the training pass runs on a reduced channel set: the reference's list through a forward-only launch of its own
        if (h->eval_N != h->var.N) {
            if (h->mfma_eval) { hpv_mfma_destroy(h->mfma_eval); h->mfma_eval = nullptr; }
            if ((rc = dalloc(h, &h->d_eval_out, (size_t)C * h->var.N))) return rc;
            if (h->backend == HPV_BACKEND_MFMA) h->mfma_eval = hpv_mfma_create(h->nd_eval, h->var.N, nullptr, false);
            h->eval_N = h->var.N;
        }
        if (h->mfma_eval) hpv_mfma_forward(h->mfma_eval, h->d_theta, h->var.X, h->d_eval_out, 0, h->stream);
        else launch_mlp_fwd_generic(h->nd_eval, h->d_theta, h->var.X, nullptr, h->d_eval_out, h->var.N, 0, h->stream);
        src = h->d_eval_out;
    } else if (h->mfma && h->backend == HPV_BACKEND_MFMA) {
        hpv_mfma_forward(h->mfma, h->d_theta, h->var.X, h->var.OUT, 0, h->stream);
    } else {
        run_fwd(h, h->var, nullptr, 0);
    }
    HIPCHK(h, hipGetLastError());
    for (int ch = 0; ch < C; ++ch)     // the device batch may carry padding / data points behind the quadrature points
        HIPCHK(h, hipMemcpyAsync(out + (size_t)ch * h->Nq, src + (size_t)ch * h->var.N, (size_t)h->Nq * sizeof(double),
                                 hipMemcpyDeviceToHost, h->stream));
    HIPCHK(h, hipStreamSynchronize(h->stream));
    return 0;
}

#if defined(HPV_FZ_TIMING) || defined(HPV_PJ_TIMING)
// (debug hook of the timing builds only -- scripts/fz_timing.py, scripts/pj_timing.py: raw read of the adjoint channel buffer /
//  the activation store the instrumented kernels stamp)
int hpv_debug_read_out(hpv_handle h, double* out, size_t n) {
    if (!h || !out || !h->var.GBAR) return -1;
    const double* src = h->var.GBAR;
    size_t have = (size_t)h->nd_var.C * (size_t)h->var.N;
    if (getenv("HPV_DEBUG_READ_STORE") && h->mfma && hpv_mfma_activation_store(h->mfma)) {
        src = hpv_mfma_activation_store(h->mfma);
        have = hpv_mfma_activation_store_doubles(h->mfma);
    }
    if (getenv("HPV_DEBUG_READ_CHANNELS") && h->var.OUT) { src = h->var.OUT; have = (size_t)h->nd_var.C * (size_t)h->var.N; }
    if (n > have) return fail(h, -1, "debug read of %zu doubles from a buffer of %zu", n, have);
    return hipMemcpy(out, src, n * sizeof(double), hipMemcpyDeviceToHost) == hipSuccess ? 0 : -2;
}
#endif
int hpv_pass_structure(hpv_handle h) { return h ? h->pass_structure : -1; }
int hpv_kernel_variant(hpv_handle h, char* buf, size_t n) {
    if (!h || !buf || n == 0) return -1;
    snprintf(buf, n, "%s", h->variant);
    return 0;
}
const char* hpv_build_info(void) {
    static std::string info;
    static std::once_flag once;
    std::call_once(once, [] {
        info = std::string("k_iter_fused=") + hpv_fused_build_state() + ";k_iter_fused_gen=" + hpv_fused_gen_build_state() +
               ";k_iter_tall=" + hpv_tall_build_state() + ";test_hooks=";
#ifdef HPV_TEST_HOOKS
        info += "1";
#else
        info += "0";
#endif
        info += ";experiments=";
#ifdef HPV_EXPERIMENTS
        info += "1";
#else
        info += "0";
#endif
    });
    return info.c_str();
}
int hpv_rule_advice(int device, int dim, int q, int ntx, int nty, long n_elem_shard, int exact_counts, int n_hidden, int* q_dev, int* nt_dev) {
    if (!q_dev || !nt_dev || (dim != 1 && dim != 2) || q < 1 || ntx < 1 || n_elem_shard < 0) return -1;
    *q_dev = q; *nt_dev = ntx;
    int n_cus = 256;
    hipDeviceProp_t prop;
    if (device >= 0 && hipGetDeviceProperties(&prop, device) == hipSuccess && prop.multiProcessorCount > 0) n_cus = prop.multiProcessorCount;
    else (void)hipGetLastError();
    if (dim == 1) {                                   // kernels_tile.hip: 80 points / 60 test functions (P1:237-238)
        if (q > 80 || ntx > 60) return 0;
        if (q == 80) { *nt_dev = 60; return 0; }      // fewer test functions: the first 60, per-element counts
        if (n_elem_shard <= hpv_rule1d_pad_max(q, n_cus) && n_elem_shard <= hpv_elem_resident_max(1, 80, n_cus)) { *q_dev = 80; *nt_dev = 60; }
        return 0;
    }
    static const int rules[4][2] = {{10, 5}, {12, 6}, {16, 8}, {20, 10}};     // kernels_fused.hip (k_iter_small, FZ_SHAPES)
    for (const auto& r : rules) {
        const bool counts_ok = exact_counts ? (ntx == r[1] && nty == r[1]) : (ntx <= r[1] && nty <= r[1]);
        if (q <= r[0] && counts_ok) {
            // (pad only while ONE WORKGROUP PER ELEMENT of that rule's kernel would take the shard -- the plan evaluated with the
            //  network's depth and the element loop as built, exactly as launch_iter_fused evaluates it: where the loop (plan 2) or
            //  the separate launches (plan 0) take the grid, the padded points cost more than the structure saves; advisor, round 5)
            const int Lh = n_hidden > 0 ? n_hidden : 3;
            const bool takes = r[0] == 10 ? n_elem_shard <= hpv_elem_resident_max(2, 10, n_cus)
                                          : (hpv_fused_grid_plan(r[0], Lh, n_elem_shard, n_cus, hpv_fused_loop_built()) | 2) == 3;      // plans 1 and 3
            if (q < r[0] && takes) *q_dev = r[0];
            break;
        }
    }
    return 0;
}
int hpv_grid_plan(int device, int q, int n_hidden, long n_elem_shard) {
    if ((q != 12 && q != 16 && q != 20) || n_hidden < 2 || n_hidden > 3 || n_elem_shard < 1) return -1;
    int n_cus = 256;
    hipDeviceProp_t prop;
    if (device >= 0 && hipGetDeviceProperties(&prop, device) == hipSuccess && prop.multiProcessorCount > 0) n_cus = prop.multiProcessorCount;
    else (void)hipGetLastError();
    return hpv_fused_grid_plan(q, n_hidden, n_elem_shard, n_cus, hpv_fused_loop_built());
}
int hpv_updates_applied(hpv_handle h, long long* n) {
    if (!h || !n) return -1;
    unsigned long long v = 0;
    HIPCHK(h, hipMemcpyAsync(&v, h->d_nupd, sizeof(v), hipMemcpyDeviceToHost, h->stream));
    HIPCHK(h, hipStreamSynchronize(h->stream));
    *n = (long long)v;
    return 0;
}
int hpv_shared_element_kernels(hpv_handle h) { return !h ? -1 : (h->shared_elem_ok ? 1 : 0); }
int hpv_set_shared_element_kernels(hpv_handle h, int on) {
    if (!h) return -1;
    HIPCHK(h, hipStreamSynchronize(h->stream));
    h->shared_elem_ok = on != 0;
    if (h->mfma) hpv_mfma_set_split_ok(h->mfma, h->shared_elem_ok);
    drop_graph(h);      // captured iterations hold the launch structure chosen before
    return 0;
}
int hpv_graphs_in_use(hpv_handle h) { return !h ? -1 : ((h->use_graph && h->own_stream && (h->g_stepK || h->g_rem[1] || h->g_rem[2] || h->g_rem[3] || h->g_rem[4] || h->g_rem[5] || h->g_rem[6] || h->g_rem[7])) ? 1 : 0); }
int hpv_backend_in_use(hpv_handle h) {
    if (!h) return -1;
    if (h->cfg.scheme == HPV_SCHEME_VPINN && h->have_quad && h->have_tables && h->have_elems && h->batch_dirty) {
        int rc = assemble_batches(h);   // decides the backend; surfaces "MFMA not available" early
        if (rc) return rc;
    }
    return h->backend;
}

}  // extern "C"
