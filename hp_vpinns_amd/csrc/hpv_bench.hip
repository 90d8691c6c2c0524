// Timing, benchmark and debug hooks of the C-ABI (include/hpvpinn.h, "introspection for tests / benchmarks"): the device
// activation probe, kernel-class timers, the stand-alone projection launches of the HBM-roofline measurement.  Split out of
// hpv_api.hip in round 4.
#include "hpv_ctx.h"

using namespace hpvd;

extern "C" {

int hpv_debug_activation(hpv_handle h, const double* x, int n, double* a, double* a1, double* ref) {
    if (!h || !x || !a || !a1 || !ref || n < 1) return -1;
    double* d = nullptr;
    int rc = dalloc(h, &d, (size_t)4 * n);
    if (rc) return rc;
    rc = upload(h, d, x, (size_t)n);
    if (!rc) {
        launch_debug_act(h->cfg.act, d, n, d + n, d + 2 * (size_t)n, d + 3 * (size_t)n, h->stream);
        hipError_t e = hipMemcpyAsync(a, d + n, (size_t)n * sizeof(double), hipMemcpyDeviceToHost, h->stream);
        if (e == hipSuccess) e = hipMemcpyAsync(a1, d + 2 * (size_t)n, (size_t)n * sizeof(double), hipMemcpyDeviceToHost, h->stream);
        if (e == hipSuccess) e = hipMemcpyAsync(ref, d + 3 * (size_t)n, (size_t)n * sizeof(double), hipMemcpyDeviceToHost, h->stream);
        if (e == hipSuccess) e = hipStreamSynchronize(h->stream);
        if (e != hipSuccess) rc = fail(h, -2, "debug_activation failed: %s", hipGetErrorString(e));
    }
    (void)hipFree(d);
    return rc;
}

int hpv_enable_timing(hpv_handle h, int on) {
    if (!h) return -1;
    HIPCHK(h, hipStreamSynchronize(h->stream));
    h->timing = on != 0;
    for (auto& t : h->timers) { t.used = 0; t.total_ms = 0.0; t.launches = 0; }
    return 0;
}

int hpv_kernel_time_ms(hpv_handle h, int which, double* avg_ms, long* launches) {
    if (!h || which < 0 || which > 2) return -1;
    HIPCHK(h, hipStreamSynchronize(h->stream));
    TimerClass& t = h->timers[which];
    for (size_t i = 0; i < t.used; i += 2) { float ms = 0; (void)hipEventElapsedTime(&ms, t.ev[i], t.ev[i + 1]); t.total_ms += ms; }
    t.used = 0;
    if (avg_ms) *avg_ms = t.launches ? t.total_ms / (double)t.launches : 0.0;
    if (launches) *launches = t.launches;
    return 0;
}

int hpv_bench_projection(hpv_handle h, long n_elem, int reps, double* avg_ms, double* bytes_per_launch) {
    return hpv_bench_residual(h, n_elem, reps, 1, avg_ms, bytes_per_launch);
}

static int bench_residual_impl(hpv_handle h, long n_elem, int reps, int do_adjoint, double* avg_ms, double* bytes_per_launch, double* sums);
int hpv_bench_residual(hpv_handle h, long n_elem, int reps, int do_adjoint, double* avg_ms, double* bytes_per_launch) {
    return bench_residual_impl(h, n_elem, reps, do_adjoint, avg_ms, bytes_per_launch, nullptr);
}
int hpv_bench_residual_checksums(hpv_handle h, long n_elem, int do_adjoint, double* sums6) {
    if (!sums6) return -1;
    return bench_residual_impl(h, n_elem, 1, do_adjoint, nullptr, nullptr, sums6);
}
static int bench_residual_impl(hpv_handle h, long n_elem, int reps, int do_adjoint, double* avg_ms, double* bytes_per_launch, double* sums) {
    if (!h) return -1;
    if (!h->have_quad || !h->have_tables) return fail(h, -3, "set quadrature and tables first");
    if (n_elem < 1 || reps < 1) return fail(h, -1, "bad arguments");
    const ProjDesc& pd = h->pd;
    const long NQ = (long)pd.qx * pd.qy, NR = (long)pd.ntx * pd.nty, N = n_elem * NQ;
    const int C = pd.C;
    double *OUT = nullptr, *GB = nullptr, *R = nullptr, *F = nullptr, *coef = nullptr, *le = nullptr, *de = nullptr;
    int rc = 0;
    rc |= dalloc(h, &OUT, (size_t)C * N); rc |= dalloc(h, &GB, (size_t)C * N);
    rc |= dalloc(h, &R, (size_t)n_elem * NR); rc |= dalloc(h, &F, (size_t)n_elem * NR);
    rc |= dalloc(h, &coef, (size_t)pd.nterms * n_elem); rc |= dalloc(h, &le, (size_t)n_elem); rc |= dalloc(h, &de, (size_t)n_elem);
    if (!rc) {
        // deterministic pseudo-random fill on the host in chunks (seeded LCG -> uniform(-1,1))
        std::vector<double> buf((size_t)1 << 20);
        unsigned long long s = 1234;
        auto fill = [&](double* dst, size_t n) {
            for (size_t o = 0; o < n; o += buf.size()) {
                size_t m = std::min(buf.size(), n - o);
                for (size_t i = 0; i < m; ++i) { s = s * 6364136223846793005ULL + 1442695040888963407ULL; buf[i] = (double)(long long)(s >> 11) / 4503599627370496.0 - 1.0; }
                (void)hipMemcpy(dst + o, buf.data(), m * sizeof(double), hipMemcpyHostToDevice);
            }
        };
        fill(OUT, (size_t)C * N); fill(F, (size_t)n_elem * NR);
        std::vector<double> c((size_t)pd.nterms * n_elem, 0.25);
        (void)hipMemcpy(coef, c.data(), c.size() * sizeof(double), hipMemcpyHostToDevice);
        const double* eps_ptr = h->has_eps ? h->d_theta + h->P : nullptr;
        ProjDesc p2 = pd; p2.edge = 0;
        hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
        (void)hipMemset(GB, 0, (size_t)C * N * sizeof(double));
        auto go = [&]() {
            if (h->cfg.backend == HPV_BACKEND_GENERIC ||
                !launch_project_tp(p2, OUT, GB, R, F, coef, n_elem, h->d_wtx, h->d_wty, eps_ptr, le, de, N, n_elem, do_adjoint ? 1 : 0, h->stream))
                launch_project(p2, OUT, GB, R, F, coef, n_elem, h->d_wtx, h->d_wty, eps_ptr, le, de, N, n_elem, do_adjoint ? 1 : 0, nullptr, nullptr, nullptr, nullptr, h->stream);
        };
        go();
        (void)hipEventRecord(e0, h->stream);
        for (int i = 0; i < reps; ++i) go();
        (void)hipEventRecord(e1, h->stream);
        (void)hipStreamSynchronize(h->stream);
        float ms = 0; (void)hipEventElapsedTime(&ms, e0, e1);
        if (avg_ms) *avg_ms = ms / reps;
        (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
        hipError_t e = hipGetLastError();
        if (e != hipSuccess) rc = fail(h, -2, "projection bench failed: %s", hipGetErrorString(e));
        if (!rc && sums) {      // what the launch produced, condensed: sum R, sum R^2, sum loss_e, sum |gbar|, sum gbar^2, sum over e of e-weighted loss
            std::vector<double> hr((size_t)n_elem * NR), hl((size_t)n_elem), hg(do_adjoint ? (size_t)C * N : 0);
            (void)hipMemcpy(hr.data(), R, hr.size() * sizeof(double), hipMemcpyDeviceToHost);
            (void)hipMemcpy(hl.data(), le, hl.size() * sizeof(double), hipMemcpyDeviceToHost);
            if (do_adjoint) (void)hipMemcpy(hg.data(), GB, hg.size() * sizeof(double), hipMemcpyDeviceToHost);
            for (int i = 0; i < 6; ++i) sums[i] = 0.0;
            for (double v : hr) { sums[0] += v; sums[1] += v * v; }
            for (size_t i = 0; i < hl.size(); ++i) { sums[2] += hl[i]; sums[5] += hl[i] * (double)(i % 97); }
            for (double v : hg) { sums[3] += std::fabs(v); sums[4] += v * v; }
        }
    }
    // algorithmic bytes: read the integrated channels + F, write R (8 (C_u N + 2 N_R)); with the adjoint
    // also write the adjoint channels
    int cu = 0;
    for (int ch = 0; ch < C; ++ch) { bool used = false; for (int t = 0; t < pd.nterms; ++t) if (pd.t[t].a0[ch] != 0.0 || pd.t[t].a1[ch] != 0.0) used = true; cu += used; }
    if (bytes_per_launch) *bytes_per_launch = 8.0 * ((do_adjoint ? 2.0 : 1.0) * cu * (double)N + 2.0 * (double)n_elem * NR);
    double* ptrs[] = {OUT, GB, R, F, coef, le, de};
    for (double* p : ptrs) if (p) (void)hipFree(p);
    return rc;
}

}  // extern "C"
