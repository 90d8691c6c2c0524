// Multi-GPU exchanges of the packed buffer (include/hpvpinn.h: hpv_rccl_*, hpv_p2p_*): the in-library RCCL all-reduce (librccl
// dlopen'ed at first use) and the peer-mapped mailbox exchange.  Split out of hpv_api.hip in round 4.
#include "hpv_ctx.h"

using namespace hpvd;

namespace {

// RCCL entry points resolved at run time.  A copy that is already loaded (torch ships one) is reused, so that one
// process never runs two collective libraries.
struct RcclApi {
    void* lib = nullptr;
    decltype(&ncclGetUniqueId) GetUniqueId = nullptr;
    decltype(&ncclCommInitRank) CommInitRank = nullptr;
    decltype(&ncclAllReduce) AllReduce = nullptr;
    decltype(&ncclCommDestroy) CommDestroy = nullptr;
    decltype(&ncclGetErrorString) GetErrorString = nullptr;
    decltype(&ncclCommCount) CommCount = nullptr;          // optional (hpv_rccl_info): what the communicator itself reports
    decltype(&ncclCommUserRank) CommUserRank = nullptr;
    bool ok = false;
    std::string why;     // what went wrong, captured where it went wrong (dlerror() is one-shot and goes stale)
};
RcclApi& rccl_api() {
    static RcclApi api;
    static std::once_flag once;
    std::call_once(once, [] {
        const char* names[] = {"librccl.so.1", "librccl.so"};
        for (const char* n : names) if (!api.lib) api.lib = dlopen(n, RTLD_NOW | RTLD_NOLOAD | RTLD_GLOBAL);
        for (const char* n : names)
            if (!api.lib) {
                api.lib = dlopen(n, RTLD_NOW | RTLD_GLOBAL);
                if (!api.lib) { const char* e = dlerror(); api.why = e ? e : "dlopen failed"; }
            }
        if (!api.lib) return;
        api.why.clear();
        auto sym = [&](const char* name) -> void* {
            void* p = dlsym(api.lib, name);
            if (!p && api.why.empty()) { const char* e = dlerror(); api.why = e ? e : (std::string("symbol missing: ") + name); }
            return p;
        };
        api.GetUniqueId = (decltype(api.GetUniqueId))sym("ncclGetUniqueId");
        api.CommInitRank = (decltype(api.CommInitRank))sym("ncclCommInitRank");
        api.AllReduce = (decltype(api.AllReduce))sym("ncclAllReduce");
        api.CommDestroy = (decltype(api.CommDestroy))sym("ncclCommDestroy");
        api.GetErrorString = (decltype(api.GetErrorString))sym("ncclGetErrorString");
        api.CommCount = (decltype(api.CommCount))dlsym(api.lib, "ncclCommCount");
        api.CommUserRank = (decltype(api.CommUserRank))dlsym(api.lib, "ncclCommUserRank");
        api.ok = api.GetUniqueId && api.CommInitRank && api.AllReduce && api.CommDestroy && api.GetErrorString;
    });
    return api;
}


}  // namespace

namespace hpvd {
// ncclAllReduce as the library issues it.  -DHPV_TEST_HOOKS builds (libhpvpinn_testhooks.so) can make it fail on demand --
// HPV_TEST_RCCL_FAIL="capture": every call on a capturing stream fails (a collective that refuses stream capture);
// HPV_TEST_RCCL_FAIL="eager:k": the k-th call outside a capture fails (k >= 1) -- the product library has no such switch.
ncclResult_t rccl_allreduce(hpv_ctx* h, void* buf, size_t n, hipStream_t s) {
#ifdef HPV_TEST_HOOKS
    if (const char* e = getenv("HPV_TEST_RCCL_FAIL")) {
        hipStreamCaptureStatus st = hipStreamCaptureStatusNone;
        (void)hipStreamIsCapturing(s, &st);
        static long eager_calls = 0;
        if (!strncmp(e, "capture", 7)) { if (st == hipStreamCaptureStatusActive) return ncclInvalidUsage; }
        else if (!strncmp(e, "eager:", 6) && st != hipStreamCaptureStatusActive) { if (++eager_calls == atol(e + 6)) return ncclSystemError; }
    }
#endif
    return rccl_api().AllReduce(buf, buf, n, ncclDouble, ncclSum, h->rccl_comm, s);
}

const char* rccl_error_string(ncclResult_t r) { return rccl_api().ok ? rccl_api().GetErrorString(r) : "librccl not loaded"; }

// after a synchronisation point: did an exchange give up waiting for a peer?
int p2p_check(hpv_ctx* h) {
    if (!h->p2p_on) return 0;
    int err = 0;
    HIPCHK(h, hipMemcpy(&err, h->d_p2p_err, sizeof(int), hipMemcpyDeviceToHost));
    if (err) return fail(h, -5, "in-library exchange: a peer did not arrive (rank %d of %d)", h->pp.rank, h->pp.world);
    return 0;
}
}  // namespace hpvd

extern "C" {

// ---- in-library exchange (multi-GPU, one process per GPU on one node) ----
}  // extern "C"
namespace hpvd {
void p2p_release(hpv_ctx* h) {
    for (void*& m : h->p2p_maps) if (m) { (void)hipIpcCloseMemHandle(m); m = nullptr; }
    if (h->d_inbox) (void)hipFree(h->d_inbox);
    if (h->d_flag) (void)hipFree(h->d_flag);
    if (h->d_p2p_counter) (void)hipFree(h->d_p2p_counter);
    if (h->d_p2p_err) (void)hipFree(h->d_p2p_err);
    h->d_inbox = nullptr; h->d_flag = nullptr; h->d_p2p_counter = nullptr; h->d_p2p_err = nullptr;
    h->p2p_on = false;
}
}  // namespace hpvd
extern "C" {

int hpv_p2p_export(hpv_handle h, int world, int rank, void* handles128) {
    if (!h || !handles128 || world < 1 || world > HPV_P2P_MAX || rank < 0 || rank >= world) return -1;
    if (!h->have_params) return fail(h, -3, "hpv_set_params has not been called");
    p2p_release(h);
    drop_graph(h);
    const int n = h->Ptot + 4;
    const size_t nb = (size_t)2 * world * n * sizeof(double), fb = (size_t)2 * world * sizeof(unsigned long long);
    // uncached (fine-grained) device memory: peers write it over xGMI while this rank polls it
    if (hipExtMallocWithFlags((void**)&h->d_inbox, nb, hipDeviceMallocUncached) != hipSuccess) { (void)hipGetLastError(); HIPCHK(h, hipMalloc((void**)&h->d_inbox, nb)); }
    if (hipExtMallocWithFlags((void**)&h->d_flag, fb, hipDeviceMallocUncached) != hipSuccess) { (void)hipGetLastError(); HIPCHK(h, hipMalloc((void**)&h->d_flag, fb)); }
    HIPCHK(h, hipMalloc((void**)&h->d_p2p_counter, sizeof(unsigned long long)));
    HIPCHK(h, hipMalloc((void**)&h->d_p2p_err, sizeof(int)));
    HIPCHK(h, hipMemset(h->d_inbox, 0, nb));
    HIPCHK(h, hipMemset(h->d_flag, 0, fb));
    HIPCHK(h, hipMemset(h->d_p2p_counter, 0, sizeof(unsigned long long)));
    HIPCHK(h, hipMemset(h->d_p2p_err, 0, sizeof(int)));
    HIPCHK(h, hipDeviceSynchronize());
    hipIpcMemHandle_t hi, hf;
    HIPCHK(h, hipIpcGetMemHandle(&hi, h->d_inbox));
    HIPCHK(h, hipIpcGetMemHandle(&hf, h->d_flag));
    static_assert(sizeof(hipIpcMemHandle_t) == 64, "IPC handle size");
    memcpy(handles128, &hi, 64);
    memcpy((char*)handles128 + 64, &hf, 64);
    h->pp = P2PArgs{};
    h->pp.world = world; h->pp.rank = rank; h->pp.n = n;
    h->pp.counter = h->d_p2p_counter; h->pp.err = h->d_p2p_err;
    {   // wall-clock budget of one exchange wait (ranks may be skewed by host work); HPV_P2P_TIMEOUT_MS overrides
        double ms = 20000.0;
        if (const char* e = getenv("HPV_P2P_TIMEOUT_MS")) { const double v = atof(e); if (v > 0.0) ms = v; }
        h->pp.timeout_ticks = (unsigned long long)(ms * 1e5);
    }
    return 0;
}

int hpv_p2p_connect(hpv_handle h, const void* handles) {
    if (!h || !handles || !h->d_inbox) return -1;
    const int W = h->pp.world, me = h->pp.rank;
    for (int r = 0; r < W; ++r) {
        if (r == me) { h->pp.inbox[r] = h->d_inbox; h->pp.flag[r] = h->d_flag; continue; }
        hipIpcMemHandle_t hi, hf;
        memcpy(&hi, (const char*)handles + (size_t)r * 128, 64);
        memcpy(&hf, (const char*)handles + (size_t)r * 128 + 64, 64);
        void *pi = nullptr, *pf = nullptr;
        hipError_t e = hipIpcOpenMemHandle(&pi, hi, hipIpcMemLazyEnablePeerAccess);
        if (e == hipSuccess) e = hipIpcOpenMemHandle(&pf, hf, hipIpcMemLazyEnablePeerAccess);
        if (e != hipSuccess) { (void)hipGetLastError(); p2p_release(h); return fail(h, -2, "hipIpcOpenMemHandle (rank %d): %s", r, hipGetErrorString(e)); }
        h->p2p_maps[2 * r] = pi; h->p2p_maps[2 * r + 1] = pf;
        h->pp.inbox[r] = (double*)pi; h->pp.flag[r] = (unsigned long long*)pf;
    }
    drop_graph(h);
    h->p2p_on = true;
    return 0;
}

int hpv_p2p_disconnect(hpv_handle h) {
    if (!h) return -1;
    (void)hipStreamSynchronize(h->stream);
    drop_graph(h);
    p2p_release(h);
    return 0;
}

// Known-answer exchange: RB[i] = (rank + 1) + 1e-3 i on every rank -> out[i] = W (W + 1) / 2 + W 1e-3 i; status = the
// device-side timeout flag.  Collective: every rank must call it the same number of times.
int hpv_p2p_selftest(hpv_handle h, double* out, size_t n, int* timed_out) {
    if (!h || !out || !timed_out || !h->p2p_on || n != (size_t)h->pp.n) return -1;
    std::vector<double> v(n);
    for (size_t i = 0; i < n; ++i) v[i] = (double)(h->pp.rank + 1) + 1e-3 * (double)i;
    int rc = upload(h, h->d_RB, v.data(), n);
    if (rc) return rc;
    launch_p2p_exchange(h->pp, h->d_RB, nullptr, h->P, h->Ptot, h->stream);
    HIPCHK(h, hipMemcpyAsync(out, h->d_RB, n * sizeof(double), hipMemcpyDeviceToHost, h->stream));
    HIPCHK(h, hipMemcpyAsync(timed_out, h->d_p2p_err, sizeof(int), hipMemcpyDeviceToHost, h->stream));
    HIPCHK(h, hipStreamSynchronize(h->stream));
    return 0;
}

// ---- in-library RCCL all-reduce (multi-GPU default: one process per GPU, communicator owned by the handle) ----
}  // extern "C"
namespace hpvd {
void rccl_release(hpv_ctx* h) {
    if (h->rccl_comm) { (void)rccl_api().CommDestroy(h->rccl_comm); h->rccl_comm = nullptr; }
    h->rccl_on = false;
}
}  // namespace hpvd
extern "C" {

int hpv_rccl_available(void) { return rccl_api().ok ? 1 : 0; }

int hpv_rccl_unique_id(hpv_handle h, void* id128) {
    if (!h || !id128) return -1;
    static_assert(sizeof(ncclUniqueId) == 128, "ncclUniqueId size");
    if (!rccl_api().ok) return fail(h, -6, "librccl.so could not be loaded: %s", rccl_api().why.c_str());
    ncclUniqueId id;
    ncclResult_t r = rccl_api().GetUniqueId(&id);
    if (r != ncclSuccess) return fail(h, -6, "ncclGetUniqueId failed: %s", rccl_api().GetErrorString(r));
    memcpy(id128, &id, 128);
    return 0;
}

int hpv_rccl_connect(hpv_handle h, int world, int rank, const void* id128) {
    if (!h || !id128 || world < 1 || rank < 0 || rank >= world) return -1;
    if (h->rccl_abandoned.load()) return fail(h, -6, "this handle abandoned an RCCL call earlier (hpv_rccl_abandon)");
    if (!rccl_api().ok) return fail(h, -6, "librccl.so could not be loaded: %s", rccl_api().why.c_str());
    if (!h->have_params) return fail(h, -3, "hpv_set_params has not been called");
    HIPCHK(h, hipSetDevice(h->cfg.device));
    HIPCHK(h, hipStreamSynchronize(h->stream));
    drop_graph(h);
    rccl_release(h);
    ncclUniqueId id;
    memcpy(&id, id128, 128);
    // The blocking part.  The caller may run this on a helper thread and stop waiting (hpv_rccl_abandon): from here on nothing of
    // the handle is touched unless the token is still clear -- a communicator that comes up late is destroyed again and the
    // handle stays unconnected (advisor, round 3: a late success used to set rccl_on in the middle of the fallback's training).
    ncclComm_t comm = nullptr;
#ifdef HPV_TEST_HOOKS
    if (const char* e = getenv("HPV_TEST_RCCL_CONNECT_DELAY_MS")) usleep((useconds_t)(atof(e) * 1000.0));
#endif
    ncclResult_t r = rccl_api().CommInitRank(&comm, world, id, rank);
    if (h->rccl_abandoned.load()) {
        if (r == ncclSuccess && comm) (void)rccl_api().CommDestroy(comm);
        return -6;
    }
    if (r != ncclSuccess) return fail(h, -6, "ncclCommInitRank failed: %s", rccl_api().GetErrorString(r));
    h->rccl_comm = comm;
    h->rccl_world = world; h->rccl_rank = rank;
    h->rccl_on = true;
    return 0;
}

// The caller gave up waiting for a hpv_rccl_connect / hpv_rccl_selftest that blocks on another thread.  Thread-safe (the only
// entry point that is); the abandoned call returns -6 without touching the handle again.  A connect that was abandoned leaves
// the handle usable (unconnected); after an abandoned self-test the stream may be blocked behind a collective that never
// completes -- the caller must not use this handle any more (the Python classes build a fresh one).
int hpv_rccl_abandon(hpv_handle h) {
    if (!h) return -1;
    h->rccl_abandoned.store(1);
    return 0;
}

int hpv_rccl_disconnect(hpv_handle h) {
    if (!h) return -1;
    (void)hipStreamSynchronize(h->stream);
    drop_graph(h);
    rccl_release(h);
    return 0;
}

// Known-answer all-reduce: RB[i] = (rank + 1) + 1e-3 i on every rank -> out[i] = W (W + 1) / 2 + W 1e-3 i.  Collective.
int hpv_rccl_selftest(hpv_handle h, double* out, size_t n) {
    if (!h || !out || !h->rccl_on || n != (size_t)h->Ptot + 4) return -1;
    HIPCHK(h, hipSetDevice(h->cfg.device));   // (may run on a helper thread of the caller: the current device is per thread)
    std::vector<double> v(n);
    for (size_t i = 0; i < n; ++i) v[i] = (double)(h->rccl_rank + 1) + 1e-3 * (double)i;
    int rc = upload(h, h->d_RB, v.data(), n);
    if (rc) return rc;
    ncclResult_t r = rccl_allreduce(h, h->d_RB, n, h->stream);
    if (h->rccl_abandoned.load()) return -6;      // the caller stopped waiting: the handle is no longer ours to touch
    if (r != ncclSuccess) return fail(h, -6, "ncclAllReduce failed: %s", rccl_api().GetErrorString(r));
    hipError_t e1 = hipMemcpyAsync(out, h->d_RB, n * sizeof(double), hipMemcpyDeviceToHost, h->stream);
    hipError_t e2 = e1 == hipSuccess ? hipStreamSynchronize(h->stream) : e1;
    if (h->rccl_abandoned.load()) return -6;
    if (e2 != hipSuccess) return fail(h, -2, "rccl self-test copy failed: %s", hipGetErrorString(e2));
    return 0;
}

// What the COMMUNICATOR reports (ncclCommCount / ncclCommUserRank), not what the caller asked for: a reader of the bench line
// can check that RCCL saw N ranks.  Local call.
int hpv_rccl_info(hpv_handle h, int* world, int* rank) {
    if (!h || !world || !rank) return -1;
    *world = 0; *rank = -1;
    if (!h->rccl_on) return 0;
    RcclApi& api = rccl_api();
    if (!api.CommCount || !api.CommUserRank) return fail(h, -6, "librccl has no ncclCommCount / ncclCommUserRank");
    ncclResult_t r = api.CommCount(h->rccl_comm, world);
    if (r == ncclSuccess) r = api.CommUserRank(h->rccl_comm, rank);
    if (r != ncclSuccess) return fail(h, -6, "ncclCommCount failed: %s", api.GetErrorString(r));
    return 0;
}

// The collective ALONE: `reps` eager ncclAllReduce(sum, double) calls of a buffer of the packed buffer's size (a scratch copy: the
// handle's state is not touched) back to back on the handle's stream between one hipEvent pair, after `reps / 10 + 1` untimed
// ones.  Collective: every rank calls it with the same `reps`.  avg_us = microseconds per all-reduce on THIS rank.
int hpv_rccl_time_allreduce(hpv_handle h, int reps, double* avg_us) {
    if (!h || !avg_us || reps < 1 || !h->rccl_on) return -1;
    if (h->rccl_abandoned.load()) return -6;
    HIPCHK(h, hipSetDevice(h->cfg.device));
    const size_t n = (size_t)h->Ptot + 4;
    double* scratch = nullptr;
    HIPCHK(h, hipMalloc((void**)&scratch, n * sizeof(double)));
    hipEvent_t e0 = nullptr, e1 = nullptr;
    int rc = 0;
    auto done = [&](int code) {
        if (e0) (void)hipEventDestroy(e0);
        if (e1) (void)hipEventDestroy(e1);
        (void)hipFree(scratch);
        return code;
    };
    if (hipMemsetAsync(scratch, 0, n * sizeof(double), h->stream) != hipSuccess || hipEventCreate(&e0) != hipSuccess ||
        hipEventCreate(&e1) != hipSuccess)
        return done(fail(h, -2, "hpv_rccl_time_allreduce: set-up failed"));
    for (int pass = 0; pass < 2 && !rc; ++pass) {
        const int k = pass ? reps : reps / 10 + 1;
        if (pass) (void)hipEventRecord(e0, h->stream);
        for (int i = 0; i < k && !rc; ++i) {
            ncclResult_t r = rccl_allreduce(h, scratch, n, h->stream);
            if (r != ncclSuccess) rc = fail(h, -6, "ncclAllReduce failed: %s", rccl_api().GetErrorString(r));
        }
        if (pass) (void)hipEventRecord(e1, h->stream);
    }
    hipError_t e = hipStreamSynchronize(h->stream);
    if (!rc && e != hipSuccess) rc = fail(h, -2, "hpv_rccl_time_allreduce: %s", hipGetErrorString(e));
    if (!rc) {
        float ms = 0.f;
        (void)hipEventElapsedTime(&ms, e0, e1);
        *avg_us = 1e3 * (double)ms / (double)reps;
    }
    return done(rc);
}

int hpv_exchange_in_use(hpv_handle h) { return !h ? -1 : (h->rccl_on ? 1 : (h->p2p_on ? 2 : 0)); }

}  // extern "C"
