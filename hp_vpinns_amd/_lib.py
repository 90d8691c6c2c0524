"""ctypes binding of libhpvpinn.so (include/hpvpinn.h).

The library is the product: if it is missing, cannot be loaded, or reports no HIP device,
every call raises -- there is no CPU fallback (the oracle under `oracle/` is test
infrastructure and is never imported from here).
"""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("HPV_LIBRARY") or os.path.join(_HERE, "libhpvpinn.so")   # (override: another build of the same library)

HPV_MAX_LAYERS = 16
HIST_CAP = 4096          # HPV_HIST_CAP of csrc/hpv_internal.h: loss-history entries the device keeps
PDE_POISSON1D, PDE_POISSON2D, PDE_ADVDIFF = 0, 1, 2
ACT_TANH, ACT_SIN = 0, 1
BACKEND_AUTO, BACKEND_GENERIC, BACKEND_MFMA = 0, 1, 2
SCHEME_VPINN, SCHEME_PINN = 0, 1

# every symbol include/hpvpinn.h declares (tests check the .so exports all of them)
EXPORTS = [
    "hpv_create", "hpv_destroy", "hpv_last_error", "hpv_set_stream", "hpv_set_quadrature",
    "hpv_set_tables", "hpv_set_elements", "hpv_set_rhs", "hpv_set_data", "hpv_num_params",
    "hpv_set_params", "hpv_get_params", "hpv_loss_and_grad", "hpv_step", "hpv_forward_backward",
    "hpv_reduce_buffer", "hpv_apply_adam", "hpv_eval_loss", "hpv_read_loss", "hpv_sync",
    "hpv_predict", "hpv_get_residuals", "hpv_backend_in_use", "hpv_pass_structure", "hpv_set_active_tests", "hpv_enable_timing",
    "hpv_kernel_time_ms", "hpv_time_iteration_kernel", "hpv_bench_projection", "hpv_debug_activation", "hpv_get_state", "hpv_set_state",
    "hpv_assemble_rhs", "hpv_set_collocation", "hpv_gll_rule", "hpv_test_tables",
    "hpv_step_record", "hpv_history_reset", "hpv_history_read",
    "hpv_p2p_export", "hpv_p2p_connect", "hpv_p2p_selftest", "hpv_p2p_disconnect",
    "hpv_eval_channels", "hpv_bench_residual",
    "hpv_set_collocation_shard", "hpv_rccl_unique_id", "hpv_rccl_connect", "hpv_rccl_selftest", "hpv_rccl_disconnect", "hpv_exchange_in_use",
    "hpv_rccl_available", "hpv_graphs_in_use", "hpv_updates_applied", "hpv_set_shared_element_kernels", "hpv_shared_element_kernels",
    "hpv_kernel_variant", "hpv_build_info", "hpv_rccl_abandon", "hpv_bench_residual_checksums", "hpv_rule_advice",
    "hpv_rccl_info", "hpv_rccl_time_allreduce", "hpv_grid_plan",
]


def rule_advice(device, dim, q, ntx, nty, n_elem_shard, exact_counts=False, n_hidden=0):
    """hpv_rule_advice: (q_dev, nt_dev) -- the instantiated rule a shard's rule should be zero-weight padded to (q_dev == q: leave it
    alone) and, in 1-D, the test-function count the device tables should have.  The limits are the launch functions' own."""
    qd, nd = C.c_int(0), C.c_int(0)
    rc = load().hpv_rule_advice(int(device), int(dim), int(q), int(ntx), int(nty), int(n_elem_shard), 1 if exact_counts else 0, int(n_hidden),
                                C.byref(qd), C.byref(nd))
    if rc:
        raise HpvError(f"hpv_rule_advice({dim}, {q}, {ntx}, {nty}, {n_elem_shard}) returned {rc}")
    return qd.value, nd.value


def grid_plan(device, q, n_hidden, n_elem_shard):
    """hpv_grid_plan: 0 separate launches | 1 one workgroup per element | 2 element loop | 3 full rounds + split tail."""
    rc = load().hpv_grid_plan(int(device), int(q), int(n_hidden), int(n_elem_shard))
    if rc < 0:
        raise HpvError(f"hpv_grid_plan({q}, {n_hidden}, {n_elem_shard}) returned {rc}")
    return rc


class HpvConfig(C.Structure):
    _fields_ = [
        ("pde", C.c_int), ("var_form", C.c_int), ("act", C.c_int), ("n_layers", C.c_int),
        ("layers", C.c_int * HPV_MAX_LAYERS),
        ("lr", C.c_double), ("beta1", C.c_double), ("beta2", C.c_double), ("eps", C.c_double),
        ("lossb_weight", C.c_double), ("V", C.c_double),
        ("device", C.c_int), ("backend", C.c_int), ("scheme", C.c_int),
    ]


class HpvError(RuntimeError):
    """A libhpvpinn entry point returned non-zero; `code` is that value (include/hpvpinn.h lists them)."""
    code = None

EXCHANGE_TIMEOUT = -7      # an in-kernel exchange between the workgroups of one element timed out (hpv_step and friends)


TEST_HOOKS_LIB_PATH = os.path.join(_HERE, "libhpvpinn_testhooks.so")   # the -DHPV_TEST_HOOKS build (fault-injection knobs; tests only)

_lib = None
_libs = {}                # path -> loaded library (the product library and, in the test suite, the test-hooks build beside it)
_current = [None]         # path new Handles bind to (None: LIB_PATH); see `library`
_dp = C.POINTER(C.c_double)


class library:
    """Context manager: Handles created inside bind to another build of the library (e.g. TEST_HOOKS_LIB_PATH).  Both builds
    are linked -Bsymbolic, so two of them may live in one process; a Handle keeps the library it was created with."""

    def __init__(self, path):
        self.path = path

    def __enter__(self):
        self.prev = _current[0]
        _current[0] = self.path
        return load()

    def __exit__(self, *exc):
        _current[0] = self.prev
        return False


def load():
    """Load libhpvpinn.so (after torch, so that both share one HIP runtime)."""
    global _lib
    path = _current[0] or LIB_PATH
    if path in _libs:
        return _libs[path]
    if not os.path.exists(path):
        raise HpvError(f"{path} is missing: build it with hp_vpinns_amd/csrc/build.sh "
                       "(or __graft_entry__.build()); there is no CPU fallback")
    try:  # torch first: its bundled libamdhip64 (same soname) then serves both
        import torch  # noqa: F401
    except Exception:  # pragma: no cover - torch is optional for the single-GPU path
        pass
    lib = C.CDLL(path, mode=C.RTLD_GLOBAL if path == LIB_PATH else C.RTLD_LOCAL)
    h = C.c_void_p
    lib.hpv_create.argtypes = [C.POINTER(h), C.POINTER(HpvConfig)]
    lib.hpv_destroy.argtypes = [h]
    lib.hpv_destroy.restype = None
    lib.hpv_last_error.argtypes = [h]
    lib.hpv_last_error.restype = C.c_char_p
    lib.hpv_set_stream.argtypes = [h, C.c_void_p]
    lib.hpv_set_quadrature.argtypes = [h, _dp, _dp, C.c_int, _dp, _dp, C.c_int]
    lib.hpv_set_tables.argtypes = [h, _dp, _dp, _dp, C.c_int, _dp, _dp, _dp, C.c_int, _dp]
    lib.hpv_set_elements.argtypes = [h, _dp, C.c_int, _dp, C.c_int, C.c_int, C.c_int]
    lib.hpv_set_rhs.argtypes = [h, _dp, C.c_size_t]
    lib.hpv_set_data.argtypes = [h, _dp, _dp, C.c_int]
    lib.hpv_set_collocation.argtypes = [h, _dp, _dp, C.c_int]
    lib.hpv_num_params.argtypes = [h]
    lib.hpv_num_params.restype = C.c_size_t
    lib.hpv_set_params.argtypes = [h, _dp, C.c_size_t]
    lib.hpv_get_params.argtypes = [h, _dp, C.c_size_t]
    lib.hpv_loss_and_grad.argtypes = [h, _dp, _dp]
    lib.hpv_step.argtypes = [h, C.c_int, _dp]
    lib.hpv_forward_backward.argtypes = [h]
    lib.hpv_reduce_buffer.argtypes = [h, C.POINTER(C.c_void_p), C.POINTER(C.c_size_t)]
    lib.hpv_apply_adam.argtypes = [h]
    lib.hpv_eval_loss.argtypes = [h]
    lib.hpv_read_loss.argtypes = [h, _dp]
    lib.hpv_sync.argtypes = [h]
    lib.hpv_predict.argtypes = [h, _dp, C.c_int, _dp]
    lib.hpv_get_residuals.argtypes = [h, _dp, C.c_size_t]
    lib.hpv_backend_in_use.argtypes = [h]
    lib.hpv_pass_structure.argtypes = [h]
    lib.hpv_set_active_tests.argtypes = [h, C.POINTER(C.c_int), C.c_int]
    lib.hpv_enable_timing.argtypes = [h, C.c_int]
    lib.hpv_kernel_time_ms.argtypes = [h, C.c_int, _dp, C.POINTER(C.c_long)]
    lib.hpv_time_iteration_kernel.argtypes = [h, C.c_int, _dp]
    lib.hpv_bench_projection.argtypes = [h, C.c_long, C.c_int, _dp, _dp]
    lib.hpv_debug_activation.argtypes = [h, _dp, C.c_int, _dp, _dp, _dp]
    lib.hpv_get_state.argtypes = [h, _dp, C.c_size_t]
    lib.hpv_set_state.argtypes = [h, _dp, C.c_size_t]
    lib.hpv_assemble_rhs.argtypes = [h, _dp, C.c_size_t, _dp, C.c_size_t]
    lib.hpv_gll_rule.argtypes = [h, C.c_int, _dp, _dp]
    lib.hpv_step_record.argtypes = [h, C.c_int, _dp, _dp]
    lib.hpv_history_reset.argtypes = [h]
    lib.hpv_p2p_export.argtypes = [h, C.c_int, C.c_int, C.c_char_p]
    lib.hpv_p2p_connect.argtypes = [h, C.c_char_p]
    lib.hpv_p2p_selftest.argtypes = [h, _dp, C.c_size_t, C.POINTER(C.c_int)]
    lib.hpv_p2p_disconnect.argtypes = [h]
    lib.hpv_history_read.argtypes = [h, C.c_int, _dp, _dp]
    lib.hpv_test_tables.argtypes = [h, C.c_int, _dp, C.c_int, _dp]
    lib.hpv_eval_channels.argtypes = [h, _dp, C.c_size_t]
    lib.hpv_bench_residual.argtypes = [h, C.c_long, C.c_int, C.c_int, _dp, _dp]
    lib.hpv_set_collocation_shard.argtypes = [h, _dp, _dp, C.c_int, C.c_long]
    lib.hpv_rccl_available.argtypes = []
    lib.hpv_graphs_in_use.argtypes = [h]
    lib.hpv_updates_applied.argtypes = [h, C.POINTER(C.c_longlong)]
    lib.hpv_set_shared_element_kernels.argtypes = [h, C.c_int]
    lib.hpv_shared_element_kernels.argtypes = [h]
    lib.hpv_rccl_unique_id.argtypes = [h, C.c_char_p]
    lib.hpv_rccl_connect.argtypes = [h, C.c_int, C.c_int, C.c_char_p]
    lib.hpv_rccl_selftest.argtypes = [h, _dp, C.c_size_t]
    lib.hpv_rccl_disconnect.argtypes = [h]
    lib.hpv_exchange_in_use.argtypes = [h]
    lib.hpv_rccl_abandon.argtypes = [h]
    lib.hpv_rccl_info.argtypes = [h, C.POINTER(C.c_int), C.POINTER(C.c_int)]
    lib.hpv_rccl_time_allreduce.argtypes = [h, C.c_int, _dp]
    lib.hpv_bench_residual_checksums.argtypes = [h, C.c_long, C.c_int, _dp]
    lib.hpv_kernel_variant.argtypes = [h, C.c_char_p, C.c_size_t]
    lib.hpv_build_info.argtypes = []
    lib.hpv_grid_plan.argtypes = [C.c_int, C.c_int, C.c_int, C.c_long]
    lib.hpv_rule_advice.argtypes = [C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_long, C.c_int, C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_int)]
    lib.hpv_build_info.restype = C.c_char_p
    _libs[path] = lib
    if path == LIB_PATH:
        _lib = lib
    return lib


def build_info(lib=None):
    """{'k_iter_fused': 'ok' | 'no-quarter-tile' | 'absent', 'k_iter_tall': ..., 'test_hooks': '0' | '1'} of a loaded build."""
    raw = (lib or load()).hpv_build_info().decode()
    return dict(kv.split("=", 1) for kv in raw.split(";"))


def _p(a):
    return None if a is None else a.ctypes.data_as(_dp)


def _c(a):
    return None if a is None else np.ascontiguousarray(a, dtype=np.float64)


def _points(X, dim, what):
    """(n, dim) float64 C-contiguous point array -- the C side reads n*dim doubles, so the row length is checked here."""
    X = np.ascontiguousarray(X, dtype=np.float64)
    if X.ndim != 2 or X.shape[1] != dim:
        raise ValueError(f"{what} must have shape (n, {dim}), got {X.shape}")
    return X


_LEAKED = []       # handles abandoned with a call still inside the library (Handle.leak)


class Handle:
    """Thin RAII wrapper over an `hpv_handle`; every method maps 1:1 onto a C entry point."""

    def __init__(self, pde, var_form, act, layers, lr=1e-3, lossb_weight=1.0, V=1.0, device=0,
                 backend=BACKEND_AUTO, beta1=0.9, beta2=0.999, eps=1e-8, scheme=SCHEME_VPINN):
        self.lib = load()
        cfg = HpvConfig()
        cfg.pde, cfg.var_form, cfg.act = int(pde), int(var_form), int(act)
        layers = [int(v) for v in layers]
        if len(layers) > HPV_MAX_LAYERS:
            raise HpvError("too many layers")
        cfg.n_layers = len(layers)
        for i, v in enumerate(layers):
            cfg.layers[i] = v
        cfg.lr, cfg.beta1, cfg.beta2, cfg.eps = lr, beta1, beta2, eps
        cfg.lossb_weight, cfg.V = float(lossb_weight), float(V)
        cfg.device, cfg.backend, cfg.scheme = int(device), int(backend), int(scheme)
        self._h = C.c_void_p()
        rc = self.lib.hpv_create(C.byref(self._h), C.byref(cfg))
        if rc != 0:
            msg = self.lib.hpv_last_error(None)
            self._h = None
            raise HpvError(f"hpv_create failed ({rc}): {msg.decode() if msg else ''}")
        self.cfg = cfg
        self.layers = layers
        self._keep = []

    def _chk(self, rc):
        if rc != 0:
            msg = self.lib.hpv_last_error(self._h)
            err = HpvError(f"libhpvpinn error {rc}: {msg.decode() if msg else ''}")
            err.code = int(rc)
            raise err

    def close(self):
        if getattr(self, "_h", None):
            self.lib.hpv_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ---- set-up -------------------------------------------------------------------------
    def set_stream(self, stream_ptr):
        self._chk(self.lib.hpv_set_stream(self._h, C.c_void_p(stream_ptr)))

    def set_quadrature(self, xi, wx, yi=None, wy=None):
        xi, wx, yi, wy = _c(xi), _c(wx), _c(yi), _c(wy)
        self._chk(self.lib.hpv_set_quadrature(self._h, _p(xi), _p(wx), xi.size, _p(yi), _p(wy),
                                              1 if yi is None else yi.size))

    def set_tables(self, tx, ty=None, edge_dphi=None):
        """tx / ty: (3, ntest, q) arrays of phi, phi', phi''."""
        tx = _c(tx)
        ty = _c(ty)
        ed = _c(edge_dphi)
        a = [_p(tx[0]), _p(tx[1]), _p(tx[2]), tx.shape[1]]
        b = [None, None, None, 1] if ty is None else [_p(ty[0]), _p(ty[1]), _p(ty[2]), ty.shape[1]]
        self._chk(self.lib.hpv_set_tables(self._h, *a, *b, _p(ed)))

    def set_elements(self, gridx, gridy=None, e_begin=0, e_end=None):
        gx, gy = _c(gridx), _c(gridy)
        nex = gx.size - 1
        ney = 1 if gy is None else gy.size - 1
        if e_end is None:
            e_end = nex * ney
        self._chk(self.lib.hpv_set_elements(self._h, _p(gx), nex, _p(gy), ney, int(e_begin), int(e_end)))

    def set_rhs(self, F):
        F = _c(F)
        self._chk(self.lib.hpv_set_rhs(self._h, _p(F), 0 if F is None else F.size))

    def set_collocation(self, X, f, n_total=None):
        """X, f: this handle's collocation points; n_total: their number over all shards (default: these are all)."""
        X, f = _points(X, self.layers[0], "collocation points"), _c(f).reshape(-1)
        if f.size != X.shape[0]:
            raise ValueError("one right-hand-side value per collocation point")
        self._chk(self.lib.hpv_set_collocation_shard(self._h, _p(X), _p(f), X.shape[0],
                                                     X.shape[0] if n_total is None else int(n_total)))

    def set_data(self, X, u):
        if X is None:
            self._chk(self.lib.hpv_set_data(self._h, None, None, 0))
            return
        X, u = _points(X, self.layers[0], "data points"), _c(u).reshape(-1)
        if u.size != X.shape[0]:
            raise ValueError("one target value per data point")
        self._chk(self.lib.hpv_set_data(self._h, _p(X), _p(u), X.shape[0]))

    # ---- parameters ---------------------------------------------------------------------
    def num_params(self):
        return int(self.lib.hpv_num_params(self._h))

    def set_params(self, theta):
        theta = _c(theta).reshape(-1)
        self._chk(self.lib.hpv_set_params(self._h, _p(theta), theta.size))

    def get_params(self):
        out = np.empty(self.num_params())
        self._chk(self.lib.hpv_get_params(self._h, _p(out), out.size))
        return out

    # ---- compute ------------------------------------------------------------------------
    def loss_and_grad(self, want_grad=True):
        loss3 = np.empty(3)
        g = np.empty(self.num_params()) if want_grad else None
        self._chk(self.lib.hpv_loss_and_grad(self._h, _p(loss3), _p(g)))
        return loss3, g

    def step(self, n_iters, read_loss=True):
        loss3 = np.empty(3) if read_loss else None
        self._chk(self.lib.hpv_step(self._h, int(n_iters), _p(loss3)))
        return loss3

    def forward_backward(self):
        self._chk(self.lib.hpv_forward_backward(self._h))

    def reduce_buffer(self):
        ptr, n = C.c_void_p(), C.c_size_t()
        self._chk(self.lib.hpv_reduce_buffer(self._h, C.byref(ptr), C.byref(n)))
        return ptr.value, n.value

    def apply_adam(self):
        self._chk(self.lib.hpv_apply_adam(self._h))

    def eval_loss(self):
        self._chk(self.lib.hpv_eval_loss(self._h))

    def read_loss(self):
        loss3 = np.empty(3)
        self._chk(self.lib.hpv_read_loss(self._h, _p(loss3)))
        return loss3

    def sync(self):
        self._chk(self.lib.hpv_sync(self._h))

    def predict(self, X):
        X = _points(X, self.layers[0], "prediction points")
        out = np.empty(X.shape[0])
        self._chk(self.lib.hpv_predict(self._h, _p(X), X.shape[0], _p(out)))
        return out

    def channels(self, n_points, n_channels):
        """(C, n_points): the network value and its input-derivative channels at this handle's quadrature points,
        element-major (one forward launch; hpv_eval_channels)."""
        out = np.empty((int(n_channels), int(n_points)))
        self._chk(self.lib.hpv_eval_channels(self._h, _p(out), out.size))
        return out

    def set_active_tests(self, n_active):
        """per-element number of active test functions (1-D p-refinement); None = all."""
        if n_active is None:
            self._chk(self.lib.hpv_set_active_tests(self._h, None, 0))
            return
        a = np.ascontiguousarray(n_active, dtype=np.int32).reshape(-1)
        self._chk(self.lib.hpv_set_active_tests(self._h, a.ctypes.data_as(C.POINTER(C.c_int)), a.size))

    def residuals(self, n):
        out = np.empty(n)
        self._chk(self.lib.hpv_get_residuals(self._h, _p(out), out.size))
        return out

    def backend_in_use(self):
        rc = int(self.lib.hpv_backend_in_use(self._h))
        if rc < 0:
            self._chk(rc)
        return rc

    def enable_timing(self, on=True):
        self._chk(self.lib.hpv_enable_timing(self._h, 1 if on else 0))

    def kernel_time_ms(self, which):
        ms, n = C.c_double(), C.c_long()
        self._chk(self.lib.hpv_kernel_time_ms(self._h, int(which), C.byref(ms), C.byref(n)))
        return ms.value, n.value

    def time_iteration_kernel(self, reps):
        """Average ms of `reps` back-to-back launches of the whole-iteration kernel between ONE hipEvent pair (HpvError -4 when the
        handle's iteration is not one such launch)."""
        ms = C.c_double()
        self._chk(self.lib.hpv_time_iteration_kernel(self._h, int(reps), C.byref(ms)))
        return ms.value

    def get_state(self):
        out = np.empty(3 * self.num_params() + 2)
        self._chk(self.lib.hpv_get_state(self._h, _p(out), out.size))
        return out

    def set_state(self, state):
        state = _c(state).reshape(-1)
        self._chk(self.lib.hpv_set_state(self._h, _p(state), state.size))

    def assemble_rhs(self, f_quad, n_out):
        f_quad = _c(f_quad).reshape(-1)
        out = np.empty(int(n_out))
        self._chk(self.lib.hpv_assemble_rhs(self._h, _p(f_quad), f_quad.size, _p(out), out.size))
        return out

    def step_record(self, n):
        """n Adam iterations; ((n, 3) array {loss, lossb, lossv}, (n,) epsilon) after each update (one extra forward pass
        in total)."""
        out, eps = np.empty((int(n), 3)), np.empty(int(n))
        self._chk(self.lib.hpv_step_record(self._h, int(n), _p(out), _p(eps)))
        return out, eps

    def history_reset(self):
        self._chk(self.lib.hpv_history_reset(self._h))

    def history_read(self, n):
        out, eps = np.empty((int(n), 3)), np.empty(int(n))
        self._chk(self.lib.hpv_history_read(self._h, int(n), _p(out), _p(eps)))
        return out, eps

    def rccl_available(self):
        """librccl can be loaded in this process (local check, no collective)."""
        return int(self.lib.hpv_rccl_available()) == 1

    def rccl_unique_id(self):
        buf = C.create_string_buffer(128)
        self._chk(self.lib.hpv_rccl_unique_id(self._h, buf))
        return buf.raw

    def rccl_connect(self, world, rank, uid):
        self._chk(self.lib.hpv_rccl_connect(self._h, int(world), int(rank), bytes(uid)))

    def rccl_selftest(self, n):
        out = np.empty(int(n))
        self._chk(self.lib.hpv_rccl_selftest(self._h, _p(out), out.size))
        return out

    def kernel_variant(self):
        """Name(s) of the kernel instantiation(s) the most recent reverse-mode pass launched (hpv_kernel_variant)."""
        buf = C.create_string_buffer(320)
        self._chk(self.lib.hpv_kernel_variant(self._h, buf, 320))
        return buf.value.decode()

    def build_info(self):
        return build_info(self.lib)

    def rccl_info(self):
        """(world, rank) as the connected communicator itself reports them (ncclCommCount / ncclCommUserRank); (0, -1): none."""
        w, r = C.c_int(0), C.c_int(-1)
        self._chk(self.lib.hpv_rccl_info(self._h, C.byref(w), C.byref(r)))
        return w.value, r.value

    def rccl_time_allreduce(self, reps=200):
        """Microseconds per eager all-reduce of a packed-buffer-sized scratch buffer (the collective alone).  Collective call."""
        us = C.c_double(0.0)
        self._chk(self.lib.hpv_rccl_time_allreduce(self._h, int(reps), C.byref(us)))
        return us.value

    def rccl_abandon(self):
        """Thread-safe: stop waiting for a blocking rccl_connect / rccl_selftest that runs on a helper thread."""
        self.lib.hpv_rccl_abandon(self._h)

    def leak(self):
        """Never destroy this handle (a helper thread may still be inside the library with it)."""
        _LEAKED.append(self._h)
        self._h = None

    def pass_structure(self):
        """'separate' | 'fused-reverse' | 'whole-iteration' | 'whole-iteration-split' | 'whole-iteration-tile' | 'whole-iteration-tall' |
        'whole-iteration-element' (or None)."""
        return {0: "separate", 1: "fused-reverse", 2: "whole-iteration", 3: "whole-iteration-split",
                4: "whole-iteration-tile", 5: "whole-iteration-tall", 6: "whole-iteration-element"}.get(int(self.lib.hpv_pass_structure(self._h)))

    def graphs_in_use(self):
        """hpv_step replays captured iteration graphs (False: eager launches, e.g. a collective that refused stream capture)."""
        return int(self.lib.hpv_graphs_in_use(self._h)) == 1

    def updates_applied(self):
        """Parameter updates applied through this handle so far (synchronises)."""
        n = C.c_longlong(0)
        self._chk(self.lib.hpv_updates_applied(self._h, C.byref(n)))
        return int(n.value)

    def shared_element_kernels(self):
        """False once the handle stays on launch structures without an in-kernel exchange (set, or after a timeout fallback)."""
        return int(self.lib.hpv_shared_element_kernels(self._h)) == 1

    def set_shared_element_kernels(self, on):
        """False: only launch structures without an in-kernel exchange from now on (what HPV_FUSE=s selects at creation)."""
        self._chk(self.lib.hpv_set_shared_element_kernels(self._h, 1 if on else 0))

    def rccl_disconnect(self):
        self._chk(self.lib.hpv_rccl_disconnect(self._h))

    def exchange_in_use(self):
        return {0: "none", 1: "rccl", 2: "p2p"}[int(self.lib.hpv_exchange_in_use(self._h))]

    def p2p_export(self, world, rank):
        buf = C.create_string_buffer(128)
        self._chk(self.lib.hpv_p2p_export(self._h, int(world), int(rank), buf))
        return buf.raw

    def p2p_connect(self, handles):
        self._chk(self.lib.hpv_p2p_connect(self._h, bytes(handles)))

    def p2p_selftest(self, n):
        out, flag = np.empty(int(n)), C.c_int(0)
        self._chk(self.lib.hpv_p2p_selftest(self._h, _p(out), out.size, C.byref(flag)))
        return out, int(flag.value)

    def p2p_disconnect(self):
        self._chk(self.lib.hpv_p2p_disconnect(self._h))

    def gll_rule(self, q):
        """(nodes, weights) of the q-point Gauss-Lobatto-Legendre rule, computed on the device."""
        xi, w = np.empty(int(q)), np.empty(int(q))
        self._chk(self.lib.hpv_gll_rule(self._h, int(q), _p(xi), _p(w)))
        return xi, w

    def test_tables(self, ntest, xi):
        """(3, ntest, q): phi, phi', phi'' of the Legendre-difference test functions at `xi`, computed on the device."""
        xi = _c(xi).reshape(-1)
        tab = np.empty((3, int(ntest), xi.size))
        self._chk(self.lib.hpv_test_tables(self._h, int(ntest), _p(xi), xi.size, _p(tab)))
        return tab

    def debug_activation(self, x):
        x = _c(x).reshape(-1)
        a, a1, ref = np.empty_like(x), np.empty_like(x), np.empty_like(x)
        self._chk(self.lib.hpv_debug_activation(self._h, _p(x), x.size, _p(a), _p(a1), _p(ref)))
        return a, a1, ref

    def bench_checksums(self, n_elem, do_adjoint=True):
        """Checksums of one stand-alone projection launch on the seeded synthetic batch (hpv_bench_residual_checksums)."""
        out = np.empty(6)
        self._chk(self.lib.hpv_bench_residual_checksums(self._h, int(n_elem), 1 if do_adjoint else 0, _p(out)))
        return out

    def bench_projection(self, n_elem, reps=10, do_adjoint=True):
        ms, by = C.c_double(), C.c_double()
        self._chk(self.lib.hpv_bench_residual(self._h, int(n_elem), int(reps), 1 if do_adjoint else 0, C.byref(ms), C.byref(by)))
        return ms.value, by.value
