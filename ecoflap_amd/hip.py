"""ctypes binding of libecoflap_hip.so (include/ecoflap_hip.h) over PyTorch-ROCm tensors.

This is the ONLY compute backend of the product.  There is no CPU fallback: if
the shared library is missing, or a tensor handed to a kernel is not on the GPU,
the call raises.  PyTorch is plumbing here — it owns device memory and the HIP
stream; the kernels receive raw pointers (`tensor.data_ptr()`) and
`torch.cuda.current_stream().cuda_stream`.
"""
import ctypes
import os

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("ECOFLAP_HIP_LIB", os.path.join(_HERE, "libecoflap_hip.so"))

DTYPE_CODE = {torch.float32: 0, torch.float16: 1, torch.bfloat16: 2}

RED_ABSW_ABSG, RED_SQW_SQG, RED_ABSG, RED_ABSW, RED_SQW = 0, 1, 2, 3, 4

EXPORTS = [
    "ecoflap_version", "ecoflap_error_string", "ecoflap_zo_perturb", "ecoflap_zo_perturb_triple",
    "ecoflap_zo_perturb_units", "ecoflap_zo_perturb_units_timed", "ecoflap_null_launch_timed", "ecoflap_zo_perturb_layers",
    "ecoflap_zo_perturb_layers_z", "ecoflap_torch_normal_threads", "ecoflap_zo_fill_normal_torch",
    "ecoflap_zo_perturb_torch", "ecoflap_zo_perturb_layers_torch", "ecoflap_zo_torch_radius_sweep",
    "ecoflap_torch_layer_items",
    "ecoflap_zo_fill_normal", "ecoflap_philox_u32", "ecoflap_absprod_reduce_workspace_bytes",
    "ecoflap_absprod_reduce", "ecoflap_absprod_reduce_multi_workspace_bytes",
    "ecoflap_absprod_reduce_multi", "ecoflap_absprod_reduce_mixed", "ecoflap_colsqnorm_workspace_bytes",
    "ecoflap_colsqnorm_accum", "ecoflap_colsqnorm_accum_dev",
    "ecoflap_colsqnorm_multi_workspace_bytes", "ecoflap_colsqnorm_accum_multi", "ecoflap_colsq_replay",
    "ecoflap_wanda_workspace_bytes", "ecoflap_wanda_prune_rows",
    "ecoflap_wanda_prune_matrix", "ecoflap_wanda_prune_nm", "ecoflap_wanda_block_workspace_bytes",
    "ecoflap_wanda_prune_block", "ecoflap_wanda_fallback_counts",
    "ecoflap_mask_mul", "ecoflap_allocate_sparsity",
    "ecoflap_sparsegpt_workspace_bytes", "ecoflap_sparsegpt_block", "ecoflap_sparsegpt_block_nm",
    "ecoflap_hessian_workspace_bytes", "ecoflap_hessian_accum", "ecoflap_cholesky_workspace_bytes", "ecoflap_cholesky_f32",
    "ecoflap_cholesky_inverse_workspace_bytes", "ecoflap_cholesky_inverse_f32",
    "ecoflap_grad_accum_multi", "ecoflap_global_prune_workspace_bytes",
    "ecoflap_global_threshold_prune", "ecoflap_global_prune_protected_workspace_bytes",
    "ecoflap_global_threshold_prune_protected", "ecoflap_count_zeros_multi",
]


class EcoflapHipError(RuntimeError):
    pass


class WandaItem(ctypes.Structure):
    """`ecoflap_wanda_item` (include/ecoflap_hip.h)."""
    _fields_ = [("w", ctypes.c_void_p), ("scaler_row", ctypes.c_void_p),
                ("rows", ctypes.c_int64), ("cols", ctypes.c_int64), ("k", ctypes.c_int64),
                ("mask_out", ctypes.c_void_p), ("dtype", ctypes.c_int), ("mode", ctypes.c_int)]


class ColsqItem(ctypes.Structure):
    """`ecoflap_colsq_item` (include/ecoflap_hip.h)."""
    _fields_ = [("scaler_row", ctypes.c_void_p), ("x", ctypes.c_void_p),
                ("tokens", ctypes.c_int64), ("cols", ctypes.c_int64),
                ("nsamples_before", ctypes.c_int64), ("nsamples_dev", ctypes.c_void_p),
                ("batch", ctypes.c_int64), ("raw", ctypes.c_int)]


COLSQ_MAX_ITEMS = 16
_COLSQ_BYTES = ctypes.sizeof(ColsqItem)
_COLSQ_PACK = __import__("struct").Struct("@PPqqqPqi").pack_into     # ColsqItem's fields, native alignment
assert [getattr(ColsqItem, f).offset for f, _ in ColsqItem._fields_] == [0, 8, 16, 24, 32, 40, 48, 56]
_COLSQ_TYPES = {}


def _colsq_array(n):
    t = _COLSQ_TYPES.get(n)
    if t is None:
        t = _COLSQ_TYPES[n] = ColsqItem * n
    return t()


WANDA_MAX_ITEMS = 16
WANDA_ROWS, WANDA_MATRIX = 0, 1


_lib = None


def load_library():
    """dlopen the in-tree library; raises if it has not been built (`make -C ecoflap_amd/csrc`)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise EcoflapHipError(
            f"{LIB_PATH} not found: build it with `python -c 'import __graft_entry__ as g; "
            "g.build()'` or `make -C ecoflap_amd/csrc`. There is no CPU fallback.")
    lib = ctypes.CDLL(LIB_PATH)
    vp, i64, u64, f32, f64, ci, sz = (ctypes.c_void_p, ctypes.c_int64, ctypes.c_uint64,
                                      ctypes.c_float, ctypes.c_double, ctypes.c_int,
                                      ctypes.c_size_t)
    lib.ecoflap_version.restype = ctypes.c_char_p
    lib.ecoflap_error_string.restype = ctypes.c_char_p
    lib.ecoflap_error_string.argtypes = [ci]
    lib.ecoflap_zo_perturb.argtypes = [vp, i64, ci, f32, f32, u64, vp, vp]
    lib.ecoflap_zo_perturb_triple.argtypes = [vp, vp, vp, vp, i64, ci, f32, u64, vp, vp]
    lib.ecoflap_zo_perturb_units.argtypes = [vp, i64, ci, f32, ci, vp, vp, vp, vp, vp]
    lib.ecoflap_zo_perturb_units_timed.argtypes = [vp, i64, ci, f32, ci, vp, vp, vp, vp, vp, vp, vp]
    lib.ecoflap_null_launch_timed.argtypes = [vp, vp, vp]
    lib.ecoflap_zo_perturb_layers.argtypes = [vp, ci, i64, ci, f32, vp, vp, vp]
    lib.ecoflap_zo_perturb_layers_z.argtypes = [vp, ci, i64, ci, f32, vp, vp, vp]
    lib.ecoflap_zo_fill_normal.argtypes = [vp, i64, ci, u64, vp]
    lib.ecoflap_torch_normal_threads.restype = i64
    lib.ecoflap_torch_normal_threads.argtypes = [i64, ci, ci]
    lib.ecoflap_zo_fill_normal_torch.argtypes = [vp, i64, ci, u64, i64, vp]
    lib.ecoflap_zo_perturb_torch.argtypes = [vp, i64, ci, f32, f32, u64, i64, vp]
    lib.ecoflap_zo_perturb_layers_torch.argtypes = [vp, ci, i64, ci, f32, vp, vp, vp]
    lib.ecoflap_zo_torch_radius_sweep.argtypes = [u64, u64, vp, vp]
    lib.ecoflap_torch_layer_items.restype = i64
    lib.ecoflap_torch_layer_items.argtypes = [i64, i64, ci]
    lib.ecoflap_philox_u32.argtypes = [vp, i64, u64, vp]
    lib.ecoflap_absprod_reduce_workspace_bytes.restype = sz
    lib.ecoflap_absprod_reduce_workspace_bytes.argtypes = [i64]
    lib.ecoflap_absprod_reduce.argtypes = [vp, vp, i64, ci, ci, ci, vp, vp, sz, vp]
    lib.ecoflap_absprod_reduce_multi_workspace_bytes.restype = sz
    lib.ecoflap_absprod_reduce_multi_workspace_bytes.argtypes = [ci]
    lib.ecoflap_absprod_reduce_multi.argtypes = [vp, ci, i64, ci, ci, ci, vp, vp, sz, vp]
    lib.ecoflap_absprod_reduce_mixed.argtypes = [vp, vp, ci, ci, vp, vp, sz, vp]
    lib.ecoflap_colsqnorm_workspace_bytes.restype = sz
    lib.ecoflap_colsqnorm_workspace_bytes.argtypes = [i64, i64]
    lib.ecoflap_colsqnorm_accum.argtypes = [vp, vp, i64, i64, ci, i64, i64, vp, sz, vp]
    lib.ecoflap_colsqnorm_accum_dev.argtypes = [vp, vp, i64, i64, ci, vp, i64, vp, sz, vp]
    lib.ecoflap_colsqnorm_multi_workspace_bytes.restype = sz
    lib.ecoflap_colsqnorm_multi_workspace_bytes.argtypes = [vp, ci]
    lib.ecoflap_colsqnorm_accum_multi.argtypes = [vp, ci, ci, vp, sz, vp]
    lib.ecoflap_colsq_replay.argtypes = [vp, vp, vp, ci, i64, i64, i64, vp]
    lib.ecoflap_wanda_workspace_bytes.restype = sz
    lib.ecoflap_wanda_workspace_bytes.argtypes = [i64, i64]
    lib.ecoflap_wanda_prune_rows.argtypes = [vp, vp, i64, i64, ci, i64, vp, vp, sz, vp]
    lib.ecoflap_wanda_prune_matrix.argtypes = [vp, vp, i64, i64, ci, i64, vp, vp, sz, vp]
    lib.ecoflap_wanda_prune_nm.argtypes = [vp, vp, i64, i64, ci, ci, ci, vp, vp]
    lib.ecoflap_wanda_block_workspace_bytes.restype = sz
    lib.ecoflap_wanda_block_workspace_bytes.argtypes = [vp, ci]
    lib.ecoflap_wanda_prune_block.argtypes = [vp, ci, vp, sz, vp]
    lib.ecoflap_wanda_fallback_counts.argtypes = [vp, ci]
    lib.ecoflap_mask_mul.argtypes = [vp, vp, i64, ci, vp]
    lib.ecoflap_grad_accum_multi.argtypes = [vp, ci, vp]
    lib.ecoflap_global_prune_workspace_bytes.restype = sz
    lib.ecoflap_global_threshold_prune.argtypes = [vp, ci, ci, f32, i64, i64, vp, sz, vp]
    lib.ecoflap_global_prune_protected_workspace_bytes.restype = sz
    lib.ecoflap_global_prune_protected_workspace_bytes.argtypes = [ci]
    lib.ecoflap_global_threshold_prune_protected.argtypes = [vp, ci, ci, f32, i64, i64, vp, vp, sz, vp]
    lib.ecoflap_count_zeros_multi.argtypes = [vp, ci, vp, vp]
    lib.ecoflap_sparsegpt_workspace_bytes.restype = sz
    lib.ecoflap_cholesky_workspace_bytes.restype = sz
    lib.ecoflap_cholesky_workspace_bytes.argtypes = []
    lib.ecoflap_cholesky_f32.argtypes = [vp, i64, i64, ci, vp, vp, sz, vp]
    lib.ecoflap_cholesky_inverse_workspace_bytes.restype = sz
    lib.ecoflap_cholesky_inverse_workspace_bytes.argtypes = [i64]
    lib.ecoflap_cholesky_inverse_f32.argtypes = [vp, i64, i64, vp, i64, vp, sz, vp]
    lib.ecoflap_sparsegpt_block.argtypes = [vp, i64, i64, vp, i64, i64, ci, i64, vp, vp, vp, vp, sz, vp]
    lib.ecoflap_sparsegpt_block_nm.argtypes = [vp, i64, i64, vp, i64, i64, ci, ci, ci, vp, vp, vp]
    lib.ecoflap_hessian_workspace_bytes.restype = sz
    lib.ecoflap_hessian_workspace_bytes.argtypes = [i64, i64]
    lib.ecoflap_hessian_accum.argtypes = [vp, vp, i64, i64, ci, i64, i64, vp, sz, vp]
    lib.ecoflap_allocate_sparsity.argtypes = [vp, vp, ci, i64, f64, vp, vp]
    _lib = lib
    return lib


def _check(rc, what):
    if rc != 0:
        msg = load_library().ecoflap_error_string(rc).decode()
        raise EcoflapHipError(f"{what} failed: {msg} (code {rc})")


def _gpu(t, name):
    if not isinstance(t, torch.Tensor) or t.device.type != "cuda":
        raise EcoflapHipError(
            f"{name} must be a GPU tensor: the ECoFLaP kernels are HIP-only (no CPU fallback)")
    if not t.is_contiguous():
        raise EcoflapHipError(f"{name} must be contiguous")
    if t.dtype not in DTYPE_CODE and t.dtype not in (torch.uint8, torch.int32, torch.int64,
                                                     torch.float64):
        raise EcoflapHipError(f"{name}: unsupported dtype {t.dtype}")
    return t


def _stream():
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


def _ptr(t):
    return ctypes.c_void_p(t.data_ptr()) if t is not None else None


class Workspace:
    """Caller-owned scratch, grown on demand and reused (no allocation inside launches).
    zeroed=True: filled with zeros when (re)allocated, for kernels that keep self-resetting
    ticket counters in it (K6) — such a workspace is not shared with other kernels."""

    def __init__(self, zeroed=False):
        self.buf = None
        self.zeroed = zeroed

    def get(self, nbytes, device):
        nbytes = max(int(nbytes), 256)
        if self.buf is None or self.buf.numel() < nbytes or self.buf.device != device:
            alloc = torch.zeros if self.zeroed else torch.empty
            self.buf = alloc(nbytes, dtype=torch.uint8, device=device)
        return self.buf


TORCH_Z = "torch"      # z argument of the K1 methods: torch.normal's own device stream, in registers


class HipKernels:
    """The HIP backend behind LayerSparsity / the Wanda pruners."""

    name = "hip"

    def __init__(self):
        self.lib = load_library()
        self.ws = Workspace()
        self.k6_ws = Workspace(zeroed=True)
        self.k6_multi_ws = Workspace(zeroed=True)     # its own layout: never shared with k6_ws
        self.hess_ws = Workspace()
        self._pair_tables = {}

    PAIR_TABLES_KEPT = 8      # >= the first-order loop's lanes (default 3), each with its own buffers

    def _pointer_table(self, kind, rows, device):
        """(host, device) int64 tables of `rows`, cached per distinct row set (the lanes of the
        first-order loop rotate: each has its own static gradient buffers, so consecutive calls
        alternate between a few tables).  A new table goes up from pinned memory without
        blocking the host, which has other lanes' replays to queue."""
        key = (kind, rows, device)
        hit = self._pair_tables.pop(key, None)
        if hit is None:
            host = torch.tensor(rows, dtype=torch.int64)
            if device.type == "cuda":
                host = host.pin_memory()
            hit = (host, host.to(device, non_blocking=True))
            while len(self._pair_tables) >= self.PAIR_TABLES_KEPT:
                self._pair_tables.pop(next(iter(self._pair_tables)))      # least recently used
        self._pair_tables[key] = hit          # most recently used last
        return hit

    # ---- K1 ---------------------------------------------------------------------------
    def zo_perturb(self, w, scaling_factor, zo_eps, seed, z=None):
        _gpu(w, "w")
        if isinstance(z, str):
            assert z == TORCH_Z
            _check(self.lib.ecoflap_zo_perturb_torch(
                _ptr(w), w.numel(), DTYPE_CODE[w.dtype], float(scaling_factor), float(zo_eps),
                int(seed), self.torch_normal_threads(w.numel(), w.device), _stream()),
                "ecoflap_zo_perturb_torch")
            return
        if z is not None:
            _gpu(z, "z")
            if z.dtype != w.dtype or z.numel() != w.numel():
                raise EcoflapHipError("z must match w in dtype and numel")
        _check(self.lib.ecoflap_zo_perturb(_ptr(w), w.numel(), DTYPE_CODE[w.dtype],
                                           float(scaling_factor), float(zo_eps), int(seed),
                                           _ptr(z), _stream()), "ecoflap_zo_perturb")

    def zo_perturb_triple(self, w_in, w_plus, w_minus, w_restored, zo_eps, seed, z=None):
        for t, n in ((w_in, "w_in"), (w_plus, "w_plus"), (w_minus, "w_minus"),
                     (w_restored, "w_restored")):
            if t is None and n in ("w_plus", "w_minus"):
                continue                      # drift-only form
            _gpu(t, n)
            if t.dtype != w_in.dtype or t.numel() != w_in.numel():
                raise EcoflapHipError("triple buffers must match w_in in dtype and numel")
        if isinstance(z, str):
            assert z == TORCH_Z
            self.zo_perturb_layers_torch([(w_in, w_restored, [seed], [w_plus], [w_minus])], zo_eps)
            return
        if z is not None:
            _gpu(z, "z")
        _check(self.lib.ecoflap_zo_perturb_triple(
            _ptr(w_in), _ptr(w_plus), _ptr(w_minus), _ptr(w_restored), w_in.numel(),
            DTYPE_CODE[w_in.dtype], float(zo_eps), int(seed), _ptr(z), _stream()),
            "ecoflap_zo_perturb_triple")

    MAX_UNITS = 32

    def zo_perturb_units(self, w, zo_eps, seeds, w_plus, w_minus, z=None, events=None):
        """All units of one layer in one launch (chunks of MAX_UNITS); w is updated in place
        to the final drifted weights, unit u's theta+/theta- land in w_plus[u]/w_minus[u]
        (None, None = drift only).  events: optional callable -> (start, stop) raw hipEvent_t
        handles per launch (ecoflap_zo_perturb_units_timed; bench.py's roofline leg)."""
        _gpu(w, "w")
        n_units = len(seeds)
        assert len(w_plus) == n_units and len(w_minus) == n_units
        if isinstance(z, str):
            assert z == TORCH_Z
            for c0 in range(0, n_units, self.MAX_UNITS):
                c1 = min(n_units, c0 + self.MAX_UNITS)
                self.zo_perturb_layers_torch([(w, w, seeds[c0:c1], w_plus[c0:c1], w_minus[c0:c1])],
                                             zo_eps, events=events)
            return
        for t in list(w_plus) + list(w_minus) + (list(z) if z is not None else []):
            if t is not None:
                _gpu(t, "unit buffer")
                if t.dtype != w.dtype or t.numel() != w.numel():
                    raise EcoflapHipError("unit buffers must match w in dtype and numel")
        for c0 in range(0, n_units, self.MAX_UNITS):
            c1 = min(n_units, c0 + self.MAX_UNITS)
            m = c1 - c0
            seeds_a = (ctypes.c_uint64 * m)(*[int(x) for x in seeds[c0:c1]])
            plus_a = (ctypes.c_void_p * m)(*[t.data_ptr() if t is not None else None
                                             for t in w_plus[c0:c1]])
            minus_a = (ctypes.c_void_p * m)(*[t.data_ptr() if t is not None else None
                                              for t in w_minus[c0:c1]])
            z_a = None
            if z is not None:
                z_a = (ctypes.c_void_p * m)(*[t.data_ptr() for t in z[c0:c1]])
            if events is not None:
                ev_start, ev_stop = events()
                _check(self.lib.ecoflap_zo_perturb_units_timed(
                    _ptr(w), w.numel(), DTYPE_CODE[w.dtype], float(zo_eps), m, seeds_a, plus_a,
                    minus_a, z_a, _stream(), ev_start, ev_stop), "ecoflap_zo_perturb_units_timed")
                continue
            _check(self.lib.ecoflap_zo_perturb_units(
                _ptr(w), w.numel(), DTYPE_CODE[w.dtype], float(zo_eps), m, seeds_a, plus_a,
                minus_a, z_a, _stream()), "ecoflap_zo_perturb_units")

    def zo_perturb_layers(self, layers, zo_eps, events=None):
        """Block-batched K1: layers = [(w_in, w_final, seeds, w_plus, w_minus)] (in-register z)
        or [(w_in, w_final, seeds, w_plus, w_minus, zs)] (z supplied per unit: the parity mode),
        all of one dtype and one of the two forms, each with at most MAX_UNITS units (None, None
        = drift only); one launch for all of them, drifted weights into w_final (w_in is left
        untouched).  events: optional callable -> (start, stop) raw hipEvent_t handles."""
        if len(layers[0]) > 5 and isinstance(layers[0][5], str):
            assert all(item[5] == TORCH_Z for item in layers)
            return self.zo_perturb_layers_torch([item[:5] for item in layers], zo_eps, events=events)
        has_z = len(layers[0]) > 5 and layers[0][5] is not None
        U = self.MAX_UNITS
        row_len = 5 + (4 if has_z else 3) * U
        dt = layers[0][0].dtype
        rows, total = [], 0
        for item in layers:
            w_in, w_final, seeds, w_plus, w_minus = item[:5]
            zs = item[5] if len(item) > 5 else None
            if (zs is not None) != has_z:
                raise EcoflapHipError("zo_perturb_layers: z for every layer of the launch or for none")
            _gpu(w_in, "w_in")
            _gpu(w_final, "w_final")
            n_units = len(seeds)
            if n_units > U or len(w_plus) != n_units or len(w_minus) != n_units:
                raise EcoflapHipError("zo_perturb_layers: at most MAX_UNITS units per layer")
            if w_in.dtype != dt or w_final.dtype != dt or w_final.numel() != w_in.numel():
                raise EcoflapHipError("zo_perturb_layers: one dtype per launch, w_final like w_in")
            if w_final.data_ptr() == w_in.data_ptr():
                raise EcoflapHipError("zo_perturb_layers: w_final must not alias w_in")
            for t in list(w_plus) + list(w_minus):
                if t is not None:
                    _gpu(t, "unit buffer")
                    if t.dtype != dt or t.numel() != w_in.numel():
                        raise EcoflapHipError("unit buffers must match w_in in dtype and numel")
            if has_z:
                if len(zs) != n_units:
                    raise EcoflapHipError("zo_perturb_layers: one z per unit")
                for z in zs:
                    _gpu(z, "z")
                    if z.dtype != dt or z.numel() != w_in.numel() or z.data_ptr() % 16:
                        raise EcoflapHipError("z must match w_in in dtype and numel, 16-byte aligned")
            per_vec = 16 // w_in.element_size()
            row = [w_in.data_ptr(), w_final.data_ptr(), w_in.numel(), n_units, total]
            row += [(int(x) & (2 ** 64 - 1)) - (2 ** 64 if (int(x) & (2 ** 63)) else 0) for x in seeds]
            row += [0] * (U - n_units)
            row += [t.data_ptr() if t is not None else 0 for t in w_plus] + [0] * (U - n_units)
            row += [t.data_ptr() if t is not None else 0 for t in w_minus] + [0] * (U - n_units)
            if has_z:
                row += [z.data_ptr() for z in zs] + [0] * (U - n_units)
            assert len(row) == row_len
            rows.append(row)
            total += max(1, -(-(w_in.numel() // per_vec) // 128))
        # pinned staging + asynchronous copy: a pageable host-to-device copy would make the host
        # wait here for everything queued on the stream before it (the caching host allocator
        # keeps the pinned block alive until the copy has run)
        table = torch.tensor(rows, dtype=torch.int64).pin_memory().to(layers[0][0].device,
                                                                      non_blocking=True)
        ev = events() if events is not None else (None, None)
        fn = self.lib.ecoflap_zo_perturb_layers_z if has_z else self.lib.ecoflap_zo_perturb_layers
        _check(fn(_ptr(table), len(rows), total, DTYPE_CODE[dt], float(zo_eps), _stream(),
                  ev[0], ev[1]),
               "ecoflap_zo_perturb_layers_z" if has_z else "ecoflap_zo_perturb_layers")

    # ---- K1 with the reference's own draw (torch.normal's device stream) in registers ---------
    def torch_normal_threads(self, n, device):
        """Threads of the launch torch.normal(size = n) makes on `device` (ATen's
        calc_execution_policy): the element -> (Philox subsequence, round, word) map hangs on it."""
        key = torch.device(device).index
        props = getattr(self, "_dev_props", None)
        if props is None:
            props = self._dev_props = {}
        if key not in props:
            pr = torch.cuda.get_device_properties(device)
            props[key] = (int(pr.multi_processor_count), int(pr.max_threads_per_multi_processor))
        return int(self.lib.ecoflap_torch_normal_threads(int(n), *props[key]))

    def zo_fill_normal_torch(self, z_out, seed):
        """z_out <- what torch.manual_seed(seed); torch.normal(0, 1, z_out.shape, dtype) returns on
        this device (regenerated by this library, not by torch)."""
        _gpu(z_out, "z_out")
        _check(self.lib.ecoflap_zo_fill_normal_torch(
            _ptr(z_out), z_out.numel(), DTYPE_CODE[z_out.dtype], int(seed),
            self.torch_normal_threads(z_out.numel(), z_out.device), _stream()),
            "ecoflap_zo_fill_normal_torch")

    def torch_stream_matches(self, device):
        """Does `zo_fill_normal_torch` reproduce THIS torch's torch.normal on THIS device bit for
        bit?  Checked once per device and process (three dtypes, a ragged single-round size and
        a size that spans rounds, two seeds); torch's generators are left as they were found."""
        key = ("torch_stream", torch.device(device).index)
        cache = getattr(self, "_probe_cache", None)
        if cache is None:
            cache = self._probe_cache = {}
        if key in cache:
            return cache[key]
        # torch.manual_seed reseeds the CPU generator and the generator of EVERY GPU: all of them
        # are put back (a single-process multi-GPU caller's other streams included)
        cpu_state = torch.get_rng_state()
        gpu_states = torch.cuda.get_rng_state_all()
        ok = True
        try:
            t_full = self.torch_normal_threads(1 << 40, device)
            for dt in (torch.float32, torch.float16, torch.bfloat16):
                for n, seed in ((1003, 123456789), (4 * t_full + 4104, 987654321)):
                    torch.manual_seed(seed)
                    want = torch.normal(mean=0, std=1, size=(n,), device=device, dtype=dt)
                    got = torch.empty_like(want)
                    self.zo_fill_normal_torch(got, seed)
                    ok = ok and bool(torch.equal(want.view(torch.int16 if dt != torch.float32 else torch.int32),
                                                 got.view(torch.int16 if dt != torch.float32 else torch.int32)))
        finally:
            torch.set_rng_state(cpu_state)
            torch.cuda.set_rng_state_all(gpu_states)
        cache[key] = ok
        return ok

    def torch_radius_sweep(self, first_word=0, n_words=1 << 32, device=None):
        """How many 32-bit words of [first_word, first_word + n_words) give a Box-Muller radius
        that differs in any bit from rocRAND's own instruction sequence (the kernels compute it
        with a shorter one; include/ecoflap_hip.h).  The whole domain by default."""
        dev = torch.device("cuda", torch.cuda.current_device()) if device is None else torch.device(device)
        differ = torch.zeros(1, dtype=torch.int64, device=dev)
        with torch.cuda.device(dev):
            _check(self.lib.ecoflap_zo_torch_radius_sweep(int(first_word), int(n_words), _ptr(differ), _stream()),
                   "ecoflap_zo_torch_radius_sweep")
        return int(differ.item())

    def zo_perturb_layers_torch(self, layers, zo_eps, events=None):
        """Block-batched K1 with z = torch.normal's draw regenerated in registers: layers =
        [(w_in, w_final, seeds, w_plus, w_minus)], one dtype, at most MAX_UNITS units per layer;
        w_final may be w_in, a unit's w_plus may alias w_in."""
        U = self.MAX_UNITS
        row_len = 6 + 3 * U
        dt = layers[0][0].dtype
        rows, total = [], 0
        for w_in, w_final, seeds, w_plus, w_minus in layers:
            _gpu(w_in, "w_in")
            _gpu(w_final, "w_final")
            n_units = len(seeds)
            if n_units > U or len(w_plus) != n_units or len(w_minus) != n_units:
                raise EcoflapHipError("zo_perturb_layers_torch: at most MAX_UNITS units per layer")
            if w_in.dtype != dt or w_final.dtype != dt or w_final.numel() != w_in.numel():
                raise EcoflapHipError("zo_perturb_layers_torch: one dtype per launch, w_final like w_in")
            if w_in.data_ptr() % 16 or w_final.data_ptr() % 16:
                raise EcoflapHipError("weights must be 16-byte aligned")
            for t in list(w_plus) + list(w_minus):
                if t is not None:
                    _gpu(t, "unit buffer")
                    if t.dtype != dt or t.numel() != w_in.numel() or t.data_ptr() % 16:
                        raise EcoflapHipError("unit buffers must match w_in in dtype and numel, 16-byte aligned")
            for a, b in zip(w_plus, w_minus):
                if (a is None) != (b is None):
                    raise EcoflapHipError("theta+ and theta- of a unit: both or neither")
            n = w_in.numel()
            if n >= 2 ** 31:       # (torch draws such a tensor in several launches at advancing offsets)
                raise EcoflapHipError("zo_perturb_layers_torch: tensors of 2^31 elements or more take the "
                                      "materialised draw")
            threads = self.torch_normal_threads(n, w_in.device)
            items = int(self.lib.ecoflap_torch_layer_items(n, threads, DTYPE_CODE[dt]))
            if items <= 0:
                raise EcoflapHipError("zo_perturb_layers_torch: no work items for this tensor / thread count")
            row = [w_in.data_ptr(), w_final.data_ptr(), n, n_units, total, threads]
            row += [(int(x) & (2 ** 64 - 1)) - (2 ** 64 if (int(x) & (2 ** 63)) else 0) for x in seeds]
            row += [0] * (U - n_units)
            row += [t.data_ptr() if t is not None else 0 for t in w_plus] + [0] * (U - n_units)
            row += [t.data_ptr() if t is not None else 0 for t in w_minus] + [0] * (U - n_units)
            assert len(row) == row_len
            rows.append(row)
            total += items
        table = torch.tensor(rows, dtype=torch.int64).pin_memory().to(layers[0][0].device,
                                                                      non_blocking=True)
        ev = events() if events is not None else (None, None)
        _check(self.lib.ecoflap_zo_perturb_layers_torch(
            _ptr(table), len(rows), total, DTYPE_CODE[dt], float(zo_eps), _stream(), ev[0], ev[1]),
            "ecoflap_zo_perturb_layers_torch")

    def zo_fill_normal(self, z_out, seed):
        _gpu(z_out, "z_out")
        _check(self.lib.ecoflap_zo_fill_normal(_ptr(z_out), z_out.numel(),
                                               DTYPE_CODE[z_out.dtype], int(seed), _stream()),
               "ecoflap_zo_fill_normal")

    def philox_u32(self, n, seed, device="cuda"):
        out = torch.empty(n, dtype=torch.int32, device=device)
        _check(self.lib.ecoflap_philox_u32(_ptr(out), n, int(seed), _stream()),
               "ecoflap_philox_u32")
        return out

    # ---- K3+K4 ------------------------------------------------------------------------
    def absprod_reduce(self, w, g, mode, out_accum):
        """out_accum (float64[1], GPU) += sum_e f(w_e, g_e)."""
        ref = w if w is not None else g
        _gpu(ref, "w/g")
        _gpu(out_accum, "out_accum")
        if out_accum.dtype != torch.float64:
            raise EcoflapHipError("out_accum must be float64")
        n = ref.numel()
        nb = self.lib.ecoflap_absprod_reduce_workspace_bytes(n)
        ws = self.ws.get(nb, ref.device)
        _check(self.lib.ecoflap_absprod_reduce(
            _ptr(_gpu(w, "w")) if w is not None else None,
            _ptr(_gpu(g, "g")) if g is not None else None, n,
            DTYPE_CODE[w.dtype] if w is not None else 0,
            DTYPE_CODE[g.dtype] if g is not None else 0, int(mode), _ptr(out_accum), _ptr(ws),
            ws.numel(), _stream()), "ecoflap_absprod_reduce")

    def absprod_reduce_multi(self, table, max_numel, dtype_w, dtype_g, mode, out_accum):
        """table: int64[n_layers, 3] GPU rows {w_ptr, g_ptr, numel}; out_accum float64[n_layers]."""
        _gpu(table, "table")
        _gpu(out_accum, "out_accum")
        n_layers = table.shape[0]
        nb = self.lib.ecoflap_absprod_reduce_multi_workspace_bytes(n_layers)
        ws = self.ws.get(nb, table.device)
        _check(self.lib.ecoflap_absprod_reduce_multi(
            _ptr(table), n_layers, int(max_numel), DTYPE_CODE[dtype_w], DTYPE_CODE[dtype_g],
            int(mode), _ptr(out_accum), _ptr(ws), ws.numel(), _stream()),
            "ecoflap_absprod_reduce_multi")

    def absprod_reduce_pairs(self, weights, grads, mode, out_accum):
        """out_accum[l] += sum_e f(weights[l], grads[l]) for every layer l: ONE multi-tensor
        launch for the whole model, whatever its mix of dtypes (ecoflap_absprod_reduce_mixed)."""
        _gpu(out_accum, "out_accum")
        if out_accum.dtype != torch.float64 or out_accum.numel() != len(weights):
            raise EcoflapHipError("out_accum must be float64[len(weights)]")
        keep_alive, rows = [], []
        for w, g in zip(weights, grads):
            if not g.is_contiguous():
                g = g.contiguous()
                keep_alive.append(g)
            _gpu(w, "w")
            _gpu(g, "g")
            if g.numel() != w.numel():
                raise EcoflapHipError("gradient and weight differ in size")
            dw, dg = DTYPE_CODE[w.dtype], DTYPE_CODE[g.dtype]
            if mode >= 3:            # ABSW / SQW: g unused
                dg = dw
            elif mode == 2:          # ABSG: w unused
                dw = dg
            rows.append((w.data_ptr(), g.data_ptr(), w.numel(), dw | (dg << 8)))
        rows = tuple(rows)
        # the pointer table is the same for every batch when the gradients live in a captured
        # graph's static buffers: building it afresh is a blocking H2D copy per batch that
        # stalls the host behind the replay it has just queued
        host, table = self._pointer_table("mixed", rows, out_accum.device)
        nb = self.lib.ecoflap_absprod_reduce_multi_workspace_bytes(len(rows))
        ws = self.ws.get(nb, out_accum.device)
        _check(self.lib.ecoflap_absprod_reduce_mixed(
            _ptr(table), host.data_ptr(), len(rows), int(mode), _ptr(out_accum), _ptr(ws), ws.numel(),
            _stream()), "ecoflap_absprod_reduce_mixed")

    # ---- K6 ---------------------------------------------------------------------------
    def colsqnorm_accum(self, scaler_row, x2d, nsamples_before, batch):
        _gpu(scaler_row, "scaler_row")
        _gpu(x2d, "x")
        tokens, cols = x2d.shape
        nb = self.lib.ecoflap_colsqnorm_workspace_bytes(tokens, cols)
        ws = self.k6_ws.get(nb, x2d.device)      # zeroed when (re)allocated: K6's ticket counters
        _check(self.lib.ecoflap_colsqnorm_accum(
            _ptr(scaler_row), _ptr(x2d), tokens, cols, DTYPE_CODE[x2d.dtype],
            int(nsamples_before), int(batch), _ptr(ws), ws.numel(), _stream()),
            "ecoflap_colsqnorm_accum")

    graph_safe = True     # launches only (no host copies / syncs): usable under HIP-graph capture

    def colsqnorm_workspace(self, tokens, cols, device):
        """A private workspace for `colsqnorm_accum_dev` (a captured graph keeps its address)."""
        nb = self.lib.ecoflap_colsqnorm_workspace_bytes(tokens, cols)
        return torch.zeros(max(int(nb), 16), dtype=torch.uint8, device=device)    # tickets start at 0

    def colsqnorm_accum_dev(self, scaler_row, x2d, n_dev, batch, ws):
        _gpu(scaler_row, "scaler_row")
        _gpu(x2d, "x")
        tokens, cols = x2d.shape
        _check(self.lib.ecoflap_colsqnorm_accum_dev(
            _ptr(scaler_row), _ptr(x2d), tokens, cols, DTYPE_CODE[x2d.dtype], _ptr(n_dev),
            int(batch), _ptr(ws), ws.numel(), _stream()), "ecoflap_colsqnorm_accum_dev")

    def colsqnorm_accum_multi(self, items, ws=None):
        """ONE launch for all hooked inputs of a block (one calibration sample).
        items: [(scaler_row, x2d, nsamples_before, n_dev or None, batch, raw)], one dtype;
        ws: private workspace (captured graphs keep its address) or None = a cached one.
        -> the workspace used."""
        out = None
        by_dtype = {}
        for it in items:
            by_dtype.setdefault(it[1].dtype, []).append(it)
        for dt, group in by_dtype.items():
            for g0 in range(0, len(group), COLSQ_MAX_ITEMS):
                chunk = group[g0:g0 + COLSQ_MAX_ITEMS]
                # the records are written with ONE struct.pack_into each (a decoder block's 11
                # inputs: 6 us of marshalling instead of 24 through ctypes field assignments —
                # more than the launch's own 7 us on the device when the call is not replayed
                # from a graph)
                arr = _colsq_array(len(chunk))
                off = 0
                for row, x2d, n_before, n_dev, batch, raw in chunk:
                    if not (row.is_cuda and x2d.is_cuda):
                        _gpu(row, "scaler_row")
                        _gpu(x2d, "x")
                    if not (row.is_contiguous() and x2d.is_contiguous()) or x2d.dim() != 2:
                        raise EcoflapHipError("scaler_row / x must be contiguous, x two-dimensional")
                    tokens, cols = x2d.shape
                    _COLSQ_PACK(arr, off, row.data_ptr(), x2d.data_ptr(), tokens, cols, int(n_before),
                                0 if n_dev is None else n_dev.data_ptr(), int(batch), 1 if raw else 0)
                    off += _COLSQ_BYTES
                nb = self.lib.ecoflap_colsqnorm_multi_workspace_bytes(arr, len(chunk))
                if ws is not None and len(by_dtype) == 1 and len(group) <= COLSQ_MAX_ITEMS:
                    use = ws
                else:
                    use = self.k6_multi_ws.get(nb, chunk[0][1].device)
                _check(self.lib.ecoflap_colsqnorm_accum_multi(
                    arr, len(chunk), DTYPE_CODE[dt], _ptr(use), use.numel(), _stream()),
                    "ecoflap_colsqnorm_accum_multi")
                out = use
        return out

    def colsqnorm_multi_workspace(self, items):
        """Private zeroed workspace for `colsqnorm_accum_multi(items, ws)` under graph capture
        (None when the items do not fit one launch: mixed dtypes / more than 16 inputs)."""
        if len({it[1].dtype for it in items}) != 1 or len(items) > COLSQ_MAX_ITEMS:
            return None
        arr = (ColsqItem * len(items))()
        for slot, (row, x2d, *_rest) in zip(arr, items):
            slot.tokens, slot.cols = x2d.shape
        nb = self.lib.ecoflap_colsqnorm_multi_workspace_bytes(arr, len(items))
        return torch.zeros(max(int(nb), 16), dtype=torch.uint8, device=items[0][1].device)

    def colsq_replay(self, scaler_row, sq, batches, nsamples_before=0):
        """scaler_row <- the running mean over per-batch statistics sq[j] (rows of a [J, ld]
        fp32 tensor, first `cols` entries used) in order, batch sizes `batches` (list of int)."""
        _gpu(scaler_row, "scaler_row")
        if sq.device.type != "cuda" or sq.dtype != torch.float32 or sq.stride(-1) != 1:
            raise EcoflapHipError("sq must be a GPU fp32 tensor with unit inner stride")
        J, cols = sq.shape
        assert cols == scaler_row.numel() and J == len(batches)
        b = torch.tensor([int(v) for v in batches], dtype=torch.int64).to(sq.device, non_blocking=False)
        _check(self.lib.ecoflap_colsq_replay(_ptr(scaler_row), _ptr(sq), _ptr(b), J, cols,
                                             sq.stride(0), int(nsamples_before), _stream()),
               "ecoflap_colsq_replay")

    # ---- K7 ---------------------------------------------------------------------------
    def _wanda(self, fn, name, w, scaler_row, k, mask_out):
        _gpu(w, "w")
        _gpu(scaler_row, "scaler_row")
        if mask_out is not None:
            _gpu(mask_out, "mask_out")
        rows, cols = w.shape
        nb = self.lib.ecoflap_wanda_workspace_bytes(rows, cols)
        ws = self.ws.get(nb, w.device)
        _check(fn(_ptr(w), _ptr(scaler_row), rows, cols, DTYPE_CODE[w.dtype], int(k),
                  _ptr(mask_out), _ptr(ws), ws.numel(), _stream()), name)

    def wanda_prune_rows(self, w, scaler_row, k, mask_out=None):
        self._wanda(self.lib.ecoflap_wanda_prune_rows, "ecoflap_wanda_prune_rows", w, scaler_row,
                    k, mask_out)

    def wanda_prune_matrix(self, w, scaler_row, k, mask_out=None):
        self._wanda(self.lib.ecoflap_wanda_prune_matrix, "ecoflap_wanda_prune_matrix", w,
                    scaler_row, k, mask_out)

    def wanda_prune_nm(self, w, scaler_row, n, m, mask_out=None):
        """Structured n:m selection (wanda_pruner.py:265-270): in every group of m consecutive
        columns of a row the n smallest metrics are zeroed, in place."""
        _gpu(w, "w")
        _gpu(scaler_row, "scaler_row")
        if scaler_row.dtype != torch.float32:
            raise EcoflapHipError("scaler_row must be float32")
        if mask_out is not None:
            _gpu(mask_out, "mask_out")
        rows, cols = w.shape
        _check(self.lib.ecoflap_wanda_prune_nm(_ptr(w), _ptr(scaler_row), rows, cols, DTYPE_CODE[w.dtype],
                                               int(n), int(m), _ptr(mask_out) if mask_out is not None else None,
                                               _stream()), "ecoflap_wanda_prune_nm")

    def wanda_fallback_counts(self, reset=True):
        """(bracket misses, crowded bins): matrices the matrix-mode selection handed to its exact
        on-device fallback since the last reset (include/ecoflap_hip.h).  Synchronises."""
        out = (ctypes.c_uint * 4)()
        _check(self.lib.ecoflap_wanda_fallback_counts(out, 1 if reset else 0), "ecoflap_wanda_fallback_counts")
        return int(out[1]) + int(out[3]), int(out[2])

    def wanda_prune_block(self, items):
        """items: [(w, scaler_row, mode, k, mask_out_or_None)], mode "rows" / "matrix": all
        the Linears of one transformer block, pruned by shared launches
        (ecoflap_wanda_prune_block); same results as one call per matrix."""
        for c0 in range(0, len(items), WANDA_MAX_ITEMS):
            chunk = items[c0:c0 + WANDA_MAX_ITEMS]
            arr = (WandaItem * len(chunk))()
            for a, (w, scaler_row, mode, k, mask_out) in zip(arr, chunk):
                _gpu(w, "w")
                _gpu(scaler_row, "scaler_row")
                if mask_out is not None:
                    _gpu(mask_out, "mask_out")
                if scaler_row.dtype != torch.float32:
                    raise EcoflapHipError("scaler_row must be float32")
                a.w, a.scaler_row = w.data_ptr(), scaler_row.data_ptr()
                a.rows, a.cols = w.shape
                a.k = int(k)
                a.mask_out = mask_out.data_ptr() if mask_out is not None else None
                a.dtype = DTYPE_CODE[w.dtype]
                a.mode = {"rows": WANDA_ROWS, "matrix": WANDA_MATRIX}[mode]
            nb = self.lib.ecoflap_wanda_block_workspace_bytes(ctypes.byref(arr), len(chunk))
            ws = self.ws.get(nb, chunk[0][0].device)
            _check(self.lib.ecoflap_wanda_prune_block(ctypes.byref(arr), len(chunk), _ptr(ws),
                                                      ws.numel(), _stream()),
                   "ecoflap_wanda_prune_block")

    # ---- Real-* global iterative pruning ---------------------------------------------------------
    def grad_accum_multi(self, accs, grads):
        """accs[l] (fp32) += |grads[l]| for every layer, one launch."""
        rows, keep = [], []
        for a, g in zip(accs, grads):
            _gpu(a, "acc")
            if not g.is_contiguous():
                g = g.contiguous()
                keep.append(g)
            rows.append((a.data_ptr(), _gpu(g, "grad").data_ptr(), a.numel(), DTYPE_CODE[g.dtype]))
        rows = tuple(rows)
        _, table = self._pointer_table("grad_accum", rows, accs[0].device)
        _check(self.lib.ecoflap_grad_accum_multi(_ptr(table), len(rows), _stream()),
               "ecoflap_grad_accum_multi")

    def global_threshold_prune(self, weights, accs, masks, mode, n_batches, k, protect_counts=None):
        """One round of get_mask + `W *= mask` over all layers: masks (uint8) and weights are
        updated in place; k = num_to_zero_out.  protect_counts: per layer num_to_set of get_mask's
        protection step (int(numel * (1 - max_sparsity_per_layer))), None / all 0 = none."""
        if accs is None:             # mode 3: the score is the signed weight itself
            accs = [None] * len(weights)
        rows = [[_gpu(w, "w").data_ptr(), 0 if a is None else _gpu(a, "acc").data_ptr(),
                 _gpu(m, "mask").data_ptr(), w.numel(), DTYPE_CODE[w.dtype]]
                for w, a, m in zip(weights, accs, masks)]
        dev = weights[0].device
        table = torch.tensor(rows, dtype=torch.int64, device=dev)
        total = sum(r[3] for r in rows)
        if total >= 2 ** 32 - 1:
            # histogram bins and the in-block scans of the selection kernels are 32-bit
            raise EcoflapHipError(
                f"global threshold over {total} elements: one selection covers at most 2^32 - 2 "
                "(4 294 967 294) elements (BLIP-2 FlanT5-XL has 3 701 932 032); prune per "
                "sub-model (prune_per_model) or layer-wise, or split the call")
        if protect_counts is not None and any(int(c) > 0 for c in protect_counts):
            ranks = torch.tensor([(r[3] - int(c) + 1) if int(c) > 0 else 0
                                  for r, c in zip(rows, protect_counts)], dtype=torch.int64, device=dev)
            nb = self.lib.ecoflap_global_prune_protected_workspace_bytes(len(rows))
            ws = self.ws.get(nb, dev)
            _check(self.lib.ecoflap_global_threshold_prune_protected(
                _ptr(table), len(rows), int(mode), float(n_batches), int(k), int(total),
                _ptr(ranks), _ptr(ws), ws.numel(), _stream()),
                "ecoflap_global_threshold_prune_protected")
            return
        ws = self.ws.get(self.lib.ecoflap_global_prune_workspace_bytes(), dev)
        _check(self.lib.ecoflap_global_threshold_prune(
            _ptr(table), len(rows), int(mode), float(n_batches), int(k), int(total), _ptr(ws),
            ws.numel(), _stream()), "ecoflap_global_threshold_prune")

    def count_zeros_multi(self, tensors):
        rows = [[_gpu(t, "w").data_ptr(), t.numel(), DTYPE_CODE[t.dtype]] for t in tensors]
        dev = tensors[0].device
        table = torch.tensor(rows, dtype=torch.int64, device=dev)
        out = torch.zeros(len(rows), dtype=torch.int64, device=dev)
        _check(self.lib.ecoflap_count_zeros_multi(_ptr(table), len(rows), _ptr(out), _stream()),
               "ecoflap_count_zeros_multi")
        return out.cpu().tolist()

    # ---- SparseGPT ------------------------------------------------------------------------
    def sparsegpt_block(self, W, Hinv, i1, count, k, err_out, mask_out=None):
        """One <=128-column block of SparseGPT.fasterprune on the fp32 working copy W."""
        _gpu(W, "W"), _gpu(Hinv, "Hinv"), _gpu(err_out, "err_out")
        if W.dtype != torch.float32 or Hinv.dtype != torch.float32:
            raise EcoflapHipError("SparseGPT works on fp32 copies, as the reference does")
        ws = self.ws.get(self.lib.ecoflap_sparsegpt_workspace_bytes(), W.device)
        _check(self.lib.ecoflap_sparsegpt_block(
            _ptr(W), W.shape[0], W.stride(0), _ptr(Hinv), Hinv.stride(0), int(i1), int(count),
            int(k), None, _ptr(err_out), _ptr(mask_out), _ptr(ws), ws.numel(), _stream()),
            "ecoflap_sparsegpt_block")

    def sparsegpt_block_nm(self, W, Hinv, i1, count, n, m, err_out, mask_out=None):
        """The block step under n:m sparsity (sparsegpt_pruner.py:196-198)."""
        _gpu(W, "W"), _gpu(Hinv, "Hinv"), _gpu(err_out, "err_out")
        if W.dtype != torch.float32 or Hinv.dtype != torch.float32:
            raise EcoflapHipError("SparseGPT works on fp32 copies, as the reference does")
        _check(self.lib.ecoflap_sparsegpt_block_nm(
            _ptr(W), W.shape[0], W.stride(0), _ptr(Hinv), Hinv.stride(0), int(i1), int(count),
            int(n), int(m), _ptr(err_out), _ptr(mask_out), _stream()), "ecoflap_sparsegpt_block_nm")

    # ---- K8 ---------------------------------------------------------------------------
    def cholesky(self, H, upper=False):
        """-> (factor, info): torch.linalg.cholesky_ex(H, upper=upper) by this library's blocked fp32
        kernel (csrc/cholesky.hip) on a COPY of H — H itself is left as it is, for the caller's
        damped retry.  info: python int, 0 or the 1-based index of the first non-positive pivot
        (one host sync, as `int(info)` after cholesky_ex is)."""
        if not isinstance(H, torch.Tensor) or H.device.type != "cuda":
            _gpu(H, "H")
        if H.dim() != 2 or H.shape[0] != H.shape[1] or H.dtype != torch.float32:
            raise EcoflapHipError("cholesky: a square fp32 matrix")
        # (torch.cholesky_inverse hands back a column-major tensor: the copy is row-major either way)
        L = H.clone(memory_format=torch.contiguous_format)
        info = torch.empty(1, dtype=torch.int32, device=H.device)
        ws = torch.empty(int(self.lib.ecoflap_cholesky_workspace_bytes()), dtype=torch.uint8, device=H.device)
        _check(self.lib.ecoflap_cholesky_f32(_ptr(L), L.shape[0], L.stride(0), int(bool(upper)), _ptr(info),
                                             _ptr(ws), ws.numel(), _stream()), "ecoflap_cholesky_f32")
        return L, int(info.item())

    def cholesky_inverse(self, L):
        """torch.cholesky_inverse(L) for a lower fp32 factor on the GPU (csrc/cholesky.hip): a new,
        symmetric, row-major tensor."""
        if not isinstance(L, torch.Tensor) or L.device.type != "cuda":
            _gpu(L, "L")
        if L.dim() != 2 or L.shape[0] != L.shape[1] or L.dtype != torch.float32:
            raise EcoflapHipError("cholesky_inverse: a square fp32 matrix")
        Lc = L if L.is_contiguous() else L.contiguous()
        n = Lc.shape[0]
        out = torch.empty((n, n), dtype=torch.float32, device=L.device)
        nb = int(self.lib.ecoflap_cholesky_inverse_workspace_bytes(n))
        ws = torch.empty(max(nb, 16), dtype=torch.uint8, device=L.device)
        _check(self.lib.ecoflap_cholesky_inverse_f32(_ptr(Lc), n, Lc.stride(0), _ptr(out), out.stride(0), _ptr(ws),
                                                     ws.numel(), _stream()), "ecoflap_cholesky_inverse_f32")
        return out

    def hessian_accum(self, H, x2d, nsamples_before, batch):
        """H <- n/(n+b) H + 2/(n+b) x^T x on the matrix cores (fp16 / bf16 x, fp32 H)."""
        _gpu(H, "H")
        _gpu(x2d, "x")
        if H.dtype != torch.float32 or x2d.dtype not in (torch.float16, torch.bfloat16):
            raise EcoflapHipError("hessian_accum: H fp32, x fp16 / bf16")
        tokens, cols = x2d.shape
        if tuple(H.shape) != (cols, cols):
            raise EcoflapHipError("hessian_accum: H must be [cols, cols]")
        nb = self.lib.ecoflap_hessian_workspace_bytes(tokens, cols)
        ws = self.hess_ws.get(nb, x2d.device)
        _check(self.lib.ecoflap_hessian_accum(
            _ptr(H), _ptr(x2d), tokens, cols, DTYPE_CODE[x2d.dtype], int(nsamples_before),
            int(batch), _ptr(ws), ws.numel(), _stream()), "ecoflap_hessian_accum")

    def mask_mul(self, g, keep_mask):
        _gpu(g, "g")
        _gpu(keep_mask, "keep_mask")
        _check(self.lib.ecoflap_mask_mul(_ptr(g), _ptr(keep_mask), g.numel(),
                                         DTYPE_CODE[g.dtype], _stream()), "ecoflap_mask_mul")


def allocate_sparsity(group_scores, group_num_params, total_parameters_to_keep,
                      max_sparsity_per_layer):
    """K5 on the host through the C ABI -> (list of python floats, list of keep counts)."""
    import numpy as np
    lib = load_library()
    sc = np.ascontiguousarray(group_scores, dtype=np.float32)
    nums = np.ascontiguousarray(group_num_params, dtype=np.int64)
    out = np.zeros(len(sc), dtype=np.float32)
    keep = np.zeros(len(sc), dtype=np.float64)
    _check(lib.ecoflap_allocate_sparsity(sc.ctypes.data, nums.ctypes.data, len(sc),
                                         int(total_parameters_to_keep),
                                         float(max_sparsity_per_layer), out.ctypes.data,
                                         keep.ctypes.data), "ecoflap_allocate_sparsity")
    return [float(v) for v in out], [float(v) for v in keep]
