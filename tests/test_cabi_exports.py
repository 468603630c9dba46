"""CPU: the C-ABI library loads and exports every symbol include/ecoflap_hip.h declares
(no compute calls — there is no GPU here), and the product refuses to run without it."""
import ctypes
import os
import re
import sys

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    text = open(os.path.join(ROOT, "include", "ecoflap_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(ecoflap_[a-z0-9_]+)\s*\(", text)))


def test_every_declared_symbol_is_exported():
    from ecoflap_amd import hip
    lib = ctypes.CDLL(hip.LIB_PATH)
    names = declared_symbols()
    assert len(names) >= 17
    for n in names:
        assert hasattr(lib, n), n
    assert sorted(hip.EXPORTS) == names
    shape_ops = re.findall(r"\b(ecoflap_[a-z0-9_]+)\s*\(", re.sub(
        r"/\*.*?\*/", "", open(os.path.join(ROOT, "include", "ecoflap_shape_ops.h")).read(), flags=re.S))
    # plumbing ops of the shape modules: in the same library, except the pinned-solution Linears,
    # which sit in a library of their own (it links hipBLASLt; the pruner ABI library does not)
    gemm = ctypes.CDLL(os.path.join(os.path.dirname(hip.LIB_PATH), "libecoflap_gemm.so"))
    in_gemm = ("ecoflap_linear_pinned", "ecoflap_linear_tune", "ecoflap_linear_library_version")
    for n in shape_ops:
        assert hasattr(gemm if n.startswith(in_gemm) else lib, n), n
    assert len([n for n in shape_ops if n.startswith(in_gemm)]) == 4
    lib.ecoflap_version.restype = ctypes.c_char_p
    assert b"gfx950" in lib.ecoflap_version()


def test_cpu_tensors_are_refused_loudly():
    from ecoflap_amd import hip
    k = hip.HipKernels()
    w = torch.zeros(64)
    with pytest.raises(hip.EcoflapHipError, match="GPU tensor"):
        k.zo_perturb(w, 1, 1e-3, 1)
    with pytest.raises(hip.EcoflapHipError):
        k.wanda_prune_rows(torch.zeros(4, 8), torch.zeros(8), 2)


def test_missing_library_is_an_error(monkeypatch, tmp_path):
    from ecoflap_amd import hip
    monkeypatch.setattr(hip, "_lib", None)
    monkeypatch.setattr(hip, "LIB_PATH", str(tmp_path / "nope.so"))
    with pytest.raises(hip.EcoflapHipError, match="no CPU fallback"):
        hip.HipKernels()


def test_allocator_through_cabi_matches_reference_goldens(golden_dir):
    import numpy as np
    from ecoflap_amd import hip
    g = np.load(os.path.join(golden_dir, "g4_allocator.npz"))
    for i in range(int(g["n"])):
        sp, mx, keep = g[f"{i}_meta"]
        out, _ = hip.allocate_sparsity(g[f"{i}_scores"], g[f"{i}_nums"], int(keep), float(mx))
        assert np.array_equal(np.array(out), g[f"{i}_out"], equal_nan=True), str(g[f"{i}_tag"])


def test_allocator_through_cabi_equals_the_oracle_on_random_problems():
    """400 fresh random allocation problems (group counts 1-400, parameter counts 1e2-1e8, scores
    with ties, zeros, huge ranges and NaN, targets 0.1-0.95, caps 0.5-1.0): the C++ allocator
    behind `ecoflap_allocate_sparsity` == the oracle restatement of
    `compute_the_sparsity_per_group` (UPop/pruners/layer_single_base_pruner.py:247-314), every
    float bit for bit (the oracle itself is pinned to 193 reference cases in g4)."""
    import numpy as np
    from ecoflap_amd import hip
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import allocator as oracle_alloc
    rng = np.random.default_rng(20261004)
    checked = 0
    for case in range(400):
        G = int(rng.integers(1, 400)) if case % 7 else int(rng.integers(1, 4))
        nums = (10 ** rng.uniform(2, 8, size=G)).astype(np.int64)
        kind = case % 6
        if kind == 0:
            scores = rng.random(G)
        elif kind == 1:
            scores = np.round(rng.random(G) * 4) / 4                      # ties and zeros
        elif kind == 2:
            scores = 10 ** rng.uniform(-12, 12, size=G)                   # huge range
        elif kind == 3:
            scores = rng.random(G) * (rng.random(G) < 0.3)                # mostly zero
        elif kind == 4:
            scores = np.full(G, 1.0)
        else:
            scores = rng.random(G)
            scores[rng.integers(0, G)] = np.nan
        scores = scores.astype(np.float32)
        mx = float(rng.choice([0.5, 0.6, 0.8, 0.9, 1.0]))
        target = float(rng.uniform(0.1, min(0.95, mx)))
        keep = int(nums.sum() * (1 - target))
        try:
            want, _ = oracle_alloc.compute_sparsity_per_group(keep, scores, nums, mx, max_iters=2000)
        except RuntimeError:
            continue                                                      # (the reference would loop forever)
        got, _ = hip.allocate_sparsity(scores, nums, keep, mx)
        fg, fw = np.array(got, dtype=np.float32), np.array(want, dtype=np.float32)
        nan = np.isnan(fw)                                                 # (a NaN is a NaN, whatever its sign bit)
        assert np.array_equal(nan, np.isnan(fg)), (case, G, kind, mx, target)
        assert np.array_equal(fg[~nan].view(np.uint32), fw[~nan].view(np.uint32)), (case, G, kind, mx, target)
        checked += 1
    assert checked > 300


def test_argument_errors_do_not_touch_the_gpu():
    from ecoflap_amd import hip
    lib = hip.load_library()
    assert lib.ecoflap_zo_perturb(None, 16, 0, 1.0, 1e-3, 1, None, None) == -2      # ENULL
    assert lib.ecoflap_zo_perturb(ctypes.c_void_p(16), 16, 7, 1.0, 1e-3, 1, None, None) == -1
    assert lib.ecoflap_zo_perturb(ctypes.c_void_p(8), 16, 0, 1.0, 1e-3, 1, None, None) == -5
    assert lib.ecoflap_zo_perturb(None, 0, 0, 1.0, 1e-3, 1, None, None) == 0        # empty input
    assert lib.ecoflap_wanda_prune_matrix(ctypes.c_void_p(16), ctypes.c_void_p(16), 4, 4, 0, 16,
                                          None, ctypes.c_void_p(16), 1 << 20, None) == -3
    assert b"aligned" in lib.ecoflap_error_string(-5)


def test_argument_errors_of_the_later_entry_points():
    """Validation comes before any launch: the multi-tensor / global-threshold / device-counter
    entry points reject bad arguments with the documented codes without a GPU."""
    from ecoflap_amd import hip
    lib = hip.load_library()
    vp = ctypes.c_void_p
    assert lib.ecoflap_grad_accum_multi(None, 0, None) == 0                       # no layers
    assert lib.ecoflap_grad_accum_multi(None, 3, None) == -2                      # ENULL
    assert lib.ecoflap_grad_accum_multi(None, -1, None) == -3                     # ESIZE
    ws = lib.ecoflap_global_prune_workspace_bytes()
    assert ws >= 3 * 2048 * 4
    args = lambda **kw: [kw.get("table", vp(16)), kw.get("n", 2), kw.get("mode", 0), 3.0,   # noqa: E731
                         kw.get("k", 5), kw.get("total", 100), kw.get("ws", vp(16)),
                         kw.get("ws_bytes", ws), None]
    assert lib.ecoflap_global_threshold_prune(*args(mode=4)) == -4                # EMODE
    assert lib.ecoflap_global_threshold_prune(*args(k=0)) == -3                   # rank out of range
    assert lib.ecoflap_global_threshold_prune(*args(k=101)) == -3
    assert lib.ecoflap_global_threshold_prune(*args(total=1 << 33)) == -3         # 32-bit bins
    assert lib.ecoflap_global_threshold_prune(*args(table=None)) == -2
    assert lib.ecoflap_global_threshold_prune(*args(ws_bytes=16)) == -6           # EWORKSPACE
    assert lib.ecoflap_count_zeros_multi(None, 0, None, None) == 0
    assert lib.ecoflap_count_zeros_multi(vp(16), 2, None, None) == -2
    # K6 with the sample count on the device: the counter pointer is mandatory
    assert lib.ecoflap_colsqnorm_accum_dev(vp(16), vp(16), 8, 8, 0, None, 1, vp(16), 1 << 20, None) == -2
    assert lib.ecoflap_colsqnorm_accum_dev(vp(16), vp(16), 8, 8, 0, vp(16), 0, vp(16), 1 << 20, None) == -3
    assert lib.ecoflap_colsqnorm_accum_dev(vp(16), vp(16), 8, 8, 0, vp(16), 1, vp(16), 4, None) == -6
    # plumbing ops of the shape modules
    lib.ecoflap_add_layernorm.argtypes = [vp, vp, vp, vp, vp, vp, ctypes.c_int64, ctypes.c_int64,
                                          ctypes.c_float, ctypes.c_int, vp]
    assert lib.ecoflap_add_layernorm(vp(16), None, vp(16), vp(16), None, vp(16), 4, 12, 1e-6, 1, None) == -3
    assert lib.ecoflap_add_layernorm(vp(16), None, vp(16), vp(16), None, vp(16), 4, 16, 1e-6, 0, None) == -1
    assert lib.ecoflap_add_layernorm(vp(16), vp(16), vp(16), vp(16), None, vp(16), 4, 16, 1e-6, 1, None) == -2


def test_argument_errors_of_the_round2_entry_points():
    """Block-level K7, the MFMA Hessian, the protected global threshold and the timed K1 launch:
    bad arguments are rejected with the documented codes before anything is launched."""
    from ecoflap_amd import hip
    lib = hip.load_library()
    vp = ctypes.c_void_p

    def items(**kw):
        arr = (hip.WandaItem * 2)()
        for a in arr:
            a.w, a.scaler_row, a.rows, a.cols, a.k, a.mask_out, a.dtype, a.mode = (
                16, 16, 4, 8, 2, None, 2, hip.WANDA_ROWS)
        for k, v in kw.items():
            setattr(arr[1], k, v)
        return arr

    ok = items()
    need = lib.ecoflap_wanda_block_workspace_bytes(ctypes.byref(ok), 2)
    assert need >= 2 * (256 + 3 * 2048 * 4)
    call = lambda arr, n=2, ws=vp(16), nb=need: lib.ecoflap_wanda_prune_block(   # noqa: E731
        ctypes.byref(arr) if arr is not None else None, n, ws, nb, None)
    assert call(ok, n=0) == 0                                                   # empty block
    assert call(ok, n=hip.WANDA_MAX_ITEMS + 1) == -3
    assert call(None) == -2
    assert call(items(dtype=9)) == -1
    assert call(items(mode=5)) == -4
    assert call(items(k=-1)) == -3
    assert call(items(cols=20000)) == -3                                         # rows-mode width limit
    assert call(items(mode=hip.WANDA_MATRIX, k=32)) == -3                        # sorted[k] out of range
    assert call(items(w=None)) == -2
    assert call(ok, ws=None) == -2
    assert call(ok, nb=64) == -6
    # MFMA Hessian: 16-bit activations only (fp32 keeps the library GEMM)
    hws = lib.ecoflap_hessian_workspace_bytes(100, 256)
    assert hws == 256 * 128 * 2                                                  # Xt [C, 128] 16-bit
    h = lambda **kw: lib.ecoflap_hessian_accum(                                  # noqa: E731
        kw.get("H", vp(16)), kw.get("x", vp(16)), kw.get("tokens", 100), 256, kw.get("dt", 2),
        kw.get("n", 0), kw.get("b", 8), kw.get("ws", vp(16)), kw.get("nb", hws), None)
    assert h(dt=0) == -1
    assert h(tokens=0) == -3 and h(b=0) == -3 and h(n=-1) == -3
    assert h(H=None) == -2 and h(ws=None) == -2
    assert h(ws=vp(8)) == -5
    assert h(nb=16) == -6
    # protected global threshold
    pws = lib.ecoflap_global_prune_protected_workspace_bytes(3)
    assert pws >= 4 * 3 * 2048 * 4
    p = lambda **kw: lib.ecoflap_global_threshold_prune_protected(               # noqa: E731
        kw.get("table", vp(16)), 3, kw.get("mode", 0), 2.0, kw.get("k", 5), 100,
        kw.get("ranks", vp(16)), vp(16), kw.get("nb", pws), None)
    assert p(mode=7) == -4 and p(k=0) == -3 and p(ranks=None) == -2 and p(nb=1024) == -6
    # timed K1: the event pair is mandatory
    seeds = (ctypes.c_uint64 * 1)(1)
    ptrs = (ctypes.c_void_p * 1)(32)
    assert lib.ecoflap_zo_perturb_units_timed(vp(16), 64, 2, 1e-3, 1, seeds, ptrs, ptrs, None, None,
                                              None, vp(16)) == -2
    assert lib.ecoflap_zo_perturb_units(vp(16), 1 << 41, 2, 1e-3, 1, seeds, ptrs, ptrs, None, None) == -3
    # block-batched K1: table of layers in device memory
    lay = lambda **kw: lib.ecoflap_zo_perturb_layers(                            # noqa: E731
        kw.get("table", vp(16)), kw.get("n", 2), kw.get("rows", 64), kw.get("dt", 2), 1e-3, None,
        kw.get("ev0", None), kw.get("ev1", None))
    assert lay(n=0) == 0 and lay(rows=0) == 0                                    # empty block
    assert lay(dt=9) == -1
    assert lay(n=-1) == -3 and lay(rows=-4) == -3 and lay(rows=1 << 31) == -3
    assert lay(table=None) == -2
    assert lay(ev0=vp(16)) == -2                                                 # half an event pair


def test_argument_errors_and_workspace_sizes_of_this_rounds_entry_points():
    """n:m selection, the mixed-dtype reduce and the multi-compare reject bad arguments without a
    GPU; the Hessian workspace covers the transposed copy, plus the K-slice slabs where the
    256-wide kernel cuts left-over tiles (C = 6144 at 8 samples per call: 44 tiles x 5 slices x
    256 KB on a 256-CU part) or the four partial Hessians of the K-split small shapes."""
    from ecoflap_amd import hip
    lib = hip.load_library()
    vp = ctypes.c_void_p
    p = vp(4096)
    assert lib.ecoflap_wanda_prune_nm(p, p, 4, 8, 0, 0, 4, None, None) == -3          # n = 0
    assert lib.ecoflap_wanda_prune_nm(p, p, 4, 8, 0, 3, 2, None, None) == -3          # m < n
    assert lib.ecoflap_wanda_prune_nm(p, p, 4, 8, 0, 2, 32, None, None) == -3         # m > 16
    assert lib.ecoflap_wanda_prune_nm(p, p, 4, 9, 0, 2, 4, None, None) == -3          # ragged group of 1 < n
    assert lib.ecoflap_wanda_prune_nm(None, p, 4, 8, 0, 2, 4, None, None) == -2
    assert lib.ecoflap_wanda_prune_nm(p, p, 0, 8, 0, 2, 4, None, None) == 0            # empty
    assert lib.ecoflap_sparsegpt_block_nm(p, 4, 128, p, 128, 0, 128, 2, 32, p, None, None) == -3
    assert lib.ecoflap_sparsegpt_block_nm(p, 4, 128, p, 128, 0, 5, 2, 4, p, None, None) == -3
    assert lib.ecoflap_sparsegpt_block_nm(None, 4, 128, p, 128, 0, 128, 2, 4, p, None, None) == -2
    host = (ctypes.c_int64 * 4)(4096, 4096, 16, 1 | (2 << 8))                         # fp16 W, bf16 g
    assert lib.ecoflap_absprod_reduce_mixed(p, host, 1, 0, p, p, 1 << 20, None) == -1
    assert lib.ecoflap_absprod_reduce_mixed(p, host, 1, 9, p, p, 1 << 20, None) == -4
    assert lib.ecoflap_absprod_reduce_mixed(p, host, 0, 0, p, p, 1 << 20, None) == 0
    lib.ecoflap_hessian_workspace_bytes.restype = ctypes.c_size_t
    lib.ecoflap_hessian_workspace_bytes.argtypes = [ctypes.c_int64, ctypes.c_int64]
    xt = lambda t, c: c * ((t + 63) // 64 * 64) * 2                                   # noqa: E731
    one = lib.ecoflap_hessian_workspace_bytes(2056, 6144)
    eight = lib.ecoflap_hessian_workspace_bytes(8 * 2056, 6144)
    assert xt(2056, 6144) <= one < xt(2056, 6144) + 4096                              # 128-wide kernel: the copy only
    assert eight >= xt(8 * 2056, 6144) + 44 * 5 * 256 * 256 * 4                       # + the slabs
    small = lib.ecoflap_hessian_workspace_bytes(8 * 2056, 1408)
    assert small >= xt(8 * 2056, 1408) + 4 * 1408 * 1408 * 4                          # + four partial Hessians
    assert lib.ecoflap_hessian_workspace_bytes(0, 6144) == 0
