"""CPU: the C-ABI library loads and exports every symbol include/ecoflap_hip.h declares
(no compute calls — there is no GPU here), and the product refuses to run without it."""
import ctypes
import os
import re

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
    for n in shape_ops:          # plumbing ops of the shape modules live in the same library
        assert hasattr(lib, n), n
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
