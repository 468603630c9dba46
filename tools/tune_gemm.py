#!/usr/bin/env python3
"""Time every hipBLASLt solution on the forward's 16-bit weight shapes (csrc/gemm_pinned.hip::
ecoflap_linear_tune) and print, per shape, the fastest ones with their repeatability / batch
invariance next to the heuristic's own first choice.  A MEASUREMENT, not a product path: round 4
tried to ship its result as a per-shape solution table and dropped it
(profiles/r04_secondary/gemm_tuning.md): at 16 evaluations the heuristic's choice is within the
run-to-run noise of the fastest of the ~230 solutions for every shape, and one solution pinned
for every row count loses at the small ones.  `--write PATH` stores, keyed by the library's
version, the solution with the smallest t(16 m) + t(m) / 2 among the repeatable and batch-invariant
ones where it beats the heuristic's first choice by 5 % or more.

    python3 tools/tune_gemm.py [--top 16] [--write gpurun_out/gemm_table.json]
"""
import ctypes
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402  (loads torch's own libhipblaslt first: the one that serves at run time)
from ecoflap_amd import blas_guard  # noqa: E402,F401  (TENSILE_STREAMK_DATA_PARALLEL=1 before the first GEMM)
from ecoflap_amd import hip as _hip  # noqa: E402

SHAPES = [  # (what, N, K, dtype, has_bias, probe rows = one evaluation at batch size 8)
    ("ViT-g qkv", 4224, 1408, torch.float16, False, 2056),
    ("ViT-g proj", 1408, 1408, torch.float16, True, 2056),
    ("ViT-g fc1", 6144, 1408, torch.float16, True, 2056),
    ("ViT-g fc2", 1408, 6144, torch.float16, True, 2056),
    ("FlanT5 q/k/v/o (encoder)", 2048, 2048, torch.bfloat16, False, 384),
    ("FlanT5 wi_0/wi_1 (encoder)", 5120, 2048, torch.bfloat16, False, 384),
    ("FlanT5 wo (encoder)", 2048, 5120, torch.bfloat16, False, 384),
    ("FlanT5 lm_head", 32128, 2048, torch.bfloat16, False, 128),
]


def main():
    top = int(sys.argv[sys.argv.index("--top") + 1]) if "--top" in sys.argv else 16
    write = sys.argv[sys.argv.index("--write") + 1] if "--write" in sys.argv else None
    lib = ctypes.CDLL(os.path.join(ROOT, "ecoflap_amd", "libecoflap_gemm.so"))
    i64, ci = ctypes.c_int64, ctypes.c_int
    ip, fp = ctypes.POINTER(ctypes.c_int), ctypes.POINTER(ctypes.c_float)
    lib.ecoflap_linear_tune.argtypes = [i64, i64, i64, ci, ci, ci, ci, ip, fp, fp, ip, ctypes.c_char_p, ci, ip, ip]
    lib.ecoflap_linear_library_version.argtypes = [ctypes.c_char_p, ci]
    torch.zeros(1, device="cuda")
    vbuf = ctypes.create_string_buffer(256)
    lib.ecoflap_linear_library_version(vbuf, 256)
    version = vbuf.value.decode()
    print("hipBLASLt", version, flush=True)
    out = {}
    for what, N, K, dt, has_bias, m in SHAPES:
        idx, flags = (ctypes.c_int * top)(), (ctypes.c_int * top)()
        ub, us = (ctypes.c_float * top)(), (ctypes.c_float * top)()
        names = ctypes.create_string_buffer(top * 256)
        n, nc = ctypes.c_int(0), ctypes.c_int(0)
        code = _hip.DTYPE_CODE[dt]
        rc = lib.ecoflap_linear_tune(m, N, K, code, int(has_bias), code, top, idx, ub, us, flags, names, 256,
                                     ctypes.byref(n), ctypes.byref(nc))
        print(f"== {what}: {N} x {K} {str(dt).split('.')[-1]} bias {has_bias}, rows {m} and {16 * m}: rc {rc}, "
              f"{nc.value} solutions ran", flush=True)
        fl = 2.0 * 16 * m * N * K
        best = first = None
        for r in range(n.value):
            nm = names.raw[r * 256:(r + 1) * 256].split(b"\0")[0].decode(errors="replace")
            f = flags[r]
            print(f"   {idx[r]:7d}  16m {ub[r]:8.1f} us ({fl / ub[r] / 1e6:6.0f} TF/s)  m {us[r]:7.1f} us  "
                  f"{'repeatable' if f & 1 else 'NOT-repeatable'} {'invariant' if f & 2 else 'NOT-invariant'}"
                  f"{'  <- heuristic first choice' if f & 4 else ''}  {nm[:100]}", flush=True)
            row = {"index": idx[r], "us_16m": round(ub[r], 1), "us_m": round(us[r], 1), "name": nm[:120],
                   "cost": ub[r] + 0.5 * us[r]}
            if f & 4:
                first = row
            if (f & 3) == 3 and (best is None or row["cost"] < best["cost"]):
                best = row
        if best is not None and first is not None and best["cost"] <= 0.95 * first["cost"]:
            best = dict(best, library_first_choice={k: first[k] for k in ("index", "us_16m", "us_m", "name")},
                        probe_rows=m)
            del best["cost"]
            out[f"{N}x{K} {str(dt).split('.')[-1]} {'bias' if has_bias else 'nobias'}"] = best
            print(f"   -> table: {best['index']}", flush=True)
        else:
            print("   -> the library's own choice stays", flush=True)
    doc = {"written_by": "tools/tune_gemm.py on an MI355X (gfx950); indices are valid for the named library only",
           "libraries": {version: out}}
    if write:
        with open(write, "w") as f:
            json.dump(doc, f, indent=1)
    print(json.dumps(doc, indent=1))


main()
