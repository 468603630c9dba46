"""The build's own blocked fp32 Cholesky (csrc/cholesky.hip) next to torch.linalg (rocSOLVER) at the
BLIP-2 Hessian sizes, lower and upper, and the whole damped-Cholesky chain of SparseGPT.fasterprune
(sparsegpt_pruner.py:113-162: factor, inverse from the factor, upper factor of the inverse) with the
library's `cholesky_inverse` kept in the middle.  Errors against the fp64 factor.

    python3 tools/diag/own_cholesky_bench.py
"""
import os
import sys

os.environ.setdefault("TENSILE_STREAMK_DATA_PARALLEL", "1")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch  # noqa: E402

from ecoflap_amd import hip  # noqa: E402


def timed(fn, reps=5):
    fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(reps):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); out = fn(); b.record(); torch.cuda.synchronize()
        ts.append(a.elapsed_time(b))
    return sorted(ts)[len(ts) // 2], out


def main():
    kern = hip.HipKernels()
    torch.manual_seed(0)
    info = torch.zeros(1, dtype=torch.int32, device="cuda")
    ws16 = torch.empty(16384, dtype=torch.uint8, device="cuda")

    def own(H, upper):
        L = H.clone(memory_format=torch.contiguous_format)
        rc = kern.lib.ecoflap_cholesky_f32(L.data_ptr(), L.shape[0], L.stride(0), int(upper), info.data_ptr(),
                                           ws16.data_ptr(), ws16.numel(), torch.cuda.current_stream().cuda_stream)
        assert rc == 0
        return L

    for n in (768, 1408, 2048, 5120, 6144):
        X = torch.randn(4 * n, n, device="cuda")
        H = (X.t() @ X) / (4 * n) + 0.01 * torch.eye(n, device="cuda")
        H = ((H + H.t()) / 2).contiguous()
        ref = torch.linalg.cholesky(H.double())
        t_lib, L_lib = timed(lambda: torch.linalg.cholesky_ex(H)[0])
        t_own, L_own = timed(lambda: own(H, False))
        assert int(info) == 0
        sc = float(ref.abs().max())
        e_lib, e_own = float((L_lib.double() - ref).abs().max()) / sc, float((L_own.double() - ref).abs().max()) / sc
        t_inv, Hi = timed(lambda: torch.cholesky_inverse(L_own))
        t_inv_own, Hi_own = timed(lambda: kern.cholesky_inverse(L_own))
        ref_i = torch.cholesky_inverse(L_own.double())
        sci = float(ref_i.abs().max())
        ei_lib, ei_own = float((Hi.double() - ref_i).abs().max()) / sci, float((Hi_own.double() - ref_i).abs().max()) / sci
        del ref_i
        t_ulib, _ = timed(lambda: torch.linalg.cholesky_ex(Hi, upper=True)[0])
        t_uown, U_own = timed(lambda: own(Hi, True))
        assert int(info) == 0
        chain_lib = t_lib + t_inv + t_ulib
        chain_own = t_own + t_inv_own + t_uown
        print(f"n={n:5d}: cholesky lower  library {t_lib:7.2f} ms (err {e_lib:.1e})  own {t_own:7.2f} ms (err {e_own:.1e}) "
              f"= {n ** 3 / 3 / t_own / 1e9:5.2f} TFLOP/s | upper  library {t_ulib:7.2f}  own {t_uown:7.2f} | "
              f"inverse  library {t_inv:6.2f} (err {ei_lib:.1e})  own {t_inv_own:6.2f} (err {ei_own:.1e}) | chain  library {chain_lib:7.2f} ms  own {chain_own:7.2f} ms", flush=True)
        del X, H, ref, L_lib, L_own, Hi, U_own
        torch.cuda.empty_cache()


if __name__ == "__main__":
    main()
