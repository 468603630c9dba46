"""SparseGPT's three factorisations (sparsegpt_pruner.py:84-110: Cholesky, inverse from the factor,
Cholesky of the inverse, upper) at the BLIP-2 Hessian sizes: torch.linalg (rocSOLVER) against a
right-looking blocked form whose flops go through the library's fp32 GEMM."""
import os, sys, time
os.environ.setdefault("TENSILE_STREAMK_DATA_PARALLEL", "1")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch


def timed(fn, reps=3):
    fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(reps):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); out = fn(); b.record(); torch.cuda.synchronize()
        ts.append(a.elapsed_time(b))
    return sorted(ts)[len(ts) // 2], out


def blocked_cholesky_lower(A, nb):
    """right-looking: A = L L^T, L returned in a new tensor (lower), fp32"""
    n = A.shape[0]
    L = A.clone()
    for k in range(0, n, nb):
        e = min(k + nb, n)
        Lkk = torch.linalg.cholesky(L[k:e, k:e])
        L[k:e, k:e] = Lkk
        if e < n:
            # L[e:, k:e] = A[e:, k:e] Lkk^{-T}
            P = torch.linalg.solve_triangular(Lkk, L[e:, k:e].t(), upper=False).t()
            L[e:, k:e] = P
            L[e:, e:].addmm_(P, P.t(), alpha=-1.0)
    return torch.tril(L)


def inverse_from_factor_trsm(L):
    n = L.shape[0]
    eye = torch.eye(n, device=L.device, dtype=L.dtype)
    Linv = torch.linalg.solve_triangular(L, eye, upper=False)
    return Linv.t() @ Linv


def main():
    torch.manual_seed(0)
    for n in (1408, 2048, 5120, 6144):
        X = torch.randn(4 * n, n, device="cuda")
        H = (X.t() @ X) / (4 * n) + 0.01 * torch.eye(n, device="cuda")
        t_l, L = timed(lambda: torch.linalg.cholesky_ex(H, upper=False)[0])
        t_i, Hi = timed(lambda: torch.cholesky_inverse(L))
        t_u, U = timed(lambda: torch.linalg.cholesky_ex(Hi, upper=True)[0])
        print(f"n={n}: torch.linalg  chol_lower {t_l:8.2f} ms  cholesky_inverse {t_i:8.2f} ms  chol_upper {t_u:8.2f} ms  "
              f"sum {t_l + t_i + t_u:8.2f} ms  ({4 * n ** 3 / 3 / (t_l + t_i + t_u) / 1e9:.2f} TFLOP/s)")
        for nb in (128, 256, 512):
            t_b, Lb = timed(lambda: blocked_cholesky_lower(H, nb))
            err = float((Lb - L).abs().max() / L.abs().max())
            print(f"        blocked nb={nb}: {t_b:8.2f} ms  rel diff to torch {err:.2e}")
        t_s, His = timed(lambda: inverse_from_factor_trsm(L))
        print(f"        inverse via trsm(I) + gemm: {t_s:8.2f} ms  rel diff {float((His - Hi).abs().max() / Hi.abs().max()):.2e}")
        t_cs, Hcs = timed(lambda: torch.cholesky_solve(torch.eye(n, device='cuda'), L))
        print(f"        inverse via cholesky_solve(I): {t_cs:8.2f} ms  rel diff {float((Hcs - Hi).abs().max() / Hi.abs().max()):.2e}")
        # batched: three 1408 / 2048 factorisations at once (a block's independent Hessians)
        if n <= 2048:
            Hb = torch.stack([H, H * 1.1, H * 0.9])
            t_bb, _ = timed(lambda: torch.linalg.cholesky_ex(Hb)[0])
            print(f"        torch.linalg batched x3: {t_bb:8.2f} ms")
        del X, H, L, Hi, U
        torch.cuda.empty_cache()


if __name__ == "__main__":
    main()
