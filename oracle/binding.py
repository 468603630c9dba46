"""ctypes binding of oracle/libecoflap_oracle.so over torch CPU tensors."""
import ctypes
import os
import subprocess

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "libecoflap_oracle.so")
DT = {torch.float32: 0, torch.float16: 1, torch.bfloat16: 2}


def build(force=False):
    src = os.path.join(_HERE, "ecoflap_oracle.c")
    if force or not os.path.exists(_SO) or os.path.getmtime(_SO) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "-s"])
    return _SO


def load():
    if not os.path.exists(_SO):
        build()
    return Oracle(ctypes.CDLL(_SO))


def _p(t):
    return ctypes.c_void_p(t.data_ptr())


def _cpu(t):
    assert t.device.type == "cpu" and t.is_contiguous(), "oracle works on contiguous CPU tensors"
    return t


class Oracle:
    def __init__(self, lib):
        self.lib = lib
        L = lib
        i64, f32, u64, vp, ci = (ctypes.c_int64, ctypes.c_float, ctypes.c_uint64,
                                 ctypes.c_void_p, ctypes.c_int)
        L.oracle_zo_perturb.argtypes = [vp, i64, ci, f32, f32, vp]
        L.oracle_zo_perturb_triple.argtypes = [vp, vp, vp, vp, i64, ci, f32, vp]
        L.oracle_philox_u32.argtypes = [vp, i64, u64, ci]
        L.oracle_philox4x32.argtypes = [vp, vp, vp, ci]
        L.oracle_normal_stream.argtypes = [vp, vp, i64, ci, u64, ci]
        L.oracle_absprod_reduce.argtypes = [vp, vp, i64, ci, ci, ci]
        L.oracle_absprod_reduce.restype = ctypes.c_double
        L.oracle_colsqnorm_accum.argtypes = [vp, vp, i64, i64, ci, i64, i64]
        L.oracle_colsq_raw.argtypes = [vp, vp, i64, i64, ci]
        L.oracle_colsq_replay.argtypes = [vp, vp, vp, i64, i64, i64, i64]
        L.oracle_wanda_prune_rows.argtypes = [vp, vp, i64, i64, ci, i64, vp]
        L.oracle_wanda_prune_matrix.argtypes = [vp, vp, i64, i64, ci, i64, vp]
        L.oracle_wanda_prune_nm.argtypes = [vp, vp, i64, i64, ci, ci, ci, vp]
        L.oracle_mask_mul.argtypes = [vp, vp, i64, ci]
        L.oracle_round_array.argtypes = [vp, i64, ci]
        L.oracle_sparsegpt_block.argtypes = [vp, i64, i64, vp, i64, i64, ci, i64, vp, vp]
        L.oracle_sparsegpt_block_nm.argtypes = [vp, i64, i64, vp, i64, i64, ci, ci, ci, vp, vp]

    def zo_perturb(self, w, scaling_factor, zo_eps, z):
        _cpu(w), _cpu(z)
        assert z.dtype == w.dtype and z.numel() == w.numel()
        self.lib.oracle_zo_perturb(_p(w), w.numel(), DT[w.dtype], scaling_factor, zo_eps, _p(z))

    def zo_perturb_triple(self, w, zo_eps, z):
        _cpu(w), _cpu(z)
        plus, minus, rest = torch.empty_like(w), torch.empty_like(w), torch.empty_like(w)
        self.lib.oracle_zo_perturb_triple(_p(w), _p(plus), _p(minus), _p(rest), w.numel(),
                                          DT[w.dtype], zo_eps, _p(z))
        return plus, minus, rest

    def philox_u32(self, n, seed, rounds):
        out = torch.empty(n, dtype=torch.int32)
        self.lib.oracle_philox_u32(_p(out), n, seed, rounds)
        return out

    def normal_stream(self, n, dtype, seed, rounds, want_f32=False):
        """The in-register z stream of K1 for (seed, n, dtype): tensor of `dtype`, and with
        want_f32 also the fp32 values before the storage rounding."""
        z = torch.empty(n, dtype=dtype)
        z32 = torch.empty(n, dtype=torch.float32) if want_f32 else None
        self.lib.oracle_normal_stream(_p(z32) if want_f32 else None, _p(z), n, DT[dtype], seed,
                                      rounds)
        return (z, z32) if want_f32 else z

    def philox4x32(self, ctr, key, rounds):
        import ctypes
        c = (ctypes.c_uint32 * 4)(*ctr)
        k = (ctypes.c_uint32 * 2)(*key)
        out = (ctypes.c_uint32 * 4)()
        self.lib.oracle_philox4x32(out, c, k, rounds)
        return [int(v) for v in out]

    def absprod_reduce(self, w, g, mode):
        n = (w if w is not None else g).numel()
        wp = _p(_cpu(w)) if w is not None else None
        gp = _p(_cpu(g)) if g is not None else None
        return self.lib.oracle_absprod_reduce(
            wp, gp, n, DT[w.dtype] if w is not None else 0, DT[g.dtype] if g is not None else 0,
            mode)

    def colsqnorm_accum(self, scaler_row, x2d, n_before, batch):
        _cpu(scaler_row), _cpu(x2d)
        tokens, cols = x2d.shape
        self.lib.oracle_colsqnorm_accum(_p(scaler_row), _p(x2d), tokens, cols, DT[x2d.dtype],
                                        n_before, batch)

    def colsq_raw(self, out_row, x2d):
        _cpu(out_row), _cpu(x2d)
        tokens, cols = x2d.shape
        self.lib.oracle_colsq_raw(_p(out_row), _p(x2d), tokens, cols, DT[x2d.dtype])

    def colsq_replay(self, scaler_row, sq, batches, n_before=0):
        _cpu(scaler_row), _cpu(sq)
        assert sq.dtype == torch.float32 and sq.dim() == 2
        b = torch.tensor([int(v) for v in batches], dtype=torch.int64)
        self.lib.oracle_colsq_replay(_p(scaler_row), _p(sq), _p(b), sq.shape[0], sq.shape[1],
                                     sq.shape[1], n_before)

    def wanda_prune_rows(self, w, scaler_row, k, want_mask=True):
        _cpu(w), _cpu(scaler_row)
        rows, cols = w.shape
        mask = torch.zeros(rows, cols, dtype=torch.uint8) if want_mask else None
        self.lib.oracle_wanda_prune_rows(_p(w), _p(scaler_row), rows, cols, DT[w.dtype], k,
                                         _p(mask) if want_mask else None)
        return mask

    def wanda_prune_matrix(self, w, scaler_row, k, want_mask=True):
        _cpu(w), _cpu(scaler_row)
        rows, cols = w.shape
        mask = torch.zeros(rows, cols, dtype=torch.uint8) if want_mask else None
        self.lib.oracle_wanda_prune_matrix(_p(w), _p(scaler_row), rows, cols, DT[w.dtype], k,
                                           _p(mask) if want_mask else None)
        return mask

    def wanda_prune_nm(self, w, scaler_row, n, m, want_mask=True):
        _cpu(w), _cpu(scaler_row)
        rows, cols = w.shape
        if not 0 < n <= m <= 64 or (cols % m and cols % m < n):
            raise ValueError("n:m selection needs 0 < n <= m <= 64 and no group shorter than n")
        mask = torch.zeros(rows, cols, dtype=torch.uint8) if want_mask else None
        self.lib.oracle_wanda_prune_nm(_p(w), _p(scaler_row), rows, cols, DT[w.dtype], n, m,
                                       _p(mask) if want_mask else None)
        return mask

    def mask_mul(self, g, keep):
        _cpu(g), _cpu(keep)
        self.lib.oracle_mask_mul(_p(g), _p(keep), g.numel(), DT[g.dtype])

    def sparsegpt_block(self, W, Hinv, i1, count, k, err_out, mask_out=None):
        """W, Hinv: fp32 CPU tensors (row strides taken from the tensors), in place on W."""
        assert W.dtype == torch.float32 and Hinv.dtype == torch.float32
        assert W.stride(1) == 1 and Hinv.stride(1) == 1
        self.lib.oracle_sparsegpt_block(_p(W), W.shape[0], W.stride(0), _p(Hinv), Hinv.stride(0),
                                        i1, count, k, _p(err_out),
                                        _p(mask_out) if mask_out is not None else None)

    def sparsegpt_block_nm(self, W, Hinv, i1, count, n, m, err_out, mask_out=None):
        assert W.dtype == torch.float32 and Hinv.dtype == torch.float32
        assert W.stride(1) == 1 and Hinv.stride(1) == 1
        if not 0 < n <= m <= 64 or (count % m and count % m < n):
            raise ValueError("n:m needs 0 < n <= m <= 64 and no group shorter than n")
        self.lib.oracle_sparsegpt_block_nm(_p(W), W.shape[0], W.stride(0), _p(Hinv), Hinv.stride(0),
                                           i1, count, n, m, _p(err_out),
                                           _p(mask_out) if mask_out is not None else None)

    def round_array(self, x, dtype):
        _cpu(x)
        self.lib.oracle_round_array(_p(x), x.numel(), DT[dtype])
