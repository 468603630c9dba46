"""Fused forward ops for the shape modules (include/ecoflap_shape_ops.h): used on the GPU,
for 16-bit tensors, when autograd is off (the zeroth-order loop); everything else takes the
plain torch op chain — this is model plumbing, not the pruner's compute path."""
import ctypes

import torch

from .. import hip as _hip

_lib = None

# diagnostics (tools/diag/transient_hunt.py --subops): while a stage graph is captured, the
# EVA block records its intermediate tensors here (name -> tensor, kept alive so the graph
# pool never reuses them); None in production
TRACE_SINK = None


def trace(name, t):
    if TRACE_SINK is not None:
        TRACE_SINK[name] = t
    return t


def _get():
    global _lib
    if _lib is None:
        lib = _hip.load_library()
        vp, i64, f32, ci = ctypes.c_void_p, ctypes.c_int64, ctypes.c_float, ctypes.c_int
        lib.ecoflap_t5_rmsnorm.argtypes = [vp, vp, vp, i64, i64, f32, ci, vp]
        lib.ecoflap_t5_add_rmsnorm.argtypes = [vp, vp, vp, vp, vp, i64, i64, f32, ci, vp]
        lib.ecoflap_gelu_mul.argtypes = [vp, vp, vp, i64, ci, vp]
        lib.ecoflap_add_layernorm.argtypes = [vp, vp, vp, vp, vp, vp, i64, i64, f32, ci, vp]
        lib.ecoflap_add_bias_layernorm.argtypes = [vp, vp, vp, vp, vp, vp, vp, i64, i64, f32, ci, vp]
        lib.ecoflap_bias_gelu.argtypes = [vp, vp, vp, i64, i64, ci, vp]
        lib.ecoflap_bias_add_residual.argtypes = [vp, vp, vp, vp, i64, i64, ci, vp]
        lib.ecoflap_linear_f32.argtypes = [vp, vp, vp, vp, i64, i64, i64, vp]
        lib.ecoflap_qkv_bias_add.argtypes = [vp, vp, vp, i64, i64, ci, vp]
        lib.ecoflap_vit_attention.argtypes = [vp, vp, i64, i64, i64, i64, f32, ci, vp]
        lib.ecoflap_multi_copy.argtypes = [vp, ci, vp]
        lib.ecoflap_multi_compare.argtypes = [vp, ci, vp, vp]
        _lib = lib
    return _lib


def _usable(*tensors):
    if torch.is_grad_enabled():
        return False
    t0 = tensors[0]
    if t0.device.type != "cuda" or t0.dtype not in (torch.float16, torch.bfloat16):
        return False
    return all(t.dtype == t0.dtype and t.device == t0.device for t in tensors)


def _stream():
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


def t5_rmsnorm(x, weight, eps):
    """-> y or None (None: caller runs the torch op chain)."""
    if not _usable(x, weight) or x.shape[-1] % 8 != 0:
        return None
    xc = x if x.is_contiguous() else x.contiguous()
    y = torch.empty_like(xc)
    d = xc.shape[-1]
    rc = _get().ecoflap_t5_rmsnorm(xc.data_ptr(), weight.data_ptr(), y.data_ptr(),
                                   xc.numel() // d, d, float(eps), _hip.DTYPE_CODE[xc.dtype],
                                   _stream())
    if rc != 0:
        raise _hip.EcoflapHipError(f"ecoflap_t5_rmsnorm failed ({rc})")
    return y


def t5_add_rmsnorm(x, residual, weight, eps):
    """(x + residual, rmsnorm(x + residual) * weight) in one pass -> (sum, y), or None (caller runs
    the add and the norm: CPU, autograd, other dtypes).  Same bits as the two separate ops."""
    if (not _usable(x, weight) or not _usable(residual, weight) or x.shape != residual.shape
            or x.dtype != residual.dtype or x.shape[-1] % 8 != 0):
        return None
    xc = x if x.is_contiguous() else x.contiguous()
    rc_ = residual if residual.is_contiguous() else residual.contiguous()
    s, y = torch.empty_like(xc), torch.empty_like(xc)
    d = xc.shape[-1]
    rc = _get().ecoflap_t5_add_rmsnorm(xc.data_ptr(), rc_.data_ptr(), weight.data_ptr(), s.data_ptr(),
                                       y.data_ptr(), xc.numel() // d, d, float(eps),
                                       _hip.DTYPE_CODE[xc.dtype], _stream())
    if rc != 0:
        raise _hip.EcoflapHipError(f"ecoflap_t5_add_rmsnorm failed ({rc})")
    return s, y


def gelu_mul(a, b):
    if not _usable(a, b) or a.numel() % 8 != 0 or a.shape != b.shape:
        return None
    ac = a if a.is_contiguous() else a.contiguous()
    bc = b if b.is_contiguous() else b.contiguous()
    y = torch.empty_like(ac)
    rc = _get().ecoflap_gelu_mul(ac.data_ptr(), bc.data_ptr(), y.data_ptr(), ac.numel(),
                                 _hip.DTYPE_CODE[ac.dtype], _stream())
    if rc != 0:
        raise _hip.EcoflapHipError(f"ecoflap_gelu_mul failed ({rc})")
    return y


def add_layernorm(x, residual, norm, residual_bias=None):
    """nn.LayerNorm `norm` (fp32 parameters) of the 16-bit activation x — or of x + residual,
    also returned — as ONE kernel instead of cast / layer_norm / cast (/ add).
    residual_bias: the bias of the Linear that produced `residual` WITHOUT it (`take_pending_bias`).
    -> (x_plus_residual or x, normalised) or None (caller runs the torch ops)."""
    if (torch.is_grad_enabled() or x.device.type != "cuda"
            or x.dtype not in (torch.float16, torch.bfloat16) or x.shape[-1] % 8 != 0
            or norm.weight is None or norm.bias is None or norm.weight.dtype != torch.float32
            or not torch.is_autocast_enabled()):
        return None
    if residual is not None and (residual.dtype != x.dtype or residual.shape != x.shape):
        return None
    if residual_bias is not None and (residual is None or residual_bias.dtype != x.dtype
                                      or residual_bias.numel() != x.shape[-1]
                                      or not residual_bias.is_contiguous()):
        return None
    xc = x if x.is_contiguous() else x.contiguous()
    rc_ = None if residual is None else (residual if residual.is_contiguous() else residual.contiguous())
    y = torch.empty_like(xc)
    s = torch.empty_like(xc) if rc_ is not None else xc
    d = xc.shape[-1]
    if residual_bias is not None:
        rc = _get().ecoflap_add_bias_layernorm(
            xc.data_ptr(), rc_.data_ptr(), residual_bias.data_ptr(), norm.weight.data_ptr(),
            norm.bias.data_ptr(), s.data_ptr(), y.data_ptr(),
            xc.numel() // d, d, float(norm.eps), _hip.DTYPE_CODE[xc.dtype], _stream())
        if rc != 0:
            raise _hip.EcoflapHipError(f"ecoflap_add_bias_layernorm failed ({rc})")
        return s, y
    rc = _get().ecoflap_add_layernorm(
        xc.data_ptr(), None if rc_ is None else rc_.data_ptr(), norm.weight.data_ptr(),
        norm.bias.data_ptr(), None if rc_ is None else s.data_ptr(), y.data_ptr(),
        xc.numel() // d, d, float(norm.eps), _hip.DTYPE_CODE[xc.dtype], _stream())
    if rc != 0:
        raise _hip.EcoflapHipError(f"ecoflap_add_layernorm failed ({rc})")
    return s, y


def qkv_bias_add(qkv, q_bias, v_bias):
    """qkv (fresh GEMM output, [.., 3*dim]) += cat(q_bias, 0, v_bias).to(dtype), in place.
    -> qkv or None (caller runs the torch ops)."""
    if (torch.is_grad_enabled() or qkv.device.type != "cuda"
            or qkv.dtype not in (torch.float16, torch.bfloat16) or not qkv.is_contiguous()
            or q_bias.dtype != torch.float32 or v_bias.dtype != torch.float32
            or q_bias.numel() % 8 != 0 or qkv.shape[-1] != 3 * q_bias.numel()):
        return None
    dim = q_bias.numel()
    rc = _get().ecoflap_qkv_bias_add(qkv.data_ptr(), q_bias.data_ptr(), v_bias.data_ptr(),
                                     qkv.numel() // (3 * dim), dim, _hip.DTYPE_CODE[qkv.dtype],
                                     _stream())
    if rc != 0:
        raise _hip.EcoflapHipError(f"ecoflap_qkv_bias_add failed ({rc})")
    return qkv


def vit_attention(qkv, heads, scale):
    """softmax(q k^T * scale) v per (image, head) straight from the qkv Linear's output
    ([B, N, 3 * heads * head_dim] = [B, N, 3, heads, head_dim]) into [B, N, heads * head_dim]:
    one kernel instead of permute / contiguous copies + the library's fused attention (which runs
    the ViT-g shape, 257 tokens x 16 heads of 88, at a few percent of the MFMA rate).
    -> tensor or None (caller runs torch's scaled_dot_product_attention)."""
    import os
    if (torch.is_grad_enabled() or qkv.device.type != "cuda" or qkv.dtype != torch.float16
            or qkv.dim() != 3 or not qkv.is_contiguous() or os.environ.get("ECOFLAP_NO_FUSED_ATTENTION")):
        return None
    B, N, C3 = qkv.shape
    if C3 % (3 * heads) != 0:
        return None
    D = C3 // (3 * heads)
    if N > 288 or D > 96 or D % 8 != 0 or D < 8:
        return None
    out = torch.empty((B, N, heads * D), dtype=qkv.dtype, device=qkv.device)
    rc = _get().ecoflap_vit_attention(qkv.data_ptr(), out.data_ptr(), B, N, heads, D, float(scale),
                                      _hip.DTYPE_CODE[qkv.dtype], _stream())
    if rc == -3:        # ECOFLAP_ESIZE: the head's K image does not fit the kernel's LDS buffers
        return None
    if rc != 0:
        raise _hip.EcoflapHipError(f"ecoflap_vit_attention failed ({rc})")
    return out


class _CopyItem(ctypes.Structure):
    _fields_ = [("dst", ctypes.c_void_p), ("src", ctypes.c_void_p), ("bytes", ctypes.c_int64)]


def multi_copy(pairs):
    """[(dst, src)] same-shape, same-dtype, contiguous GPU tensors -> ONE copy launch on the
    current stream (chunks of 32); anything else falls back to `dst.copy_(src)`."""
    fast = []
    for dst, src in pairs:
        if (dst.device.type == "cuda" and src.device == dst.device and dst.dtype == src.dtype
                and dst.shape == src.shape and dst.is_contiguous() and src.is_contiguous()):
            if dst.data_ptr() != src.data_ptr() and dst.numel():
                fast.append((dst, src))
        else:
            dst.copy_(src, non_blocking=True)
    if len(fast) == 1:
        fast[0][0].copy_(fast[0][1], non_blocking=True)
        return
    for g0 in range(0, len(fast), 32):
        chunk = fast[g0:g0 + 32]
        arr = (_CopyItem * len(chunk))()
        for slot, (dst, src) in zip(arr, chunk):
            slot.dst, slot.src = dst.data_ptr(), src.data_ptr()
            slot.bytes = dst.numel() * dst.element_size()
        rc = _get().ecoflap_multi_copy(arr, len(chunk), _stream())
        if rc != 0:
            raise _hip.EcoflapHipError(f"ecoflap_multi_copy failed ({rc})")


def multi_compare(pairs, flag=None):
    """Bitwise `a == b` for all (a, b) pairs (same shape / dtype, contiguous, on the GPU) in one
    launch per 32 pairs -> int32 device tensor, non-zero when ANY pair differs (pass `flag` to
    accumulate several calls into one read-back).  Shape / dtype mismatches count as different."""
    dev = None
    for a, b in pairs:
        dev = a.device
        break
    if flag is None:
        flag = torch.zeros(1, dtype=torch.int32, device=dev if dev is not None else "cuda")
    fast = []
    for a, b in pairs:
        if a.shape != b.shape or a.dtype != b.dtype:
            flag.fill_(1)
            continue
        if not (a.is_contiguous() and b.is_contiguous()):
            a, b = a.contiguous(), b.contiguous()
        if a.numel():
            fast.append((a, b))
    for g0 in range(0, len(fast), 32):
        chunk = fast[g0:g0 + 32]
        arr = (_CopyItem * len(chunk))()
        for slot, (a, b) in zip(arr, chunk):
            slot.dst, slot.src = a.data_ptr(), b.data_ptr()
            slot.bytes = a.numel() * a.element_size()
        rc = _get().ecoflap_multi_compare(arr, len(chunk), flag.data_ptr(), _stream())
        if rc != 0:
            raise _hip.EcoflapHipError(f"ecoflap_multi_compare failed ({rc})")
    return flag


# ---- 16-bit Linears through hipBLASLt with the solution pinned per weight shape -----------------
# (csrc/gemm_pinned.hip, include/ecoflap_shape_ops.h: why, and how the solution is chosen)
_gemm = None            # libecoflap_gemm.so (None: not loaded yet; False: unavailable)
_plans = {}             # (N, K, dtype) -> dict (index, name, ...) or None = keep torch's GEMM: THIS RUN's binding
# What a weight shape is bound to is decided when the shape first comes up in a run, at the row
# count it comes up with.  The decisions themselves are pure functions of their inputs (candidate
# solutions in ascending index order, the first that measures batch invariant; the library's own
# choice compared at 16 slots against one) and are remembered per (weight shape, epilogue, probe row
# count) for the life of the process; the BINDING of a weight shape to one of them is per run
# (`begin_run`, called where a pruner's prune() starts): a run in a process that has multiplied
# the same weight shape at other row counts before — another model, a test at batch size 1 —
# binds what a fresh process would bind, and ends with the same table.
_plan_memo = {}         # (N, K, dtype, has_bias, bias dtype, probe rows) -> plan dict (without the per-run part) or None
_invariance_memo = {}   # (N, K, dtype, has_bias, bias dtype, probe rows) -> bool: the library's own choice, 16 slots vs 1


def begin_run():
    """Forget which solution each weight shape is bound to (not the decisions: see above)."""
    _plans.clear()
_gemm_ws = {}           # stream -> workspace tensor


def _gemm_lib():
    global _gemm
    if _gemm is None:
        import os
        path = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "libecoflap_gemm.so")
        if os.environ.get("ECOFLAP_PINNED_GEMM", "auto") == "0":
            _gemm = False
        elif not os.path.exists(path):
            raise _hip.EcoflapHipError(f"{path} is not built (make -C ecoflap_amd/csrc); set "
                                       "ECOFLAP_PINNED_GEMM=0 to run on the framework's own GEMM choice")
        else:
            lib = ctypes.CDLL(path)
            vp, i64, ci, sz = ctypes.c_void_p, ctypes.c_int64, ctypes.c_int, ctypes.c_size_t
            ip, fp = ctypes.POINTER(ctypes.c_int), ctypes.POINTER(ctypes.c_float)
            lib.ecoflap_linear_pinned_plan.argtypes = [i64, i64, i64, ci, ci, ci, ip, ip, ip, fp, fp,
                                                       ctypes.c_char_p, ci]
            lib.ecoflap_linear_pinned.argtypes = [vp, vp, vp, vp, i64, i64, i64, ci, ci, vp, sz, vp]
            lib.ecoflap_linear_library_version.argtypes = [ctypes.c_char_p, ci]
            _gemm = lib
    return _gemm


_library_version = None


def library_version():
    """"<hipblasLtGetVersion>-<git revision>" of the hipBLASLt serving this process (None without
    the GEMM library): what a solution index is an index INTO."""
    global _library_version
    lib = _gemm_lib()
    if _library_version is None and lib:
        buf = ctypes.create_string_buffer(256)
        if lib.ecoflap_linear_library_version(buf, 256) == 0:
            _library_version = buf.value.decode(errors="replace")
    return _library_version


def gemm_report():
    """What ran the Linears, for the run summaries: library version and the plans of `pinned_plans`."""
    def label(k):
        return f"{k[0]}x{k[1]} {str(k[2]).split('.')[-1]}"
    shapes = {label(k): (None if v is None else {
        "used": v["used"], "index": v["index"], "us_at_16_slots": round(v["us_at_16_slots"], 1),
        "library_first_choice_us": round(v["library_first_choice_us"], 1), "name": v["name"][:96]})
        for k, v in pinned_plans().items()}
    try:
        version = library_version()
    except Exception:           # (the GEMM library is not built: nothing was pinned either)
        version = None
    return {"hip": getattr(torch.version, "hip", None), "hipblaslt": version, "shapes": shapes}


def _pinned_wanted(plan, has_bias):
    """Policy (deterministic: a bitwise property of the library's own choice, never a timing).
    ECOFLAP_PINNED_GEMM=1: every weight shape with a surviving solution runs it — reproducibility
    and batch invariance of the 16-bit Linears then hang on nothing outside this build, at a
    price (one 256 x 256 macro tile for every row count: FlanT5's small decoder GEMMs lose a
    third, a single evaluation's ViT-g GEMMs fill 60 % of the chip).  Default: the pinned
    solution where the framework's own GEMM choice is NOT batch invariant for this weight shape
    at the row counts the loop uses (measured when the shape first comes up: 16 slots against one
    alone) — on gfx950 the ViT-g shapes at batch size 1, where the library picks different kernels
    for 257 and 4112 rows; everything else stays on the library's choice under blas_guard's rules."""
    import os
    return (os.environ.get("ECOFLAP_PINNED_GEMM", "auto") == "1"
            or not plan["library_batch_invariant"].get(has_bias, True))


def pinned_plans():
    """{(N, K, dtype): {"index", "name", "tried", "passed", "us_at_16_slots"} or None} chosen so
    far, with `used` = what the policy makes of it now (bench.py and the run summaries record it: a
    solution index means something only together with the library version)."""
    return {k: (None if v is None else dict(
        v, used={("bias" if hb else "no bias"): _pinned_wanted(v, hb) for hb in v["library_batch_invariant"]}))
        for k, v in _plans.items()}


def linear(x, weight, bias, library_bias=None):
    """F.linear(x, weight, bias) for a 16-bit (or, outside autocast, fp32) weight on the GPU without autograd, through the
    pinned hipBLASLt solution of this weight shape -> tensor, or None (caller runs F.linear:
    CPU, fp32 weights, autograd on, no surviving candidate, ECOFLAP_PINNED_GEMM=0).  Under
    autocast to the weight's dtype the input is cast as autocast would cast it.
    library_bias: the bias the caller's fallback `F.linear` would carry (a Linear that leaves its
    bias to the consuming op asks with bias=None; whether the LIBRARY's choice is batch invariant
    has to be probed with the epilogue it would really run)."""
    if (torch.is_grad_enabled() or weight.device.type != "cuda"
            or weight.dtype not in (torch.float16, torch.bfloat16, torch.float32) or weight.dim() != 2
            or not weight.is_contiguous()):
        return None
    if weight.dtype == torch.float32:
        # (the fp32 Q-Former: outside autocast only — under autocast torch runs the Linear in 16 bits)
        if torch.is_autocast_enabled("cuda") or x.dtype != torch.float32:
            return None
    elif x.dtype != weight.dtype:
        if not (torch.is_autocast_enabled("cuda") and torch.get_autocast_dtype("cuda") == weight.dtype
                and x.is_floating_point()):
            return None
        x = x.to(weight.dtype)
    elif torch.is_autocast_enabled("cuda") and torch.get_autocast_dtype("cuda") != weight.dtype:
        return None
    if bias is not None and (bias.dtype != weight.dtype or not bias.is_contiguous()
                             or bias.data_ptr() % 16):
        return None
    lib = _gemm_lib()
    if lib is False:
        return None
    N, K = weight.shape
    per_vec = 4 if weight.dtype == torch.float32 else 8
    if x.shape[-1] != K or K % per_vec != 0 or N % per_vec != 0:
        return None
    x2 = x.reshape(-1, K)
    if not x2.is_contiguous():
        x2 = x2.contiguous()
    M = x2.shape[0]
    if M == 0:
        return None
    dt = _hip.DTYPE_CODE[weight.dtype]
    bdt = _hip.DTYPE_CODE[bias.dtype] if bias is not None else 0
    key = (N, K, weight.dtype)              # one plan per weight shape (the bias is added around the GEMM)
    if key not in _plans and weight.dtype == torch.float32:
        # no library solution to pin (gfx950: every fp32 solution is a Stream-K kernel): the
        # build's own MFMA GEMM, for the shapes it tiles
        _plans[key] = ({"index": -1, "name": "gemm_f32_nt_kernel (csrc/gemm_f32.hip)", "tried": 0,
                        "passed": 0, "us_at_16_slots": 0.0, "library_first_choice_us": 0.0,
                        "library_batch_invariant": {}}
                       if (N % 128 == 0 and K % 32 == 0) else None)
    if key not in _plans:
        mp = M if M <= 4096 else 2048
        memo_key = (N, K, weight.dtype, bias is not None, bdt, mp)
        if memo_key not in _plan_memo:
            if torch.cuda.is_current_stream_capturing():
                raise RuntimeError(f"pinned GEMM: the solution for weight shape {N}x{K} has not been "
                                   "chosen yet and a graph is being captured (run the stage eagerly once)")
            idx, tried, passed = ctypes.c_int(-1), ctypes.c_int(0), ctypes.c_int(0)
            us, default_us = ctypes.c_float(0.0), ctypes.c_float(0.0)
            name = ctypes.create_string_buffer(512)
            torch.cuda.synchronize()
            rc = lib.ecoflap_linear_pinned_plan(mp, N, K, dt, int(bias is not None), bdt,
                                                ctypes.byref(idx), ctypes.byref(tried), ctypes.byref(passed),
                                                ctypes.byref(us), ctypes.byref(default_us), name, 512)
            if rc == 0:
                _plan_memo[memo_key] = {"index": idx.value, "name": name.value.decode(errors="replace"),
                                        "tried": tried.value, "passed": passed.value, "us_at_16_slots": us.value,
                                        "library_first_choice_us": default_us.value, "probe_rows": mp}
            elif rc == -3:          # ECOFLAP_ESIZE: no candidate survived; torch's GEMM for this shape
                _plan_memo[memo_key] = None
            else:
                raise _hip.EcoflapHipError(f"ecoflap_linear_pinned_plan failed ({rc}) for {N}x{K}")
        found = _plan_memo[memo_key]
        _plans[key] = None if found is None else dict(found, library_batch_invariant={})
    plan = _plans[key]
    if plan is None:
        return None
    lib_bias = bias if bias is not None else library_bias
    has_lib_bias = lib_bias is not None
    if has_lib_bias not in plan["library_batch_invariant"]:
        # what the framework's own choice does with this weight (and this epilogue) at the row
        # counts the loop uses: 16 slots against one alone, bit for bit
        mp = M if M <= 4096 else 2048
        inv_key = (N, K, weight.dtype, has_lib_bias, lib_bias.dtype if has_lib_bias else None, mp)
        if inv_key not in _invariance_memo:
            if torch.cuda.is_current_stream_capturing():
                raise RuntimeError(f"pinned GEMM: weight shape {N}x{K} has not been probed yet and a "
                                   "graph is being captured (run the stage eagerly once)")
            import torch.nn.functional as F
            with torch.no_grad():
                xs = x2[:mp].contiguous()
                alone = F.linear(xs, weight, lib_bias)
                many = F.linear(xs.repeat(16, 1), weight, lib_bias)
                # fp32 (the Q-Former): the library treats the LAST rows of a problem differently at any
                # size, which the loop's padding slots absorb at no cost; what the padding cannot absorb
                # is a difference in the other slots (batch size 1)
                last = 7 if weight.dtype == torch.float32 else 15
                _invariance_memo[inv_key] = bool(
                    torch.equal(many[:mp], alone) and torch.equal(many[last * mp:(last + 1) * mp], alone))
                del alone, many
        plan["library_batch_invariant"][has_lib_bias] = _invariance_memo[inv_key]
    if not _pinned_wanted(plan, has_lib_bias):
        return None
    stream = torch.cuda.current_stream()
    if weight.dtype == torch.float32:
        if bias is not None and (bias.data_ptr() % 4 or not bias.is_contiguous()):
            return None
        y = torch.empty((M, N), dtype=torch.float32, device=weight.device)
        rc = _get().ecoflap_linear_f32(x2.data_ptr(), weight.data_ptr(),
                                       None if bias is None else bias.data_ptr(), y.data_ptr(), M, N, K,
                                       ctypes.c_void_p(stream.cuda_stream))
        if rc != 0:
            raise _hip.EcoflapHipError(f"ecoflap_linear_f32 failed ({rc}) for [{M}, {K}] x {N}x{K}")
        return y.view(*x.shape[:-1], N)
    ws = _gemm_ws.get(stream.cuda_stream)
    if ws is None or ws.device != weight.device:
        ws = _gemm_ws[stream.cuda_stream] = torch.empty(64 << 20, dtype=torch.uint8, device=weight.device)
    y = torch.empty((M, N), dtype=weight.dtype, device=weight.device)
    rc = lib.ecoflap_linear_pinned(x2.data_ptr(), weight.data_ptr(),
                                   None if bias is None else bias.data_ptr(), y.data_ptr(), M, N, K, dt,
                                   bdt, ws.data_ptr(), ws.numel(), ctypes.c_void_p(stream.cuda_stream))
    if rc != 0:
        raise _hip.EcoflapHipError(f"ecoflap_linear_pinned failed ({rc}) for [{M}, {K}] x {N}x{K}")
    return y.view(*x.shape[:-1], N)


def linear_or_torch(x, weight, bias):
    """The forward of every Linear of the shape modules (`pin_linears`) and of the loop's
    per-slot form of the owning Linear (pruners/prefix_cache.py): one dispatch, so the batched
    and the per-evaluation paths run the same kernel on the same bits."""
    y = linear(x, weight, bias)
    return y if y is not None else torch.nn.functional.linear(x, weight, bias)


def _pinned_forward(self, x):
    # a Linear whose consumer asked, FOR THIS CALL (`deferring_bias`: the EVA block's forward), to
    # add the bias itself in the op that consumes the output anyway; any other caller — `blk.mlp(x)`,
    # a forward hook, a feature reader — gets the full Linear.  Why at all: the pinned solutions have no bias epilogue on gfx950,
    # and a bias prefill + beta = 1 costs a write and a read of the output (measured: -10 % on
    # the bench).  The flag the consumer reads says whether THIS call left the bias out.
    if self.bias is not None and self.__dict__.get("_defer_now"):
        y = linear(x, self.weight, None, library_bias=self.bias)
        if y is not None:
            self._bias_pending = True
            return y
    self._bias_pending = False
    # (`_call_bias`: a bias that is not the module's own parameter, handed over for this call by
    # the parent — EVA's qkv Linear has bias=False and its q / v biases live in the Attention)
    return linear_or_torch(x, self.weight, self.bias if self.bias is not None else self.__dict__.get("_call_bias"))


class deferring_bias:
    """`with deferring_bias(lin): y = lin(x)` (or a call that reaches `lin`): the caller takes over
    `lin`'s bias for the calls inside — it MUST read `take_pending_bias(lin)` afterwards and add
    what that returns.  The request lives exactly as long as the block; a Linear that is not one
    of `pin_linears`' (CPU, fp32, autograd) ignores it and returns its biased output."""

    def __init__(self, *mods):
        self.mods = [m for m in mods if m.__dict__.get("_defer_bias")]

    def __enter__(self):
        for m in self.mods:
            m.__dict__["_defer_now"] = True
        return self

    def __exit__(self, *exc):
        for m in self.mods:
            m.__dict__.pop("_defer_now", None)
        return False


def take_pending_bias(mod):
    """The bias `mod`'s last forward left out (None: the output carries it already)."""
    if mod.__dict__.get("_bias_pending"):
        mod._bias_pending = False
        return mod.bias
    return None


def bias_gelu(a, bias):
    """gelu(a + bias) for a Linear output without its bias (erf GELU): one fused kernel for
    16-bit GPU tensors, the torch op chain otherwise (an fp32 Linear on the build's own GEMM)."""
    if a.dtype not in (torch.float16, torch.bfloat16) or a.device.type != "cuda" or a.shape[-1] % 8:
        return torch.nn.functional.gelu(a + bias)
    ac = a if a.is_contiguous() else a.contiguous()
    y = torch.empty_like(ac)
    d = ac.shape[-1]
    rc = _get().ecoflap_bias_gelu(ac.data_ptr(), bias.data_ptr(), y.data_ptr(), ac.numel() // d, d,
                                  _hip.DTYPE_CODE[ac.dtype], _stream())
    if rc != 0:
        raise _hip.EcoflapHipError(f"ecoflap_bias_gelu failed ({rc})")
    return y


def bias_add_residual(x, m, bias):
    """x + (m + bias) for a Linear output m without its bias: one fused kernel for 16-bit GPU
    tensors, the torch ops otherwise."""
    if (x.dtype not in (torch.float16, torch.bfloat16) or x.device.type != "cuda" or x.shape[-1] % 8
            or m.dtype != x.dtype or m.shape != x.shape):
        return x + (m + bias)
    xc = x if x.is_contiguous() else x.contiguous()
    mc = m if m.is_contiguous() else m.contiguous()
    y = torch.empty_like(xc)
    d = xc.shape[-1]
    rc = _get().ecoflap_bias_add_residual(xc.data_ptr(), mc.data_ptr(), bias.data_ptr(), y.data_ptr(),
                                          xc.numel() // d, d, _hip.DTYPE_CODE[xc.dtype], _stream())
    if rc != 0:
        raise _hip.EcoflapHipError(f"ecoflap_bias_add_residual failed ({rc})")
    return y


def patches_gemm(proj, x):
    """`proj(x).flatten(2).transpose(1, 2)` of a Conv2d whose stride is its kernel size, as a
    GEMM: [B, C, H, W] -> [B, patches, D] (rows and columns the kernel does not cover are left
    out, as the convolution leaves them out).  Why: shapes/eva_vit.py `PatchEmbed`."""
    p = proj.kernel_size[0]
    assert _is_patch_conv(proj)
    B, C, H, W = x.shape
    gh, gw = H // p, W // p
    cols = (x[:, :, :gh * p, :gw * p].reshape(B, C, gh, p, gw, p)
            .permute(0, 2, 4, 1, 3, 5).reshape(B, gh * gw, C * p * p))
    return torch.nn.functional.linear(cols, proj.weight.reshape(proj.out_channels, -1), proj.bias)


def _is_patch_conv(mod):
    if type(mod) is not torch.nn.Conv2d:
        return False
    k = mod.kernel_size
    return (k[0] == k[1] and tuple(mod.stride) == tuple(k)
            and tuple(mod.padding) == (0, 0) and tuple(mod.dilation) == (1, 1) and mod.groups == 1
            and mod.padding_mode == "zeros")


def _patch_conv_forward(self, x):
    if not x.is_cuda or x.dim() != 4:
        return torch.nn.Conv2d.forward(self, x)
    p = self.kernel_size[0]
    y = patches_gemm(self, x)
    return y.transpose(1, 2).reshape(x.shape[0], self.out_channels, x.shape[2] // p, x.shape[3] // p)


def pin_patch_convs(model):
    """For a model that is NOT one of the build's shape modules (the reference's own
    `eva_vit.VisionTransformer`, …) and is going to be scored on the GPU by several ranks, or by
    runs whose tables are compared: every `nn.Conv2d` whose stride is its kernel size (a ViT's
    patch embedding) gets an instance-level forward that runs it as one GEMM over the unfolded
    patches instead of handing it to MIOpen, whose timed Find may crown another solver — another
    rounding — in another process (profiles/NOTES_r06.md §7).  Opt-in: it changes the rounding of
    that layer (within one 16-bit ulp of each of MIOpen's three).  Same module type, parameters
    and state_dict keys.  -> how many were pinned; `unpinned_convs(model)` lists the rest."""
    import types
    n = 0
    for mod in model.modules():
        if _is_patch_conv(mod) and "forward" not in mod.__dict__:
            mod.forward = types.MethodType(_patch_conv_forward, mod)
            n += 1
    return n


def unpinned_convs(model):
    """Names of the convolutions of `model` whose GPU forward is MIOpen's (kernel chosen by
    timing on first use): what `BasePruner` warns about before a GPU run."""
    conv_types = (torch.nn.Conv1d, torch.nn.Conv2d, torch.nn.Conv3d, torch.nn.ConvTranspose1d,
                  torch.nn.ConvTranspose2d, torch.nn.ConvTranspose3d)
    return [name for name, mod in model.named_modules()
            if isinstance(mod, conv_types) and "forward" not in mod.__dict__
            and not mod.__dict__.get("_ecoflap_gemm_form")]


def pin_linears(model):
    """Every nn.Linear of `model` gets an instance-level `forward` that goes through
    `linear_or_torch`: its 16-bit GPU forward runs the pinned solution, everything else (CPU, fp32,
    autograd) F.linear as before.  The modules stay exactly `nn.Linear` (the reference's
    `find_layers` tests `type(module) in [nn.Linear]`, wanda_pruner.py:33-52), with the same
    parameters and state_dict keys; deep copies (the loop's lanes) carry the method along."""
    import types
    from .eva_vit import Block
    n = 0
    for mod in model.modules():
        if type(mod) is torch.nn.Linear and "forward" not in mod.__dict__:
            mod.forward = types.MethodType(_pinned_forward, mod)
            mod._ecoflap_pinned = True
            n += 1
        if isinstance(mod, Block):          # its forward MAY add these biases in the consuming op
            for lin in (mod.attn.proj, mod.mlp.fc1, mod.mlp.fc2):   # (eligible; asked for per call)
                if lin.bias is not None:
                    lin._defer_bias = True
    return n
