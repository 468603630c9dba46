"""numpy restatement of the reference's sparsity allocator — TEST INFRASTRUCTURE.

Follows LAVIS/lavis/compression/pruners/layer_single_base_pruner.py:247-314
(`LayerSparsity.compute_the_sparsity_per_group`) op for op, including the
dtype of every intermediate torch tensor (int64 -> float32 after the first
`keep + add`), torch's CPU float32 `sum` order (8-lane cascade, restated in
`torch_sum_f32`) and the shipped "remove excess" branch that ADDS (:301).
Pinned by tests/golden/g4_allocator.npz (outputs of the reference itself).
"""
import numpy as np

f32 = np.float32


def _ceil_log2(x):
    return 1 if x <= 2 else int(x - 1).bit_length()


def _multi_row_sum(load, size, nrows, zero):
    """ATen SumKernel.cpp `multi_row_sum`: 4-level cascade over `size` rows of `nrows` lanes."""
    levels = 4
    power = max(4, _ceil_log2(size) // levels)
    step = 1 << power
    mask0 = step - 1
    acc = [[zero.copy() if hasattr(zero, "copy") else zero for _ in range(nrows)]
           for _ in range(levels)]
    i = 0
    while i + step <= size:
        for _ in range(step):
            for k in range(nrows):
                acc[0][k] = (acc[0][k] + load(i, k)).astype(f32)
            i += 1
        for j in range(1, levels):
            for k in range(nrows):
                acc[j][k] = (acc[j][k] + acc[j - 1][k]).astype(f32)
                acc[j - 1][k] = acc[j - 1][k] * f32(0)
            if (i & (mask0 << (j * power))) != 0:
                break
    while i < size:
        for k in range(nrows):
            acc[0][k] = (acc[0][k] + load(i, k)).astype(f32)
        i += 1
    for j in range(1, levels):
        for k in range(nrows):
            acc[0][k] = (acc[0][k] + acc[j][k]).astype(f32)
    return acc[0]


def _row_sum(load, size, zero):
    ilp = 4
    size_ilp = size // ilp
    ps = _multi_row_sum(lambda i, k: load(i * ilp + k), size_ilp, ilp, zero)
    for i in range(size_ilp * ilp, size):
        ps[0] = (ps[0] + load(i)).astype(f32)
    for k in range(1, ilp):
        ps[0] = (ps[0] + ps[k]).astype(f32)
    return ps[0]


def torch_sum_f32(x, lanes=8):
    """Bit-exact restatement of torch.sum on a contiguous 1-D float32 CPU tensor
    (vectorized_inner_sum / scalar_inner_sum of ATen's cascade_sum; the kernel
    runs 8 lanes under DEFAULT, AVX2 and AVX512 dispatch alike — verified in
    tests/test_oracle_golden.py against torch itself)."""
    x = np.ascontiguousarray(x, dtype=f32)
    n = x.shape[0]
    if n >= lanes:
        nvec = n // lanes
        vacc = _row_sum(lambda i: x[i * lanes:(i + 1) * lanes], nvec, np.zeros(lanes, f32))
        final = f32(0)
        for k in range(nvec * lanes, n):
            final = f32(final + x[k])
        for k in range(lanes):
            final = f32(final + vacc[k])
        return final
    return f32(_row_sum(lambda i: np.asarray(x[i]), n, np.asarray(f32(0))))


def compute_sparsity_per_group(total_parameters_to_keep, group_scores, group_num_parameters,
                               max_sparsity_per_layer=0.8, max_iters=100000):
    """-> (sparsity float32[G] as python floats, keep vector float64[G])."""
    scores = np.array(group_scores, dtype=f32)              # torch.FloatTensor(...)      :248
    num = np.array(group_num_parameters, dtype=np.int64)    # torch.LongTensor(...)       :249
    total = int(total_parameters_to_keep)
    one_minus = f32(1 - max_sparsity_per_layer)             # python double -> float32 scalar
    num_f = num.astype(f32)                                 # int64 -> float32 promotion
    floor_keep = np.ceil(num_f * one_minus).astype(np.int32).astype(np.int64)  # :253
    keep_is_float = False
    keep_i = floor_keep.copy()                              # int64 tensor
    keep_f = None                                           # float32 tensor once promoted

    def keep_sum():
        return torch_sum_f32(keep_f) if keep_is_float else int(keep_i.sum())

    def lt_total(s):   # tensor < python int : int compare for int64 sum, float32 compare otherwise
        return (s < f32(total)) if keep_is_float else (s < total)

    np.seterr(invalid="ignore", divide="ignore")  # 0/0 -> nan exactly as torch does
    it = 0
    while lt_total(keep_sum()):                             # :255
        it += 1
        if it > max_iters:
            raise RuntimeError("allocator does not terminate on this input")
        total_ratio = torch_sum_f32(scores)                 # :256
        s = keep_sum()
        rest = f32(f32(total) - s) if keep_is_float else f32(total - s)   # :258 (-> float32 in :260)
        add = np.ceil(((scores / total_ratio).astype(f32) * rest).astype(f32))  # :260
        base = keep_f if keep_is_float else keep_i.astype(f32)
        keep_f = (base + add).astype(f32)                   # :262  (int64 + float32 -> float32)
        keep_is_float = True
        scores[keep_f >= num_f] = 0                         # :264
        keep_f = np.minimum(keep_f, num_f)                  # :266
        if torch_sum_f32(add) == 0:                         # :269 stuck branch
            cur = torch_sum_f32(keep_f)
            if cur < f32(total):
                need = f32(f32(total) - cur)
                guard = 0
                while need > 0:
                    guard += 1
                    if guard > max_iters:
                        raise RuntimeError("allocator does not terminate on this input")
                    for idx in np.nonzero(scores > 0)[0]:
                        room = f32(num_f[idx] - keep_f[idx])
                        can = room if room < need else need   # python min(need, room)
                        keep_f[idx] = f32(keep_f[idx] + can)
                        need = f32(need - can)
                        if need == 0:
                            break
        if torch_sum_f32(keep_f) > f32(total):              # :288 over-target branch
            cur = torch_sum_f32(keep_f)
            excess = f32(cur - f32(total))
            guard = 0
            while excess > 0:
                guard += 1
                if guard > max_iters:
                    raise RuntimeError("allocator does not terminate on this input")
                order = np.argsort(-keep_f, kind="stable")  # argsort(descending=True, stable=True)
                for idx in order:
                    floor_i = np.int32(f32(num_f[idx] * one_minus))   # (.int() truncates)  :299
                    room = f32(keep_f[idx] - f32(floor_i))
                    can = room if room < excess else excess
                    keep_f[idx] = f32(keep_f[idx] + can)    # the reference ADDS here        :301
                    excess = f32(excess - can)
                    if excess == 0:
                        break
    keep = keep_f if keep_is_float else keep_i.astype(f32)
    ratio = (keep / num_f).astype(f32)                      # :312
    sparsity = np.clip((f32(1) - ratio).astype(f32), f32(0), f32(1))
    return [float(v) for v in sparsity], keep.astype(np.float64)
