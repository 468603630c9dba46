"""Checker backend for the tests: the CPU oracle behind the same interface as
`ecoflap_amd.hip.HipKernels`.  Lives under tests/ on purpose — the product never
imports oracle/.  Tensors on the GPU are round-tripped through the host so the
same model forward (on the GPU) can be driven by either backend."""
import torch

import oracle as _oracle


class OracleKernels:
    name = "oracle"

    def __init__(self, z_from=None):
        self.o = _oracle.load()
        self.z_from = z_from      # callable (seed, like_tensor) -> z, used when z is None

    @staticmethod
    def _host(t):
        return t.detach().cpu().contiguous()

    def _z(self, z, seed, like):
        if z is None:
            if self.z_from is None:
                raise RuntimeError("oracle backend needs z (it has no in-register generator)")
            z = self.z_from(seed, like)
        return self._host(z)

    def zo_perturb(self, w, scaling_factor, zo_eps, seed, z=None):
        h = self._host(w)
        self.o.zo_perturb(h, float(scaling_factor), float(zo_eps), self._z(z, seed, w))
        w.copy_(h)

    def zo_perturb_triple(self, w_in, w_plus, w_minus, w_restored, zo_eps, seed, z=None):
        plus, minus, rest = self.o.zo_perturb_triple(self._host(w_in), float(zo_eps),
                                                     self._z(z, seed, w_in))
        if w_plus is not None:
            w_plus.copy_(plus)
        if w_minus is not None:
            w_minus.copy_(minus)
        w_restored.copy_(rest)

    def zo_perturb_units(self, w, zo_eps, seeds, w_plus, w_minus, z=None):
        cur = self._host(w)
        for u, seed in enumerate(seeds):
            zz = self._z(z[u] if z is not None else None, seed, w)
            plus, minus, cur = self.o.zo_perturb_triple(cur, float(zo_eps), zz)
            if w_plus[u] is not None:
                w_plus[u].copy_(plus)
                w_minus[u].copy_(minus)
        w.copy_(cur)

    MAX_UNITS = 32

    def zo_perturb_layers(self, layers, zo_eps, events=None):
        """Block-batched K1: per layer the unit chain from w_in, theta+ / theta- into the unit
        buffers, the drifted weights into w_final (w_in untouched)."""
        for item in layers:
            w_in, w_final, seeds, w_plus, w_minus = item[:5]
            zs = item[5] if len(item) > 5 else None
            cur = self._host(w_in).clone()
            for u, seed in enumerate(seeds):
                zz = self._z(zs[u] if zs is not None else None, seed, w_in)
                plus, minus, cur = self.o.zo_perturb_triple(cur, float(zo_eps), zz)
                if w_plus[u] is not None:
                    w_plus[u].copy_(plus)
                    w_minus[u].copy_(minus)
            w_final.copy_(cur)

    def absprod_reduce(self, w, g, mode, out_accum):
        v = self.o.absprod_reduce(self._host(w) if w is not None else None,
                                  self._host(g) if g is not None else None, mode)
        out_accum += v

    def absprod_reduce_pairs(self, weights, grads, mode, out_accum):
        for i, (w, g) in enumerate(zip(weights, grads)):
            out_accum[i] += self.o.absprod_reduce(self._host(w), self._host(g), mode)

    def colsqnorm_accum(self, scaler_row, x2d, nsamples_before, batch):
        s = self._host(scaler_row)
        self.o.colsqnorm_accum(s, self._host(x2d), nsamples_before, batch)
        scaler_row.copy_(s)

    def colsqnorm_accum_multi(self, items, ws=None):
        """items: [(scaler_row, x2d, nsamples_before, n_dev or None, batch, raw)]"""
        for row, x2d, n_before, n_dev, batch, raw in items:
            if raw:
                out = torch.empty(row.shape, dtype=torch.float32)
                self.o.colsq_raw(out, self._host(x2d))
                row.copy_(out)
                continue
            if n_dev is not None:
                n_before = int(n_dev.item())
                n_dev += batch
            self.colsqnorm_accum(row, x2d, n_before, batch)

    def colsq_replay(self, scaler_row, sq, batches, nsamples_before=0):
        s = self._host(scaler_row)
        self.o.colsq_replay(s, self._host(sq).contiguous(), batches, nsamples_before)
        scaler_row.copy_(s)

    def wanda_prune_rows(self, w, scaler_row, k, mask_out=None):
        h = self._host(w)
        m = self.o.wanda_prune_rows(h, self._host(scaler_row), k, want_mask=True)
        w.copy_(h)
        if mask_out is not None:
            mask_out.copy_(m)

    def wanda_prune_matrix(self, w, scaler_row, k, mask_out=None):
        h = self._host(w)
        m = self.o.wanda_prune_matrix(h, self._host(scaler_row), k, want_mask=True)
        w.copy_(h)
        if mask_out is not None:
            mask_out.copy_(m)

    def wanda_prune_nm(self, w, scaler_row, n, m, mask_out=None):
        h = self._host(w)
        mk = self.o.wanda_prune_nm(h, self._host(scaler_row), n, m, want_mask=True)
        w.copy_(h)
        if mask_out is not None:
            mask_out.copy_(mk)

    def wanda_prune_block(self, items):
        for w, scaler_row, mode, k, mask_out in items:
            (self.wanda_prune_rows if mode == "rows" else self.wanda_prune_matrix)(
                w, scaler_row, k, mask_out)

    # Real-*: the reference's own torch formulation (layer_single_base_pruner.py:156-245,
    # :446-471) on host copies — the checker for the multi-tensor HIP kernels
    def grad_accum_multi(self, accs, grads):
        for a, g in zip(accs, grads):
            a += g.detach().float().abs().to(a.device)

    def global_threshold_prune(self, weights, accs, masks, mode, n_batches, k, protect_counts=None):
        scores = []
        if accs is None:
            accs = [None] * len(weights)
        for w, a, m in zip(weights, accs, masks):
            wf = self._host(w).float()
            g = None if a is None else self._host(a) / n_batches
            if mode == 3:
                sc = wf.clone()
            elif mode == 0:
                sc = wf.abs() * g.abs()
            elif mode == 1:
                sc = (wf ** 2) * g
            else:
                sc = g.abs()
            scores.append(sc * self._host(m).float())
        if protect_counts is not None:           # get_mask's protection step (:160-167)
            for sc, num_to_set in zip(scores, protect_counts):
                if num_to_set > 0:
                    thr = torch.topk(sc.flatten(), int(num_to_set), largest=True)[0][-1]
                    sc[torch.where(sc >= thr)] = torch.finfo(sc.dtype).max
        allv = torch.cat([t.flatten() for t in scores])
        thr = torch.topk(allv, k, largest=False)[0][-1]
        for w, m, sc in zip(weights, masks, scores):
            keep = (sc > thr)
            m.copy_(keep.to(torch.uint8))
            w.copy_((self._host(w) * keep.to(w.dtype)))

    def count_zeros_multi(self, tensors):
        return [int((self._host(t) == 0).sum()) for t in tensors]

    def sparsegpt_block(self, W, Hinv, i1, count, k, err_out, mask_out=None):
        h = self._host(W)
        e = torch.empty(err_out.shape, dtype=torch.float32)
        m = torch.zeros(err_out.shape, dtype=torch.uint8) if mask_out is not None else None
        self.o.sparsegpt_block(h, self._host(Hinv), i1, count, k, e, m)
        W.copy_(h)
        err_out.copy_(e)
        if mask_out is not None:
            mask_out.copy_(m)

    def sparsegpt_block_nm(self, W, Hinv, i1, count, n, m, err_out, mask_out=None):
        h = self._host(W)
        e = torch.empty(err_out.shape, dtype=torch.float32)
        mk = torch.zeros(err_out.shape, dtype=torch.uint8) if mask_out is not None else None
        self.o.sparsegpt_block_nm(h, self._host(Hinv), i1, count, n, m, e, mk)
        W.copy_(h)
        err_out.copy_(e)
        if mask_out is not None:
            mask_out.copy_(mk)

    def mask_mul(self, g, keep_mask):
        h = self._host(g)
        self.o.mask_mul(h, self._host(keep_mask))
        g.copy_(h)


def torch_cpu_normal(seed, like):
    """z exactly as the reference draws it for a CPU parameter
    (layer_single_base_pruner.py:482-485)."""
    torch.manual_seed(seed)
    return torch.normal(mean=0, std=1, size=like.size(), device="cpu", dtype=like.dtype)


class OracleKernelsK6Synced(OracleKernels):
    """The oracle backend for an END-TO-END comparison with the HIP backend on the GPU at true
    widths.  The column statistic (K6) is a float reduction: the oracle adds the squares in
    torch's CPU order, a GPU kernel in another, and the two agree to ~1e-6, not bit for bit; at
    true row lengths one differing ulp flips a near-tie of the selection, the pruned weight
    changes the next block's activations, and from there on two independent runs drift apart
    (measured: 538 of 4 194 304 positions by FlanT5's first decoder block) — which says nothing
    about any kernel.  So every K6 call is computed BOTH ways on the same input, the two are held
    to `rtol`, and the run goes on with the HIP value: everything downstream (metric, selection,
    zeroing, the next block's forward) then sees identical statistics and has to agree bit for
    bit, with every other kernel still the oracle's."""

    def __init__(self, hip_kernels, rtol=1e-5, **kw):
        super().__init__(**kw)
        self.hip = hip_kernels
        self.rtol = rtol
        self.k6_calls = 0
        self.k6_max_rel = 0.0

    def colsqnorm_accum(self, scaler_row, x2d, nsamples_before, batch):
        mine = scaler_row.clone()
        super().colsqnorm_accum(mine, x2d, nsamples_before, batch)
        self.hip.colsqnorm_accum(scaler_row, x2d, nsamples_before, batch)
        rel = ((scaler_row - mine).abs() / mine.abs().clamp_min(1e-30)).max().item()
        self.k6_calls += 1
        self.k6_max_rel = max(self.k6_max_rel, rel)
        assert rel <= self.rtol, f"K6: HIP vs oracle differ by {rel:.3e} relative"
