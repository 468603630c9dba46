"""Stage 1 of ECoFLaP: global importance scores -> per-layer sparsity table.

Host-side mirror of the reference's `LayerSparsity`
(LAVIS/lavis/compression/pruners/layer_single_base_pruner.py:120-561): same
constructor arguments, method names and return values, so the reference's pruners
(and its tests-as-scripts) can drive it unchanged.  What differs is HOW it runs:

  * every tensor op of the hot loops is one of the HIP kernels behind the C ABI
    (include/ecoflap_hip.h); there is no torch arithmetic on weights or gradients
    and no CPU fallback;
  * the +eps / -2eps / +eps perturbation triple (:530-539) is ONE kernel pass that
    emits theta+, theta- and the reference's drifted "restored" theta into three
    buffers the parameter pointer rotates through (bit-identical to three passes);
  * losses stay on the device in a [units, 2] table — one host sync per run
    instead of one `.item()` per loss pair (:544);
  * first-order gradients are never copied to the host nor accumulated per element
    (:453-461): each batch's (W, g) pairs are reduced to one double per layer by a
    single multi-tensor launch;
  * calibration batches may be sharded over ranks (one process per GPU); the
    exchange is ONE all-reduce of the loss table / the per-layer sums over RCCL.

The order in which the global NumPy RNG is consumed, the fp32/python-float
arithmetic of the score reduction (:544-549, :362-377) and the allocator (:247-314,
through `ecoflap_allocate_sparsity`) replay the reference exactly.
"""
import time

import numpy as np
import torch

from .. import hip as _hip
from .base_pruner import capture_graph

_f32 = np.float32


def _default_batch_len(batch):
    """Sample count of a calibration batch, as the reference's loss closures report it
    (pruners/utils.py:29,42: len(samples["text_input"]); :65: len(targets))."""
    for key in ("text_input", "label", "image"):
        if key in batch:
            return len(batch[key])
    raise ValueError("cannot infer the batch length; pass batch_len_fn")


class _UniformSparsity:
    """`return_sparsity()` result when no grouping is requested (:327-331)."""

    def __init__(self, value):
        self.value = value

    def __getitem__(self, key):
        return self.value


class LayerSparsity:
    def __init__(
        self,
        model,
        data_loader,
        loss_func,
        num_samples,
        original_sparsity,
        max_sparsity_per_layer=0.8,
        score_method="GradMagSquare_avg",
        num_noise=1,
        noise_eps=1e-3,
        layer_to_group_mapping={},
        prune_per_model=False,
        per_model_group=[],
        *,
        kernels=None,
        z_source="torch",
        batch_len_fn=None,
        process_group=None,
        k1_form="block",
        couple_torch_rng=False,
        grad_graphs=True,
        checkpoint_path=None,
        checkpoint_every=32,
    ):
        """Positional arguments are the reference's (:120-135).  Keyword-only extras:

        kernels       backend object; None = the HIP library (raises if it is not built).
        z_source      "torch" (default): z = torch.normal after torch.manual_seed(seed) on the
                      parameter's device, exactly as the reference draws it (:482-485);
                      "philox": z generated in registers by the kernel, never in memory (the
                      build's own stream: opt-in, no reference run can equal its table);
                      or a callable (seed, param) -> z tensor (parity tests).
        process_group torch.distributed group to shard calibration batches over
                      (None = use the default group if initialised, else single process).
        k1_form       "block"  (default) one launch for ALL layers that re-enter the forward at
                               one stage (a transformer block): every unit's theta+ / theta-
                               precomputed into scratch, each W read once, the drifted weights
                               parked until the layer's turn is over.  Needs in-register z and a
                               loss closure that knows its stages; otherwise it runs as "units";
                      "units"  one such launch per layer;
                      "triple" one fused launch per (layer, batch, noise) unit;
                      "single" three in-place launches per unit, the reference's call pattern.
                      All three are bit-identical.
        checkpoint_path  zeroth order: every `checkpoint_every` layers the loss table of the layers
                      done so far goes to this file (the reference has no mid-stage-1
                      resume although a run takes 100 min); a later run with the same model, batches,
                      seeds and path picks up behind the last saved layer — the finished layers'
                      weights get their K1 drift back from the seeds (no forwards) — and ends with
                      the same table and the same weights, bit for bit.
        """
        self.importance_measure = {}
        self.model = model
        self.data_loader = data_loader
        self.loss_func = loss_func
        self.num_samples = num_samples
        self.original_sparsity = original_sparsity
        self.layer_to_group_mapping = layer_to_group_mapping
        self.max_sparsity_per_layer = max_sparsity_per_layer
        self.num_noise = num_noise
        self.noise_eps = noise_eps
        self.prune_per_model = prune_per_model
        self.score_method = score_method
        self.per_model_group = per_model_group
        if score_method is not None:
            self.score_compute, self.score_aggregate = score_method.split("_")
        assert self.max_sparsity_per_layer >= self.original_sparsity

        self.kernels = kernels if kernels is not None else _hip.HipKernels()
        self.z_source = z_source
        self.batch_len_fn = batch_len_fn or _default_batch_len
        # losses that draw from torch's global RNG (BLIP retrieval's hard negatives,
        # torch.multinomial): the reference reseeds that RNG inside every K1 call (:482), so both
        # losses of a pair see the same draws; replay that state right before each loss
        self.couple_torch_rng = couple_torch_rng
        # first-order passes: forward + backward of equally shaped batches captured once as a HIP
        # graph and replayed (the eager loop is launch-bound at batch 1: thousands of tiny kernels)
        self.grad_graphs = grad_graphs
        self.process_group = process_group
        self.checkpoint_path = checkpoint_path
        self.checkpoint_every = max(1, int(checkpoint_every))
        self.resumed_layers = 0            # layers a zeroth-order run took over from its checkpoint
        self.emulate_rank_world = None      # (rank, world): see `_dist`
        assert k1_form in ("units", "block", "triple", "single")
        self.k1_form = k1_form
        self.stats = {}          # wall-clock + unit counts of the last run (reference: @print_time)
        self.loss_table = None   # [units, 2] fp32 (host copy) of the last zeroth-order run
        self.seed_schedule = None

    # ------------------------------------------------------------------ distributed helpers
    def _dist(self):
        import torch.distributed as dist
        if self.emulate_rank_world is not None:
            # one rank's share of a data-parallel run without a process group (bench.py
            # --emulate-rank): its batches, its K1 chaining of the not-owned units; the
            # closing all-reduce is skipped, so the TABLE of such a run means nothing
            rank, world = self.emulate_rank_world
            return None, int(rank), int(world)
        if dist.is_available() and dist.is_initialized():
            group = self.process_group
            return dist, dist.get_rank(group), dist.get_world_size(group)
        return None, 0, 1

    def _all_reduce_sum(self, t):
        dist, _, world = self._dist()
        if world > 1 and dist is not None:
            dist.all_reduce(t, op=dist.ReduceOp.SUM, group=self.process_group)
        return t

    # ------------------------------------------------------------------ K1
    def _torch_z_in_registers(self, param):
        """z_source="torch" on the GPU: the kernels regenerate torch.normal's own device stream in
        registers (no z tensor, no library launch) — when the backend has that form and its
        start-up probe finds it equal to THIS torch's draw on THIS device bit for bit
        (`HipKernels.torch_stream_matches`); otherwise, or with ECOFLAP_TORCH_Z=materialised,
        every z is drawn by torch itself and read back from memory."""
        if self.z_source != "torch":
            return False
        mode = getattr(self, "_torch_z_mode", None)
        if mode is None:
            import os
            mode = "materialised"
            probe = getattr(self.kernels, "torch_stream_matches", None)
            if (probe is not None and param.data.device.type == "cuda"
                    and os.environ.get("ECOFLAP_TORCH_Z", "registers") != "materialised"):
                if probe(param.data.device):
                    mode = "registers"
                else:
                    import warnings
                    warnings.warn("this torch's torch.normal stream is not the one the K1 kernels "
                                  "regenerate (torch / rocRAND changed?): z is drawn by torch and "
                                  "read from memory")
            self._torch_z_mode = mode
        return mode == "registers"

    def _draw_z(self, seed, param, materialise=False):
        if self.z_source == "philox":
            return None
        if self.z_source == "torch":
            # (a tensor of 2^31 elements or more is drawn by torch in several launches at advancing
            # Philox offsets: those draws are torch's own, read from memory)
            if not materialise and param.numel() < 2 ** 31 and self._torch_z_in_registers(param):
                return _hip.TORCH_Z
            torch.manual_seed(seed)
            return torch.normal(mean=0, std=1, size=param.data.size(), device=param.data.device,
                                dtype=param.data.dtype)
        z = self.z_source(seed, param)
        return z.to(device=param.data.device, dtype=param.data.dtype).contiguous()

    def zo_perturb_parameters(self, params, random_seed=1, scaling_factor=1, zo_eps=1e-3):
        """theta <- theta + scaling_factor * z * zo_eps, in place, one HIP launch per
        parameter (same name and arguments as the reference method, :473-486)."""
        if self.z_source != "philox":
            torch.manual_seed(random_seed)
        for param in params:
            z = None
            if self.z_source == "torch":
                z = torch.normal(mean=0, std=1, size=param.data.size(),
                                 device=param.data.device, dtype=param.data.dtype)
            elif self.z_source != "philox":
                z = self._draw_z(random_seed, param)
            self.kernels.zo_perturb(param.data, scaling_factor, zo_eps, random_seed, z)

    # ------------------------------------------------------------------ schedule
    def _draw_zs(self, seeds, param):
        """z argument of a multi-unit K1 call: None (the build's own in-register stream), the
        TORCH_Z marker (torch's stream in registers) or one tensor per unit."""
        if self.z_source == "philox":
            return None
        # (2^31 elements or more: torch draws in several launches at advancing Philox offsets —
        # not the single stream the kernels regenerate; those draws are torch's own, as in _draw_z)
        if param.numel() < 2 ** 31 and self._torch_z_in_registers(param):
            return _hip.TORCH_Z
        return [self._draw_z(sd, param) for sd in seeds]

    def _select(self, layer_to_group_mapping):
        names, params = [], []
        for k, v in self.model.named_parameters():
            if k in layer_to_group_mapping:
                names.append(k)
                params.append(v)
        return names, params

    def calibration_prefix(self):
        """The batches one pass of the reference's loop visits (:519-525: it leaves the loader
        once `accum_samples >= num_samples`): the loader is consumed LAZILY up to that point —
        never `list(loader)`, a UPop entrypoint hands over its whole training loader — and the
        prefix is taken ONCE and reused for every layer and both losses of a pair.  For a
        deterministic loader (LAVIS's DataLoaderWrapper, lists) that is what the reference sees,
        which re-iterates the loader per layer; a shuffling loader would give the reference a
        fresh random prefix per layer — the same distribution, not the same batches."""
        if isinstance(self.data_loader, (list, tuple)):
            seq = self.data_loader
        else:
            seq = iter(self.data_loader)
        out, accum = [], 0
        for d in seq:
            if accum >= self.num_samples:
                break
            out.append(d)
            n = self.batch_len_fn(d)
            for _ in range(self.num_noise):           # (:525-541: every noise draw spends samples)
                if accum >= self.num_samples:
                    break
                accum += n
        return out

    def build_zeroth_order_schedule(self, names):
        """Replay the reference's loop nest (:512-549) on the host only, consuming the
        global NumPy RNG at exactly the points it does, and return the list of units
        (layer index, batch index, noise index, seed, batch_len)."""
        batches = self.calibration_prefix()
        lens = [self.batch_len_fn(b) for b in batches]
        units = []
        for li, _ in enumerate(names):
            accum = 0
            for bi in range(len(batches)):
                if accum >= self.num_samples:
                    break
                for ni in range(self.num_noise):
                    if accum >= self.num_samples:
                        break
                    seed = int(np.random.randint(1000000000))
                    units.append((li, bi, ni, seed, lens[bi]))
                    accum += lens[bi]
        return batches, units

    # ------------------------------------------------------------------ stage-1 checkpoint
    def _checkpoint_file(self, rank, world):
        if not self.checkpoint_path:
            return None
        return self.checkpoint_path if world == 1 else f"{self.checkpoint_path}.rank{rank}of{world}"

    def _run_fingerprint(self, names, params, batches):
        """What a stage-1 checkpoint must share with the run that picks it up, beyond layer names
        and seeds (those depend on --seed, the architecture and the sample count alone): the
        perturbation size, how z is drawn, the sample / noise budget, the dtypes, the STARTING
        weights of every scored layer (sum |W| per layer, one reduce launch each: another
        --vit/t5_pruned_checkpoint or another init changes it) and the first calibration batch
        (float64 sums of its tensors).  A checkpoint of a run that differs in any of these holds
        another table: it is refused, not merged."""
        import hashlib
        import json
        zsrc = self.z_source if isinstance(self.z_source, str) else "callable"
        # (the loss closure and how many evaluations it batches are part of the identity: a table
        # whose deferred lock-step checks failed is never written — `_save_stage1_checkpoint` —
        # and the rerun that failure asks for, eval_batch=1, must not pick up a table of the
        # batched run either)
        eval_batch = getattr(self.loss_func, "eval_batch", None)
        head = {"zo_eps": float(self.noise_eps), "z_source": zsrc, "num_samples": int(self.num_samples),
                "num_noise": int(self.num_noise), "n_batches": len(batches),
                "dtypes": sorted({str(p.dtype) for p in params}), "score_method": str(self.score_method),
                "loss_closure": type(self.loss_func).__name__,
                "eval_batch": None if eval_batch is None else int(eval_batch)}
        h = hashlib.sha256(json.dumps(head, sort_keys=True).encode())
        if params:
            sums = self._weight_sums(params, _hip.RED_ABSW)
            h.update(np.asarray(sums, dtype=np.float64).tobytes())
        first = batches[0] if batches else None
        tensors = []
        if isinstance(first, dict):
            tensors = [first[k] for k in sorted(first, key=str) if torch.is_tensor(first[k])]
        elif isinstance(first, (list, tuple)):
            tensors = [v for v in first if torch.is_tensor(v)]
        for t in tensors:
            h.update(str(tuple(t.shape)).encode())
            h.update(np.float64(t.double().sum().item()).tobytes())
        return h.hexdigest(), head

    def _replica_identity(self, names, params, batches, units):
        """What the replicas of a data-parallel run must have in common before the pass starts, as
        digests: the seed of every unit (drawn from the process-global NumPy generator — anything
        else that draws from it between `np.random.seed` and here shifts them) and the starting
        weights of every scored layer (sum |W|).  The reference gives each rank `seed + rank`
        (LAVIS/evaluate_blip.py:287-295) because its ranks never share a table; ranks that fill
        ONE table need one schedule.  And, where every rank holds the same first batch (the
        harness: each rank builds the whole list; not bench.py's weak-scaling shards, where a rank
        holds only its own batches), they must COMPUTE alike: the loss of that batch at the
        starting weights, evaluated once by every rank, bit for bit — equal inputs are not enough
        where a library under the forward picks its kernel per process by measuring (MIOpen's Find
        behind `nn.Conv2d` did, between eight ranks sharing one device: shapes/eva_vit.py
        `PatchEmbed`, profiles/NOTES_r06.md §7).  Replicas that differ would still all-reduce a
        table without any error — rows of different runs next to each other — so a difference
        raises here.  -> the digests, for the stage statistics (one process: the seeds only,
        unless ECOFLAP_RUN_IDENTITY=1 asks for all of them — two runs can be compared by them)."""
        import hashlib
        import os
        seeds = hashlib.sha256(np.asarray([u[3] for u in units], dtype=np.int64).tobytes())
        seeds.update(repr([str(n) for n in names]).encode())
        out = {"seeds": seeds.hexdigest()[:16]}
        dist, rank, world = self._dist()
        shared = world > 1 and dist is not None
        if not shared and os.environ.get("ECOFLAP_RUN_IDENTITY") != "1":
            return out
        weights = hashlib.sha256()
        if params and params[0].is_cuda:
            weights.update(np.asarray(self._weight_sums(params, _hip.RED_ABSW), dtype=np.float64).tobytes())
        out["start_weights"] = weights.hexdigest()[:16]
        first = batches[0] if batches else None
        vals = (sorted(first.items(), key=lambda kv: str(kv[0])) if isinstance(first, dict)
                else list(enumerate(first)) if isinstance(first, (list, tuple)) else [])
        data = hashlib.sha256(repr([(str(k), tuple(v.shape)) for k, v in vals if torch.is_tensor(v)]).encode())
        sums = [v.double().sum() for _, v in vals if torch.is_tensor(v)]
        if sums:
            data.update(torch.stack(sums).cpu().numpy().tobytes())      # one read-back
        out["first_batch"] = data.hexdigest()[:16]

        def gather(keys):
            mine = torch.tensor([int(out[k][:15], 16) for k in keys], dtype=torch.int64)
            if dist.get_backend(self.process_group) == "nccl":
                mine = mine.to(params[0].device)
            every = [torch.zeros_like(mine) for _ in range(world)]
            dist.all_gather(every, mine, group=self.process_group)
            every = [e.cpu().tolist() for e in every]
            return {k: [e[i] for e in every] for i, k in enumerate(keys)}

        def refuse(key, got, why):
            r = next(i for i, v in enumerate(got) if v != got[0])
            raise RuntimeError(f"data-parallel stage 1: rank {r} differs from rank 0 in ['{key}'] — {why}")

        if shared:
            got = gather(["seeds", "start_weights", "first_batch"])
            if len(set(got["seeds"])) > 1:
                refuse("seeds", got["seeds"], "every rank must be started with the same --seed, and "
                       "nothing may draw from the global NumPy generator before the pruner runs")
            if len(set(got["start_weights"])) > 1:
                refuse("start_weights", got["start_weights"], "every rank must load the same weights")
            if len(set(got["first_batch"])) == 1 and first is not None:
                with torch.no_grad():
                    loss0 = self.loss_func(self.model, first, params[0].device.type != "cpu")[0]
                out["first_loss"] = hashlib.sha256(
                    loss0.detach().float().cpu().numpy().tobytes()).hexdigest()[:16]
                got = gather(["first_loss"])
                if len(set(got["first_loss"])) > 1:
                    refuse("first_loss", got["first_loss"],
                           "the ranks hold the same model and the same batch but their forwards round "
                           "differently: a library picked different kernels per process (a "
                           "convolution under MIOpen's timed Find: ecoflap_amd.shapes.fused."
                           "pin_patch_convs(model) runs a patch embedding as a GEMM, "
                           "MIOPEN_DEBUG_FIND_ONLY_SOLVER pins MIOpen's choice)")
        return out

    def _save_stage1_checkpoint(self, path, done, names, units, table, complete=False):
        """Layers [0, done) are finished: their rows of the loss table (this rank's entries) and
        what identifies the run (layer names, seeds, `_run_fingerprint`).  One stream sync; written
        atomically.  complete: the pass is over — a later run with the same identity takes the
        whole table from the file and replays only the K1 drift."""
        import os
        # checks the loss closure queued without a host sync (pruners/hooked_prefix.py) are read
        # BEFORE the rows they cover reach the disk: a failed one raises here and the file keeps
        # its last verified state
        check = getattr(self.loss_func, "check_assumed", None)
        if check is not None:
            check()
        host = table.detach().float().cpu().numpy() if table is not None else np.zeros(self._table_shape,
                                                                                       np.float32)
        tmp = path + ".tmp.npz"
        np.savez(tmp, done=np.array([done]), names=np.array(names), seeds=np.array([u[3] for u in units],
                 dtype=np.int64), table=host,
                 table_dtype=np.array(str(table.dtype) if table is not None else "torch.float32"),
                 fingerprint=np.array(self._fingerprint[0]), complete=np.array([int(bool(complete))]))
        os.replace(tmp, path)
        self.stats_checkpoints = getattr(self, "stats_checkpoints", 0) + 1

    def _load_stage1_checkpoint(self, path, names, units, device):
        """-> (layers already finished, their loss table on `device` or None)"""
        import os
        import warnings
        if not path or not os.path.exists(path):
            return 0, None
        with np.load(path, allow_pickle=False) as ck:
            same = (list(ck["names"]) == list(names)
                    and np.array_equal(ck["seeds"], np.array([u[3] for u in units], dtype=np.int64))
                    and tuple(ck["table"].shape) == tuple(self._table_shape))
            if not same:
                warnings.warn(f"stage-1 checkpoint {path} belongs to another run (layers / seeds "
                              "differ): ignored")
                return 0, None
            if "fingerprint" not in ck.files or str(ck["fingerprint"]) != self._fingerprint[0]:
                warnings.warn(f"stage-1 checkpoint {path} was written by a run with other starting "
                              "weights, calibration data, eps, z source or sample budget "
                              f"(this run: {self._fingerprint[1]}): ignored")
                return 0, None
            dtype = getattr(torch, str(ck["table_dtype"]).split(".")[-1])
            table = torch.from_numpy(ck["table"].copy()).to(device=device, dtype=dtype)
            return int(ck["done"][0]), table

    # ------------------------------------------------------------------ zeroth order
    def compute_importance_scores_mezo(self, layer_to_group_mapping):
        try:
            return self._compute_importance_scores_mezo(layer_to_group_mapping)
        except BaseException:
            abort = getattr(self.loss_func, "abort_run", None)
            if abort is not None:
                abort()
            raise

    def _compute_importance_scores_mezo(self, layer_to_group_mapping):
        t0 = time.time()
        model = self.model
        model.eval()
        names, params = self._select(layer_to_group_mapping)
        device = next(iter(model.parameters())).device
        cuda_enabled = device.type != "cpu"
        zo_eps = self.noise_eps
        _, rank, world = self._dist()

        batches, units = self.build_zeroth_order_schedule(names)
        self.seed_schedule = units
        n_units = len(units)
        identity = self._replica_identity(names, params, batches, units)
        # the loss pair of a unit, in the loss tensor's OWN dtype (fp32 for every loss closure of
        # the reference; a model returning a bf16 / fp16 loss has its subtraction and division
        # done in that dtype, as `(loss1 - loss2) / (2 * zo_eps)` is at :544): allocated on the
        # first loss
        table = None
        self._table_shape = (max(n_units, 1), 2)

        by_layer = {}
        for u, unit in enumerate(units):
            by_layer.setdefault(unit[0], []).append(u)

        n_forward = 0
        if hasattr(self.loss_func, "set_working_set"):
            self.loss_func.set_working_set(len(batches))
        begin_layer = getattr(self.loss_func, "begin_layer", None)
        # "block": consecutive layers owned by the same stage of the forward share one K1 launch
        groups, stash = {}, {}
        stage_of = getattr(self.loss_func, "stage_of", None)
        if (self.k1_form == "block" and stage_of is not None
                and hasattr(self.kernels, "zo_perturb_layers")):
            start = 0
            for li in range(1, len(names) + 1):
                if (li == len(names) or stage_of(names[li]) != stage_of(names[start])
                        or params[li].dtype != params[start].dtype):
                    if li - start > 1:
                        groups[start] = list(range(start, li))
                    start = li
            # A run of fewer than three matrices (a block the caller selected only part of; a
            # whole pass has none) is a launch too small to fill the chip for long — one wave
            # round, and the clock ramp after the forward's GEMMs is a sixth of it: its K1 rides
            # with the launch of the run in front of it when that one has the dtype (legal for
            # the reason block batching is: K1 of a layer depends on its original weights and
            # seeds only, and nothing writes those before the layer's own turn).
            starts = sorted(set(groups) | {i for i in range(len(names))
                                          if not any(i in g for g in groups.values())})
            runs = [groups.get(st, [st]) for st in starts]
            merged = []
            for run_ in runs:
                if (merged and len(run_) < 3 and len(merged[-1]) >= 3
                        and params[run_[0]].dtype == params[merged[-1][0]].dtype
                        and len(merged[-1]) + len(run_) <= 16):
                    merged[-1] = merged[-1] + run_
                else:
                    merged.append(list(run_))
            groups = {g[0]: g for g in merged if len(g) > 1}
        max_units = getattr(self.kernels, "MAX_UNITS", 32)
        ck_file = self._checkpoint_file(rank, world)
        if ck_file:
            self._fingerprint = self._run_fingerprint(names, params, batches)
        resume_done, resumed = self._load_stage1_checkpoint(ck_file, names, units, device)
        if resumed is not None:
            table = resumed
        self.resumed_layers = resume_done
        for li, (name, param) in enumerate(zip(names, params)):
            home = param.data
            layer_units = by_layer.get(li, [])
            if li < resume_done:
                # finished before the interruption: its losses are in the table; its weights get
                # the drift of its K1 chain back — a function of the original weights and the
                # seeds alone — without a forward
                if layer_units:
                    seeds_ = [units[u][3] for u in layer_units]
                    none_ = [None] * len(layer_units)
                    self.kernels.zo_perturb_units(home, zo_eps, seeds_, none_, list(none_),
                                                  self._draw_zs(seeds_, param))
                continue
            if (ck_file and li > resume_done and (li - resume_done) % self.checkpoint_every == 0):
                self._save_stage1_checkpoint(ck_file, li, names, units, table)
            if begin_layer is not None:
                begin_layer(name)     # exact suffix-only re-forward (pruners/prefix_cache.py)
            owned = [(units[u][1] % world) == rank for u in layer_units]
            if li in groups and all(0 < len(by_layer.get(g, [])) <= max_units for g in groups[li]):
                # K1 of every layer of the block now, each from its ORIGINAL weights; the drifted
                # weights wait in `final` until the layer's own turn is over
                batch = []
                for g in groups[li]:
                    g_units = by_layer[g]
                    g_owned = [(units[u][1] % world) == rank for u in g_units]
                    g_home = params[g].data
                    scr = torch.empty((2 * max(sum(g_owned), 1),) + tuple(g_home.shape),
                                      dtype=g_home.dtype, device=g_home.device)
                    plus, minus, kk = [], [], 0
                    for mine in g_owned:
                        plus.append(scr[2 * kk] if mine else None)
                        minus.append(scr[2 * kk + 1] if mine else None)
                        kk += int(mine)
                    fin = torch.empty_like(g_home)
                    stash[g] = (plus, minus, fin, scr)
                    item = (g_home, fin, [units[u][3] for u in g_units], plus, minus)
                    if self.z_source != "philox":
                        # parity mode: every unit's z drawn as the reference draws it (each draw
                        # re-seeds, :482-485, so drawing the block's layers ahead of their turn
                        # changes no value); not-owned units need theirs for the drift
                        item += (self._draw_zs([units[u][3] for u in g_units], params[g]),)
                    batch.append(item)
                self.kernels.zo_perturb_layers(batch, zo_eps)
                del batch, item
            if self.k1_form in ("units", "block"):
                # one launch: theta+/theta- of every owned unit into scratch, final drifted theta
                # back into the parameter's own storage; then only forwards remain
                static_w = bool(getattr(self.loss_func, "requires_static_weights", False))
                # (a closure without graphs may batch evaluations too: pruners/hooked_prefix.py)
                paired = bool(getattr(self.loss_func, "supports_pairs", lambda: False)())
                if paired and self.couple_torch_rng:
                    if static_w:
                        raise RuntimeError(
                            "couple_torch_rng needs one loss at a time (the torch RNG is re-seeded "
                            "before each loss, :482): use a loss closure without lanes / batching")
                    paired = False
                ahead = stash.pop(li, None)
                if ahead is not None:        # K1 ran with its block: `home` still holds the originals
                    plus, minus, final, scratch = ahead
                    zs = []
                else:
                    n_owned = sum(owned)
                    scratch = torch.empty((2 * max(n_owned, 1),) + tuple(home.shape), dtype=home.dtype,
                                          device=home.device)
                    plus, minus, k = [], [], 0
                    for u, mine in zip(layer_units, owned):
                        plus.append(scratch[2 * k] if mine else None)
                        minus.append(scratch[2 * k + 1] if mine else None)
                        k += int(mine)
                    zs = self._draw_zs([units[u][3] for u in layer_units], param)
                    if layer_units:
                        self.kernels.zo_perturb_units(
                            home, zo_eps, [units[u][3] for u in layer_units], plus, minus, zs)
                    final = home.clone() if static_w else None   # graphs bake the address of `home`
                if paired:
                    self.loss_func.begin_layer_weights(name, home)
                if paired:
                    # theta+/theta- of several units on concurrent lanes (pruners/prefix_cache.py)
                    mine_js = [j for j, mine in enumerate(owned) if mine]
                    step = self.loss_func.pairs_in_flight()
                    for c0 in range(0, len(mine_js), step):
                        chunk = mine_js[c0:c0 + step]
                        items = [(batches[units[layer_units[j]][1]], plus[j], minus[j]) for j in chunk]
                        with torch.no_grad():
                            res = self.loss_func.multi(self.model, items, cuda_enabled)
                        self.loss_func.join()
                        for j, (l1, l2, batch_len) in zip(chunk, res):
                            u = layer_units[j]
                            if batch_len != units[u][4]:
                                raise RuntimeError("loss_func batch_len differs from the schedule")
                            table = self._table_for(table, l1, device)
                            table[u, 0].copy_(l1.detach(), non_blocking=True)
                            table[u, 1].copy_(l2.detach(), non_blocking=True)
                            n_forward += 2
                for j, (u, mine) in enumerate(zip(layer_units, owned)):
                    if not mine or paired:
                        continue
                    _, bi, _, _, blen = units[u]
                    if static_w:
                        home.copy_(plus[j])
                    else:
                        param.data = plus[j]    # theta + eps z
                    table = self._loss_into(table, u, 0, batches[bi], cuda_enabled, blen,
                                            rng=(units[u][3], param))
                    if static_w:
                        home.copy_(minus[j])
                    else:
                        param.data = minus[j]   # theta - eps z
                    table = self._loss_into(table, u, 1, batches[bi], cuda_enabled, blen,
                                            rng=(units[u][3], param))
                    n_forward += 2
                if static_w or ahead is not None:
                    home.copy_(final)
                if paired:
                    self.loss_func.end_layer_weights(final)
                param.data = home               # "recovered" weights, with the reference's drift
                del scratch, plus, minus, zs, final, ahead
                continue
            cur = home
            spare = [torch.empty_like(home), torch.empty_like(home)]
            for u, mine in zip(layer_units, owned):
                _, bi, _, seed, blen = units[u]
                z = self._draw_z(seed, param)
                if self.k1_form == "single":
                    # reference call pattern: three in-place passes (:530-539)
                    param.data = cur
                    self.kernels.zo_perturb(cur, 1, zo_eps, seed, z)
                    if mine:
                        table = self._loss_into(table, u, 0, batches[bi], cuda_enabled, blen, rng=(seed, param))
                    self.kernels.zo_perturb(cur, -2, zo_eps, seed, z)
                    if mine:
                        table = self._loss_into(table, u, 1, batches[bi], cuda_enabled, blen, rng=(seed, param))
                    self.kernels.zo_perturb(cur, 1, zo_eps, seed, z)
                    n_forward += 2 * int(mine)
                    continue
                if mine:
                    minus, restored = spare
                    self.kernels.zo_perturb_triple(cur, cur, minus, restored, zo_eps, seed, z)
                    param.data = cur            # theta + eps z
                    table = self._loss_into(table, u, 0, batches[bi], cuda_enabled, blen, rng=(seed, param))
                    param.data = minus          # theta - eps z
                    table = self._loss_into(table, u, 1, batches[bi], cuda_enabled, blen, rng=(seed, param))
                    param.data = restored       # "recovered" weights, with the reference's drift
                    cur, spare = restored, [minus, cur]
                    n_forward += 2
                else:
                    # another rank evaluates this batch; only carry the rounding drift so
                    # every replica ends with the weights the single-process run leaves
                    self.kernels.zo_perturb_triple(cur, None, None, spare[1], zo_eps, seed, z)
                    cur, spare = spare[1], [spare[0], cur]
                    param.data = cur
            if cur is not home:                  # keep the parameter's own storage
                home.copy_(cur)
            param.data = home
            del spare

        if ck_file and len(names) > resume_done:
            # the pass is over: mark the file complete (a rerun of the same command replays the
            # drift from it; a modified command is refused by the fingerprint)
            self._save_stage1_checkpoint(ck_file, len(names), names, units, table, complete=True)
        if getattr(self, "_torch_z_mode", None) == "registers" and units and params:
            # leave torch's generators where the reference's last K1 call leaves them (:482-485)
            self._draw_z(units[-1][3], params[units[-1][0]], materialise=True)
        t_enqueued = time.time() - t0            # host done; the device may still be replaying
        if hasattr(self.loss_func, "finish_run"):
            self.loss_func.finish_run()
        if table is None:                        # this rank evaluated nothing (more ranks than batches)
            table = torch.zeros(self._table_shape, dtype=torch.float32, device=device)
        t_ar = None
        if world > 1:
            # each entry is written by exactly one rank.  Timed for the bench line (BASELINE.md §3
            # row 4): an event pair on the stream the collective is ordered on (RCCL), and the
            # host's wall time of the call (gloo blocks the host instead)
            ev = None
            if table.is_cuda:
                ev = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
                ev[0].record()
            t_ar = time.time()
            self._all_reduce_sum(table)
            t_ar = time.time() - t_ar
            if ev is not None:
                ev[1].record()
        # (loss1 - loss2) / (2 eps) with torch's own tensor ops in the loss dtype on the loss
        # tensors' device, as the reference evaluates it per pair (:544), then `.item()` ->
        # python float; ONE sync for the run.
        projected = ((table[:, 0] - table[:, 1]) / (2 * zo_eps)).float().cpu().numpy()
        self.loss_table = table.float().cpu().numpy()
        allreduce = None
        if t_ar is not None:
            allreduce = {"bytes": int(table.numel() * table.element_size()), "host_wall_ms": 1e3 * t_ar,
                         "stream_ms": float(ev[0].elapsed_time(ev[1])) if ev is not None else None}

        grad_sum = {}
        for li, name in enumerate(names):
            acc = _f32(0)                        # gradients_dict[name], fp32 tensor (:549)
            started = False
            per_batch = {}
            for u in by_layer.get(li, []):
                per_batch.setdefault(units[u][1], []).append(u)
            for bi in sorted(per_batch):
                s = 0                            # python number accumulates |pg| over noise (:547)
                for u in per_batch[bi]:
                    s += abs(float(projected[u]))
                v = abs(_f32(s))                 # torch.FloatTensor([s]).abs()
                acc = v if not started else _f32(acc + v)
                started = True
            grad_sum[name] = acc if started else 0
        # a loader with no batch leaves the int 0 of the reference's dict (:501)

        importance = {}
        if self.score_compute == "MEZO-GradOnly":
            for name in names:
                importance[name] = torch.tensor([abs(float(grad_sum[name]))], dtype=torch.float32)
        elif self.score_compute in ("MEZO-GradMagAbs", "MEZO-GradMagSquare"):
            mode = _hip.RED_ABSW if self.score_compute == "MEZO-GradMagAbs" else _hip.RED_SQW
            sums = self._weight_sums(params, mode)
            for name, wsum in zip(names, sums):
                s = float(grad_sum[name])
                factor = abs(s) if mode == _hip.RED_ABSW else s * s
                # sum_e |W_e| * s  (:556)  /  sum_e W_e^2 * s^2  (:559), reduced on the device
                importance[name] = torch.tensor([_f32(wsum * factor)], dtype=torch.float32)
        else:
            raise ValueError(f"unknown zeroth-order score_method {self.score_method!r}")
        self.stats = {"seconds": time.time() - t0, "layers": len(names), "units": n_units,
                      "forwards": n_forward, "world_size": world, "run_identity": identity,
                      "host_enqueue_seconds": t_enqueued, "loss_table_allreduce": allreduce,
                      "z_mode": ("philox" if self.z_source == "philox" else
                                 "torch-" + getattr(self, "_torch_z_mode", "materialised")
                                 if self.z_source == "torch" else "callable")}
        return importance

    def _couple_rng(self, seed, param):
        if not self.couple_torch_rng:
            return
        torch.manual_seed(seed)
        if self.z_source != "philox":
            # leaves the generator where the reference's draw does
            self._draw_z(seed, param, materialise=True)

    def _table_for(self, table, loss, device):
        if table is None:
            table = torch.zeros(self._table_shape, dtype=loss.dtype, device=device)
        elif table.dtype != loss.dtype:
            raise RuntimeError(f"loss dtype changed within a run: {table.dtype} -> {loss.dtype}")
        return table

    def _loss_into(self, table, unit, col, batch, cuda_enabled, expected_len, rng=None):
        if rng is not None:
            self._couple_rng(*rng)
        with torch.no_grad():
            loss, batch_len = self.loss_func(self.model, batch, cuda_enabled)
        if batch_len != expected_len:
            raise RuntimeError(
                f"loss_func reported batch_len {batch_len}, schedule assumed {expected_len}; "
                "pass batch_len_fn matching the loss closure")
        table = self._table_for(table, loss, table.device if table is not None else loss.device)
        table[unit, col].copy_(loss.detach(), non_blocking=True)
        return table

    def _weight_sums(self, params, mode):
        device = params[0].device
        out = torch.zeros(len(params), dtype=torch.float64, device=device)
        for i, p in enumerate(params):
            self.kernels.absprod_reduce(p.data, None, mode, out[i:i + 1])
        return out.cpu().numpy()

    # ------------------------------------------------------------------ first order
    def compute_importance_scores(self, layer_to_group_mapping):
        t0 = time.time()
        model = self.model
        names, params = self._select(layer_to_group_mapping)
        device = next(iter(model.parameters())).device
        cuda_enabled = device.type != "cpu"
        _, rank, world = self._dist()
        mode = {"GradMagSquare": _hip.RED_SQW_SQG, "GradMagAbs": _hip.RED_ABSW_ABSG,
                "GradOnly": _hip.RED_ABSG}.get(self.score_compute)
        if mode is None:
            raise ValueError(f"unknown first-order score_method {self.score_method!r}")

        sums = torch.zeros(len(names), dtype=torch.float64, device=device)
        accum_samples = 0
        n_batches = 0
        todo = []
        for bi, d in enumerate(self.data_loader):
            if accum_samples >= self.num_samples:
                break
            accum_samples += self.batch_len_fn(d)
            n_batches += 1
            if (bi % world) == rank:
                todo.append(d)
        for _, grads in self._grads_per_batch(params, todo, cuda_enabled):
            assert len(grads) == len(names) == len(params)
            self._reduce_pairs(params, grads, mode, sums)
        self._all_reduce_sum(sums)
        host = sums.cpu().numpy()
        importance = {}
        for name, s in zip(names, host):
            # (sum_b term_b) / n_batches: the reference divides the accumulator by the batch
            # count (:461) before the product; the product is linear in it
            importance[name] = torch.tensor([_f32(s / max(n_batches, 1))], dtype=torch.float32)
        self.stats = {"seconds": time.time() - t0, "layers": len(names), "batches": n_batches,
                      "world_size": world}
        return importance

    def _grads_per_batch(self, params, todo, cuda_enabled):
        """Yields (batch, grads) for every batch of `todo`: `torch.autograd.grad(loss, params)`
        as the reference calls it (:446), eagerly — or, on the GPU when all batches have the same
        shapes, by replaying ONE captured HIP graph of forward + backward on static input
        buffers (same kernels, same order -> same bits; the gradients live in the graph's static
        buffers and are consumed by the caller's kernel before the next replay)."""
        model = self.model
        dev = params[0].device

        def tensors_of(b):
            return [v for v in (b.values() if isinstance(b, dict) else b) if torch.is_tensor(v)]

        def signature(b):
            if not isinstance(b, (dict, tuple, list)):
                return None
            items = b.items() if isinstance(b, dict) else enumerate(b)
            sig = []
            for k, v in items:
                if torch.is_tensor(v):
                    if v.device != dev:
                        return None
                    sig.append((k, tuple(v.shape), v.dtype))
                elif isinstance(v, (int, float, str, type(None))):
                    sig.append((k, v))
                else:
                    return None            # lists of per-sample counts etc.: shapes vary
            return tuple(sig)

        sigs = {signature(b) for b in todo}
        use_graph = (self.grad_graphs and dev.type == "cuda" and len(todo) >= 4
                     and len(sigs) == 1 and None not in sigs)
        if not use_graph:
            for d in todo:
                loss, batch_len = self.loss_func(model, d, cuda_enabled)
                if batch_len != self.batch_len_fn(d):
                    raise RuntimeError("loss_func batch_len differs from batch_len_fn")
                grads = torch.autograd.grad(loss, params)
                assert len(grads) == len(params)
                yield d, grads
                del grads, loss
            return
        # the captured graph bakes in the parameters' storage addresses and the loss closure:
        # a re-pointed `param.data` (model_reset's .type(), the non-static K1 forms) or another
        # loss_func must not replay it
        import os
        n_lanes = max(1, min(int(os.environ.get("ECOFLAP_GRAD_LANES", getattr(self, "grad_lanes", 3))),
                             len(todo)))
        key = (next(iter(sigs)), tuple((id(p), p.data_ptr()) for p in params), id(self.loss_func),
               id(self.model), n_lanes)
        cache = getattr(self, "_grad_graph_cache", None)
        if cache is None or cache[0] != key:
            first = todo[0]
            main = torch.cuda.current_stream()
            lanes = []
            self.stats_grad_graph = {"captured": 0, "replays": 0, "lanes": n_lanes}
            for li in range(n_lanes):
                # one captured forward + backward per lane, each on its own stream with its own
                # static inputs and gradient buffers: the batch-1 graph is latency-bound
                # (thousands of 5-10 us kernels), several of them in flight fill the device
                static = ({k: (v.clone() if torch.is_tensor(v) else v) for k, v in first.items()}
                          if isinstance(first, dict) else
                          type(first)(v.clone() if torch.is_tensor(v) else v for v in first))
                stream = torch.cuda.Stream()
                stream.wait_stream(main)
                with torch.cuda.stream(stream):       # warm-up off the capture, as torch asks
                    for _ in range(2 if li == 0 else 1):
                        loss, batch_len = self.loss_func(model, static, cuda_enabled)
                        torch.autograd.grad(loss, params)
                stream.synchronize()
                if batch_len != self.batch_len_fn(first):
                    raise RuntimeError("loss_func batch_len differs from batch_len_fn")
                graph = torch.cuda.CUDAGraph()
                with capture_graph(graph, stream=stream, capture_error_mode="thread_local"):
                    loss, _ = self.loss_func(model, static, cuda_enabled)
                    grads = torch.autograd.grad(loss, params)
                assert len(grads) == len(params)
                lanes.append((graph, static, grads, stream))
                self.stats_grad_graph["captured"] += 1
            # the graphs read the parameters in place: later rounds (Real-*: pruned weights in
            # the same storage) replay them as they are
            self._grad_graph_cache = cache = (key, lanes)
        _, lanes = cache
        main = torch.cuda.current_stream()
        in_flight = []
        for i, d in enumerate(todo):
            graph, static, grads, stream = lanes[i % len(lanes)]
            stream.wait_stream(main)      # the consumer of this lane's previous gradients is queued on `main`
            with torch.cuda.stream(stream):
                for dst, src in zip(tensors_of(static), tensors_of(d)):
                    dst.copy_(src, non_blocking=True)
                graph.replay()
            self.stats_grad_graph["replays"] += 1
            in_flight.append((d, grads, stream))
            if len(in_flight) == len(lanes):          # oldest first: batch order is kept
                d0, g0, s0 = in_flight.pop(0)
                main.wait_stream(s0)
                yield d0, g0
        for d0, g0, s0 in in_flight:
            main.wait_stream(s0)
            yield d0, g0

    def _reduce_pairs(self, params, grads, mode, sums):
        """sums[l] += sum_e f(W_l, g_l): one multi-tensor launch per dtype class."""
        self.kernels.absprod_reduce_pairs([p.data for p in params], list(grads), mode, sums)

    # ------------------------------------------------------------------ Real-* (global iterative)
    def accumulate_abs_grads(self, params):
        """Per-element `acc += |g|` over the calibration batches (:433-455) -> (accs, n_batches).
        One flat fp32 buffer in HBM (14.8 GB for BLIP-2) viewed per layer, one multi-tensor launch
        per batch.  Data-parallel: batches sharded by global index, ONE all-reduce of the flat
        buffer per round — the only bandwidth-sized collective of the framework (ring over xGMI;
        fp32 sums re-associate across ranks, so ties at the threshold may resolve differently
        from the single-process run; replicas always agree with each other)."""
        model = self.model
        device = next(iter(model.parameters())).device
        cuda_enabled = device.type != "cpu"
        _, rank, world = self._dist()
        flat = torch.zeros(sum(p.numel() for p in params), dtype=torch.float32, device=device)
        accs, o = [], 0
        for p in params:
            accs.append(flat[o:o + p.numel()].view(p.shape))
            o += p.numel()
        accum_samples, n_batches = 0, 0
        todo = []
        for bi, d in enumerate(self.data_loader):
            if accum_samples >= self.num_samples:
                break
            accum_samples += self.batch_len_fn(d)
            n_batches += 1
            if (bi % world) == rank:
                todo.append(d)
        for _, grads in self._grads_per_batch(params, todo, cuda_enabled):
            self.kernels.grad_accum_multi(accs, list(grads))
        self._all_reduce_sum(flat)
        return accs, n_batches

    def global_iterative_pruning(self, target_sparsity, dict_layers_to_prune, iteratation=1,
                                 max_sparsity_per_layer=1.0):
        """Three rounds of: first-order per-element importance -> ONE global threshold over all
        prunable elements -> prune; then the per-parameter zero fractions, weights restored
        (:199-245).  Per-element |g| accumulators (fp32, 14.8 GB for BLIP-2) live in HBM; the
        score is never materialised: the threshold kernels recompute it from (W, acc, mask)."""
        t0 = time.time()
        _, _, world = self._dist()
        names, params = self._select(dict_layers_to_prune)
        sc = self.score_compute
        mode = 1 if "GradMagSquare" in sc else (0 if "GradMagAbs" in sc else 2)     # (:463-469)
        weight_copy = [p.data.clone() for p in params]
        masks = [torch.ones(p.shape, dtype=torch.uint8, device=p.device) for p in params]
        total = sum(p.numel() for p in params)
        for i in range(1, iteratation + 1):
            p_i = target_sparsity ** (iteratation / i)                                # (:213)
            accs, n_batches = self.accumulate_abs_grads(params)    # |g| even for *Square (:452)
            k = int(p_i * total)                                                       # (:173)
            # get_mask's protection step (:160-167): the top int(numel * (1 - max)) scores of
            # every layer cannot be pruned (a no-op at the reference's own max = 1.0, :324)
            protect = [int(p.numel() * (1 - max_sparsity_per_layer)) for p in params]
            self.kernels.global_threshold_prune([p.data for p in params], accs, masks, mode,
                                                n_batches, k,
                                                protect if any(c > 0 for c in protect) else None)
            del accs
        model = self.model
        all_names, all_params = [], []
        for k_, v in model.named_parameters():
            all_names.append(k_)
            all_params.append(v.data if v.data.is_contiguous() else v.data.contiguous())
        counts = self.kernels.count_zeros_multi(all_params)
        sparsity_dict = {k_: float(_f32(c) / _f32(v.numel()))                          # (:237)
                         for k_, v, c in zip(all_names, all_params, counts)}
        for p, w in zip(params, weight_copy):
            p.data.copy_(w)                                                            # (:239-243)
        self.stats = {"seconds": time.time() - t0, "layers": len(names), "iterations": iteratation,
                      "world_size": world}
        return sparsity_dict

    # ------------------------------------------------------------------ allocation
    def compute_the_sparsity_per_group(self, total_parameters_to_keep, group_scores,
                                       group_num_parameters, max_sparsity_per_layer=0.8):
        """dict group -> sparsity (python float holding an fp32 value), via the C ABI's
        host allocator, which replays the reference's mixed-dtype arithmetic (:247-314)."""
        keys = list(group_num_parameters.keys())
        scores = [float(group_scores[k]) for k in group_scores]
        nums = [int(group_num_parameters[k]) for k in keys]
        sparsity, keep = _hip.allocate_sparsity(scores, nums, total_parameters_to_keep,
                                                max_sparsity_per_layer)
        self.last_keep = dict(zip(keys, keep))
        return dict(zip(keys, sparsity))

    def return_sparsity(self):
        t0 = time.time()
        original_sparsity = self.original_sparsity
        mapping = self.layer_to_group_mapping
        if self.score_compute.startswith("Real"):
            # the layer sparsities a real global iterative pruning would produce (:321-325)
            return self.global_iterative_pruning(
                original_sparsity, mapping, iteratation=3, max_sparsity_per_layer=1.0)
        if mapping is None or len(mapping) == 0:
            return _UniformSparsity(original_sparsity)

        dev0 = next(iter(self.model.parameters())).device
        if dev0.type == "cuda":
            from .. import blas_guard
            # reproducible GEMMs (ecoflap_amd/blas_guard.py); batch invariance is asked for only
            # by a loss closure that concatenates evaluations
            blas_guard.verify(dev0, need_batch_invariance=int(
                getattr(self.loss_func, "eval_batch", 1) or 1) > 1)
        if len(self.importance_measure) == 0:
            if self.score_compute.startswith("MEZO"):
                self.importance_measure = self.compute_importance_scores_mezo(mapping)
            else:
                self.importance_measure = self.compute_importance_scores(mapping)

        group_to_layers = {}
        for layer, group in mapping.items():
            group_to_layers.setdefault(group, []).append(layer)

        numel = {}
        total_parameters = 0
        for k, v in self.model.named_parameters():
            if k in mapping:
                numel[k] = v.numel()
                total_parameters += v.numel()
        total_parameters_to_keep = int(total_parameters * (1 - original_sparsity))

        group_scores, group_num_parameters = {}, {}
        for group, layers in group_to_layers.items():
            acc = _f32(0)                      # 0 + fp32 0-dim tensors, added in mapping order (:370)
            count = 0
            for layer in layers:
                acc = _f32(acc + _f32(self._layer_sum(layer)))
                count += numel[layer]
            if self.score_aggregate == "avg":
                acc = _f32(acc / _f32(count))  # tensor /= python int, fp32 (:375)
            group_scores[group] = acc
            group_num_parameters[group] = count

        if self.prune_per_model:
            group_sparsity = {}
            for prefix in self.per_model_group:
                sub_scores = {k: v for k, v in group_scores.items() if k.startswith(prefix)}
                sub_nums = {k: v for k, v in group_num_parameters.items() if k.startswith(prefix)}
                sub_keep = int(sum(sub_nums.values()) * (1 - original_sparsity))
                group_sparsity.update(self.compute_the_sparsity_per_group(
                    sub_keep, sub_scores, sub_nums,
                    max_sparsity_per_layer=self.max_sparsity_per_layer))
        else:
            group_sparsity = self.compute_the_sparsity_per_group(
                total_parameters_to_keep, group_scores, group_num_parameters,
                max_sparsity_per_layer=self.max_sparsity_per_layer)

        kept = 0
        for g in group_num_parameters:
            kept += (1 - group_sparsity[g]) * group_num_parameters[g]
        self.stats["kept_vs_target"] = (kept, total_parameters_to_keep)   # reference prints it (:407)
        self.stats["return_sparsity_seconds"] = time.time() - t0
        return {layer: group_sparsity[group] for layer, group in mapping.items()}

    def _layer_sum(self, layer):
        v = self.importance_measure[layer]
        if torch.is_tensor(v):
            return float(v.sum()) if v.numel() != 1 else float(v.reshape(()))
        return float(v)
