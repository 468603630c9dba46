#!/usr/bin/env python3
"""bench.py — layers scored / second, zeroth-order ECoFLaP on a BLIP-2 (EVA-ViT-g +
Q-Former + FlanT5-XL) shaped random-init model at 0.5 sparsity (BASELINE.json configs[2];
SURVEY.md §8d).

A "step" = one prunable weight matrix taken through the whole of the reference's inner
loops (layer_single_base_pruner.py:512-549): for each of its calibration batches one
+eps/-2eps/+eps perturbation triple (K1), two full forward losses, and its row of the loss
table.  The K timed steps are K matrices — runs of 6 consecutive ones at evenly strided
positions over the model's 588 (ViT, T5 encoder, T5 decoder alike); the closing all-reduce of the loss table, the single host sync,
the score reduction and the allocator are inside the timed region.

    python bench.py --gpus 1 --steps 12 --warmup 2
    python bench.py --gpus N --steps K --warmup W          (starts its own N ranks, see below)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

`--gpus N` with N > 1 and no launcher environment (WORLD_SIZE unset): this process makes no
GPU call at all; it starts N children of itself, one per GPU (RANK / LOCAL_RANK / WORLD_SIZE /
MASTER_ADDR=127.0.0.1 / MASTER_PORT set, backend nccl = RCCL), relays rank 0's JSON line and
exits with the worst child return code.

Weak scaling: every rank holds a full weight replica and scores the SAME matrices on its
own 128 calibration pairs (16 batches of 8); N ranks = 128*N pairs per matrix
(configs[3] at N=8).  `value` counts a matrix scored on 128 pairs as one unit, so the
whole-job aggregate is N * K / seconds.  One process per GPU, RCCL only for the one
all-reduce of the loss table.
"""
import os
os.environ.setdefault("TENSILE_STREAMK_DATA_PARALLEL", "1")   # before the first GEMM (ecoflap_amd/blas_guard.py)
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")      # RCCL between processes: dmabuf IPC only on this pool
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402
import torch  # noqa: E402

HBM_PEAK_GBS = 8000.0   # MI355X HBM3E spec peak (MI355X_MICROARCH.md); ~6.3 TB/s achievable


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=24)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--num-data", type=int, default=128, help="calibration pairs per rank")
    ap.add_argument("--batch-size", type=int, default=8)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--layers", default=None,
                    help="analysis only: comma-separated matrix indices to time instead of the strided "
                         "sample (--steps is set to their count; not the headline workload)")
    ap.add_argument("--cpu-baseline-layer", type=int, default=-1)
    ap.add_argument("--k1-form", default="block", choices=["block", "units", "triple", "single"],
                    help="block: one K1 launch per transformer block (all its matrices, default); "
                         "units: one per layer; triple: one fused launch per (layer,batch) unit; "
                         "single: the reference's three in-place passes")
    ap.add_argument("--full-forward", action="store_true",
                    help="two full forwards per unit like the reference, instead of the exact "
                         "suffix-only re-forward (pruners/prefix_cache.py)")
    ap.add_argument("--no-graphs", action="store_true",
                    help="launch the suffix forwards eagerly instead of replaying HIP graphs")
    ap.add_argument("--lanes", type=int, default=2, choices=[1, 2, 3, 4, 6, 8],
                    help="2: theta+ and theta- suffixes replay concurrently (second weight "
                         "replica + second stream); 1: one after the other")
    ap.add_argument("--eval-batch", type=int, default=16,
                    help="evaluations of a layer whose shared suffix runs once on their "
                         "concatenated states (exact; see pruners/prefix_cache.py)")
    ap.add_argument("--no-batched-advance", action="store_true",
                    help="A/B: move the prefix cache one batch at a time")
    ap.add_argument("--toy", action="store_true", help="tiny shapes (plumbing check only)")
    ap.add_argument("--dist-backend", default="nccl",
                    help="nccl (= RCCL) in production; 'gloo' + --same-device only to exercise "
                         "the N>1 code path on a single-GPU box")
    ap.add_argument("--same-device", action="store_true")
    ap.add_argument("--k1-blocker", default="spin", choices=["spin", "gemm", "none"],
                    help="what keeps the launch stream busy while the host enqueues the event pair "
                         "around a K1 launch: a one-thread spin kernel (default; leaves the chip's "
                         "power state alone), four 4096^3 GEMMs (round 1; they pull the clock down "
                         "for the kernel that follows), or nothing (the pair then also times the "
                         "host's enqueue gap)")
    ap.add_argument("--profile-host", default=None,
                    help="cProfile the timed region's host side into this file (diagnosis only: "
                         "the profiler slows the loop)")
    ap.add_argument("--scaling", default="weak", choices=["weak", "strong"],
                    help="weak (default, what the driver's scaling run measures): every rank scores "
                         "the matrices on its own --num-data pairs; strong: --num-data pairs in "
                         "total, split over the ranks (a user with one calibration set and N GPUs)")
    ap.add_argument("--emulate-world", type=int, default=0,
                    help="PROJECTION, one process, no process group: run rank --emulate-rank's share "
                         "of a W-rank data-parallel pass on this GPU — its shard of the global batch "
                         "list and the drift-only K1 chaining of the other ranks' units — and print "
                         "a line flagged \"projection\": true.  No collective runs and nothing "
                         "about xGMI / RCCL is measured: the line says what one rank's GPU would "
                         "spend per step, not what N GPUs achieve")
    ap.add_argument("--emulate-rank", type=int, default=0)
    ap.add_argument("--z-source", default="torch", choices=["torch", "philox"],
                    help="torch (default, what every entrypoint defaults to): the reference's own draw, "
                         "torch.manual_seed(seed) + torch.normal on the device (layer_single_base_pruner.py:"
                         "482-485), regenerated in registers by K1; philox: the build's own cheaper "
                         "in-register stream (no reference run can equal its table)")
    ap.add_argument("--unstaged", action="store_true",
                    help="analysis: the same model with its stage_plan() hidden — what a reference user "
                         "who swaps the import hands the pruners (INTEGRATION.md §A) — scored through "
                         "pruners/hooked_prefix.py (no HIP graphs; the chunk's evaluations in lock step)")
    ap.add_argument("--no-parity-leg", action="store_true",
                    help="skip the short leg after the timed region that scores two blocks in each z mode")
    ap.add_argument("--no-k1-events", action="store_true",
                    help="no event pairs, no blocker: the uninstrumented loop (rocprofv3 "
                         "cross-check of the K1 durations; the line then carries no roofline)")
    return ap.parse_args()


def strided(n_total, k, offset=0, run=6, block_starts=None):
    """k layer indices as runs of `run` consecutive matrices at evenly strided positions.
    A real pass visits the 588 matrices in order, ~6.6 per block, so per-block costs
    (prefix-cache advance, graph capture, the block's K1 launch) amortise over consecutive
    matrices; runs of 6 keep that structure in a k-matrix sample.  With `block_starts` a run
    begins at the first matrix of a transformer block, as every block of the real pass does (its
    K1 launch then covers the block's matrices of the run instead of the tail of one block and
    the head of the next)."""
    n_runs = max(1, (k + run - 1) // run)
    out = []
    for i in range(n_runs):
        start = int((i + 0.5) * n_total / n_runs) + offset
        start -= start % run
        if block_starts:
            start = max(b for b in block_starts if b <= start) + offset
        for j in range(run):
            if len(out) < k:
                out.append(min(n_total - 1, start + j))
    return out


class HipEvents:
    """Raw hipEvent_t pairs through libamdhip64 (ctypes): `ecoflap_zo_perturb_units_timed` fills
    them with the kernel's own begin / end timestamps."""

    def __init__(self):
        import ctypes
        self.ct = ctypes
        self.rt = ctypes.CDLL("libamdhip64.so")
        self.rt.hipEventCreate.argtypes = [ctypes.POINTER(ctypes.c_void_p)]
        self.rt.hipEventElapsedTime.argtypes = [ctypes.POINTER(ctypes.c_float), ctypes.c_void_p,
                                                ctypes.c_void_p]
        self.rt.hipEventSynchronize.argtypes = [ctypes.c_void_p]
        self.rt.hipEventDestroy.argtypes = [ctypes.c_void_p]

    def pair(self):
        out = []
        for _ in range(2):
            ev = self.ct.c_void_p()
            rc = self.rt.hipEventCreate(self.ct.byref(ev))
            assert rc == 0, f"hipEventCreate -> {rc}"
            out.append(ev)
        return tuple(out)

    def elapsed_us(self, start, stop):
        ms = self.ct.c_float()
        assert self.rt.hipEventSynchronize(stop) == 0
        rc = self.rt.hipEventElapsedTime(self.ct.byref(ms), start, stop)
        assert rc == 0, f"hipEventElapsedTime -> {rc}"
        return 1e3 * ms.value


class TimedKernels:
    """Wraps the HIP backend and times every K1 launch on the launch stream.

    Layer-batched form (the bench's default, `units`): the launch itself carries the event pair
    (`ecoflap_zo_perturb_units_timed` -> hipExtLaunchKernelGGL), which records the kernel's own
    begin / end timestamps — the duration rocprofv3's kernel trace reports for the same dispatch;
    nothing is queued around the launch and the loop runs as it does uninstrumented.
    The other K1 forms (`triple`, `single`) keep a torch event pair behind a one-thread spin
    kernel (see `_timed`)."""

    def __init__(self, inner):
        self.inner = inner
        self.enabled = False
        self.records = []   # (start_event, end_event, algorithmic_bytes, kind, n_launches)
        self.blocker = None
        self.spin_cycles = 600000
        self.hip_events = None
        self.unit_records = []   # (start, stop, algorithmic_bytes) raw HIP events
        self.layers_per_launch = []

    def __getattr__(self, name):
        return getattr(self.inner, name)

    def _timed(self, kind, nbytes, fn, *a, n_launches=1, **kw):
        if not self.enabled:
            return fn(*a, **kw)
        # The GPU drains its queue while Python prepares the next launch, so a bare event pair
        # would also time the host gap between its two records.  ~0.3 ms of a ONE-THREAD spin
        # kernel queued first keeps the stream busy while the host enqueues start-event, kernel
        # and stop-event back to back; it is not part of the timed span and, unlike the GEMMs
        # round 1 used here, leaves the chip's power state alone (the same bf16 K1 launch:
        # 156 us right after four 4096^3 GEMMs, 119 us after the spin; round 1, the tool is in the history).
        if self.blocker == "spin":
            torch.cuda._sleep(self.spin_cycles)
        elif self.blocker is not None:
            for _ in range(4):
                torch.mm(self.blocker, self.blocker)
        s = torch.cuda.Event(enable_timing=True)
        e = torch.cuda.Event(enable_timing=True)
        s.record()
        r = fn(*a, **kw)
        e.record()
        self.records.append((s, e, nbytes, kind, n_launches))
        return r

    def zo_perturb_triple(self, w_in, w_plus, w_minus, w_restored, zo_eps, seed, z=None):
        s = w_in.element_size()
        full = w_plus is not None
        nbytes = (4 if full else 2) * s * w_in.numel()     # read W, write theta+, theta-, theta
        return self._timed("triple" if full else "drift", nbytes, self.inner.zo_perturb_triple,
                           w_in, w_plus, w_minus, w_restored, zo_eps, seed, z)

    def zo_perturb_units(self, w, zo_eps, seeds, w_plus, w_minus, z=None):
        if not self.enabled:
            return self.inner.zo_perturb_units(w, zo_eps, seeds, w_plus, w_minus, z)
        if self.hip_events is None:
            self.hip_events = HipEvents()
        s, n, mx = w.element_size(), w.numel(), self.inner.MAX_UNITS
        chunks = [(c0, min(len(seeds), c0 + mx)) for c0 in range(0, len(seeds), mx)]
        it = iter(chunks)

        def events():
            c0, c1 = next(it)
            owned = sum(1 for t in w_plus[c0:c1] if t is not None)
            pair = self.hip_events.pair()
            # per launch: read W, write theta+/theta- of its owned units, write the drifted W
            # (+ one read of z per unit when z comes from memory: the parity mode)
            zr = (c1 - c0) if (z is not None and not isinstance(z, str)) else 0
            self.unit_records.append((pair[0], pair[1], (2 * owned + 2 + zr) * s * n))
            return pair
        return self.inner.zo_perturb_units(w, zo_eps, seeds, w_plus, w_minus, z, events=events)

    def zo_perturb_layers(self, layers, zo_eps):
        if not self.enabled:
            return self.inner.zo_perturb_layers(layers, zo_eps)
        if self.hip_events is None:
            self.hip_events = HipEvents()
        # per layer: read W, write theta+/theta- of its owned units, write the drifted W
        # (+ one read of z per unit when z comes from memory: the parity mode)
        nbytes = sum((2 * sum(1 for t in it[3] if t is not None) + 2
                      + (len(it[5]) if len(it) > 5 and isinstance(it[5], (list, tuple)) else 0))
                     * it[0].element_size() * it[0].numel() for it in layers)
        pair = self.hip_events.pair()
        self.unit_records.append((pair[0], pair[1], nbytes))
        self.layers_per_launch.append(len(layers))
        return self.inner.zo_perturb_layers(layers, zo_eps, events=lambda: pair)

    def zo_perturb(self, w, scaling_factor, zo_eps, seed, z=None):
        nbytes = 2 * w.element_size() * w.numel()          # read W, write W
        return self._timed("single", nbytes, self.inner.zo_perturb, w, scaling_factor, zo_eps,
                           seed, z)

    def event_floor_us(self, n=24):
        """Median reading of the same event pair around an EMPTY kernel (after the timed
        region): the part of every K1 reading that is not the kernel."""
        if self.hip_events is None:
            return None
        import statistics
        pairs = []
        for _ in range(n):
            p = self.hip_events.pair()
            rc = self.inner.lib.ecoflap_null_launch_timed(
                self.hip_events.ct.c_void_p(torch.cuda.current_stream().cuda_stream), p[0], p[1])
            assert rc == 0
            pairs.append(p)
        torch.cuda.synchronize()
        return statistics.median(self.hip_events.elapsed_us(a, b) for a, b in pairs)

    @staticmethod
    def box_write_gbs(nbytes=1 << 30, reps=5):
        """What THIS box writes at, measured after the timed region: a plain single-stream fill of
        1 GiB (torch events on the current stream).  K1's traffic is 32 parts write to 1 part read;
        boxes of the pool differ by +-10 % in this figure and K1's in-situ rate moves with it."""
        import statistics
        buf = torch.empty(nbytes, dtype=torch.uint8, device="cuda")
        buf.fill_(0)
        torch.cuda.synchronize()
        out = []
        for i in range(reps):
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s.record(); buf.fill_(i + 1); e.record()
            torch.cuda.synchronize()
            out.append(nbytes / (s.elapsed_time(e) * 1e-3) / 1e9)
        del buf
        return statistics.median(out)

    def summary(self, kind):
        if kind in ("units", "block"):
            recs = [(self.hip_events.elapsed_us(a, b) * 1e-6, nb, 1) for a, b, nb in self.unit_records]
        else:
            recs = [(s.elapsed_time(e) * 1e-3, b, n) for s, e, b, k, n in self.records if k == kind]
        if not recs:
            return None
        t = sum(r[0] for r in recs)
        b = sum(r[1] for r in recs)
        n = sum(r[2] for r in recs)
        return {"launches": n, "avg_us": 1e6 * t / n, "bytes_per_launch": b / n, "gbs": b / t / 1e9,
                "per_launch": [{"us": round(1e6 * r[0], 2), "bytes": r[1],
                                "frac": round(r[1] / r[0] / 1e9 / HBM_PEAK_GBS, 4)} for r in recs]}


def self_launch(n, argv, script=None, timeout=None):
    """Start `n` ranks of `script` (default: this file) as CHILD processes, one per GPU, and
    return (worst return code, rank 0's stdout).  The caller has not touched the GPU (a process
    that has initialised HIP must never be replaced by / fork another GPU user), so the children
    are plain `subprocess` children with the launcher environment torch.distributed expects."""
    import socket
    import subprocess
    with socket.socket() as sock:
        sock.bind(("127.0.0.1", 0))
        port = sock.getsockname()[1]
    import tempfile
    procs = []
    with tempfile.TemporaryFile("w+") as rank0_out:
        for r in range(n):
            env = dict(os.environ)
            env.update({"RANK": str(r), "LOCAL_RANK": str(r), "WORLD_SIZE": str(n),
                        "LOCAL_WORLD_SIZE": str(n), "MASTER_ADDR": "127.0.0.1",
                        "MASTER_PORT": str(port), "HSA_ENABLE_IPC_MODE_LEGACY": "0"})
            procs.append(subprocess.Popen(
                [sys.executable, script or os.path.abspath(__file__)] + list(argv), env=env,
                stdout=rank0_out if r == 0 else sys.stderr, stderr=sys.stderr))
        t_end = None if timeout is None else time.time() + timeout
        worst = 0
        killed_here = set()                       # ranks the launcher itself ended
        while True:
            codes = [p.poll() for p in procs]
            if all(c is not None for c in codes):
                break
            failed = any(c not in (None, 0) for c in codes)
            if failed or (t_end is not None and time.time() > t_end):
                # one rank died (the others would sit in a collective until the process group's
                # own timeout) or the caller's limit passed: end exactly the PIDs started here
                worst = 1 if failed else 124
                for i, p in enumerate(procs):
                    if p.poll() is None:
                        killed_here.add(i)
                        p.kill()
                for p in procs:
                    p.wait()
                break
            time.sleep(0.2)
        for i, p in enumerate(procs):
            rc = p.returncode
            # the code reported is that of the rank that FAILED, not the -9 of the survivors this
            # launcher killed because of it; a signal (negative) reads as 128 + signal
            if rc and worst != 124 and i not in killed_here:
                worst = max(worst, rc if rc > 0 else 128 - rc)
        rank0_out.seek(0)
        out0 = rank0_out.read()
    return worst, out0


def main():
    args = parse()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # no GPU call before this point (module imports only)
        # counting devices goes through amdsmi / sysfs on this image and does not initialise HIP
        # in the parent; where torch has to fall back to hipGetDeviceCount the check is skipped
        # (ECOFLAP_BENCH_NO_COUNT=1) and the ranks themselves report a missing device
        visible = (args.gpus if os.environ.get("ECOFLAP_BENCH_NO_COUNT")
                   else torch.cuda.device_count())
        if visible < args.gpus and not args.same_device:
            print(f"bench.py: --gpus {args.gpus} but only {visible} GPU(s) visible",
                  file=sys.stderr)
            sys.exit(2)
        rc, out0 = self_launch(args.gpus, sys.argv[1:])
        sys.stdout.write(out0)
        sys.stdout.flush()
        sys.exit(rc)
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    emulated = args.emulate_world > 1
    if emulated and (world != 1 or args.gpus != 1 or not 0 <= args.emulate_rank < args.emulate_world):
        print("bench.py: --emulate-world runs as ONE process with --gpus 1 and "
              "0 <= --emulate-rank < --emulate-world", file=sys.stderr)
        sys.exit(2)
    if world != args.gpus:
        print(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}", file=sys.stderr)
        sys.exit(2)
    assert torch.cuda.is_available(), "bench.py needs an MI355X"
    if args.same_device:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    import torch.distributed as dist
    rccl_ranks = None
    if world > 1:
        if args.dist_backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)     # RCCL over xGMI
        else:
            dist.init_process_group(args.dist_backend)
        # read back from the process group, after one real collective on it
        probe = torch.ones(1, device=dev)
        dist.all_reduce(probe)
        torch.cuda.synchronize()
        assert int(probe.item()) == world
        rccl_ranks = {"backend": dist.get_backend(), "world_size": dist.get_world_size(),
                      "all_reduce_of_ones": int(probe.item())}

    from ecoflap_amd import hip
    from ecoflap_amd.pruners import LayerSparsity
    from ecoflap_amd.pruners.losses import loss_vision_language
    from ecoflap_amd.shapes import synthetic as S
    from ecoflap_amd.shapes.blip2_t5 import blip2_flant5xl, blip2_toy

    if args.unstaged:
        from ecoflap_amd.shapes.blip2_t5 import Blip2T5
        from ecoflap_amd.shapes.unstaged import hide_stage_plan
        hide_stage_plan(Blip2T5)
    torch.manual_seed(0)
    t_build = time.time()
    with torch.device(dev):
        model = (blip2_toy(fp32=False) if args.toy else blip2_flant5xl()).eval()
    for p in model.parameters():
        p.requires_grad = False
    img = 28 if args.toy else 224
    vocab = 96 if args.toy else 32128
    # every rank its own shard of the global calibration set (global batch index = rank + N*j)
    # the world the SCHEDULE sees: the process group's, or the emulated one (no group)
    s_world, s_rank = (args.emulate_world, args.emulate_rank) if emulated else (world, rank)
    strong = args.scaling == "strong" and s_world > 1
    if strong:
        if (args.num_data // args.batch_size) % s_world != 0:
            print(f"bench.py: --scaling strong needs the {args.num_data // args.batch_size} batches "
                  f"to divide over {s_world} ranks", file=sys.stderr)
            sys.exit(2)
        args.num_data //= s_world
    batches_local = S.image_text_batches(args.num_data, args.batch_size, img_size=img, vocab=vocab,
                                         in_len=16, out_len=16, seed=42 + s_rank, device=dev)
    nb_local = len(batches_local)
    # global list seen by the schedule: batch j*world + r lives on rank r (placeholders elsewhere)
    batches = []
    for j in range(nb_local):
        for r in range(s_world):
            batches.append(batches_local[j] if r == s_rank else
                           {"text_input": batches_local[j]["text_input"]})
    num_samples_global = args.num_data * s_world

    prunable = [k for k, v in model.named_parameters()
                if v.dim() == 2 and ".block" in k and "relative_attention_bias.weight" not in k
                and (k.startswith("t5_model") or k.startswith("visual_encoder"))]
    group_of = lambda k: ".".join(k.split(".")[:4 if k.startswith("t5_model") else 3])  # noqa: E731
    full_mapping = {k: group_of(k) for k in prunable}
    n_total = len(prunable)
    block_starts = [i for i, k in enumerate(prunable)
                    if i == 0 or group_of(k) != group_of(prunable[i - 1])]
    params_by_name = dict(model.named_parameters())
    numel_total = sum(params_by_name[k].numel() for k in prunable)
    build_s = time.time() - t_build

    kern = TimedKernels(hip.HipKernels())
    kern.blocker = {"spin": "spin", "none": None,
                    "gemm": torch.randn(4096, 4096, device=dev, dtype=torch.bfloat16)
                    if args.k1_blocker == "gemm" else None}[args.k1_blocker]

    from ecoflap_amd.pruners.prefix_cache import PrefixCachedLoss

    # one loss closure for warm-up and timed region: graph captures, library heuristics for the
    # new GEMM shapes and the batch-invariance probe are one-time costs of a pruning run (588
    # layers), so the warm-up steps absorb them; the prefix cache itself is reset in between
    if args.full_forward:
        shared_loss = loss_vision_language
    elif args.unstaged:
        # what `blipt5_wanda_pruner` builds for a model without stage_plan() (pruners/wanda.py)
        from ecoflap_amd.pruners.hooked_prefix import HookedPrefixLoss
        assert not hasattr(model, "stage_plan")
        shared_loss = HookedPrefixLoss(model, loss_vision_language,
                                       ["visual_encoder.blocks", "t5_model.encoder.block",
                                        "t5_model.decoder.block"], ["ln_vision", "Qformer", "t5_proj"],
                                       eval_batch=args.eval_batch)
    else:
        shared_loss = PrefixCachedLoss(model, use_graphs=not args.no_graphs,
                                       n_lanes=args.lanes, eval_batch=args.eval_batch,
                                       batched_advance=not args.no_batched_advance)

    def run(layer_ids, timed, z_source=None, reset=None, events=True):
        z_source = args.z_source if z_source is None else z_source
        mapping = {prunable[i]: full_mapping[prunable[i]] for i in layer_ids}
        np.random.seed(42)
        loss_fn = shared_loss
        if hasattr(loss_fn, "reset"):
            if (not timed) if reset is None else reset:
                loss_fn.reset()
            # (the timed region keeps the prefix cache the warm-up left at the block in front of
            # its first matrix — the steady state of a real pass, where the cached states are
            # always one block behind; building them from the raw batches is a one-off of a
            # 588-matrix run like the graph captures)
            for key in ("stage_calls", "stage_calls_full", "advance_calls", "graph_captures",
                        "graph_replays", "capture_seconds", "batched_evals", "batched_checks",
                        "host_blocked_seconds", "guard_gpu_seconds", "lockstep_seconds", "lockstep_evals",
                        "events_total", "events_served", "forwards", "events_shared", "events_per_evaluation"):
                if key in loss_fn.stats:
                    loss_fn.stats[key] = 0
        run.loss_fns.append(loss_fn)
        ls = LayerSparsity(model, batches, loss_fn, num_samples_global, 0.5, 0.6,
                           "MEZO-GradOnly_sum", 1, 1e-3, mapping, kernels=kern,
                           z_source=z_source, k1_form=args.k1_form)
        if emulated:
            ls.emulate_rank_world = (s_rank, s_world)
        kern.enabled = timed and events and not args.no_k1_events
        out = ls.return_sparsity()
        kern.enabled = False
        run.z_modes.append(ls.stats.get("z_mode"))
        return ls, out

    run.loss_fns = []
    run.z_modes = []
    # ---- warmup (untimed) ---------------------------------------------------------------
    if args.warmup > 0:
        # the first matrix of the model (every later stage gets captured / probed once) plus the
        # matrices right in front of the timed sample's first one
        first = (int(args.layers.split(",")[0]) if args.layers
                 else strided(n_total, args.steps, block_starts=block_starts)[0])
        run(sorted(set([0] + [i for i in range(first - (args.warmup - 1), first) if i > 0]))
            if args.warmup > 1 else [0], timed=False)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()

    # ---- timed region: exactly K steps -----------------------------------------------------
    layer_ids = strided(n_total, args.steps, block_starts=block_starts)
    if args.layers:
        layer_ids = [int(x) for x in args.layers.split(",")]
        args.steps = len(layer_ids)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    if args.profile_host:
        import cProfile
        prof = cProfile.Profile()
        ls, table = prof.runcall(run, layer_ids, True)
        prof.dump_stats(args.profile_host)
    else:
        ls, table = run(layer_ids, timed=True)
    run.z_mode_timed = run.z_modes[-1]
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    import copy as _copy
    timed_stats = _copy.deepcopy(getattr(run.loss_fns[-1], "stats", None))   # before the extra legs

    class _Snap:
        stats = timed_stats or {}
    snap = _Snap() if timed_stats is not None else object()
    # ---- outside the timed region: the three ways K1 can get its z, on the same matrices ----
    # One ViT-g block and one FlanT5 decoder block, each scored three times from the same cached
    # prefix: "torch" = the reference's draw regenerated in registers (the default, the timed
    # region's mode unless --z-source says otherwise), "torch_materialised" = every z drawn by
    # torch.normal itself and read from memory (round 4's default; what the start-up probe falls
    # back to), "philox" = the build's own in-register stream.
    z_modes = None
    if not args.no_parity_leg and not args.full_forward and world == 1 and not emulated:
        kern.enabled = False
        vit_blocks = [b for b in block_starts if prunable[b].startswith("visual_encoder")]
        dec_blocks = [b for b in block_starts if ".decoder." in prunable[b]]
        picks = [bs[len(bs) // 2] for bs in (vit_blocks, dec_blocks) if bs]
        ends = {b: (block_starts[block_starts.index(b) + 1]
                    if block_starts.index(b) + 1 < len(block_starts) else n_total) for b in picks}

        class DrawTimer:
            """The reference's draw (layer_single_base_pruner.py:482-485) with a torch event pair
            around it: what the z tensors cost to produce when torch.normal's own kernels write
            each z once and K1 reads it once."""

            def __init__(self):
                self.pairs, self.bytes = [], 0

            def __call__(self, seed, param):
                s_, e_ = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                s_.record()
                torch.manual_seed(seed)
                z = torch.normal(mean=0, std=1, size=param.data.size(), device=param.data.device,
                                 dtype=param.data.dtype)
                e_.record()
                self.pairs.append((s_, e_))
                self.bytes += z.numel() * z.element_size()
                return z

        draws = DrawTimer()
        main_records, main_lpl = kern.unit_records, kern.layers_per_launch
        legs, k1_of, lpl_of, seen_mode = {}, {}, {}, {}
        modes = [("torch", "torch"), ("torch_materialised", draws), ("philox", "philox")]
        for b in picks:
            ids = list(range(b, ends[b]))
            run([b], timed=False, reset=True)               # prefix cache -> this block, untimed
            for mode, src in modes:
                kern.unit_records, kern.layers_per_launch = k1_of.setdefault(mode, []), lpl_of.setdefault(mode, [])
                torch.cuda.synchronize()
                t1 = time.perf_counter()
                # (the K1 launches carry their event pair: same instrumentation as the timed
                # region's, it costs the loop nothing)
                run(ids, timed=True, z_source=src, reset=False)
                torch.cuda.synchronize()
                legs.setdefault(mode, []).append((len(ids), time.perf_counter() - t1))
                seen_mode[mode] = run.z_modes[-1]
        z_modes = {"what": "one ViT-g block and one FlanT5 decoder block scored in each z mode from the "
                           "same cached prefix, outside the timed region",
                   "first_matrices": [prunable[b] for b in picks], "modes": {}}
        for mode, _ in modes:
            kern.unit_records, kern.layers_per_launch = k1_of[mode], lpl_of[mode]
            ksum = kern.summary(args.k1_form if args.k1_form in ("units", "block") else "units") \
                if kern.unit_records else None
            n_l = sum(n for n, _ in legs[mode])
            entry = {"z_mode": seen_mode[mode], "layers": n_l,
                     "layers_per_s": n_l / sum(t for _, t in legs[mode]),
                     "ms_per_layer": [1e3 * t / n for n, t in legs[mode]]}
            if ksum:
                entry["k1"] = {"achieved": ksum["gbs"], "peak": HBM_PEAK_GBS, "unit": "GB/s",
                               "frac": ksum["gbs"] / HBM_PEAK_GBS, "launches": ksum["launches"],
                               "avg_launch_us": ksum["avg_us"],
                               "algorithmic_bytes_per_launch": ksum["bytes_per_launch"],
                               "bytes_rule": ("sum over the launch's layers of (3*U+2)*s*numel: z read "
                                              "from memory per unit" if mode == "torch_materialised" else
                                              "sum over the launch's layers of (2*U+2)*s*numel: no z bytes"),
                               "per_launch": ksum["per_launch"]}
            if mode == "torch_materialised":
                draw_s = sum(a_.elapsed_time(b_) for a_, b_ in draws.pairs) * 1e-3
                entry["z_draws"] = {"count": len(draws.pairs), "bytes_written": draws.bytes, "seconds": draw_s,
                                    "gbs": (draws.bytes / draw_s / 1e9) if draw_s > 0 else None}
                if ksum and draw_s > 0:
                    tot_b = ksum["bytes_per_launch"] * ksum["launches"] + draws.bytes
                    tot_s = ksum["avg_us"] * ksum["launches"] * 1e-6 + draw_s
                    entry["k1_plus_draws_frac_of_peak"] = tot_b / tot_s / 1e9 / HBM_PEAK_GBS
            z_modes["modes"][mode] = entry
        kern.unit_records, kern.layers_per_launch = main_records, main_lpl

    kind = args.k1_form
    k1 = kern.summary(kind)
    drift = kern.summary("drift")
    # weak: a matrix scored on 128 pairs per rank = one unit per rank; strong: the ranks share
    # ONE calibration set, a matrix is scored once by all of them together
    value = (1 if strong else s_world) * args.steps / elapsed

    out = {
        "metric": "layers scored/sec (zeroth-order, BLIP-2 @0.5)",
        "value": value,
        "unit": "layers/s (one layer = one weight matrix scored on 128 calibration pairs)",
        "n_gpus": world,
        "rccl_ranks": (rccl_ranks["world_size"] if rccl_ranks and rccl_ranks["backend"] == "nccl"
                       else None),
        "process_group": rccl_ranks,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": 1e3 * elapsed / args.steps,
        "higher_is_better": True,
        "scaling": "strong" if strong else "weak",
        "vs_baseline": None,
        "dtype": "bf16/fp16 weights (T5 bf16, ViT-g fp16), fp32 arithmetic per op",
        "data": "synthetic",
        "config": {
            "workload": ("toy shapes" if args.toy else
                         "BASELINE configs[2]: BLIP-2 (EVA-ViT-g fp16 + Q-Former + FlanT5-XL bf16) "
                         "zeroth-order MEZO-GradOnly_sum, block groups, max 0.6, eps 1e-3, "
                         "128 synthetic image-text pairs per GPU, batch 8 -> 16 batches per layer"),
            "prunable_matrices": n_total,
            "prunable_elements": numel_total,
            "layers_timed": layer_ids,
            "pairs_per_gpu": args.num_data,
            "pairs_total": args.num_data * s_world,
            "batch_size": args.batch_size,
            "forwards_per_step": 2 * nb_local,
            "k1_form": args.k1_form,
            "z_source": args.z_source,
            "z_mode": run.z_mode_timed,
            "model_protocol": ("un-staged: stage_plan() hidden, block lists only (pruners/hooked_prefix.py)"
                               if args.unstaged else "staged: stage_plan() (pruners/prefix_cache.py)"),
            "forward_form": ("2 full forwards per unit" if args.full_forward else
                             (f"exact suffix-only re-forward through forward patches on the block lists, "
                              f"{args.eval_batch} evaluations in lock step (blocks behind the owner once "
                              "per chunk, eager launches)") if args.unstaged else
                             "exact suffix-only re-forward from the owning block (activations "
                             "at the block boundary cached per batch)"
                             + ("" if args.no_graphs else ", suffix replayed as a HIP graph")
                             + (f", {args.eval_batch} evaluations per pass: the batch-invariant "
                                "part of the suffix (FlanT5 stages) runs once on their "
                                "concatenated states, verified bit for bit against one suffix "
                                "per evaluation"
                                if (args.eval_batch > 1 and not args.no_graphs) else
                                (f", {args.lanes} evaluations in flight on concurrent lanes"
                                 if (args.lanes > 1 and not args.no_graphs) else ""))),
            "parallelism": (f"dp{s_world} (batch-sharded, one all-reduce of the loss table)"
                            + (" — EMULATED: one rank's share on one GPU, no collective" if emulated else "")),
        },
        "breakdown": {
            "model_build_s": build_s,
            "forwards_total": ls.stats.get("forwards"),
            # host side of the timed region: time spent issuing work vs. time spent blocked on the
            # device (graph captures of newly entered stages synchronise; the bitwise guard reads
            # a loss back) — the loop is device-bound when enqueue << ms_per_step
            "host_enqueue_ms_per_step": 1e3 * max(0.0, ls.stats.get("host_enqueue_seconds", 0.0)
                                                  - _blocked(snap)) / args.steps,
            "host_blocked_on_device_ms_per_step": 1e3 * _blocked(snap) / args.steps,
            # inside the timed region and inside `value`: the bitwise guard's own sequential
            # re-evaluations, one-off per entry stage (this sample: one per ~2.5 matrices; a whole
            # 588-matrix run: one per ~9)
            "guard_gpu_ms_per_step": (1e3 * getattr(snap, "stats", {}).get(
                "guard_gpu_seconds", 0.0) / args.steps),
            "cpu_model": _cpu_model_name(),
            # stage 1's one exchange: the [units, 2] loss table, all-reduced once per pass (N > 1)
            "allreduce_ms": ((ls.stats.get("loss_table_allreduce") or {}).get("stream_ms")
                             or (ls.stats.get("loss_table_allreduce") or {}).get("host_wall_ms")),
            "allreduce": ls.stats.get("loss_table_allreduce"),
            "k1_ms_per_step": (k1["avg_us"] * k1["launches"] / args.steps / 1e3) if k1 else None,
            "drift_only": drift,
            "suffix_forward": (_compact_stats(snap.stats) if hasattr(snap, "stats") else None),
            # the forward's 16-bit Linears: hipBLASLt solution pinned per weight shape
            # (csrc/gemm_pinned.hip); an index means something only with the library version
            "pinned_gemm": _pinned_gemm_report(),
        },
    }
    if k1:
        out["roofline"] = {
            "kernel": ("zo_torch_layers_kernel (ecoflap_zo_perturb_layers_torch: torch.normal's own "
                       "device stream regenerated in registers; every K1 form goes through it)"
                       if run.z_mode_timed == "torch-registers" else
                       {"units": "zo_perturb_units_kernel", "triple": "zo_perturb_triple_kernel",
                        "single": "zo_perturb_kernel",
                        "block": "zo_perturb_layers_kernel (+ zo_perturb_units_kernel for a layer "
                                 "that is alone in its block within the sample)"}[kind]
                       + ("<HAS_Z>" if run.z_mode_timed == "torch-materialised" else "")),
            "z_mode": run.z_mode_timed,
            "bound": "hbm",
            "achieved": k1["gbs"],
            "peak": HBM_PEAK_GBS,
            "unit": "GB/s",
            "frac": k1["gbs"] / HBM_PEAK_GBS,
            "traffic": load_pmc_traffic("torch_" + kind if run.z_mode_timed == "torch-registers" else kind,
                                        k1["bytes_per_launch"]),
            # NOT a reading of this run (PMC counters cannot be collected from inside the process):
            # this run's algorithmic bytes x the traffic/algorithmic ratio the committed rocprofv3
            # --pmc passes measured for the same kernel (FETCH_SIZE and WRITE_SIZE in separate
            # passes, FETCH doubled for gfx950 as MI355X_MICROARCH.md prescribes)
            "traffic_source": "derived: algorithmic bytes x the ratio in profiles/k1_pmc_traffic.json "
                              "(rocprofv3 --pmc, separate passes, of the same kernel); not a "
                              "counter reading of this run",
            "launches": k1["launches"],
            "avg_launch_us": k1["avg_us"],
            "layers_per_launch": kern.layers_per_launch if kind == "block" else None,
            "timing": ("kernel begin/end timestamps in HIP events attached to the launch "
                       "(hipExtLaunchKernelGGL)" if kind in ("units", "block") else
                       f"torch event pair behind a '{args.k1_blocker}' blocker"),
            "per_launch": k1["per_launch"],
            # not subtracted from anything: `achieved` is the raw reading
            "event_pair_floor_us": kern.event_floor_us() if kind in ("units", "block") else None,
            # context, not a peak: the same box's single-stream fill rate right after the run
            "box_fill_gbs": kern.box_write_gbs(),
            "algorithmic_bytes_per_launch": k1["bytes_per_launch"],
            "bytes_rule": {
                "units": "(2*U+2)*s*numel per launch: read W once, write theta+/theta- for each of "
                         "the layer's U units, write the final drifted W",
                "block": "sum over the launch's layers of (2*U+2)*s*numel: read W once, write "
                         "theta+/theta- for each of the layer's U units, write the drifted W",
                "triple": "4*s*numel (read W; write theta+, theta-, restored)",
                "single": "2*s*numel per pass (read W, write W)"}[kind],
        }
    if emulated:
        # NOT a measurement of N GPUs: what rank r of W would spend per step on its own GPU
        # (its 1/W of the forwards + K1 over ALL W*U units of every matrix), times W for `value`
        # under the assumption that the ranks run in perfect parallel and the closing all-reduce
        # of a 2.3-KB (strong: 75-KB) table is free
        out["projection"] = True
        out["projected"] = {
            "world": s_world, "rank": s_rank, "n_gpus_measured": 1,
            "what": "one rank's share of a dp%d pass timed on ONE GPU without a process group; "
                    "`value` = %s x steps / that time" % (s_world, "1" if strong else str(s_world)),
            "not_measured": "RCCL / xGMI transport, the other ranks, host contention of 8 processes",
            "k1_units_chained_per_matrix": nb_local * s_world,
            "k1_units_owned_per_matrix": nb_local,
        }
    if z_modes is not None:
        out["z_modes"] = z_modes
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        out["cpu_baseline"] = cpu_baseline(model, prunable, batches_local, args)
    if rank == 0:
        print(json.dumps(out))
    if world > 1:
        dist.destroy_process_group()


def _pinned_gemm_report():
    try:
        from ecoflap_amd.shapes import fused
        return fused.gemm_report()
    except Exception as e:          # never let the report break the line
        return {"error": repr(e)}


def _blocked(loss_fn):
    return float(getattr(loss_fn, "stats", {}).get("host_blocked_seconds", 0.0))


def _compact_stats(stats):
    out = dict(stats)
    names = out.pop("stages_not_batch_invariant", None)
    if names is not None:      # e.g. the 39 ViT-g blocks + the Q-Former bridge
        out["stages_not_batch_invariant"] = len(names)
        out["stages_not_batch_invariant_examples"] = sorted(names)[:2] + sorted(names)[-1:]
    return out


def load_pmc_traffic(kind, algorithmic_bytes_per_launch=None):
    """HBM bytes per launch from the committed rocprofv3 --pmc passes (profiles/
    k1_pmc_traffic.json: FETCH_SIZE and WRITE_SIZE collected in separate passes over
    tools/k1_launches.py, FETCH_SIZE doubled as MI355X_MICROARCH.md prescribes for gfx950).
    The counters cannot be read live; the measured traffic/algorithmic ratio of the same
    kernel is applied to this run's average launch."""
    path = os.path.join(ROOT, "profiles", "k1_pmc_traffic.json")
    if not os.path.exists(path) or algorithmic_bytes_per_launch is None:
        return None
    try:
        doc = json.load(open(path))
        ratio = (doc.get(kind) or ({} if kind.startswith("torch_") else doc.get("units", {}))).get(
            "traffic_over_algorithmic")
        return None if ratio is None else ratio * algorithmic_bytes_per_launch
    except Exception:
        return None


def _cpu_model_name():
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return None


def _usable_cpus():
    """CPUs this process may actually use: affinity mask capped by the cgroup CPU quota (the
    GPU boxes expose 256 logical CPUs but grant 16: 128 torch threads then run 5x SLOWER than
    16 — measured, fp32 BLIP-2 forward: 28.8 s vs 5.4 s)."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except (OSError, ValueError):
        pass
    return n


def cpu_baseline(model, prunable, batches_local, args):
    """The reference's CPU path restated, timed on the GPU box's host cores on a bounded sample
    (kind "port": the reference's Python cannot travel to this box).  One (matrix, batch) unit of
    its loop = 3 K1 calls + 2 full forwards (layer_single_base_pruner.py:530-539):
      * forward: the same torch modules in fp32 on the host threads; one untimed warm-up forward
        (first-touch, oneDNN primitive creation), then one timed;
      * K1 on three matrices spanning ViT / T5 encoder / T5 decoder, two ways: with torch CPU
        ops exactly as the reference writes it (`param.data + scaling_factor * z * zo_eps` after
        torch.manual_seed + torch.normal, :482-486; multi-threaded) and with the scalar-C oracle
        (1 thread, the checker the parity tests use).
    value = 1 / (batches per layer * (3-pass K1 by torch ops, mean of the three matrices +
    2 warm forwards)).  Checker code is used here only as the measured CPU baseline."""
    import copy
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from oracle_backend import OracleKernels, torch_cpu_normal
    from ecoflap_amd.pruners.losses import loss_vision_language

    threads_before = torch.get_num_threads()
    torch.set_num_threads(_usable_cpus())
    t0 = time.time()
    cpu_model = copy.deepcopy(model).to("cpu").float().eval()   # host forward in fp32
    cpu_batch = {k: v.cpu() for k, v in batches_local[0].items()}
    copy_s = time.time() - t0
    params = dict(cpu_model.named_parameters())
    if args.cpu_baseline_layer >= 0:
        picks = [prunable[args.cpu_baseline_layer]]
    else:
        vit = [k for k in prunable if k.startswith("visual_encoder")]
        enc = [k for k in prunable if ".encoder." in k]
        dec = [k for k in prunable if ".decoder." in k]
        picks = [g[len(g) // 2] for g in (vit, enc, dec) if g]
    kernels = OracleKernels()
    seed = 123456789
    k1_torch, k1_c, sampled = [], [], []
    for name in picks:
        p = params[name]
        # as the matrices are stored in the run (fp16 ViT / bf16 T5): K1 rounds to that dtype
        store = dict(model.named_parameters())[name].dtype
        w = p.data.to(store)
        t0 = time.perf_counter()
        for sf in (1, -2, 1):                                   # the reference's own expression
            torch.manual_seed(seed)
            z = torch.normal(mean=0, std=1, size=w.size(), device="cpu", dtype=w.dtype)
            w = w + sf * z * 1e-3
        k1_torch.append(time.perf_counter() - t0)
        w = p.data.to(store).contiguous()
        t0 = time.perf_counter()
        for sf in (1, -2, 1):
            kernels.zo_perturb(w, sf, 1e-3, seed, torch_cpu_normal(seed, w))
        k1_c.append(time.perf_counter() - t0)
        sampled.append({"name": name, "shape": list(p.shape), "dtype": str(store),
                        "k1_seconds_torch": round(k1_torch[-1], 4),
                        "k1_seconds_oracle_c": round(k1_c[-1], 4)})
    fwd = []
    with torch.no_grad():
        for _ in range(2):                                      # [0] cold (warm-up), [1] timed
            t0 = time.perf_counter()
            loss, _ = loss_vision_language(cpu_model, cpu_batch, False)
            float(loss)
            fwd.append(time.perf_counter() - t0)
    k1_mean = sum(k1_torch) / len(k1_torch)
    unit_s = k1_mean + 2 * fwd[1]
    nb = len(batches_local)
    used = torch.get_num_threads()
    torch.set_num_threads(threads_before)
    return {
        "value": 1.0 / (unit_s * nb),
        "unit": "layers/s",
        "cores": used,
        "kind": "port",
        "sample": (f"{len(picks)} of {len(prunable)} matrices (ViT / T5 encoder / T5 decoder) x 1 of "
                   f"{nb} batches: K1 as three torch-CPU passes = {k1_mean:.3f} s (mean), one warm fp32 "
                   f"forward on {used} host threads (= the cgroup's CPU quota) = {fwd[1]:.2f} s (cold: "
                   f"{fwd[0]:.2f} s); unit = K1 + 2 forwards = {unit_s:.2f} s, x{nb} batches per layer"),
        "unit_seconds": unit_s,
        "k1_seconds_torch": k1_mean,
        "k1_seconds_oracle_c": sum(k1_c) / len(k1_c),
        "forward_seconds_warm": fwd[1],
        "forward_seconds_cold": fwd[0],
        "matrices_sampled": sampled,
        "host_copy_seconds": copy_s,
        "nproc": os.cpu_count(),
        "usable_cpus": _usable_cpus(),
        "cpu_model": _cpu_model_name(),
        "torch_parallel_info": " | ".join(
            ln.strip() for ln in torch.__config__.parallel_info().splitlines()
            if "threads" in ln.lower() or "OMP_NUM" in ln or "MKL_NUM" in ln)[:400],
    }


if __name__ == "__main__":
    main()
