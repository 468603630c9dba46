#!/usr/bin/env python3
"""BASELINE configs[3] FUNCTIONALLY on the hardware a 1-GPU box has: BLIP-2 zeroth-order, 1024
calibration pairs, sharded DP=8 — as 8 ranks TIME-SHARING one MI355X (8 x ~20 GB of 288 GB), the
exchange over gloo instead of RCCL (RCCL refuses two ranks on one device) — and the same 1024
pairs in ONE process.  Both print the sha256 of the sparsity table and of the pruned weights:
equal hashes = the data-parallel path (batch sharding, drift-only K1 chaining of the other
ranks' units, the loss-table all-reduce, the per-block statistics exchange + replay of stage 2)
reproduces the one-process run bit for bit at full size.  What this does NOT exercise or
measure: the RCCL / xGMI transport and any scaling (the ranks share one GPU).

    python3 tools/run_config4.py single > profiles/r04_config4_single.json
    python3 tools/run_config4.py dp8    > profiles/r04_config4_dp8_one_gpu.json
"""
import os
os.environ.setdefault("TENSILE_STREAMK_DATA_PARALLEL", "1")   # before the first GEMM (ecoflap_amd/blas_guard.py)
import datetime
import json
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))

PAIRS = int(os.environ.get("ECOFLAP_CONFIG4_PAIRS", "1024"))
EXTRA = ["--num_data", str(PAIRS), "--num_data_first_stage", str(PAIRS), "--lanes", "1"]


def one_rank(rank, world, port, out_path):
    import torch
    import torch.distributed as dist
    torch.cuda.set_device(0)
    if world > 1:
        os.environ["MASTER_ADDR"] = "127.0.0.1"
        os.environ["MASTER_PORT"] = str(port)
        # an out-of-memory rank must fail by itself, not take the box down with it
        torch.cuda.set_per_process_memory_fraction(0.95 / world)
        dist.init_process_group("gloo", rank=rank, world_size=world,
                                timeout=datetime.timedelta(minutes=20))
    import run_config
    t0 = time.time()
    res = run_config.run("3", EXTRA)
    res["wall_seconds_rank"] = time.time() - t0
    res["world_size"] = world
    res["rank"] = rank
    res["pairs_total"] = PAIRS
    res["transport"] = ("none (one process)" if world == 1 else
                        f"gloo, {world} ranks sharing cuda:0 (functional run: no RCCL, no scaling)")
    if world > 1:
        # every replica must hold the same table and the same pruned weights
        h = torch.tensor([int(res["table_sha256"][:15], 16), int(res["pruned_weights_sha256"][:15], 16)],
                         dtype=torch.int64)
        all_h = [torch.zeros_like(h) for _ in range(world)]
        dist.all_gather(all_h, h)
        res["replicas_agree"] = bool(all(torch.equal(x, all_h[0]) for x in all_h))
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0:
        with open(out_path, "w") as f:
            json.dump(res, f, default=str)


def main():
    mode = sys.argv[1]
    out_path = os.path.join(ROOT, "gpurun_out", f"config4_{mode}.json")
    os.makedirs(os.path.dirname(out_path), exist_ok=True)
    if mode == "single":
        one_rank(0, 1, 0, out_path)
    else:
        world = int(mode[2:])
        import socket
        import torch.multiprocessing as mp
        with socket.socket() as s:
            s.bind(("127.0.0.1", 0))
            port = s.getsockname()[1]
        # (the parent makes no GPU call: the ranks are spawned)
        mp.spawn(one_rank, args=(world, port, out_path), nprocs=world, join=True)
    print(open(out_path).read())


if __name__ == "__main__":
    main()
