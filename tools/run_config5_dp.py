#!/usr/bin/env python3
"""BASELINE configs[4] (BLIP VQA: ECoFLaP intended-mode stage 1 on the task loss + Wanda local
prune + one masked fine-tune step) in its data-parallel form, FUNCTIONALLY on one GPU: N ranks
time-sharing cuda:0 over gloo against the same run in one process — equal sparsity table and equal
pruned weights (sha256).  Like tools/run_config4.py this exercises everything but the RCCL / xGMI
transport and measures no scaling.

    python3 tools/run_config5_dp.py single
    python3 tools/run_config5_dp.py dp8
"""
import os
os.environ.setdefault("TENSILE_STREAMK_DATA_PARALLEL", "1")   # before the first GEMM (ecoflap_amd/blas_guard.py)
import datetime
import hashlib
import json
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "UPop"))

ARGS = ["--stage1", "intended", "--num_data", os.environ.get("ECOFLAP_CONFIG5_SAMPLES", "128")]


def one_rank(rank, world, port, out_path):
    import torch
    import torch.distributed as dist
    torch.cuda.set_device(0)
    if world > 1:
        os.environ["MASTER_ADDR"] = "127.0.0.1"
        os.environ["MASTER_PORT"] = str(port)
        torch.cuda.set_per_process_memory_fraction(0.95 / world)
        dist.init_process_group("gloo", rank=rank, world_size=world,
                                timeout=datetime.timedelta(minutes=15))
    import _entry
    t0 = time.time()
    model, table = _entry.run("vqa", ARGS)
    torch.cuda.synchronize()
    wall = time.time() - t0
    h = hashlib.sha256()
    blocks = {k: v for k, v in sorted(model.state_dict().items()) if v.dim() == 2}
    for k, v in blocks.items():
        h.update(v.detach().cpu().contiguous().view(torch.uint8).numpy().tobytes())
    prunable = {k: v for k, v in blocks.items() if ".blocks." in k or ".layer." in k}
    res = {"config": "5 (BLIP VQA, intended-mode stage 1 + Wanda)", "world_size": world, "rank": rank,
           "samples": int(ARGS[-1]), "wall_seconds": wall,
           "table_entries": len(table) if isinstance(table, dict) else 0,
           "distinct_sparsities": len(set(table.values())) if isinstance(table, dict) else 0,
           "table_sha256": hashlib.sha256(repr(sorted(table.items())).encode()).hexdigest()
           if isinstance(table, dict) else None,
           "pruned_weights_sha256": h.hexdigest(),
           "pruned_fraction": sum(int((v == 0).sum()) for v in prunable.values())
           / max(1, sum(v.numel() for v in prunable.values())),
           "peak_mem_gb": torch.cuda.max_memory_allocated() / 1e9,
           "transport": ("none (one process)" if world == 1 else
                         f"gloo, {world} ranks sharing cuda:0 (functional run: no RCCL, no scaling)")}
    if world > 1:
        t = torch.tensor([int(res["table_sha256"][:15], 16), int(res["pruned_weights_sha256"][:15], 16)],
                         dtype=torch.int64)
        allt = [torch.zeros_like(t) for _ in range(world)]
        dist.all_gather(allt, t)
        res["replicas_agree"] = bool(all(torch.equal(x, allt[0]) for x in allt))
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0:
        with open(out_path, "w") as f:
            json.dump(res, f, default=str)


def main():
    mode = sys.argv[1]
    out_path = os.path.join(ROOT, "gpurun_out", f"config5_{mode}.json")
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
        mp.spawn(one_rank, args=(world, port, out_path), nprocs=world, join=True)
    print(open(out_path).read())


if __name__ == "__main__":
    main()
