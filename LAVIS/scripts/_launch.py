"""Shared launcher body: same role as the reference's `subprocess.call("python -m
torch.distributed.run --nproc_per_node=1 ... evaluate_*.py ...")` lines, pointed at the
build's harness (ecoflap_amd/harness.py).  Usage of every script: `python <script> GPU PORT`
(the reference's two positional arguments); extra arguments are passed through."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def launch(shape, flags):
    gpu = sys.argv[1] if len(sys.argv) > 1 else "0"
    port = sys.argv[2] if len(sys.argv) > 2 else "12341"
    extra = " ".join(sys.argv[3:])
    program = (f"HIP_VISIBLE_DEVICES={gpu} {sys.executable} -m torch.distributed.run"
               f" --nproc_per_node=1 --master-addr 127.0.0.1 --master_port {port}"
               f" -m ecoflap_amd.harness --shape {shape} {flags} {extra}")
    print(program)
    return subprocess.call(program, shell=True, cwd=ROOT)
