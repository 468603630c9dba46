"""The reference's own model classes through the build's boundary: runs
tests/golden/reference_model_through_boundary.py where the reference tree is present — the build
container; `/root/reference` does not exist on the GPU box, and nothing marked `gpu` reads it."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = "/root/reference"


@pytest.mark.skipif(not os.path.isdir(os.path.join(REF, "LAVIS")), reason="the reference tree is not on this box")
def test_reference_owned_eva_vit_and_t5_through_the_boundary_equal_the_reference_pruners():
    """The reference's eva_vit.VisionTransformer (three forward forms) and vendored modeling_t5
    T5ForConditionalGeneration: reference pruner == build pruner through HookedPrefixLoss at
    eval_batch 1 and 4 — losses, sparsity table, pruned state_dict, bit for bit."""
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "golden", "reference_model_through_boundary.py")],
                       capture_output=True, text=True, timeout=900)
    tail = "\n".join((r.stdout + r.stderr).splitlines()[-15:])
    assert r.returncode == 0, tail
    assert "RESULT: all equal" in r.stdout, tail
    assert r.stdout.count("EQUAL (losses, table, state_dict: bit for bit)") == 8, tail
