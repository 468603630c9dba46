"""Uniform-sparsity Wanda on the EVA-CLIP vision tower (reference: LAVIS/scripts/eva_clip/wanda.py:9-22; the build's ViT shape has 12 blocks).
Parameters of the job: LAVIS/scripts/_launch.py::JOBS["eva_clip/wanda"]."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from _launch import run  # noqa: E402

sys.exit(run("eva_clip/wanda"))
