"""ECoFLaP zeroth-order + Wanda on the EVA-CLIP vision tower (reference: LAVIS/scripts/eva_clip/ecoflap.py:9-31; the build's ViT shape has 12 blocks).
Parameters of the job: LAVIS/scripts/_launch.py::JOBS["eva_clip/ecoflap"]."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from _launch import run  # noqa: E402

sys.exit(run("eva_clip/ecoflap"))
