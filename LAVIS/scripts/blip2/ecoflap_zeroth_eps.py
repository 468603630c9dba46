"""Perturbation-size sweep of the zeroth-order stage on BLIP-2 (reference: LAVIS/scripts/blip2/ecoflap_zeroth_eps.py:9-32; its `olmezo-gradient_sum` is the pre-release name of `MEZO-GradOnly_sum`, which is what the shipped LayerSparsity understands).
Parameters of the job: LAVIS/scripts/_launch.py::JOBS["blip2/ecoflap_zeroth_eps"]."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from _launch import run  # noqa: E402

sys.exit(run("blip2/ecoflap_zeroth_eps"))
