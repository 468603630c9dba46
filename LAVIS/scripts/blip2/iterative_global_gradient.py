"""Iterative global |grad|*|W| baseline on BLIP-2, three rounds, one threshold per sub-model (reference: LAVIS/scripts/blip2/iterative_global_gradient.py:9-28).
Parameters of the job: LAVIS/scripts/_launch.py::JOBS["blip2/iterative_global_gradient"]."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from _launch import run  # noqa: E402

sys.exit(run("blip2/iterative_global_gradient"))
