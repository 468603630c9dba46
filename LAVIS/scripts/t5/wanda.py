"""Uniform-sparsity Wanda on FlanT5-XL (reference: LAVIS/scripts/t5/wanda.py:9-22).
Parameters of the job: LAVIS/scripts/_launch.py::JOBS["t5/wanda"]."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from _launch import run  # noqa: E402

sys.exit(run("t5/wanda"))
