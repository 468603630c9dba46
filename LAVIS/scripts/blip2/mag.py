"""Global magnitude baseline on BLIP-2 (reference: LAVIS/scripts/blip2/mag.py:9-24).
Parameters of the job: LAVIS/scripts/_launch.py::JOBS["blip2/mag"]."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from _launch import run  # noqa: E402

sys.exit(run("blip2/mag"))
