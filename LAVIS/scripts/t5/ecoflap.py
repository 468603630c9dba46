"""ECoFLaP zeroth-order + Wanda on FlanT5 (reference: LAVIS/scripts/t5/ecoflap.py:10-31).
Parameters of the job: LAVIS/scripts/_launch.py::JOBS["t5/ecoflap"]."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from _launch import run  # noqa: E402

sys.exit(run("t5/ecoflap"))
