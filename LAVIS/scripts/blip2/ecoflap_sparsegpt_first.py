"""ECoFLaP (first-order table) + SparseGPT on BLIP-2 (reference: LAVIS/scripts/blip2/ecoflap_sparsegpt_first.py:10-24; table as the third argument, else stage 1 runs in the same job).
Parameters of the job: LAVIS/scripts/_launch.py::JOBS["blip2/ecoflap_sparsegpt_first"]."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from _launch import run  # noqa: E402

sys.exit(run("blip2/ecoflap_sparsegpt_first"))
