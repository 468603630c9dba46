"""ECoFLaP (zeroth-order table) + SparseGPT on BLIP-2 (reference: LAVIS/scripts/blip2/ecoflap_sparsegpt_zeroth.py:10-28 - there the table comes from a previous ecoflap_zeroth run through --sparsity_dict; here stage 1 runs in the same job unless a table is passed as the third argument).
Parameters of the job: LAVIS/scripts/_launch.py::JOBS["blip2/ecoflap_sparsegpt_zeroth"]."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from _launch import run  # noqa: E402

sys.exit(run("blip2/ecoflap_sparsegpt_zeroth"))
