"""ECoFLaP on BLIP NLVR2 — the build's counterpart of the reference's
UPop/ecoflap_compress_nlvr.py (pruner construction :239-254, task "nlvr").  As shipped this
entrypoint stops at the sample-count assertion of the first ViT block
(UPop/pruners/wanda_pruner.py:496-497) unless asserts are disabled; `--stage1 intended` uses the
count the ViT actually sees.

    python [-O] UPop/ecoflap_compress_nlvr.py --p 0.5 [--stage1 intended] [--toy]
"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from _entry import run  # noqa: E402


def main(argv=None):
    return run("nlvr", argv)


if __name__ == "__main__":
    main()
