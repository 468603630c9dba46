"""ECoFLaP on BLIP captioning — the build's counterpart of the reference's
UPop/ecoflap_compress_caption.py (pruner construction :231-246, task "coco").

    python UPop/ecoflap_compress_caption.py --p 0.5 [--stage1 intended] [--finetune_steps 2] [--toy]
"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from _entry import run  # noqa: E402


def main(argv=None):
    return run("coco", argv)


if __name__ == "__main__":
    main()
