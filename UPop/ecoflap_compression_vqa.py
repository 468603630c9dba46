"""ECoFLaP on BLIP-VQA — the build's counterpart of the reference's
UPop/ecoflap_compression_vqa.py (pruner construction :256-271, masks :312-315, masked
fine-tune step :124-129).

    python UPop/ecoflap_compression_vqa.py --p 0.5 [--stage1 intended] [--finetune_steps 2] [--toy]
"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from _entry import run  # noqa: E402


def main(argv=None):
    return run("vqa", argv)


if __name__ == "__main__":
    main()
