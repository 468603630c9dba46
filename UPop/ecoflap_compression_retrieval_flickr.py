"""ECoFLaP on BLIP image-text retrieval — the build's counterpart of the reference's
UPop/ecoflap_compression_retrieval_flickr.py (pruner construction :357-372, task "retrieval",
loss = `forward_itm` with in-batch hard negatives).

    python UPop/ecoflap_compression_retrieval_flickr.py --p 0.5 [--stage1 intended] [--toy]
"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from _entry import run  # noqa: E402


def main(argv=None):
    return run("retrieval", argv)


if __name__ == "__main__":
    main()
