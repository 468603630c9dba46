"""ecoflap_amd — MI355X-native ECoFLaP scoring + pruning hot path.

Drop-in for the reference's pruner API for this path and nothing else
(LAVIS/lavis/compression/__init__.py:29-46 `load_pruner`,
LAVIS/lavis/common/registry.py:113-137): same pruner names, same config keys,
`prune() -> (model, sparsity_dict)`.  Compute runs in hand-written HIP kernels
for gfx950 behind a C ABI (include/ecoflap_hip.h, ecoflap_amd/libecoflap_hip.so);
there is no CPU fallback.
"""
import sys

from . import blas_guard

blas_guard.configure()      # before this process's first GEMM (see blas_guard.py)

from .registry import registry  # noqa: F401
from .pruners import (  # noqa: F401  (importing registers the pruners)
    BasePruner, BLIPT5LayerWandaPruner, LayerSparsity, T5LayerWandaPruner, VITLayerWandaPruner,
    WrappedGPT,
)

__all__ = ["BasePruner", "LayerSparsity", "load_pruner", "registry"]


def load_pruner(name, model, data_loader, cfg_path=None, cfg=None):
    """registry lookup + construction with the config dict as keywords; an unknown
    name or keyword prints the available pruners and exits with status 1, as the
    reference does (compression/__init__.py:37-44)."""
    if cfg_path is not None:
        import yaml
        with open(cfg_path, "r") as f:
            cfg = yaml.safe_load(f)
    if cfg is None:
        cfg = {}
    try:
        pruner = registry.get_pruner_class(name)(model=model, data_loader=data_loader, **cfg)
    except TypeError as e:
        print(f"Pruner {name} not found ({e}). Available pruners:\n"
              + ", ".join(registry.list_pruners()))
        sys.exit(1)
    return pruner
