from .base_pruner import BasePruner, LayerWiseBasePruner  # noqa: F401
from .layer_sparsity import LayerSparsity  # noqa: F401
from .wanda import (  # noqa: F401
    BLIPT5LayerWandaPruner, T5LayerWandaPruner, VITLayerWandaPruner, WrappedGPT, find_layers,
    get_module_recursive,
)
from .losses import loss_language, loss_vision, loss_vision_language  # noqa: F401
from .upop import (  # noqa: F401,E402
    BLIPBertLayerWandaPruner, apply_masks_to_grads, pruning_masks, task_forward,
)
from .sparsegpt import (  # noqa: F401,E402
    BLIPT5LayerSparseGPTPruner, SparseGPT, T5LayerSparseGPTPruner, VITLayerSparseGPTPruner,
)
from .global_pruner import (  # noqa: F401,E402
    BLIPT5GlobalGradMagAbsPruner, BLIPT5GlobalMagPruner, BLIPT5GlobalMeZoPruner, BLIPT5GlobalPruner,
)
