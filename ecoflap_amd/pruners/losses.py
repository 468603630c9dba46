"""Loss closures handed to stage 1, `(model, batch, cuda_enabled) -> (0-dim loss, batch_len)` —
the three the LAVIS pruners use (LAVIS/lavis/compression/pruners/utils.py:21-67: the language
and vision-language ones read the model's own "loss"; the vision one rebuilds a cross entropy
from `predict()`'s x100-scaled zero-shot logits)."""
import torch
import torch.nn.functional as F


def prepare_sample(samples, cuda_enabled=True, device=None):
    """Move a batch's tensors to the model's device (lavis.datasets.data_utils.prepare_sample)."""
    if not cuda_enabled or device is None:
        return samples
    return {k: (v.to(device, non_blocking=True) if torch.is_tensor(v) else v)
            for k, v in samples.items()}


def _on_model_device(model, samples, cuda_enabled):
    return prepare_sample(samples, cuda_enabled, next(iter(model.parameters())).device)


def _model_loss(count_key):
    """Closure for models whose forward returns {"loss": ...}; batch_len = len(batch[count_key])."""
    def closure(model, samples, cuda_enabled):
        batch = _on_model_device(model, samples, cuda_enabled)
        return model(batch)["loss"], len(batch[count_key])
    return closure


loss_vision_language = _model_loss("text_input")
loss_vision_language.__name__ = "loss_vision_language"
loss_language = _model_loss("text_input")
loss_language.__name__ = "loss_language"


def loss_vision(model, samples, cuda_enabled):
    """-mean(log softmax(predictions / 100)[target]) over the batch (utils.py:47-67: probabilities
    first, then the log of the picked ones — kept in that order for the same roundings)."""
    out = model.predict(_on_model_device(model, samples, cuda_enabled))
    targets = out["targets"]
    picked = F.softmax(out["predictions"] / 100, -1).gather(1, targets.view(-1, 1)).squeeze(1)
    return -(picked.log().mean()), len(targets)
