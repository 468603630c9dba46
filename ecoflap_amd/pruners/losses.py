"""Loss closures handed to stage 1: `(model, batch, cuda_enabled) -> (loss, batch_len)`
(LAVIS/lavis/compression/pruners/utils.py:21-67)."""
import torch


def prepare_sample(samples, cuda_enabled=True, device=None):
    """Move a batch's tensors to the model's device (lavis.datasets.data_utils.prepare_sample)."""
    if not cuda_enabled or device is None:
        return samples
    return {k: (v.to(device, non_blocking=True) if torch.is_tensor(v) else v)
            for k, v in samples.items()}


def _device_of(model):
    return next(iter(model.parameters())).device


def loss_vision_language(model, samples, cuda_enabled):
    samples = prepare_sample(samples, cuda_enabled, _device_of(model))
    loss = model(samples)["loss"]
    return loss, len(samples["text_input"])


def loss_language(model, samples, cuda_enabled):
    samples = prepare_sample(samples, cuda_enabled, _device_of(model))
    loss = model(samples)["loss"]
    return loss, len(samples["text_input"])


def loss_vision(model, samples, cuda_enabled):
    """Cross entropy of the zero-shot logits, undoing predict()'s x100 (utils.py:47-67)."""
    samples = prepare_sample(samples, cuda_enabled, _device_of(model))
    outputs = model.predict(samples)
    logits = outputs["predictions"] / 100
    targets = outputs["targets"]
    probs = torch.nn.functional.softmax(logits, -1)
    batch_index = torch.arange(len(targets)).to(targets.device)
    log_probs = probs[batch_index, targets].log()
    return -log_probs.mean(), len(targets)
