"""Loss closures handed to stage 1, `(model, batch, cuda_enabled) -> (0-dim loss, batch_len)` —
the three the LAVIS pruners use (LAVIS/lavis/compression/pruners/utils.py:21-67: the language
and vision-language ones read the model's own "loss"; the vision one rebuilds a cross entropy
from `predict()`'s x100-scaled zero-shot logits) and CoOp's CLIP contrastive closure
(CoOp/trainers/zsclip.py:73-91)."""
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


def clip_contrastive(prompt_tokens):
    """CoOp's zero-shot CLIP closure (CoOp/trainers/zsclip.py:73-91), handed to its pruner as
    `forward_to_cache(model, batch, device)`: every image of the batch is paired with the
    prompt of its OWN label ("a photo of a {class}."), the two-tower model returns the
    image->text and text->image logit matrices, and the loss is the symmetric InfoNCE — the mean
    of the two cross entropies against the diagonal, both taken in fp32.

    prompt_tokens: LongTensor [num_classes, context] — the tokenised prompt of every class.  The
    reference tokenises the formatted strings inside the closure (`clip.tokenize`, CoOp's
    vendored BPE); the build takes ids, as every synthetic shape here does.
    -> closure(model, batch{"img","label"}, device) -> (0-dim fp32 loss, batch_len)."""
    def forward_to_cache(model, batch, device):
        labels = batch["label"]
        text = prompt_tokens.to(labels.device)[labels].to(device)
        per_image, per_text = model(batch["img"].to(device), text)
        diagonal = torch.arange(per_image.shape[0], device=device, dtype=torch.long)
        both = F.cross_entropy(per_image.float(), diagonal) + F.cross_entropy(per_text.float(), diagonal)
        return both / 2, labels.shape[0]
    return forward_to_cache
