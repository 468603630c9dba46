"""Synthetic calibration batches: a FIXED, re-iterable list of
batches reused for every layer (the reference re-iterates its DataLoader per
layer, layer_single_base_pruner.py:517; `DataLoaderWrapper` yields the first
num_data//batch_size batches, LAVIS/lavis/runners/runner_base.py:672-690).
Text is pre-tokenised ids (pad id 0 never drawn)."""
import torch


def image_text_batches(num_data, batch_size, img_size=224, vocab=32128, in_len=16, out_len=16,
                       seed=42, device="cpu", image_dtype=torch.float32):
    g = torch.Generator().manual_seed(seed)
    batches = []
    for _ in range(num_data // batch_size):
        batches.append({
            "image": torch.randn(batch_size, 3, img_size, img_size, generator=g)
                          .to(image_dtype).to(device),
            "text_input": torch.randint(2, vocab, (batch_size, in_len), generator=g).to(device),
            "text_output": torch.randint(2, vocab, (batch_size, out_len), generator=g).to(device),
        })
    return batches


def text_batches(num_data, batch_size, vocab=32128, in_len=32, out_len=16, seed=42, device="cpu"):
    g = torch.Generator().manual_seed(seed)
    return [{
        "text_input": torch.randint(2, vocab, (batch_size, in_len), generator=g).to(device),
        "text_output": torch.randint(2, vocab, (batch_size, out_len), generator=g).to(device),
    } for _ in range(num_data // batch_size)]


def image_label_batches(num_data, batch_size, img_size=224, num_classes=1000, seed=42,
                        device="cpu"):
    g = torch.Generator().manual_seed(seed)
    return [{
        "image": torch.randn(batch_size, 3, img_size, img_size, generator=g).to(device),
        "label": torch.randint(0, num_classes, (batch_size,), generator=g).to(device),
    } for _ in range(num_data // batch_size)]
