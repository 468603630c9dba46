"""Host-side helpers of the zeroth-order loop and the stage-2 block loop that only run on the
GPU path in production (graph families, slot layout of batched evaluations, calibration
uniformity): their pure-tensor logic, on CPU."""
import torch

from ecoflap_amd.pruners import prefix_cache as PC
from ecoflap_amd.pruners.wanda import _uniform_calibration


def test_family_ignores_values_but_not_shapes():
    a = {"image": torch.zeros(8, 3, 4, 4), "text_input": torch.zeros(8, 5, dtype=torch.long)}
    b = {"image": torch.ones(8, 3, 4, 4), "text_input": torch.ones(8, 5, dtype=torch.long)}
    c = {"image": torch.ones(8, 3, 4, 4), "text_input": torch.ones(8, 6, dtype=torch.long)}
    assert PC._family(a) == PC._family(b) != PC._family(c)
    # VQA tuples: per-question answer counts matter through their sum only
    v1 = (torch.zeros(2, 3), torch.zeros(2, 4), torch.zeros(3, 5), torch.zeros(3), [1, 2])
    v2 = (torch.zeros(2, 3), torch.zeros(2, 4), torch.zeros(3, 5), torch.zeros(3), [2, 1])
    v3 = (torch.zeros(2, 3), torch.zeros(2, 4), torch.zeros(4, 5), torch.zeros(4), [2, 2])
    assert PC._family(v1) == PC._family(v2) != PC._family(v3)


def test_slot_layout_round_trip():
    B, k = 2, 4
    states = [{"x": torch.full((B, 3), float(i)), "mask": torch.full((B, 1, 1, 3), float(-i)),
               "bias": torch.arange(5.0).view(1, 5), "n": 7, "pair": [torch.full((B,), float(i))]}
              for i in range(k)]
    cat = PC._cat_states(states, B)
    assert cat["x"].shape == (k * B, 3) and cat["bias"].shape == (1, 5) and cat["n"] == 7
    for i in range(k):
        sl = PC._slice_state(cat, i, B, k)
        assert torch.equal(sl["x"], states[i]["x"]) and torch.equal(sl["mask"], states[i]["mask"])
        assert torch.equal(sl["pair"][0], states[i]["pair"][0])
        assert sl["bias"] is cat["bias"]
    # writing one evaluation into its slot leaves the others alone; shared tensors follow slot 0
    new = {"x": torch.full((B, 3), 9.0), "mask": torch.full((B, 1, 1, 3), 9.0),
           "bias": torch.zeros(1, 5), "n": 7, "pair": [torch.full((B,), 9.0)]}
    PC._copy_slot(cat, new, 2, B)
    assert torch.equal(PC._slice_state(cat, 2, B, k)["x"], new["x"])
    assert torch.equal(PC._slice_state(cat, 1, B, k)["x"], states[1]["x"])
    assert torch.equal(cat["bias"], torch.arange(5.0).view(1, 5))
    PC._copy_slot(cat, new, 0, B)
    assert torch.equal(cat["bias"], torch.zeros(1, 5))


def test_uniform_calibration_detection():
    inps = [torch.zeros(1, 4, 8) for _ in range(3)]
    caches = [{"attention_mask": torch.zeros(1, 1, 1, 4), "position_bias": None, "flag": False}
              for _ in range(3)]
    cpu_tensor_kwargs = _uniform_calibration(inps, caches, 3)
    assert cpu_tensor_kwargs is False            # kwargs tensors must live on the GPU to be replayed
    caches2 = [{"position_bias": None, "flag": False} for _ in range(3)]
    assert _uniform_calibration(inps, caches2, 3) is True
    ragged = [torch.zeros(1, 4, 8), torch.zeros(1, 5, 8), torch.zeros(1, 4, 8)]
    assert _uniform_calibration(ragged, caches2, 3) is False
    lists = [{"encoder_hidden_states": [torch.zeros(1, 2, 8)] * 2} for _ in range(3)]
    assert _uniform_calibration(inps, lists, 3) is False     # NLVR's twin states: eager path
