"""Row A1 / N4 (CPU): the EVA ViT-g restatement (``mraudio_amd/models/eva_vit.py``) pinned to the in-image structural
stand-in for LAVIS' ``create_eva_vit_g`` -- ``transformers.InstructBlipVisionModel`` -- on seeded weights:
against the committed vectors (``tests/golden/vit_g.npz``, made by ``tools/make_vit_golden.py``) and live.
LAVIS itself is absent: parity with the reference's own encoder build is unpinned."""
import json
import os

import numpy as np
import pytest
import torch

from mraudio_amd.models.eva_vit import EvaViTg
from tools.make_vit_golden import DEPTH, FRAMES, FULL_DEPTH, FULL_INPUT_SEED, FULL_WEIGHT_SEED, ROWS, WEIGHT_SEED, hf_reference, make_frames


@pytest.fixture(scope="module")
def vit():
    torch.set_num_threads(8)
    return EvaViTg(depth=DEPTH).eval().init_seeded_(WEIGHT_SEED)


def test_vit_matches_the_committed_hf_vectors(vit, golden_dir):
    gold = np.load(os.path.join(golden_dir, "vit_g.npz"))
    meta = json.loads(str(gold["meta"]))
    assert meta["depth"] == DEPTH and meta["rows"] == ROWS
    with torch.no_grad():
        y = vit(make_frames())
    assert y.shape == (2, 257, 1408)
    assert np.abs(y[:, ROWS].numpy() - gold["rows"]).max() < 1e-4           # fp32 on both sides, |y| <= 7
    assert np.abs(y.sum(-1).numpy() - gold["token_sum"]).max() < 5e-3       # every token, through its checksums
    assert np.abs(y.abs().sum(-1).numpy() - gold["token_abs_sum"]).max() < 5e-3


def test_vit_matches_hf_live_and_the_name_map_round_trips(vit):
    pytest.importorskip("transformers")
    frames = make_frames(1, seed=9)
    ref = hf_reference(vit, frames)
    with torch.no_grad():
        assert (vit(frames) - ref).abs().max().item() < 1e-5
    other = EvaViTg(depth=DEPTH).eval()
    other.load_hf_state_dict(vit.hf_state_dict())
    for (k, a), (_, b) in zip(vit.state_dict().items(), other.state_dict().items()):
        assert torch.equal(a, b), k
    sd = vit.hf_state_dict()
    sd["encoder.layers.0.self_attn.qkv.bias"] = sd["encoder.layers.0.self_attn.qkv.bias"] + 1.0
    with pytest.raises(ValueError):
        other.load_hf_state_dict(sd)                                         # a key bias cannot be represented


def test_full_depth_restatement_matches_the_committed_39_layer_hf_vectors(golden_dir):
    """The 39-block encoder bench.py times: the restatement against ``tests/golden/vit_g_d39.npz`` (HF ``InstructBlipVisionModel``,
    39 layers, made by ``tools/make_vit_golden.py full``).  ~30 s of seeded weight drawing + ~1 TFLOP of fp32 CPU work."""
    gold = np.load(os.path.join(golden_dir, "vit_g_d39.npz"))
    meta = json.loads(str(gold["meta"]))
    assert meta["depth"] == FULL_DEPTH and meta["weight_seed"] == FULL_WEIGHT_SEED and meta["input_seed"] == FULL_INPUT_SEED
    torch.set_num_threads(8)
    vit = EvaViTg(depth=FULL_DEPTH).eval().init_seeded_(FULL_WEIGHT_SEED)
    with torch.no_grad():
        y = vit(make_frames(FRAMES, FULL_INPUT_SEED))
    assert np.abs(y[:, ROWS].numpy() - gold["rows"]).max() < 5e-4            # fp32 on both sides, |y| <= 26, 39 blocks
    assert np.abs(y.sum(-1).numpy() - gold["token_sum"]).max() < 5e-2
    assert (np.abs(y.pow(2).sum(-1).numpy() - gold["token_sq_sum"]) / gold["token_sq_sum"]).max() < 1e-4
