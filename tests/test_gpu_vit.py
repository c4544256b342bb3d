"""``-m gpu``: row A1 / N4 -- the EVA ViT-g encoder on the HIP extension (``mra_vit_*``, ``csrc/vit.hip``) against the
committed HF-generated vectors (``tests/golden/vit_g.npz``: transformers ``InstructBlipVisionModel`` on the seeded weights)
and against the fp32 torch restatement on the CPU; then through ``XInstructBLIP`` from raw frames.

Tolerance: MFMA operands are f16 (the reference builds the encoder with ``precision="fp16"``), accumulation, residual
stream, LayerNorm and softmax statistics fp32.  Three blocks of width 1408 land at ~1e-3 of the activation scale:
|d| <= 2e-2 on |y| <= 7, relative Frobenius error < 2e-3."""
import json
import os

import numpy as np
import pytest
import torch

from mraudio_amd.models.eva_vit import EvaViTg, HipEvaViTg, create_eva_vit_g
from tools.make_vit_golden import DEPTH, ROWS, WEIGHT_SEED, make_frames

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available()
    return torch.device("cuda:0")


@pytest.fixture(scope="module")
def pair(dev):
    ref = EvaViTg(depth=DEPTH).eval().init_seeded_(WEIGHT_SEED)
    hip = HipEvaViTg(depth=DEPTH, device=dev).eval()
    hip.load_state_dict(ref.state_dict())
    return ref, hip


def test_hip_vit_matches_the_hf_vectors_and_the_fp32_restatement(pair, golden_dir, dev):
    ref, hip = pair
    gold = np.load(os.path.join(golden_dir, "vit_g.npz"))
    assert json.loads(str(gold["meta"]))["depth"] == DEPTH
    frames = make_frames()
    y = hip(frames.to(dev)).cpu()
    assert y.shape == (2, 257, 1408) and y.dtype == torch.float32 and torch.isfinite(y).all()
    assert np.abs(y[:, ROWS].numpy() - gold["rows"]).max() < 2e-2
    assert np.abs(y.sum(-1).numpy() - gold["token_sum"]).max() < 0.5          # 1408-term sums of f16-rounded values
    with torch.no_grad():
        want = ref(frames)
    assert (y - want).abs().max().item() < 2e-2
    rel = ((y - want).norm() / want.norm()).item()
    assert rel < 2e-3, rel
    # f16 frames are accepted as they are; a different batch composition gives the same rows (frames are independent)
    y16 = hip(frames.half().to(dev)).cpu()
    assert (y16 - y).abs().max().item() < 2e-2
    more = torch.cat([make_frames(3, seed=21), frames])
    ym = hip(more.to(dev)).cpu()
    assert (ym[3:] - y).abs().max().item() < 1e-4


def test_hip_vit_with_the_reference_precision_residual(pair, dev):
    """``residual="op"``: the residual stream in f16, every add rounding to 11 bits -- what LAVIS' ``precision="fp16"`` encoder does.
    Against the fp32 restatement the bar is the f16 resolution of values up to ~7 accumulated over 2 x DEPTH adds."""
    ref, _ = pair
    hip16 = HipEvaViTg(depth=DEPTH, device=dev, residual="op").eval()
    hip16.load_state_dict(ref.state_dict())
    frames = make_frames()
    y = hip16(frames.to(dev))
    assert y.dtype == torch.float16 and y.shape == (2, 257, 1408)
    with torch.no_grad():
        want = ref(frames)
    y = y.float().cpu()
    assert torch.isfinite(y).all() and (y - want).abs().max().item() < 4e-2
    assert ((y - want).norm() / want.norm()).item() < 4e-3


def test_hip_vit_ragged_batches_and_errors(pair, dev):
    from mraudio_amd import MraError

    ref, hip = pair
    one = make_frames(1, seed=4)
    with torch.no_grad():
        want = ref(one)
    assert (hip(one.to(dev)).cpu() - want).abs().max().item() < 2e-2
    assert hip(torch.zeros(0, 3, 224, 224, device=dev)).shape == (0, 257, 1408)
    fresh = HipEvaViTg(depth=1, device=dev)
    fresh._dirty, fresh._ver = False, sum(p._version for p in fresh.parameters())   # nothing uploaded: the library must refuse to run
    with pytest.raises(MraError):
        fresh(one.to(dev))
    assert isinstance(create_eva_vit_g(224, 0, False, "fp16", backend="hip", depth=1, device=dev), HipEvaViTg)


def test_raw_frames_through_the_model_use_the_batched_hip_encoder(dev):
    """samples["video"] [B, 3, T, 224, 224] -> one [B * T] batch through the HIP ViT -> video_ln -> Q-Former -> scores:
    equal to feeding the fp32 restatement's per-frame features as ``video_embeds`` (reference :262-306)."""
    from mraudio_amd.models.xinstructblip import XInstructBLIP

    ref = EvaViTg(depth=2).eval().init_seeded_(3)
    hip = HipEvaViTg(depth=2, device=dev).eval()
    hip.load_state_dict(ref.state_dict())
    model = XInstructBLIP(seed=0, perturb=True, device=dev, modalities=["video"], video_encoder=hip)
    B, T = 2, 3
    g = torch.Generator().manual_seed(8)
    video = torch.randn(B, 3, T, 224, 224, generator=g)
    prompts = ["Query: a dog runs.\nRelevant windows: ", "Query: the door closes.\nRelevant windows: "]
    base = {"text_input": prompts, "timestamps": [[0, 2, 4], [1, 3, 5]], "duration": [6, 6]}
    model.encode_chunk = 4                     # 6 frames -> chunks of 4 + 2
    out = model.encode_fuse({**base, "video": video})
    with torch.no_grad():
        feats = torch.stack([torch.stack([ref(video[b, :, t][None])[0] for t in range(T)]) for b in range(B)])   # the reference's per-frame calls
    want = model.encode_fuse({**base, "video_embeds": feats})
    assert (out["z"]["video"] - want["z"]["video"]).abs().max().item() < 2e-2
    scale = want["fused"].abs().max().item()
    assert (out["fused"] - want["fused"]).abs().max().item() <= 2e-3 * scale


def test_other_frame_sizes_take_the_generic_attention_kernel(dev):
    """224 x 224 frames (257 tokens) run the attention core specialised on the sequence length; any other geometry takes the
    generic instantiation (run-time masks, every key fragment): 112 x 112 -> 65 tokens, 154 x 154 -> 122 tokens (a partially
    valid key fragment), against the fp32 restatement with the same weights."""
    for img, seed in ((112, 21), (154, 22)):
        ref = EvaViTg(img_size=img, depth=2).eval().init_seeded_(seed)
        hip = HipEvaViTg(img_size=img, depth=2, device=dev).eval()
        hip.load_state_dict(ref.state_dict())
        x = torch.randn(3, 3, img, img, generator=torch.Generator().manual_seed(seed))
        with torch.no_grad():
            want = ref(x)
            got = hip(x.to(dev)).float().cpu()
        assert got.shape == want.shape == (3, (img // 14) ** 2 + 1, 1408)
        err = (got - want).abs()
        assert err.max().item() < 2e-2 and err.mean().item() < 2e-3, (img, err.max().item(), err.mean().item())
