"""Golden vectors for the EVA ViT-g restatement (row A1 / N4), generated in the build container.

Source of truth: ``transformers.InstructBlipVisionModel`` (HF modeling_instructblip.py:392-440; 1408 wide, 16 heads x 88,
MLP 6144, 14 x 14 patches of a 224 x 224 frame -> 257 tokens, pre-LN blocks, exact GELU, LayerNorm eps 1e-6) -- the
in-image structural stand-in SURVEY 8(c) names for LAVIS' ``create_eva_vit_g`` (reference models/xinstructblip.py:658-666;
LAVIS itself is absent, so parity with the reference's own encoder stays unpinned).  Full width, DEPTH layers (the
geometry per layer is what is pinned; 39 identical layers add nothing but time).  Weights are NOT stored: they are
re-derived from the seed by ``EvaViTg.init_seeded_``; the frames from ``make_frames``.  Stored: the last encoder layer's
output BEFORE HF's ``post_layernorm`` (that LayerNorm is the reference's separate ``video_ln``) at a few token rows, plus
per-token checksums of all 257 tokens.

    python tools/make_vit_golden.py        # writes tests/golden/vit_g.npz
    python tools/make_vit_golden.py full   # writes tests/golden/vit_g_d39.npz (all 39 blocks, ~1 TFLOP of fp32 CPU work, ~9 GB of RAM)
"""
from __future__ import annotations

import json
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

DEPTH, WEIGHT_SEED, INPUT_SEED, FRAMES = 3, 5, 6, 2
ROWS = [0, 1, 128, 256]
# the full 39-block encoder (what bench.py times): 2 frames, rows + per-token checksums only -> tests/golden/vit_g_d39.npz
FULL_DEPTH, FULL_WEIGHT_SEED, FULL_INPUT_SEED = 39, 11, 12


def make_frames(n: int = FRAMES, seed: int = INPUT_SEED) -> torch.Tensor:
    """Seeded frames, normalised-pixel range, [n, 3, 224, 224] fp32 (shared by the generator and the tests)."""
    return torch.randn(n, 3, 224, 224, generator=torch.Generator().manual_seed(seed))


def hf_reference(vit, frames):
    from transformers import InstructBlipVisionConfig, InstructBlipVisionModel

    cfg = InstructBlipVisionConfig(hidden_size=1408, intermediate_size=6144, num_hidden_layers=len(vit.blocks), num_attention_heads=16,
                                   image_size=224, patch_size=14, hidden_act="gelu", layer_norm_eps=1e-6, qkv_bias=True)
    hf = InstructBlipVisionModel(cfg).eval()
    sd = vit.hf_state_dict()
    sd["post_layernorm.weight"], sd["post_layernorm.bias"] = torch.ones(1408), torch.zeros(1408)
    hf.load_state_dict(sd, strict=True)
    with torch.no_grad():
        out = hf(pixel_values=frames, output_hidden_states=True)
    return out.hidden_states[-1]          # last encoder layer, before post_layernorm


if __name__ == "__main__":
    from mraudio_amd.models.eva_vit import EvaViTg

    torch.set_num_threads(8)
    if len(sys.argv) > 1 and sys.argv[1] == "full":
        vit = EvaViTg(depth=FULL_DEPTH).eval().init_seeded_(FULL_WEIGHT_SEED)
        frames = make_frames(FRAMES, FULL_INPUT_SEED)
        ref = hf_reference(vit, frames)
        with torch.no_grad():
            own = vit(frames)
        print("depth 39, restatement vs HF: max|d|", (own - ref).abs().max().item(), "on |y| max", ref.abs().max().item(), "rms", ref.pow(2).mean().sqrt().item())
        np.savez_compressed(os.path.join(ROOT, "tests", "golden", "vit_g_d39.npz"),
                            meta=np.array(json.dumps(dict(depth=FULL_DEPTH, weight_seed=FULL_WEIGHT_SEED, input_seed=FULL_INPUT_SEED, frames=FRAMES, rows=ROWS,
                                                          source="transformers InstructBlipVisionModel, 39 layers, hidden_states[-1] (before post_layernorm)"))),
                            rows=ref[:, ROWS].numpy().astype(np.float32),
                            token_sum=ref.sum(-1).numpy().astype(np.float32), token_abs_sum=ref.abs().sum(-1).numpy().astype(np.float32),
                            token_sq_sum=ref.pow(2).sum(-1).numpy().astype(np.float32))
        print("wrote tests/golden/vit_g_d39.npz", tuple(ref.shape))
        sys.exit(0)
    vit = EvaViTg(depth=DEPTH).eval().init_seeded_(WEIGHT_SEED)
    frames = make_frames()
    ref = hf_reference(vit, frames)
    with torch.no_grad():
        own = vit(frames)
    print("restatement vs HF: max|d|", (own - ref).abs().max().item(), "on |y| max", ref.abs().max().item())
    np.savez_compressed(os.path.join(ROOT, "tests", "golden", "vit_g.npz"),
                        meta=np.array(json.dumps(dict(depth=DEPTH, weight_seed=WEIGHT_SEED, input_seed=INPUT_SEED, frames=FRAMES, rows=ROWS,
                                                      source="transformers InstructBlipVisionModel, hidden_states[-1] (before post_layernorm)"))),
                        rows=ref[:, ROWS].numpy().astype(np.float32),
                        token_sum=ref.sum(-1).numpy().astype(np.float32), token_abs_sum=ref.abs().sum(-1).numpy().astype(np.float32))
    print("wrote tests/golden/vit_g.npz", tuple(ref.shape))
