"""Generates the committed golden vectors under tests/golden/ (run once, in the build container).

Source of truth for the Q-Former vectors: the HF port of the LAVIS Q-Former that ships in this
image (``transformers.InstructBlipQFormerModel``) -- the reference's own model code is not importable
(LAVIS absent) and the reference holds no fixtures.  Weights are NOT stored: they are re-derived
from the seed by the documented recipe (``oracle/qformer_ref.py:init_weights``); inputs are
re-derived from ``input_seed`` by ``make_inputs`` below.  Stored: the outputs (a few hundred KB).

Integer known-answer vectors for the span/metric helpers come from importing the reference's
``utils/utils.py`` (with an empty ``wandb`` module in ``sys.modules``; wandb is only touched by an
unused logging helper) and ``eval/mr_utils.py`` from /root/reference -- data only is written.

    python tools/make_golden.py            # writes tests/golden/*.npz, *.json
"""
from __future__ import annotations

import json
import os
import sys
import types

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import qformer_ref as O  # noqa: E402

GOLD = os.path.join(ROOT, "tests", "golden")

CASES = {
    # name: (enc_width, kv, N, L, weight_seed, input_seed, ragged)
    "qformer_video": (1408, 257, 3, 16, 0, 11, False),
    "qformer_audio_ragged": (768, 256, 3, 12, 1, 12, True),
}


def make_inputs(cfg: O.QFormerCfg, n: int, L: int, kv: int, input_seed: int, ragged: bool):
    """Seeded inputs shared by the generator and the tests (kept here so both use one definition)."""
    g = torch.Generator().manual_seed(input_seed)
    ids = torch.randint(1000, 30000, (n, L), generator=g)
    tmask = torch.ones(n, L, dtype=torch.long)
    if ragged:
        for r in range(n):
            keep = max(1, L - 3 * r - 1)
            tmask[r, keep:] = 0
    att = torch.cat([torch.ones(n, cfg.n_query, dtype=torch.long), tmask], dim=1)
    feats = torch.randn(n, kv, cfg.enc_width, generator=g)
    return ids, tmask, att, feats


def qformer_cases():
    from transformers import InstructBlipQFormerConfig, InstructBlipQFormerModel

    for name, (E, kv, n, L, wseed, iseed, ragged) in CASES.items():
        cfg = O.QFormerCfg(enc_width=E)
        w = O.init_weights(cfg, seed=wseed, perturb=True)
        hf = InstructBlipQFormerModel(InstructBlipQFormerConfig(vocab_size=cfg.vocab, encoder_hidden_size=E)).eval()
        hf.load_state_dict(O.to_hf_state_dict(w), strict=True)
        ids, tmask, att, feats = make_inputs(cfg, n, L, kv, iseed, ragged)
        enc = O.modality_layernorm(feats, w["ln.weight"], w["ln.bias"])
        q = w["query_tokens"].expand(n, -1, -1)
        with torch.no_grad():
            ref = hf(input_ids=ids, attention_mask=att, query_embeds=q, encoder_hidden_states=enc,
                     encoder_attention_mask=torch.ones(n, kv, dtype=torch.long)).last_hidden_state
        np.savez_compressed(
            os.path.join(GOLD, name + ".npz"),
            meta=np.array(json.dumps(dict(enc_width=E, kv=kv, n=n, L=L, weight_seed=wseed, input_seed=iseed,
                                          ragged=ragged, perturb=True, source="transformers InstructBlipQFormerModel"))),
            last_hidden_state=ref.numpy().astype(np.float32),
            enc_ln_row0=enc[0, 0].numpy().astype(np.float32),
        )
        print(name, tuple(ref.shape))


def integer_kats():
    ref = "/root/reference"
    if not os.path.isdir(ref):
        print("reference not present: skipping integer KATs")
        return
    sys.modules.setdefault("wandb", types.ModuleType("wandb"))
    sys.path.insert(0, ref)
    from utils.utils import convert_percentages_to_second, moment_str_to_list, post_process  # type: ignore
    from eval.mr_utils import compute_temporal_iou_batch_cross, compute_temporal_iou_batch_paired  # type: ignore

    strings = ["[[0 1] [7, 4]]", "[[3,, 9]]</s>junk", "no windows", "[[12, 5]]\n", "[[1, 2, 3]]", "[[0.5, 2]]",
               "[[3, 9]]", "[[10, 20], [30, 25]]", "[[ 4 , 8 ]]", "[[7]]", "[[5, 5]]", "[[0, 1],, [2, 3]]"]
    kat = {"post_process": [], "convert_percentages": [], "iou_cross": [], "iou_paired": []}
    for s in strings:
        p = post_process(s)
        kat["post_process"].append({"in": s, "post": p, "list": moment_str_to_list(p)})
    for s, d in [("[[0.25, 0.5]]", 150), ("[[0.1, 0.9], [0.5, 0.75]]", 60), ("nothing", 10)]:
        kat["convert_percentages"].append({"in": s, "duration": d, "out": convert_percentages_to_second(s, d)})
    a = np.array([[0, 10], [2, 5]], dtype=float)
    b = np.array([[5, 15], [0, 10], [3, 4]], dtype=float)
    iou, union = compute_temporal_iou_batch_cross(a, b)
    kat["iou_cross"].append({"a": a.tolist(), "b": b.tolist(), "iou": np.asarray(iou).tolist()})
    p1 = np.array([[0, 10], [2, 5], [1, 2]], dtype=float)
    p2 = np.array([[5, 15], [2, 5], [3, 4]], dtype=float)
    kat["iou_paired"].append({"a": p1.tolist(), "b": p2.tolist(), "iou": np.asarray(compute_temporal_iou_batch_paired(p1, p2)).tolist()})
    with open(os.path.join(GOLD, "integer_kats.json"), "w") as f:
        json.dump(kat, f, indent=1)
    print("integer KATs", {k: len(v) for k, v in kat.items()})


if __name__ == "__main__":
    os.makedirs(GOLD, exist_ok=True)
    torch.set_num_threads(8)
    qformer_cases()
    integer_kats()
