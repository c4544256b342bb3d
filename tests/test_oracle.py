"""CPU suite: the oracle against the committed golden vectors (made from the HF port of the LAVIS
Q-Former, tools/make_golden.py), against the live HF model, and its own small-case properties."""
import json
import os

import numpy as np
import pytest
import torch

from oracle import qformer_ref as O
from tools.make_golden import CASES, make_inputs


@pytest.mark.parametrize("name", sorted(CASES))
def test_oracle_matches_golden(name, golden_dir):
    E, kv, n, L, wseed, iseed, ragged = CASES[name]
    gold = np.load(os.path.join(golden_dir, name + ".npz"))
    meta = json.loads(str(gold["meta"]))
    assert (meta["enc_width"], meta["kv"], meta["n"], meta["L"]) == (E, kv, n, L)
    cfg = O.QFormerCfg(enc_width=E)
    w = O.init_weights(cfg, seed=wseed, perturb=True)
    ids, tmask, att, feats = make_inputs(cfg, n, L, kv, iseed, ragged)
    enc = O.modality_layernorm(feats, w["ln.weight"], w["ln.bias"])
    np.testing.assert_allclose(enc[0, 0].numpy(), gold["enc_ln_row0"], rtol=0, atol=1e-6)
    h = O.qformer_forward(w, cfg, ids, att, w["query_tokens"].expand(n, -1, -1), enc)
    ref = torch.from_numpy(gold["last_hidden_state"])
    valid = att.bool()  # padded text positions are don't-care (their mask value differs: -1e4 vs finfo.min)
    assert (h - ref)[valid].abs().max().item() < 2e-5
    assert (h[:, :32] - ref[:, :32]).abs().max().item() < 2e-5


def test_oracle_matches_live_hf_small():
    from transformers import InstructBlipQFormerConfig, InstructBlipQFormerModel

    cfg = O.QFormerCfg(enc_width=256, layers=4, vocab=500, max_pos=64)
    w = O.init_weights(cfg, seed=3, perturb=True)
    hf = InstructBlipQFormerModel(InstructBlipQFormerConfig(vocab_size=cfg.vocab, encoder_hidden_size=256, num_hidden_layers=4,
                                                            max_position_embeddings=64)).eval()
    hf.load_state_dict(O.to_hf_state_dict(w), strict=True)
    g = torch.Generator().manual_seed(5)
    n, L, kv = 4, 7, 19
    ids = torch.randint(1, 500, (n, L), generator=g)
    tm = torch.ones(n, L, dtype=torch.long)
    tm[1, 4:] = 0
    tm[3, 1:] = 0
    att = torch.cat([torch.ones(n, 32, dtype=torch.long), tm], 1)
    enc = torch.randn(n, kv, 256, generator=g)
    q = w["query_tokens"].expand(n, -1, -1)
    with torch.no_grad():
        ref = hf(input_ids=ids, attention_mask=att, query_embeds=q, encoder_hidden_states=enc,
                 encoder_attention_mask=torch.ones(n, kv, dtype=torch.long)).last_hidden_state
    mine = O.qformer_forward(w, cfg, ids, att, q, enc)
    assert (mine - ref)[att.bool()].abs().max().item() < 2e-5
    # no text at all (L = 0): queries only
    mine0 = O.qformer_forward(w, cfg, ids[:, :0], att[:, :32], q, enc)
    with torch.no_grad():
        ref0 = hf(input_ids=None, attention_mask=att[:, :32], query_embeds=q, encoder_hidden_states=enc,
                  encoder_attention_mask=torch.ones(n, kv, dtype=torch.long)).last_hidden_state
    assert (mine0 - ref0).abs().max().item() < 2e-5


def test_reorder_and_repeat_quirk():
    # reference models/xinstructblip.py:283 written out literally
    for bs, num in [(1, 4), (2, 3), (2, 20), (3, 5)]:
        lit = [j_ + r for r, j in enumerate([[i * bs for i in range(num)]] * bs) for j_ in j]
        assert O.reorder_indices(bs, num) == lit
        frames = torch.arange(num * bs).view(num, bs)          # frames[i][r] = id of (position i, sample r)
        cat = frames.reshape(-1)
        got = cat[torch.tensor(lit)].view(bs, num)
        assert torch.equal(got, frames.t())
        ids = torch.arange(bs).view(bs, 1)
        assert ids.repeat(num, 1).view(-1).tolist() == O.repeat_text_rows(bs, num)


def test_modality_layernorm_matches_torch():
    g = torch.Generator().manual_seed(0)
    x = torch.randn(3, 5, 1408, generator=g) * 3 + 1
    w, b = torch.randn(1408, generator=g), torch.randn(1408, generator=g)
    ref = torch.nn.functional.layer_norm(x, (1408,), w, b, 1e-5)
    assert (O.modality_layernorm(x, w, b) - ref).abs().max().item() < 1e-5
    assert O.modality_layernorm(x.half(), w, b).dtype == torch.float16


def test_scorer_and_span_small_cases():
    z = torch.zeros(2, 3, 4)
    z[0, 1] = torch.tensor([1.0, 0, 0, 0])
    z[0, 2] = torch.tensor([0, 2.0, 0, 0])
    z[1, 0] = torch.tensor([-1.0, 0, 0, 0])
    t = torch.tensor([[3.0, 0, 0, 0], [1.0, 0, 0, 0]])
    sim, logit = O.cosine_scores(z, t)
    assert sim[0].tolist() == [0.0, 1.0, 0.0] and logit.tolist() == [1.0, 0.0]
    assert sim[1, 0].item() == -1.0
    # spans: first argmax wins ties, grows while >= lo + alpha * (hi - lo)
    assert O.span_from_logits(torch.tensor([0.0, 0.6, 1.0, 0.5, 0.1])) == (1, 3)
    assert O.span_from_logits(torch.tensor([1.0, 0.0, 1.0])) == (0, 0)
    assert O.span_from_logits(torch.tensor([0.3])) == (0, 0)
    assert O.span_from_logits(torch.tensor([2.0, 2.0, 2.0])) == (0, 2)
    assert O.span_from_logits(torch.tensor([0.0, 0.2, 0.9, 1.0]), alpha=0.1) == (1, 3)
    assert O.spans_to_text([(1, 3)], [[0, 2, 5, 7, 9]]) == ["[[2, 7]]"]
    f = O.fuse_logits([torch.tensor([1.0, 2.0]), torch.tensor([3.0, 6.0])])
    assert f.tolist() == [2.0, 4.0]


def test_fp64_and_fp32_oracle_agree():
    cfg = O.QFormerCfg(enc_width=128, layers=2, vocab=100, max_pos=32)
    w = O.init_weights(cfg, seed=1, perturb=True)
    g = torch.Generator().manual_seed(2)
    ids = torch.randint(1, 100, (2, 5), generator=g)
    att = torch.ones(2, 37, dtype=torch.long)
    enc = torch.randn(2, 9, 128, generator=g)
    q = w["query_tokens"].expand(2, -1, -1)
    a = O.qformer_forward(w, cfg, ids, att, q, enc)
    b = O.qformer_forward(w, cfg, ids, att, q, enc, dtype=torch.float64)
    assert (a - b).abs().max().item() < 1e-5
