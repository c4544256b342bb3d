"""Parity tests proper (``-m gpu``): the HIP path, called through the C ABI, against the CPU oracle
on the same seeded inputs, against the committed golden vectors, and -- at BASELINE.json's full
sizes -- through size-independent properties.

Tolerances (north_star: similarity logits within 1e-3 relative; integer spans bit-exact):
  * hidden states: MFMA operands are f16 (11-bit significand), everything else fp32; 12 layers of
    f16-operand GEMMs land at ~1e-3 of the hidden-state scale -> max|d| <= 1e-2 on |z| <= ~8.
  * similarity logits: |d| <= 1e-3 * max|logit| (relative to the logit scale), asserted below.
  * integers (spans, reorder): exact.
"""
import json
import os

import numpy as np
import pytest
import torch

from oracle import qformer_ref as O
from tools.make_golden import CASES, make_inputs

pytestmark = pytest.mark.gpu

Z_ATOL = 1e-2
LOGIT_RTOL = 1e-3


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available(), "gpu tests need a GPU"
    return torch.device("cuda:0")


def build_qformer(dev, enc_width, seed, perturb=True, op_dtype=torch.float16, **over):
    from mraudio_amd.qformer import QFormer, QFormerConfig, draw_seeded

    cfg = QFormerConfig(enc_width=enc_width, op_dtype=op_dtype, **over)
    qf = QFormer(cfg, device=dev)
    g = qf.init_seeded_(seed=seed, perturb=perturb)
    extras = {
        "query_tokens": draw_seeded(g, (1, cfg.n_query, cfg.hidden), "w", perturb),
        "ln.weight": draw_seeded(g, (enc_width,), "g", perturb),
        "ln.bias": draw_seeded(g, (enc_width,), "z", perturb),
        "llm_proj.weight": draw_seeded(g, (cfg.llm_hidden, cfg.hidden), "w", perturb),
        "llm_proj.bias": draw_seeded(g, (cfg.llm_hidden,), "b", perturb),
    }
    for k, v in extras.items():
        qf.push(k, v)
    qf.sync_weights()
    assert qf.missing() == []
    return qf, cfg


def oracle_cfg(cfg):
    return O.QFormerCfg(hidden=cfg.hidden, heads=cfg.heads, inter=cfg.inter, layers=cfg.layers, cross_freq=cfg.cross_freq,
                        enc_width=cfg.enc_width, vocab=cfg.vocab, max_pos=cfg.max_pos, llm_hidden=cfg.llm_hidden)


@pytest.fixture(scope="module")
def video(dev):
    qf, cfg = build_qformer(dev, 1408, seed=0)
    return qf, cfg, O.init_weights(oracle_cfg(cfg), seed=0, perturb=True)


@pytest.fixture(scope="module")
def audio(dev):
    qf, cfg = build_qformer(dev, 768, seed=1)
    return qf, cfg, O.init_weights(oracle_cfg(cfg), seed=1, perturb=True)


def test_extension_is_the_hip_library():
    from mraudio_amd import _lib

    assert os.path.exists(_lib.LIB_PATH)
    assert b"gfx950" in _lib.lib().mra_version()


def test_modality_ln_and_reorder(video, dev):
    qf, cfg, w = video
    g = torch.Generator().manual_seed(3)
    bs, num, kv = 2, 3, 257
    frames = [torch.randn(bs, kv, 1408, generator=g) * 2 + 0.5 for _ in range(num)]  # per-position encoder outputs
    raw = torch.cat(frames)
    idx = torch.tensor(O.reorder_indices(bs, num))
    ref = O.modality_layernorm(raw, w["ln.weight"], w["ln.bias"])[idx]
    for dt, tol in ((torch.float32, 2e-3), (torch.float16, 4e-3)):
        got = qf.modality_ln(raw.to(dev).to(dt), item_index=idx.to(dev)).float().cpu()
        ref_dt = O.modality_layernorm(raw.to(dt).float(), w["ln.weight"], w["ln.bias"])[idx]
        assert got.shape == ref.shape
        assert (got - ref_dt).abs().max().item() < tol  # f16 output rounding of |y| <~ 6
    # identity index == no index
    a = qf.modality_ln(raw.to(dev))
    b = qf.modality_ln(raw.to(dev), item_index=torch.arange(bs * num, device=dev))
    assert torch.equal(a, b)


@pytest.mark.parametrize("name", sorted(CASES))
def test_qformer_matches_golden_and_oracle(name, video, audio, dev, golden_dir):
    E, kv, n, L, wseed, iseed, ragged = CASES[name]
    qf, cfg, w = video if E == 1408 else audio
    ocfg = oracle_cfg(cfg)
    ids, tmask, att, feats = make_inputs(ocfg, n, L, kv, iseed, ragged)
    gold = torch.from_numpy(np.load(os.path.join(golden_dir, name + ".npz"))["last_hidden_state"])
    enc = qf.modality_ln(feats.to(dev))
    res = qf.forward_fused(ids.to(dev), att.to(dev), enc, want_query=True, want_full=True, want_cls=True)
    full = res["full"].cpu()
    valid = att.bool()
    assert (full - gold)[valid].abs().max().item() < Z_ATOL
    assert torch.equal(res["query"].cpu(), full[:, :32])
    assert torch.equal(res["cls"].cpu(), full[:, 32])
    # the cheap variants (no last-layer text FFN / [CLS] row only) give the same numbers
    only_q = qf.forward_fused(ids.to(dev), att.to(dev), enc, want_query=True)["query"].cpu()
    assert torch.equal(only_q, full[:, :32])
    q_cls = qf.forward_fused(ids.to(dev), att.to(dev), enc, want_query=True, want_cls=True)
    assert torch.equal(q_cls["query"].cpu(), full[:, :32])
    assert (q_cls["cls"].cpu() - full[:, 32]).abs().max().item() < 1e-5
    # oracle run on the exact f16-rounded features the kernel saw: isolates Q-Former error
    enc_ref = O.modality_layernorm(feats, w["ln.weight"], w["ln.bias"])
    h = O.qformer_forward(w, ocfg, ids, att, w["query_tokens"].expand(n, -1, -1), enc_ref)
    assert (full - h)[valid].abs().max().item() < Z_ATOL
    rel = ((full - h)[valid].norm() / h[valid].norm()).item()
    assert rel < 2e-3, rel


def test_bert_call_signature_like_the_reference(video, dev):
    qf, cfg, w = video
    ocfg = oracle_cfg(cfg)
    n, L, kv = 2, 6, 40
    ids, tmask, att, feats = make_inputs(ocfg, n, L, kv, 21, True)
    enc = O.modality_layernorm(feats, w["ln.weight"], w["ln.bias"])
    q = w["query_tokens"].expand(n, -1, -1)
    out = qf.bert(ids.to(dev), attention_mask=att.to(dev), query_embeds=q.to(dev).repeat(1, 1, 1),
                  encoder_hidden_states=enc.to(dev), encoder_attention_mask=torch.ones(n, kv, dtype=torch.long, device=dev),
                  return_dict=True)
    h = O.qformer_forward(w, ocfg, ids, att, q, enc)
    assert out.last_hidden_state.shape == (n, 32 + L, 768)
    assert (out.last_hidden_state.cpu() - h)[att.bool()].abs().max().item() < Z_ATOL
    # a different query_embeds tensor per item is honoured
    q2 = q.clone()
    q2[1] = q2[1] * 1.5
    out2 = qf.bert(ids.to(dev), attention_mask=att.to(dev), query_embeds=q2.to(dev), encoder_hidden_states=enc.to(dev))
    h2 = O.qformer_forward(w, ocfg, ids, att, q2, enc)
    assert (out2.last_hidden_state.cpu() - h2)[att.bool()].abs().max().item() < Z_ATOL
    with pytest.raises(ValueError):
        qf.bert(ids.to(dev), attention_mask=att.to(dev), query_embeds=q.to(dev))
    with pytest.raises(NotImplementedError):
        bad = torch.ones(n, kv, dtype=torch.long, device=dev)
        bad[0, 3] = 0
        qf.bert(ids.to(dev), attention_mask=att.to(dev), query_embeds=q.to(dev), encoder_hidden_states=enc.to(dev),
                encoder_attention_mask=bad)


def test_edge_cases(video, dev):
    from mraudio_amd import MraError

    qf, cfg, w = video
    ocfg = oracle_cfg(cfg)
    # empty batch
    e = qf.forward_fused(torch.zeros(0, 4, dtype=torch.long, device=dev), None, torch.zeros(0, 5, 1408, device=dev, dtype=torch.float16))
    assert e["query"].shape == (0, 32, 768)
    # no text (L = 0), one encoder token (kv = 1), one item
    g = torch.Generator().manual_seed(9)
    enc = torch.randn(1, 1, 1408, generator=g)
    q = w["query_tokens"].expand(1, -1, -1)
    got = qf.forward_fused(None, None, enc.to(dev).half(), want_query=True)["query"].cpu()
    ref = O.qformer_forward(w, ocfg, torch.zeros(1, 0, dtype=torch.long), torch.ones(1, 32, dtype=torch.long), q, enc.half().float())
    assert (got - ref).abs().max().item() < Z_ATOL
    # longest prompt the reference allows (max_txt_len 128 -> S = 160), heavy padding on one row
    n, L, kv = 2, 128, 33
    ids, tmask, att, feats = make_inputs(ocfg, n, L, kv, 5, False)
    att[1, 32 + 7:] = 0
    encf = O.modality_layernorm(feats, w["ln.weight"], w["ln.bias"])
    full = qf.forward_fused(ids.to(dev), att.to(dev), encf.to(dev).half(), want_full=True)["full"].cpu()
    ref = O.qformer_forward(w, ocfg, ids, att, w["query_tokens"].expand(n, -1, -1), encf.half().float())
    assert (full - ref)[att.bool()].abs().max().item() < Z_ATOL
    # bad shapes fail loudly
    with pytest.raises(MraError):
        qf.forward_fused(ids.to(dev), att[:, :-1].to(dev), encf.to(dev).half())
    with pytest.raises(MraError):
        qf.forward_fused(ids.to(dev), att.to(dev), encf[..., :100].to(dev).half())
    with pytest.raises(MraError):
        qf.push("bert.not.a.parameter", torch.zeros(4, device=dev))


def test_llm_projection(video, dev):
    qf, cfg, w = video
    g = torch.Generator().manual_seed(4)
    bs, num = 2, 3
    z = torch.randn(bs * num, 32 + 5, 768, generator=g)
    ref = O.llm_project(z, w, bs, num)
    got = qf.llm_proj(z[:, :32].to(dev)).reshape(bs, num * 32, -1).cpu()
    assert got.shape == ref.shape == (bs, num * 32, 4096)
    assert (got - ref).abs().max().item() < 3e-3  # |y| ~ 0.6, f16 operands


def test_scorer_fuse_span_bit_exact_integers(dev):
    from mraudio_amd import scorer

    g = torch.Generator().manual_seed(8)
    z = torch.randn(40, 32, 768, generator=g)
    t = torch.randn(40, 768, generator=g)
    sim_r, logit_r = O.cosine_scores(z, t)
    sim, logit = scorer.cosine_scores(z.to(dev), t.to(dev))
    assert (sim.cpu() - sim_r).abs().max().item() < 1e-6 and (logit.cpu() - logit_r).abs().max().item() < 1e-6
    _, logit1 = scorer.cosine_scores(z.to(dev), t[:1].to(dev), want_sim=False)
    assert (logit1.cpu() - O.cosine_scores(z, t[:1])[1]).abs().max().item() < 1e-6
    # fusion + span on IDENTICAL fp32 logits must be bit-exact / integer-exact
    a, b = torch.randn(60, generator=g), torch.randn(60, generator=g)
    for wts in (None, (0.3, 0.7)):
        fr = O.fuse_logits([a, b], wts)
        fg = scorer.fuse_logits([a.to(dev), b.to(dev)], wts).cpu()
        assert torch.equal(fr, fg)
    x = torch.randn(7, 61, generator=g)
    x[2, 5] = x[2, 40] = 9.0          # tie -> first argmax
    x[3] = 1.25                        # constant row -> whole video
    for alpha in (0.5, 0.1, 0.9):
        got = scorer.spans_from_logits(x.reshape(-1).to(dev), 7, 61, alpha).cpu().tolist()
        ref = [list(O.span_from_logits(x[v], alpha)) for v in range(7)]
        assert got == ref
    assert scorer.spans_to_text([(1, 3)], [[0, 2, 5, 7]]) == O.spans_to_text([(1, 3)], [[0, 2, 5, 7]])


def test_end_to_end_encode_fuse_score_vs_oracle(dev):
    """BASELINE config 3 shape at small N: both modalities, LN -> Q-Former -> score -> fuse -> span."""
    from mraudio_amd.models.xinstructblip import ENC_WIDTH, XInstructBLIP

    model = XInstructBLIP(seed=0, perturb=True, device=dev)
    bs, num = 2, 3
    g = torch.Generator().manual_seed(17)
    feats = {"video": torch.randn(bs, num, 257, 1408, generator=g), "audio": torch.randn(bs, num, 256, 768, generator=g)}
    prompts = ["Query: a person opens the door.\nRelevant windows: ", "Query: someone is cooking in the kitchen while music plays.\nRelevant windows: "]
    samples = {"video_embeds": feats["video"], "audio_embeds": feats["audio"], "text_input": prompts,
               "timestamps": [[0, 3, 6], [1, 4, 8]], "duration": [9, 10]}
    out = model.encode_fuse(samples, want_llm=True)
    text = model.tokenizer(prompts, padding="longest", truncation=True, max_length=128, return_tensors="pt")
    rows = O.repeat_text_rows(bs, num)       # the reference's .repeat(T, 1) pairing
    ids, tm = text.input_ids[rows], text.attention_mask[rows]
    assert torch.equal(ids, text.input_ids.repeat(num, 1))
    cfgs = {m: O.QFormerCfg(enc_width=ENC_WIDTH[m]) for m in ("video", "audio")}
    ws = {"video": O.init_weights(cfgs["video"], seed=0, perturb=True), "audio": O.init_weights(cfgs["audio"], seed=1, perturb=True)}
    ref = O.encode_fuse_score(ws, cfgs, {m: feats[m].reshape(bs * num, *feats[m].shape[2:]) for m in ("audio", "video")}, ids, tm, bs, num)
    for m in ("video", "audio"):
        assert (out["z"][m].cpu() - ref["z"][m]).abs().max().item() < Z_ATOL
        scale = ref["logit"][m].abs().max().item()
        assert (out["logit"][m].cpu() - ref["logit"][m]).abs().max().item() <= LOGIT_RTOL * scale, (m, scale)
        assert (out["sim"][m].cpu() - ref["sim"][m]).abs().max().item() <= LOGIT_RTOL * ref["sim"][m].abs().max().item()
        yl = O.llm_project(torch.cat([ref["z"][m], torch.zeros(bs * num, 1, 768)], 1), ws[m], bs, num)
        assert (out["inputs_llm"][m].cpu() - yl).abs().max().item() < 5e-3
        assert out["atts_llm"][m].shape == (bs, num * 32)
    scale = ref["fused"].abs().max().item()
    assert (out["fused"].cpu() - ref["fused"]).abs().max().item() <= LOGIT_RTOL * scale
    assert [tuple(s) for s in out["spans"].cpu().tolist()] == [tuple(s) for s in ref["spans"]]
    strings = model.generate(samples)
    assert strings == O.spans_to_text(ref["spans"], samples["timestamps"])
    from mraudio_amd.utils import spans as sp
    assert all(sp.moment_str_to_list(sp.post_process(s))[0][0] >= 0 for s in strings)
    # checkpoint key names of the reference round-trip through load_state_dict
    sd = model.state_dict()
    for k in ("video_Qformer.bert.encoder.layer.0.crossattention.self.key.weight", "audio_Qformer.bert.embeddings.LayerNorm.bias",
              "video_query_tokens", "audio_ln.weight", "video_llm_proj.bias", "audio_Qformer.bert.encoder.layer.11.output_query.dense.weight"):
        assert k in sd, k
    assert sd["video_Qformer.bert.encoder.layer.0.crossattention.self.key.weight"].shape == (768, 1408)
    assert "video_Qformer.bert.encoder.layer.1.crossattention.self.key.weight" not in sd
    msg = model.load_state_dict(sd, strict=True)
    assert list(msg.missing_keys) == [] and list(msg.unexpected_keys) == []
    again = model.encode_fuse(samples)
    assert torch.equal(again["fused"], out["fused"])
    # aligned text pairing differs from the reference's quirk only when prompts differ
    assert model.forward({})["loss"].item() == 0.0


def test_full_size_properties(video, dev):
    """BASELINE sizes (32 clips; 32-frame clips = 8224 video tokens): properties that need no oracle
    run, plus an oracle spot check of two items at Kv = 8224."""
    qf, cfg, w = video
    ocfg = oracle_cfg(cfg)
    n, L, kv = 32, 32, 257
    ids, tmask, att, feats = make_inputs(ocfg, n, L, kv, 31, False)
    enc = qf.modality_ln(feats.to(dev))
    z = qf.forward_fused(ids.to(dev), att.to(dev), enc, want_query=True, want_cls=True)
    assert torch.isfinite(z["query"]).all()
    # items are independent: a permutation of the batch permutes the output (same tiles, same order of
    # accumulation per element -> tight tolerance), and a half batch equals the matching half
    perm = torch.randperm(n, generator=torch.Generator().manual_seed(1))
    zp = qf.forward_fused(ids[perm].to(dev), att[perm].to(dev), enc[perm.to(dev)], want_query=True)["query"]
    assert (zp - z["query"][perm.to(dev)]).abs().max().item() < 1e-4
    zh = qf.forward_fused(ids[:16].to(dev), att[:16].to(dev), enc[:16], want_query=True)["query"]
    assert (zh - z["query"][:16]).abs().max().item() < 1e-4
    # determinism
    z2 = qf.forward_fused(ids.to(dev), att.to(dev), enc, want_query=True)["query"]
    assert torch.equal(z2, z["query"])
    # 32-frame clips: Kv = 32 * 257; two items against the oracle
    n2, kv2 = 2, 8224
    ids2, _, att2, feats2 = make_inputs(ocfg, n2, 16, kv2, 41, False)
    enc2 = qf.modality_ln(feats2.to(dev))
    got = qf.forward_fused(ids2.to(dev), att2.to(dev), enc2, want_query=True)["query"].cpu()
    ref = O.qformer_forward(w, ocfg, ids2, att2, w["query_tokens"].expand(n2, -1, -1), O.modality_layernorm(feats2, w["ln.weight"], w["ln.bias"]))[:, :32]
    assert (got - ref).abs().max().item() < Z_ATOL
    # concatenating the SAME 257 tokens 32 times leaves softmax(QK)V unchanged: Kv = 8224 == Kv = 257
    rep = feats[:2].repeat(1, 32, 1)
    a = qf.forward_fused(ids[:2].to(dev), att[:2].to(dev), qf.modality_ln(rep.to(dev)), want_query=True)["query"]
    assert (a - z["query"][:2]).abs().max().item() < 5e-3


def test_bf16_operands_run_and_are_close(dev):
    qf, cfg = build_qformer(dev, 768, seed=1, op_dtype=torch.bfloat16)
    w = O.init_weights(oracle_cfg(cfg), seed=1, perturb=True)
    ocfg = oracle_cfg(cfg)
    ids, tmask, att, feats = make_inputs(ocfg, 2, 8, 64, 3, True)
    enc = qf.modality_ln(feats.to(dev))
    assert enc.dtype == torch.bfloat16
    got = qf.forward_fused(ids.to(dev), att.to(dev), enc, want_query=True)["query"].cpu()
    ref = O.qformer_forward(w, ocfg, ids, att, w["query_tokens"].expand(2, -1, -1), O.modality_layernorm(feats, w["ln.weight"], w["ln.bias"]))[:, :32]
    rel = ((got - ref).norm() / ref.norm()).item()
    assert rel < 2e-2, rel  # bf16 has 8 significand bits: ~8x the f16 error; f16 is the default for that reason
    for mode in ("fold", "fold384"):           # the folded cross-attention in bf16, both tile families
        qf.set_cross_mode(mode)
        got = qf.forward_fused(ids.to(dev), att.to(dev), enc, want_query=True)["query"].cpu()
        rel = ((got - ref).norm() / ref.norm()).item()
        assert rel < 2e-2, (mode, rel)
    qf.set_cross_mode("auto")


def test_folded_cross_attention_matches_kv_cache_path_and_oracle(video, audio, dev):
    """The folded cross-attention (S = (Q W_k) enc^T, ctx = (P enc) W_v^T + b_v; automatic from Kv >= 2048) is the
    same arithmetic re-associated: forced on at small and ragged Kv it must agree with the K/V-cache path and with
    the oracle to the usual bar, and the automatic switch must pick it for a long sequence."""
    for (qf, cfg, w), kv, n, L in ((video, 257, 3, 9), (audio, 100, 2, 5), (video, 130, 2, 4), (video, 200, 7, 4)):
        ocfg = oracle_cfg(cfg)
        ids, tmask, att, feats = make_inputs(ocfg, n, L, kv, 11, True)
        enc = qf.modality_ln(feats.to(dev))
        qf.set_cross_mode("kv_cache")
        ref = qf.forward_fused(ids.to(dev), att.to(dev), enc, want_query=True, want_cls=True)
        qf.set_cross_mode("fold")
        got = qf.forward_fused(ids.to(dev), att.to(dev), enc, want_query=True, want_cls=True)
        qf.set_cross_mode("fold_stream")    # video geometry: the streaming kernels; audio (E = 768): falls back to "fold"
        strm = qf.forward_fused(ids.to(dev), att.to(dev), enc, want_query=True, want_cls=True)
        qf.set_cross_mode("fold_rescale_pass")   # the split softmax with its rescale pass over P instead of the factors inside P.enc
        resc = qf.forward_fused(ids.to(dev), att.to(dev), enc, want_query=True, want_cls=True)
        qf.set_cross_mode("auto")
        assert (resc["query"] - ref["query"]).abs().max().item() < 5e-3 and (resc["cls"] - ref["cls"]).abs().max().item() < 5e-3
        assert (strm["query"] - ref["query"]).abs().max().item() < 5e-3 and (strm["cls"] - ref["cls"]).abs().max().item() < 5e-3
        assert (got["query"] - ref["query"]).abs().max().item() < 5e-3, (kv, (got["query"] - ref["query"]).abs().max().item())
        assert (got["cls"] - ref["cls"]).abs().max().item() < 5e-3
        enc_ref = O.modality_layernorm(feats, w["ln.weight"], w["ln.bias"])
        h = O.qformer_forward(w, ocfg, ids, att, w["query_tokens"].expand(n, -1, -1), enc_ref)
        assert (got["query"].cpu() - h[:, :32]).abs().max().item() < Z_ATOL
        rel = ((got["query"].cpu() - h[:, :32]).norm() / h[:, :32].norm()).item()
        assert rel < 2e-3, rel
    # long sequence: automatic = folded; against the cache path on the same inputs
    qf, cfg, w = video
    g = torch.Generator().manual_seed(5)
    feats = torch.randn(2, 2100, 1408, generator=g)
    ids = torch.randint(1000, 30000, (2, 7), generator=g)
    att = torch.ones(2, 39, dtype=torch.long)
    enc = qf.modality_ln(feats.to(dev))
    auto = qf.forward_fused(ids.to(dev), att.to(dev), enc, want_query=True)["query"]
    qf.set_cross_mode("kv_cache")
    ref = qf.forward_fused(ids.to(dev), att.to(dev), enc, want_query=True)["query"]
    qf.set_cross_mode("fold")
    forced = qf.forward_fused(ids.to(dev), att.to(dev), enc, want_query=True)["query"]
    qf.set_cross_mode("auto")
    assert torch.equal(auto, forced)
    assert (auto - ref).abs().max().item() < 5e-3


def test_modality_streams_are_joined_before_fusion(dev):
    """The modality Q-Formers run on side streams (one of them high priority); fusion and span selection run on the
    caller's stream and must wait for BOTH.  Two different batches back to back with a long video sequence: a missing
    join would fuse the second batch's audio logits with the first batch's (stale) video logits."""
    from mraudio_amd.models.xinstructblip import XInstructBLIP

    model = XInstructBLIP(seed=4, perturb=True, device=dev)
    g = torch.Generator().manual_seed(9)
    n, L = 8, 6
    ids = torch.randint(1000, 30000, (n, L), generator=g).to(dev)
    tmask = torch.ones(n, L, dtype=torch.long, device=dev)
    batches = [{"video": torch.randn(n, 2304, 1408, generator=g).to(dev).half(), "audio": torch.randn(n, 64, 768, generator=g).to(dev).half()}
               for _ in range(2)]
    model.overlap_modalities = False
    want = [model.fuse_score(b, ids, tmask, bs=1, num=n)["fused"].clone() for b in batches]
    assert not torch.allclose(want[0], want[1])
    model.overlap_modalities = True
    for _ in range(3):
        got = [model.fuse_score(b, ids, tmask, bs=1, num=n)["fused"].clone() for b in batches]
        assert torch.equal(got[0], want[0]) and torch.equal(got[1], want[1])


def test_folded_path_on_a_small_configuration(dev):
    """Not the reference geometry (4 heads -> 128 (head, query) rows, 2 layers, E = 256): the folded path then runs on its
    fallback tiles (128 x 128, fp32 score rows + softmax kernel) and must still match the K/V-cache path and the oracle."""
    qf, cfg = build_qformer(dev, 256, seed=5, hidden=256, heads=4, inter=512, layers=2, llm_hidden=256)
    ocfg = oracle_cfg(cfg)
    w = O.init_weights(ocfg, seed=5, perturb=True)
    ids, tmask, att, feats = make_inputs(ocfg, 5, 7, 333, 2, True)
    enc = qf.modality_ln(feats.to(dev))
    qf.set_cross_mode("kv_cache")
    ref = qf.forward_fused(ids.to(dev), att.to(dev), enc, want_query=True)["query"]
    qf.set_cross_mode("fold")
    got = qf.forward_fused(ids.to(dev), att.to(dev), enc, want_query=True)["query"]
    assert (got - ref).abs().max().item() < 5e-3
    h = O.qformer_forward(w, ocfg, ids, att, w["query_tokens"].expand(5, -1, -1), O.modality_layernorm(feats, w["ln.weight"], w["ln.bias"]))
    assert (got.cpu() - h[:, :32]).abs().max().item() < Z_ATOL


def test_layer_chain_on_the_ring_tiles_and_with_the_layernorm_inside_the_launch(video, dev):
    """Round 3: at ~2 k rows the chain's GEMMs run on ``gemm_ring_kernel``'s exact-fit tiles (``chain_ring`` mask, default 7: QKV 144 x 128,
    FFN-up 192 x 128, residual projections 96 x 64); bit 3 adds the LayerNorm inside the residual projection's launch (EPI_RES_LN, opt-in).
    32 items x (32 + 32) rows = the bench's chain shape; Kv = 257 (K/V-cache path).  Every mask must reproduce the round-2 tiles (mask 0) to
    fp32 summation order and meet the oracle bars; the launch counters say the ring kernels really ran."""
    from mraudio_amd import _lib as L

    qf, cfg, w = video
    n, Lt, kv = 32, 32, 257
    g = torch.Generator().manual_seed(21)
    feats = torch.randn(n, kv, 1408, generator=g)
    ids = torch.randint(1000, 30000, (n, Lt), generator=g)
    att = torch.ones(n, 32 + Lt, dtype=torch.long)
    att[3, 32 + 20:] = 0                       # one ragged prompt
    enc = qf.modality_ln(feats.to(dev))
    torch.set_num_threads(max(1, min(32, len(os.sched_getaffinity(0)))))
    with torch.no_grad():
        ref = O.qformer_forward(w, oracle_cfg(cfg), ids, att, w["query_tokens"].expand(n, -1, -1), enc.float().cpu())
    GF_RING = (11, 12, 13)                      # GemmFamily: ring 144 x 128, 192 x 128, 96 x 64
    outs = {}
    for mask in (0, 7, 15):
        qf.set_option("chain_ring", mask)
        before = [sum(L.gemm_launches(f, e) for e in (0, 1, 2, 9)) for f in GF_RING]
        res = qf.forward_fused(ids.to(dev), att.to(dev), enc, want_full=True)
        torch.cuda.synchronize()
        after = [sum(L.gemm_launches(f, e) for e in (0, 1, 2, 9)) for f in GF_RING]
        ran = [a - b for a, b in zip(after, before)]
        # QKV and FFN-up once per layer; the 96 x 64 tile: attention-output + FFN-down per layer, with the LayerNorm inside also the 6 cross-attention outputs
        assert ran == {0: [0, 0, 0], 7: [12, 12, 24], 15: [12, 12, 30]}[mask], (mask, ran)
        if mask == 15:
            assert L.gemm_launches(13, 9) > 0                                       # EPI_RES_LN launches (all but the last layer's down-projection)
        outs[mask] = res["full"].cpu()
        assert torch.isfinite(outs[mask]).all()
        assert (outs[mask] - ref)[att.bool()].abs().max().item() < Z_ATOL, mask
    qf.set_option("chain_ring", 7)
    valid = att.bool()
    # different fp32 summation orders (the ring kernel adds the even and the odd K tiles separately), amplified by the f16 rounding of the next
    # operands: 3.0e-3 measured (r03h) on |z| <= 8, the size of the folded-vs-cache difference; the oracle bar above is what counts
    d70, d157 = (outs[7] - outs[0])[valid].abs().max().item(), (outs[15] - outs[7])[valid].abs().max().item()
    print('chain_ring 7 vs 0:', d70, ' 15 vs 7:', d157)
    assert d70 < 6e-3 and d157 < 6e-3


def test_split_precision_chain_on_both_encoder_widths(video, audio, dev):
    """``set_cross_precision("split")`` on the diffuse synthetic weights must change nothing beyond rounding (it carries MORE bits) and
    work for both encoder widths: video (E = 1408: K-major P.enc, row factors in registers) and audio (E = 768: the 128 x 128 P.enc
    fallback with an enc^T copy and the rescale pass), at the reference's item shape (the folded form is forced) and on a ragged prompt."""
    for (qf, cfg, w), kv in ((video, 257), (audio, 256)):
        ocfg = oracle_cfg(cfg)
        n, Lt = 5, 12
        g = torch.Generator().manual_seed(31)
        feats = torch.randn(n, kv, cfg.enc_width, generator=g)
        ids = torch.randint(1000, 30000, (n, Lt), generator=g)
        att = torch.ones(n, 32 + Lt, dtype=torch.long)
        att[2, 32 + 7:] = 0
        enc = qf.modality_ln(feats.to(dev))
        with torch.no_grad():
            ref = O.qformer_forward(w, ocfg, ids, att, w["query_tokens"].expand(n, -1, -1), enc.float().cpu())
        base = qf.forward_fused(ids.to(dev), att.to(dev), enc, want_full=True)["full"].cpu()
        qf.set_cross_precision("split")
        try:
            got = qf.forward_fused(ids.to(dev), att.to(dev), enc, want_full=True)["full"].cpu()
        finally:
            qf.set_cross_precision("op")
        valid = att.bool()
        assert torch.isfinite(got).all()
        e_split, e_base = (got - ref)[valid].abs().max().item(), (base - ref)[valid].abs().max().item()
        print(f"split precision E {cfg.enc_width}: max|d| vs oracle {e_split:.2e} (operand-dtype chain {e_base:.2e})")
        assert e_split < Z_ATOL and e_split < 1.5 * e_base + 1e-3
        again = qf.forward_fused(ids.to(dev), att.to(dev), enc, want_full=True)["full"].cpu()   # back on the default path, same workspace
        assert torch.equal(again, base)


def test_pair_forward_equals_the_two_separate_forwards(dev):
    """``mra_qformer_forward_pair`` (both modality Q-Formers in one launch sequence: chain GEMMs grouped over the lanes, attention cores and
    LayerNorms over 2 N items) against the two ``mra_qformer_forward`` calls it replaces -- same kernels, same summation order per lane:
    a ragged prompt, Kv below / above the fold switch, the bench's chain shape (32 items), the sharded-style cls-only last layer."""
    from mraudio_amd.models.xinstructblip import ENC_WIDTH, XInstructBLIP

    model = XInstructBLIP(seed=5, perturb=True, device=dev)
    for n, L, kvv, kva in ((5, 12, 300, 64), (3, 7, 2100, 40), (32, 32, 257, 256)):
        g = torch.Generator().manual_seed(100 + n)
        feats = {"video": torch.randn(n, kvv, ENC_WIDTH["video"], generator=g).half().to(dev), "audio": torch.randn(n, kva, ENC_WIDTH["audio"], generator=g).half().to(dev)}
        ids = torch.randint(1000, 30000, (n, L), generator=g).to(dev)
        tmask = torch.ones(n, L, dtype=torch.long, device=dev)
        tmask[n // 2, L - 3:] = 0
        outs = {}
        for pair in (False, True):
            model.pair_forward = pair
            o = model.fuse_score(feats, ids, tmask, bs=1, num=n)
            torch.cuda.synchronize()
            outs[pair] = o
        for m in ("video", "audio"):
            dz = (outs[True]["z"][m] - outs[False]["z"][m]).abs().max().item()
            dc = (outs[True]["cls"][m] - outs[False]["cls"][m]).abs().max().item()
            assert dz <= 1e-5 and dc <= 1e-5, (n, m, dz, dc)
        assert torch.equal(outs[True]["spans"], outs[False]["spans"])
        assert (outs[True]["fused"] - outs[False]["fused"]).abs().max().item() <= 1e-6
    model.pair_forward = False
