"""CPU oracle for the cross-modal encode/fuse/score hot path.  TEST INFRASTRUCTURE ONLY.

Only ``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of ``bench.py`` may
import this module, and only as the checker.  The product path (``mraudio_amd``) never imports it
and raises when the HIP extension is missing.

What it restates (plain ``torch`` fp32/fp64 on the CPU, no HIP, no fused ops):

* A2  modality LayerNorm          reference ``models/xinstructblip.py:822-828`` (fp32 math, eps 1e-5)
* A3  sample-major reorder        reference ``models/xinstructblip.py:281-285`` / ``:456-460``
* A4  Q-Former forward            reference call site ``models/xinstructblip.py:286-293``; the layer
      arithmetic lives in un-vendored, un-pinned LAVIS (``lavis.models.blip2_models.Qformer``,
      installed from git HEAD, reference ``README.md:5``).  The published algorithm is restated here
      from the BLIP-2/InstructBLIP Q-Former definition (BERT-base, cross-attention every 2nd layer,
      separate query/text feed-forward) and cross-checked against the HF port that ships in this
      image (``transformers`` ``InstructBlipQFormerModel``, ``modeling_instructblip.py:446-857``).
* A5  slice + LLM projection      reference ``models/xinstructblip.py:303-306``
* A6  cosine scorer + span        NOT IN THE REFERENCE (the reference decodes spans with a 7B LLM);
      defined by this build, see ``cosine_scores`` / ``span_from_logits``.

Pinning status: the reference ships no tests, fixtures or golden vectors for this path
 and its own model code cannot be imported (LAVIS absent).  A4 is pinned
against the HF port on shared seeded weights (live: ``tests/test_oracle.py``; committed vectors: ``tests/golden/qformer_*.npz``
made by ``tools/make_golden.py``);
A6 has no external pin at all: "parity unpinned" for the scorer, by construction.
"""
from __future__ import annotations

import math
from dataclasses import dataclass, field
from typing import Dict, List, Optional, Sequence, Tuple

import torch

Tensor = torch.Tensor


# --------------------------------------------------------------------------------------------
# configuration + seeded weights
# --------------------------------------------------------------------------------------------
@dataclass
class QFormerCfg:
    """Shape of one modality Q-Former (reference ``models/xinstructblip.py:614-627``)."""

    hidden: int = 768
    heads: int = 12
    inter: int = 3072
    layers: int = 12
    cross_freq: int = 2          # cross-attention on layers i % cross_freq == 0
    enc_width: int = 1408        # 1408 EVA ViT-g (video), 768 BEATs (audio)
    n_query: int = 32
    vocab: int = 30523           # bert-base-uncased 30522 + [DEC]
    max_pos: int = 512
    ln_eps: float = 1e-12
    llm_hidden: int = 4096

    @property
    def head_dim(self) -> int:
        return self.hidden // self.heads

    def cross_layers(self) -> List[int]:
        return [i for i in range(self.layers) if i % self.cross_freq == 0]


def weight_names(cfg: QFormerCfg) -> List[Tuple[str, Tuple[int, ...], str]]:
    """(LAVIS-style key, shape, kind) in the fixed order the seeded recipe draws them.

    Keys follow the checkpoint naming the reference's loader routes
    (``models/xinstructblip.py:644-653,778-812``): ``bert.embeddings...``, ``bert.encoder.layer.{i}...``.
    kind: 'w' weight matrix / embedding, 'b' bias, 'g' LayerNorm gain, 'z' LayerNorm bias.
    """
    H, I, E = cfg.hidden, cfg.inter, cfg.enc_width
    out: List[Tuple[str, Tuple[int, ...], str]] = [
        ("bert.embeddings.word_embeddings.weight", (cfg.vocab, H), "w"),
        ("bert.embeddings.position_embeddings.weight", (cfg.max_pos, H), "w"),
        ("bert.embeddings.LayerNorm.weight", (H,), "g"),
        ("bert.embeddings.LayerNorm.bias", (H,), "z"),
    ]
    for i in range(cfg.layers):
        p = f"bert.encoder.layer.{i}."
        for n in ("query", "key", "value"):
            out += [(p + f"attention.self.{n}.weight", (H, H), "w"), (p + f"attention.self.{n}.bias", (H,), "b")]
        out += [
            (p + "attention.output.dense.weight", (H, H), "w"),
            (p + "attention.output.dense.bias", (H,), "b"),
            (p + "attention.output.LayerNorm.weight", (H,), "g"),
            (p + "attention.output.LayerNorm.bias", (H,), "z"),
        ]
        if i % cfg.cross_freq == 0:
            out += [
                (p + "crossattention.self.query.weight", (H, H), "w"),
                (p + "crossattention.self.query.bias", (H,), "b"),
                (p + "crossattention.self.key.weight", (H, E), "w"),
                (p + "crossattention.self.key.bias", (H,), "b"),
                (p + "crossattention.self.value.weight", (H, E), "w"),
                (p + "crossattention.self.value.bias", (H,), "b"),
                (p + "crossattention.output.dense.weight", (H, H), "w"),
                (p + "crossattention.output.dense.bias", (H,), "b"),
                (p + "crossattention.output.LayerNorm.weight", (H,), "g"),
                (p + "crossattention.output.LayerNorm.bias", (H,), "z"),
            ]
        for suf in ("", "_query"):
            out += [
                (p + f"intermediate{suf}.dense.weight", (I, H), "w"),
                (p + f"intermediate{suf}.dense.bias", (I,), "b"),
                (p + f"output{suf}.dense.weight", (H, I), "w"),
                (p + f"output{suf}.dense.bias", (H,), "b"),
                (p + f"output{suf}.LayerNorm.weight", (H,), "g"),
                (p + f"output{suf}.LayerNorm.bias", (H,), "z"),
            ]
    return out


def init_weights(cfg: QFormerCfg, seed: int = 0, perturb: bool = False) -> Dict[str, Tensor]:
    """Seeded random weights (there are no pretrained weights in this container).

    Recipe (documented so the GPU box can re-derive the tensors the golden vectors were made with):
    one ``torch.Generator`` seeded with ``seed``; tensors are drawn on the CPU in ``weight_names``
    order, then the extras below.  ``perturb=False`` is the BERT init the reference's config implies
    (``initializer_range`` 0.02, reference ``models/xinstructblip.py:627``): matrices N(0, 0.02),
    biases 0, LayerNorm gain 1 / bias 0.  ``perturb=True`` additionally draws biases N(0, 0.02),
    gains 1 + N(0, 0.1) and LayerNorm biases N(0, 0.05) so that every parameter influences the
    output (a zero bias cannot catch a dropped bias add).
    """
    g = torch.Generator().manual_seed(seed)
    w: Dict[str, Tensor] = {}

    def draw(shape, kind):
        if kind == "w":
            return torch.randn(shape, generator=g, dtype=torch.float32) * 0.02
        if kind == "b":
            return torch.randn(shape, generator=g) * 0.02 if perturb else torch.zeros(shape)
        if kind == "g":
            return 1.0 + torch.randn(shape, generator=g) * 0.1 if perturb else torch.ones(shape)
        return torch.randn(shape, generator=g) * 0.05 if perturb else torch.zeros(shape)

    for name, shape, kind in weight_names(cfg):
        w[name] = draw(shape, kind)
    # extras outside the Q-Former proper (reference ``models/xinstructblip.py:624-627,678-735``)
    w["query_tokens"] = torch.randn((1, cfg.n_query, cfg.hidden), generator=g) * 0.02
    w["ln.weight"] = draw((cfg.enc_width,), "g")
    w["ln.bias"] = draw((cfg.enc_width,), "z")
    w["llm_proj.weight"] = draw((cfg.llm_hidden, cfg.hidden), "w")
    w["llm_proj.bias"] = draw((cfg.llm_hidden,), "b")
    return w


def to_hf_state_dict(w: Dict[str, Tensor]) -> Dict[str, Tensor]:
    """Rename LAVIS keys to the HF port's (``attention.self.*`` -> ``attention.attention.*``,
    ``embeddings.LayerNorm`` -> ``embeddings.layernorm``, no ``bert.`` prefix)."""
    out = {}
    for k, v in w.items():
        if not k.startswith("bert."):
            continue
        k2 = k[len("bert."):]
        k2 = k2.replace("attention.self.", "attention.attention.")
        k2 = k2.replace("embeddings.LayerNorm", "embeddings.layernorm")
        out[k2] = v
    return out


# --------------------------------------------------------------------------------------------
# A2 / A3
# --------------------------------------------------------------------------------------------
def modality_layernorm(x: Tensor, weight: Tensor, bias: Tensor, eps: float = 1e-5) -> Tensor:
    """``LayerNorm.forward`` of the reference (``models/xinstructblip.py:822-828``): statistics and
    affine in fp32 over the last axis, biased variance, result cast back to the input dtype."""
    xf = x.to(torch.float32)
    mu = xf.mean(dim=-1, keepdim=True)
    var = ((xf - mu) ** 2).mean(dim=-1, keepdim=True)
    y = (xf - mu) / torch.sqrt(var + eps) * weight.to(torch.float32) + bias.to(torch.float32)
    return y.to(x.dtype)


def reorder_indices(bs: int, num: int) -> List[int]:
    """Frame-major -> sample-major gather of the reference (``models/xinstructblip.py:283``):
    ``cat`` of ``num`` per-position ``[bs, ...]`` blocks is indexed so that row ``r*num + i`` holds
    sample ``r``, position ``i``."""
    return [i * bs + r for r in range(bs) for i in range(num)]


def repeat_text_rows(bs: int, num: int) -> List[int]:
    """Which prompt each Q-Former row gets: the reference tiles text with ``.repeat(num, 1)``
    (``models/xinstructblip.py:287-288``), i.e. row ``k`` carries prompt ``k % bs`` although its
    features belong to sample ``k // num`` (identical when bs == 1)."""
    return [k % bs for k in range(bs * num)]


# --------------------------------------------------------------------------------------------
# A4  Q-Former
# --------------------------------------------------------------------------------------------
def _ln(x: Tensor, g: Tensor, b: Tensor, eps: float) -> Tensor:
    mu = x.mean(dim=-1, keepdim=True)
    var = ((x - mu) ** 2).mean(dim=-1, keepdim=True)
    return (x - mu) / torch.sqrt(var + eps) * g + b


def _gelu_erf(x: Tensor) -> Tensor:
    return 0.5 * x * (1.0 + torch.erf(x / math.sqrt(2.0)))


def _mha(xq: Tensor, xkv: Tensor, wq, bq, wk, bk, wv, bv, heads: int, add_mask: Optional[Tensor]) -> Tensor:
    """softmax(Q K^T / sqrt(d) + mask) V, heads split on the channel axis (HF :446-515)."""
    n, sq, h = xq.shape
    skv = xkv.shape[1]
    d = h // heads
    q = (xq @ wq.T + bq).view(n, sq, heads, d).permute(0, 2, 1, 3)
    k = (xkv @ wk.T + bk).view(n, skv, heads, d).permute(0, 2, 1, 3)
    v = (xkv @ wv.T + bv).view(n, skv, heads, d).permute(0, 2, 1, 3)
    s = (q @ k.transpose(-1, -2)) / math.sqrt(d)
    if add_mask is not None:
        s = s + add_mask
    p = torch.softmax(s, dim=-1)
    return (p @ v).permute(0, 2, 1, 3).reshape(n, sq, h)


def qformer_forward(
    w: Dict[str, Tensor],
    cfg: QFormerCfg,
    input_ids: Tensor,
    attention_mask: Tensor,
    query_embeds: Tensor,
    encoder_hidden_states: Tensor,
    encoder_attention_mask: Optional[Tensor] = None,
    dtype: torch.dtype = torch.float32,
    collect: Optional[Dict[str, Tensor]] = None,
) -> Tensor:
    """``{modality}_Qformer.bert(...)`` of the reference (``models/xinstructblip.py:286-293``).

    input_ids [N, L] int64, attention_mask [N, n_query + L] (1 = attend), query_embeds [N, 32, H],
    encoder_hidden_states [N, Kv, E], encoder_attention_mask [N, Kv] or None (reference: all ones).
    Returns last_hidden_state [N, 32 + L, H].  Dropout is inactive (eval).  Padding uses the LAVIS
    additive form ``(1 - mask) * -10000`` (HF uses ``finfo.min``; equal after softmax unless a whole
    row is masked, which cannot happen because the 32 queries are never masked).
    ``collect`` (optional dict) receives per-layer intermediates for kernel bring-up.
    """
    W = {k: v.to(dtype) for k, v in w.items() if k.startswith("bert.")}
    n, L = input_ids.shape
    Q = query_embeds.shape[1]
    eps = cfg.ln_eps
    # embeddings: word + absolute position for the text, none for the queries (HF :728-757)
    pos = torch.arange(L)
    text = W["bert.embeddings.word_embeddings.weight"][input_ids] + W["bert.embeddings.position_embeddings.weight"][pos]
    h = torch.cat([query_embeds.to(dtype), text], dim=1)
    h = _ln(h, W["bert.embeddings.LayerNorm.weight"], W["bert.embeddings.LayerNorm.bias"], eps)
    if collect is not None:
        collect["emb"] = h.clone()
    self_mask = ((1.0 - attention_mask.to(dtype)) * -10000.0)[:, None, None, :]
    cross_mask = None
    if encoder_attention_mask is not None:
        cross_mask = ((1.0 - encoder_attention_mask.to(dtype)) * -10000.0)[:, None, None, :]
    enc = encoder_hidden_states.to(dtype)
    for i in range(cfg.layers):
        p = f"bert.encoder.layer.{i}."
        a = _mha(h, h,
                 W[p + "attention.self.query.weight"], W[p + "attention.self.query.bias"],
                 W[p + "attention.self.key.weight"], W[p + "attention.self.key.bias"],
                 W[p + "attention.self.value.weight"], W[p + "attention.self.value.bias"],
                 cfg.heads, self_mask)
        a = a @ W[p + "attention.output.dense.weight"].T + W[p + "attention.output.dense.bias"]
        h1 = _ln(a + h, W[p + "attention.output.LayerNorm.weight"], W[p + "attention.output.LayerNorm.bias"], eps)
        hq = h1[:, :Q]
        if i % cfg.cross_freq == 0:
            c = _mha(hq, enc,
                     W[p + "crossattention.self.query.weight"], W[p + "crossattention.self.query.bias"],
                     W[p + "crossattention.self.key.weight"], W[p + "crossattention.self.key.bias"],
                     W[p + "crossattention.self.value.weight"], W[p + "crossattention.self.value.bias"],
                     cfg.heads, cross_mask)
            c = c @ W[p + "crossattention.output.dense.weight"].T + W[p + "crossattention.output.dense.bias"]
            hq = _ln(c + hq, W[p + "crossattention.output.LayerNorm.weight"], W[p + "crossattention.output.LayerNorm.bias"], eps)
        # query feed-forward (HF :640-645) and text feed-forward with its own weights (HF :647-654)
        fq = _gelu_erf(hq @ W[p + "intermediate_query.dense.weight"].T + W[p + "intermediate_query.dense.bias"])
        fq = fq @ W[p + "output_query.dense.weight"].T + W[p + "output_query.dense.bias"]
        oq = _ln(fq + hq, W[p + "output_query.LayerNorm.weight"], W[p + "output_query.LayerNorm.bias"], eps)
        if L > 0:
            ht = h1[:, Q:]
            ft = _gelu_erf(ht @ W[p + "intermediate.dense.weight"].T + W[p + "intermediate.dense.bias"])
            ft = ft @ W[p + "output.dense.weight"].T + W[p + "output.dense.bias"]
            ot = _ln(ft + ht, W[p + "output.LayerNorm.weight"], W[p + "output.LayerNorm.bias"], eps)
            h = torch.cat([oq, ot], dim=1)
        else:
            h = oq
        if collect is not None:
            collect[f"layer{i}.attn"] = h1.clone()
            collect[f"layer{i}.out"] = h.clone()
    return h


# --------------------------------------------------------------------------------------------
# A5  slice + projection
# --------------------------------------------------------------------------------------------
def llm_project(last_hidden_state: Tensor, w: Dict[str, Tensor], bs: int, num: int, n_query: int = 32) -> Tensor:
    """``llm_proj(last_hidden_state[:, :32])`` reshaped ``[bs, num*32, llm_hidden]``
    (reference ``models/xinstructblip.py:303-305``)."""
    z = last_hidden_state[:, :n_query, :]
    y = z @ w["llm_proj.weight"].T.to(z.dtype) + w["llm_proj.bias"].to(z.dtype)
    return y.reshape(bs, num, n_query, -1).reshape(bs, num * n_query, -1)


# --------------------------------------------------------------------------------------------
# A6  scorer (build-defined; no reference counterpart)
# --------------------------------------------------------------------------------------------
def cosine_scores(z: Tensor, t: Tensor, eps: float = 1e-8) -> Tuple[Tensor, Tensor]:
    """clip x query cosine similarity.

    z [N, Q, H] query embeddings (``last_hidden_state[:, :32]``), t [N, H] or [1, H] text vector
    (``last_hidden_state[:, 32]``, the prompt's [CLS] output of the same forward).
    sim[n, q] = <z_nq, t_n> / (max(|z_nq|, eps) * max(|t_n|, eps));  logit[n] = max_q sim[n, q]
    (the BLIP-2 image-text-contrast reduction over queries).  Returns (sim [N, Q], logit [N]).
    """
    zf = z.to(torch.float32)
    tf = t.to(torch.float32).expand(z.shape[0], -1)
    dot = (zf * tf[:, None, :]).sum(-1)
    zn = zf.pow(2).sum(-1).sqrt().clamp_min(eps)
    tn = tf.pow(2).sum(-1).sqrt().clamp_min(eps)
    sim = dot / (zn * tn[:, None])
    return sim, sim.max(dim=1).values


def fuse_logits(per_modality: Sequence[Tensor], weights: Optional[Sequence[float]] = None) -> Tensor:
    """Late fusion of per-modality clip logits: weighted sum in fp32, modality order as given
    (default: equal weights 1/M).  Sequential left-to-right accumulation, so a kernel can match
    it bit for bit."""
    m = len(per_modality)
    ws = [1.0 / m] * m if weights is None else list(weights)
    acc = torch.zeros_like(per_modality[0], dtype=torch.float32)
    for x, wt in zip(per_modality, ws):
        acc = acc + x.to(torch.float32) * torch.tensor(wt, dtype=torch.float32)
    return acc


def span_from_logits(logits: Tensor, alpha: float = 0.5) -> Tuple[int, int]:
    """Integer clip span of one video from its T clip logits (fp32).

    peak = first argmax; thr = lo + alpha * (hi - lo) computed in fp32 with one fused step
    ``thr = fma(alpha, hi - lo, lo)`` restated as ``lo + alpha * (hi - lo)`` (two roundings; the
    kernel does the same two roundings); the span grows left and right from the peak while the
    neighbour's logit is >= thr.  Returns inclusive clip indices (start, end)."""
    x = logits.to(torch.float32).contiguous()
    T = x.numel()
    hi = x.max()
    lo = x.min()
    peak = int(torch.nonzero(x == hi)[0, 0])
    thr = lo + torch.tensor(alpha, dtype=torch.float32) * (hi - lo)
    s = peak
    while s - 1 >= 0 and bool(x[s - 1] >= thr):
        s -= 1
    e = peak
    while e + 1 < T and bool(x[e + 1] >= thr):
        e += 1
    return s, e


def spans_to_text(spans: Sequence[Tuple[int, int]], timestamps: Sequence[Sequence[int]]) -> List[str]:
    """Span indices -> the ``"[[start, end]]"`` strings the reference's LLM is trained to emit and
    ``utils/utils.py:66-132,364-415`` parse; seconds come from ``samples["timestamps"]``
    (``utils/mr_dataset.py:44``)."""
    out = []
    for (s, e), ts in zip(spans, timestamps):
        out.append(f"[[{int(ts[s])}, {int(ts[e])}]]")
    return out


# --------------------------------------------------------------------------------------------
# whole path, one modality / both
# --------------------------------------------------------------------------------------------
def encode_fuse_score(
    weights: Dict[str, Dict[str, Tensor]],
    cfgs: Dict[str, QFormerCfg],
    feats: Dict[str, Tensor],
    input_ids: Tensor,
    text_mask: Tensor,
    bs: int,
    num: int,
    dtype: torch.dtype = torch.float32,
    alpha: float = 0.5,
):
    """LN -> Q-Former -> slice -> cosine score -> fuse -> span, per the reference orchestration
    (``models/xinstructblip.py:255-306``) with features already sample-major ``[bs*num, Kv, E]``
    and ``input_ids`` / ``text_mask`` already expanded to one row per item ``[bs*num, L]``.

    Returns dict(z={m: [N,32,H]}, sim={m: [N,32]}, logit={m: [N]}, fused [N], spans [(s,e)]*bs)."""
    out = {"z": {}, "sim": {}, "logit": {}, "cls": {}}
    n = bs * num
    for m, x in feats.items():
        cfg, w = cfgs[m], weights[m]
        enc = modality_layernorm(x.to(torch.float32), w["ln.weight"], w["ln.bias"])
        att = torch.cat([torch.ones(n, cfg.n_query, dtype=text_mask.dtype), text_mask], dim=1)
        q = w["query_tokens"].expand(n, -1, -1)
        h = qformer_forward(w, cfg, input_ids, att, q, enc, None, dtype=dtype)
        z = h[:, : cfg.n_query].to(torch.float32)
        t = h[:, cfg.n_query].to(torch.float32)
        sim, logit = cosine_scores(z, t)
        out["z"][m], out["sim"][m], out["logit"][m], out["cls"][m] = z, sim, logit, t
    fused = fuse_logits([out["logit"][m] for m in feats])
    out["fused"] = fused
    out["spans"] = [span_from_logits(fused[r * num:(r + 1) * num], alpha) for r in range(bs)]
    return out
