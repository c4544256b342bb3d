"""Precision study (CPU, test infrastructure; not collected by pytest): which f16 rounding points of the cross-attention chain bound the
similarity-logit error under PEAKED attention?

The HIP pipeline's rounding points are emulated on the oracle's arithmetic (fp32 matmuls = fp32 accumulation; ``r16`` = round to
f16 and back) and switched off one group at a time.  ``python tests/study_peaked_precision.py`` prints, per (Kv, gain) case, the errors of

    base        every rounding point of the shipped f16 path (folded form)
    +q'         Q' = Q_h W_k,h kept as f16 hi + lo (the scores product runs two MFMA passes against the same enc slab)
    +q          additionally the cross-query projection output Q kept hi + lo
    +h          additionally the hidden state entering the cross-query projection hi + lo
    +w          additionally W_cq and W_k as hi + lo (every operand of the score chain carries ~22 bits)
    +p          additionally P~ hi + lo into P.enc

against the fp32 oracle.  The numbers decide what the "precise" cross mode has to carry.
"""
import math
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import qformer_ref as O  # noqa: E402


def r16(x):
    return x.half().float()


def hl(x):
    """f16 hi + lo pair: ~22 significant bits."""
    hi = r16(x)
    return hi + r16(x - hi)


def emulated_forward(w, cfg, ids, att, query, enc, prec=()):
    """qformer_forward with the shipped kernels' rounding points; ``prec`` = set of points kept at hi+lo precision."""
    prec = set(prec)
    W = {k: v for k, v in w.items() if k.startswith("bert.")}
    n, L = ids.shape
    Q, H, heads = query.shape[1], cfg.hidden, cfg.heads
    eps = cfg.ln_eps
    wr = lambda k: r16(W[k])
    text = W["bert.embeddings.word_embeddings.weight"][ids] + W["bert.embeddings.position_embeddings.weight"][torch.arange(L)]
    h = O._ln(torch.cat([query, text], 1), W["bert.embeddings.LayerNorm.weight"], W["bert.embeddings.LayerNorm.bias"], eps)
    self_mask = ((1.0 - att.float()) * -10000.0)[:, None, None, :]
    LOG2E = 1.4426950408889634

    def attn_core(q, k, v, mask):   # q, k, v already f16-rounded [n, heads, s, 64]; fp32 statistics, P f16 into PV
        s = q @ k.transpose(-1, -2) / 8.0
        if mask is not None:
            s = s + mask
        p = torch.softmax(s, -1)
        return r16(r16(p) @ v)

    for i in range(cfg.layers):
        p_ = f"bert.encoder.layer.{i}."
        h16 = r16(h)
        qkv = [r16(h16 @ wr(p_ + f"attention.self.{nm}.weight").T + W[p_ + f"attention.self.{nm}.bias"]).view(n, -1, heads, 64).permute(0, 2, 1, 3)
               for nm in ("query", "key", "value")]
        a = attn_core(qkv[0], qkv[1], qkv[2], self_mask).permute(0, 2, 1, 3).reshape(n, -1, H)
        a = a @ wr(p_ + "attention.output.dense.weight").T + W[p_ + "attention.output.dense.bias"]
        h1 = O._ln(a + h, W[p_ + "attention.output.LayerNorm.weight"], W[p_ + "attention.output.LayerNorm.bias"], eps)
        hq = h1[:, :Q]
        if i % cfg.cross_freq == 0:
            hin = hl(hq) if "h" in prec else r16(hq)
            wq = hl(W[p_ + "crossattention.self.query.weight"]) if "w" in prec else wr(p_ + "crossattention.self.query.weight")
            wk = hl(W[p_ + "crossattention.self.key.weight"]) if "w" in prec else wr(p_ + "crossattention.self.key.weight")
            qc = hin @ wq.T + W[p_ + "crossattention.self.query.bias"]
            qc = hl(qc) if "q" in prec else r16(qc)
            qh = qc.view(n, Q, heads, 64).permute(0, 2, 1, 3)                       # [n, heads, 32, 64]
            wkh = wk.view(heads, 64, -1)                                             # [heads, 64, E]
            qp = torch.einsum("nhqd,hde->nhqe", qh, wkh)                             # Q' [n, heads, 32, E]
            qp = hl(qp) if "q'" in prec else r16(qp)
            s = torch.einsum("nhqe,nke->nhqk", qp, enc) * (0.125 * LOG2E)            # log2 units, fp32
            # split softmax over 176-column tiles: P~ = f16(exp2(s - m_tile)), factors f16(exp2(m_tile - m_row) / L)
            kv = s.shape[-1]
            nt = (kv + 175) // 176
            pad = nt * 176 - kv
            sp = torch.nn.functional.pad(s, (0, pad), value=-3e38).view(n, heads, Q, nt, 176)
            mt = sp.max(-1, keepdim=True).values
            pt = torch.exp2(sp - mt)
            pt = hl(pt) if "p" in prec else r16(pt)
            lt = pt.sum(-1, keepdim=True)
            mrow = mt.max(-2, keepdim=True).values
            fac = torch.exp2(mt - mrow)
            Lrow = (fac * lt).sum(-2, keepdim=True)
            g = fac / Lrow
            g = g if "p" in prec else r16(g)
            pt = pt * g
            pt = hl(pt) if "p" in prec else r16(pt)                                  # v_pk_mul_f16 on the fragment
            pn = pt.view(n, heads, Q, nt * 176)[..., :kv]
            u = r16(torch.einsum("nhqk,nke->nhqe", pn, enc))                          # U [n, heads, 32, E]
            wvh = wr(p_ + "crossattention.self.value.weight").view(heads, 64, -1)
            ctx = torch.einsum("nhqe,hde->nhqd", u, wvh) + W[p_ + "crossattention.self.value.bias"].view(heads, 1, 64)[None]
            ctx = r16(ctx).permute(0, 2, 1, 3).reshape(n, Q, H)
            c = ctx @ wr(p_ + "crossattention.output.dense.weight").T + W[p_ + "crossattention.output.dense.bias"]
            hq = O._ln(c + hq, W[p_ + "crossattention.output.LayerNorm.weight"], W[p_ + "crossattention.output.LayerNorm.bias"], eps)
        fq = r16(O._gelu_erf(r16(hq) @ wr(p_ + "intermediate_query.dense.weight").T + W[p_ + "intermediate_query.dense.bias"]))
        fq = fq @ wr(p_ + "output_query.dense.weight").T + W[p_ + "output_query.dense.bias"]
        oq = O._ln(fq + hq, W[p_ + "output_query.LayerNorm.weight"], W[p_ + "output_query.LayerNorm.bias"], eps)
        ht = h1[:, Q:]
        ft = r16(O._gelu_erf(r16(ht) @ wr(p_ + "intermediate.dense.weight").T + W[p_ + "intermediate.dense.bias"]))
        ft = ft @ wr(p_ + "output.dense.weight").T + W[p_ + "output.dense.bias"]
        ot = O._ln(ft + ht, W[p_ + "output.LayerNorm.weight"], W[p_ + "output.LayerNorm.bias"], eps)
        h = torch.cat([oq, ot], 1)
    return h


def peaked_weights(cfg, seed, gain):
    w = O.init_weights(cfg, seed=seed, perturb=True)
    for i in cfg.cross_layers():
        for nme in ("query", "key"):
            for part in ("weight", "bias"):
                k = f"bert.encoder.layer.{i}.crossattention.self.{nme}.{part}"
                w[k] = w[k] * gain
    return w


def main():
    torch.set_num_threads(8)
    cfg = O.QFormerCfg(enc_width=1408)
    cases = [(300, 4.0), (300, 5.0), (2100, 4.0), (2100, 5.0), (2100, 7.0)]
    ladders = [("base", ()), ("+q'", ("q'",)), ("+q", ("q'", "q")), ("+h", ("q'", "q", "h")), ("+w", ("q'", "q", "h", "w")), ("+p", ("q'", "q", "h", "w", "p"))]
    for kv, gain in cases:
        w = peaked_weights(cfg, 3, gain)
        n, L = 3, 8
        g = torch.Generator().manual_seed(11)
        feats = torch.randn(n, kv, 1408, generator=g)
        ids = torch.randint(1000, 30000, (n, L), generator=g)
        att = torch.ones(n, 32 + L, dtype=torch.long)
        enc = r16(O.modality_layernorm(feats, w["ln.weight"], w["ln.bias"]))   # the f16 operand the kernels read
        q = w["query_tokens"].expand(n, -1, -1)
        with torch.no_grad():
            href = O.qformer_forward(w, cfg, ids, att, q, enc)
            sim_ref, logit_ref = O.cosine_scores(href[:, :32], href[:, 32])
            for name, prec in ladders:
                h = emulated_forward(w, cfg, ids, att, q, enc, prec)
                sim, logit = O.cosine_scores(h[:, :32], h[:, 32])
                dz = (h[:, :32] - href[:, :32]).abs().max().item()
                rel = ((h[:, :32] - href[:, :32]).norm() / href[:, :32].norm()).item()
                dl = (logit - logit_ref).abs().max().item() / logit_ref.abs().max().item()
                ds = (sim - sim_ref).abs().max().item() / sim_ref.abs().max().item()
                print(f"kv {kv:5d} gain {gain:3.0f} {name:5s} dz {dz:.2e} rel {rel:.2e} dlogit {dl:.2e} dsim {ds:.2e}", flush=True)


if __name__ == "__main__":
    main()
