"""Clip x query cosine scorer, late fusion and integer span pick on the HIP extension (A6).

The reference has no scorer: it prompts a 7B LLM with the projected query embeddings and parses
``"[[s, e]]"`` text (``models/xinstructblip.py:346-397``, ``utils/utils.py:66-132``).  The north
star replaces that stage; the definition (and its CPU checker) is ``oracle/qformer_ref.py``:
``sim[n, q] = cos(z[n, q], t[n])``, ``logit[n] = max_q sim[n, q]``, weighted sum over modalities,
span = grow around the first argmax while the neighbour's logit >= lo + alpha * (hi - lo).
"""
from __future__ import annotations

import ctypes as C
from typing import List, Optional, Sequence, Tuple

import torch

from ._lib import MraError, check, current_stream, lib, ptr


def cosine_scores(z: torch.Tensor, t: torch.Tensor, want_sim: bool = True) -> Tuple[Optional[torch.Tensor], torch.Tensor]:
    """z [N, Q, H] fp32, t [N, H] or [1, H] fp32 -> (sim [N, Q] or None, logit [N])."""
    if z.dim() != 3 or t.dim() != 2 or t.shape[1] != z.shape[2] or t.shape[0] not in (1, z.shape[0]):
        raise MraError(f"cosine_scores: bad shapes z {tuple(z.shape)} t {tuple(t.shape)}")
    z = z.to(torch.float32).contiguous()
    t = t.to(device=z.device, dtype=torch.float32).contiguous()
    n, q, h = (int(x) for x in z.shape)
    sim = torch.empty(n, q, dtype=torch.float32, device=z.device) if want_sim else None
    logit = torch.empty(n, dtype=torch.float32, device=z.device)
    with torch.cuda.device(z.device):
        check(lib().mra_cosine_score(ptr(z), ptr(t), int(t.shape[0]), n, q, h, ptr(sim), ptr(logit), current_stream()),
              "mra_cosine_score")
    return sim, logit


def fuse_logits(per_modality: Sequence[torch.Tensor], weights: Optional[Sequence[float]] = None) -> torch.Tensor:
    """sum_m w[m] * logits[m], fp32, left to right (default w = 1/M)."""
    xs = [x.to(torch.float32).contiguous() for x in per_modality]
    if not xs or len(xs) > 4:
        raise MraError("fuse_logits takes 1..4 modalities")
    n = int(xs[0].numel())
    if any(x.numel() != n or x.device != xs[0].device for x in xs):
        raise MraError("fuse_logits: mismatched logits")
    out = torch.empty(n, dtype=torch.float32, device=xs[0].device)
    ptrs = (C.c_void_p * len(xs))(*[x.data_ptr() for x in xs])
    w = None if weights is None else (C.c_float * len(xs))(*[float(v) for v in weights])
    with torch.cuda.device(out.device):
        check(lib().mra_fuse_logits(ptrs, w, len(xs), n, ptr(out), current_stream()), "mra_fuse_logits")
    return out


def spans_from_logits(logits: torch.Tensor, videos: int, clips: int, alpha: float = 0.5) -> torch.Tensor:
    """logits [videos * clips] fp32 -> int32 [videos, 2] inclusive (start, end) clip indices."""
    x = logits.to(torch.float32).contiguous()
    if x.numel() != videos * clips:
        raise MraError(f"spans_from_logits: {x.numel()} logits for {videos} x {clips}")
    out = torch.empty(videos, 2, dtype=torch.int32, device=x.device)
    with torch.cuda.device(x.device):
        check(lib().mra_span_from_logits(ptr(x), videos, clips, float(alpha), ptr(out), current_stream()),
              "mra_span_from_logits")
    return out


def spans_to_text(spans: Sequence[Sequence[int]], timestamps: Sequence[Sequence[int]]) -> List[str]:
    """Clip spans -> the ``"[[start, end]]"`` strings (seconds) the reference's generate() returns
    and ``post_process`` / ``moment_str_to_list`` parse (``evaluate.py:48``); seconds come from
    ``samples["timestamps"]`` (``utils/mr_dataset.py:44``)."""
    return [f"[[{int(ts[int(s)])}, {int(ts[int(e)])}]]" for (s, e), ts in zip(spans, timestamps)]
