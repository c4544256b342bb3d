"""Video processors with the reference's signatures (``processors/alpro_processors.py:14-85``).

Video decoding (decord) and the LAVIS Alpro transforms are outside the hot path (SURVEY.md section 2,
row 10); what the path consumes from them is integer: the sampled frame indices and, through
``utils/mr_dataset.py:44``, the per-position timestamps.  Those are restated here exactly.  The
decoder is pluggable: ``reader(vpath, height, width) -> (frames uint8/float [T, H, W, C], fps)``.
"""
from __future__ import annotations

import random as rnd
from typing import Callable, List, Optional, Sequence, Tuple

import numpy as np
import torch


def frame_indices(vlen: int, n_frms: int, sampling: str = "uniform", rng: Optional[rnd.Random] = None):
    """Reference ``load_video`` (``:19-33``): ``n = min(n_frms, vlen)``; uniform =
    ``np.linspace(0, vlen, n, endpoint=False).astype(int)``; random = one frame per interval of
    ``np.linspace(0, vlen, n + 1).astype(int)`` (``low`` itself when the interval is empty)."""
    n = min(n_frms, vlen)
    if sampling == "uniform":
        return np.linspace(start=0, stop=vlen, num=n, endpoint=False).astype(int)
    if sampling == "random":
        r = rng or rnd
        iv = np.linspace(start=0, stop=vlen, num=n + 1).astype(int)
        return [int(lo) if lo == hi else r.choice(range(int(lo), int(hi))) for lo, hi in zip(iv[:-1], iv[1:])]
    raise NotImplementedError(f"Sampling strategy '{sampling}' is not implemented.")


def timestamps_from_indices(indices: Sequence[int], fps: float) -> List[int]:
    """``utils/mr_dataset.py:44``: ``[round(idx / fps) for idx in indices]`` (Python banker's rounding)."""
    return [round(int(i) / fps) for i in indices]


def _normalise(frames: torch.Tensor, mean, std) -> torch.Tensor:
    m = torch.tensor(mean if mean is not None else (0.48145466, 0.4578275, 0.40821073)).view(3, 1, 1, 1)
    s = torch.tensor(std if std is not None else (0.26862954, 0.26130258, 0.27577711)).view(3, 1, 1, 1)
    return (frames / 255.0 - m) / s


class _StampsProcessor:
    sampling = "uniform"

    def __init__(self, image_size=224, mean=None, std=None, n_frms=60, full_video=True, reader: Optional[Callable] = None):
        self.image_size, self.mean, self.std, self.n_frms, self.full_video = image_size, mean, std, n_frms, full_video
        self.reader = reader

    def __call__(self, vpath) -> Tuple[torch.Tensor, object, float]:
        if self.reader is None:
            raise RuntimeError("no video reader configured (decord is not part of this build); pass reader=")
        frames, fps = self.reader(vpath, self.image_size, self.image_size)
        vlen = int(frames.shape[0])
        indices = frame_indices(vlen, self.n_frms, self.sampling)
        clip = torch.as_tensor(np.asarray(frames)[np.asarray(indices)]).permute(3, 0, 1, 2).float()  # (C, T, H, W)
        transformed = _normalise(clip, self.mean, self.std)
        pad = self.n_frms - transformed.shape[1]
        if pad > 0:  # repeat the last frame (reference :79-83)
            transformed = torch.cat([transformed, transformed[:, -1:].repeat(1, pad, 1, 1)], dim=1)
        return transformed, indices, fps


class AlproVideoEvalProcessor_Stamps(_StampsProcessor):
    """Reference ``:64-85``: uniform sampling, returns ``(frames [C,T,H,W] float32, indices, fps)``."""
    sampling = "uniform"


class AlproVideoTrainProcessor_Stamps(_StampsProcessor):
    """Reference ``:40-62``: one random frame per interval."""
    sampling = "random"

    def __init__(self, image_size=224, mean=None, std=None, min_scale=0.9, max_scale=1.0, n_frms=60, full_video=True, reader=None):
        super().__init__(image_size=image_size, mean=mean, std=std, n_frms=n_frms, full_video=full_video, reader=reader)
        self.min_scale, self.max_scale = min_scale, max_scale
