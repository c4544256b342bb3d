"""Moment-retrieval dataset with the reference's record layout (``utils/mr_dataset.py:7-119``).

A record is what ``XInstructBLIP.generate`` / ``forward`` read: ``text_input`` (the two-line prompt of
``:96-98``), ``text_output`` (``str(relevant_windows)``), ``video`` [C, T, H, W], ``audio`` [T, F, 128],
``timestamps`` (``round(index / fps)`` per sampled frame, ``:44``), ``duration``, ``qid``, ``query``, ``vid``.

Decoding is not part of the hot path and neither ffmpeg nor decord exists in this image, so the decoders
are the pluggable processors of ``mraudio_amd.processors``; clipping by ``start`` / ``end`` (the reference
shells out to ffmpeg, ``:25-35``) is handed to the processor as a keyword when it accepts one.  For a
corpus whose encoder outputs were extracted once, ``embeds_root`` replaces both processors: the record then
carries ``video_embeds`` [T, 257, 1408] / ``audio_embeds`` [T, Kv, 768] (what BASELINE configs 1-5 feed).
"""
from __future__ import annotations

import inspect
import json
import os
from typing import Callable, Dict, List, Optional

import torch
from torch.utils.data import Dataset

QUERY_PREFIX = "Query: "
TASK_PROMPT = "Given the video and the query, find the relevant windows.\nRelevant windows: "


def build_prompt(query: str) -> str:
    """``utils/mr_dataset.py:96-98``."""
    return QUERY_PREFIX + query + "\n" + TASK_PROMPT


class MRDataset(Dataset):
    def __init__(self, vis_root: str, ann_path: str, video_processor: Optional[Callable], audio_processor: Optional[Callable],
                 model: str = "X-InstructBLIP", embeds_root: Optional[str] = None):
        self.vis_root, self.video_processor, self.audio_processor, self.model = vis_root, video_processor, audio_processor, model
        self.embeds_root = embeds_root
        with open(ann_path, "r") as f:
            self.annotation = [json.loads(line) for line in f if line.strip()]

    def __len__(self) -> int:
        return len(self.annotation)

    @staticmethod
    def _call(proc: Callable, path: str, ann: dict):
        if "start" in ann and "clip" in inspect.signature(proc.__call__ if not inspect.isfunction(proc) else proc).parameters:
            return proc(path, clip=(float(ann["start"]), float(ann["end"])))
        return proc(path)

    def __getitem__(self, index: int) -> Dict[str, object]:
        ann = self.annotation[index]
        rec: Dict[str, object] = {"text_input": build_prompt(ann["query"]), "text_output": str(ann["relevant_windows"]),
                                  "duration": ann["duration"], "qid": ann["qid"], "query": ann["query"], "vid": ann["vid"]}
        if self.embeds_root is not None:
            blob = torch.load(os.path.join(self.embeds_root, ann["vid"] + ".pt"), map_location="cpu", weights_only=True)
            for k in ("video_embeds", "audio_embeds"):
                if k in blob:
                    rec[k] = blob[k]
            rec["timestamps"] = [int(t) for t in blob["timestamps"]]
            return rec
        path = os.path.join(self.vis_root, ann["vid"] + ".mp4")
        if self.audio_processor is not None:
            rec["audio"] = self._call(self.audio_processor, path, ann)
        video, indices, fps = self._call(self.video_processor, path, ann)
        rec["video"] = video
        rec["timestamps"] = [round(idx / fps) for idx in indices]
        return rec


class SyntheticMRDataset(Dataset):
    """Seeded stand-in corpus of pre-extracted features (no dataset exists offline): ``n`` videos of ``T``
    positions, one target window each; the target positions' features carry a common direction so a trained
    scorer has something to find."""

    def __init__(self, n: int = 8, T: int = 20, seed: int = 0, duration: int = 40, kv_video: int = 257, kv_audio: int = 256,
                 modalities=("video", "audio"), signal: float = 0.0):
        self.n, self.T, self.seed, self.duration = n, T, seed, duration
        self.kv = {"video": (kv_video, 1408), "audio": (kv_audio, 768)}
        self.modalities, self.signal = tuple(modalities), signal

    def __len__(self) -> int:
        return self.n

    def __getitem__(self, i: int) -> Dict[str, object]:
        g = torch.Generator().manual_seed(self.seed * 100003 + i)
        step = self.duration / self.T
        s = int(torch.randint(0, self.T - 2, (1,), generator=g))
        e = min(self.T - 1, s + 1 + int(torch.randint(1, max(2, self.T // 3), (1,), generator=g)))
        ts = [round(k * step) for k in range(self.T)]
        rec: Dict[str, object] = {"text_input": build_prompt(f"synthetic event number {i}"), "text_output": str([[ts[s], ts[e]]]),
                                  "timestamps": ts, "duration": self.duration, "qid": i, "query": f"synthetic event number {i}", "vid": f"syn{i}"}
        for m in self.modalities:
            kv, width = self.kv[m]
            x = torch.randn(self.T, kv, width, generator=g)
            if self.signal:
                x[s:e + 1] += self.signal * torch.randn(1, 1, width, generator=torch.Generator().manual_seed(self.seed + 7))
            rec[f"{m}_embeds"] = x
        return rec


def collate_fn(batch: List[dict]) -> Dict[str, object]:
    """``utils/mr_dataset.py:113-119``: tensors are stacked, everything else stays a list."""
    return {k: (torch.stack([b[k] for b in batch], dim=0) if isinstance(batch[0][k], torch.Tensor) else [b[k] for b in batch])
            for k in batch[0]}


def prepare_sample(samples: dict, device=None) -> dict:
    """What LAVIS ``prepare_sample(samples, cuda_enabled=True)`` does for the trainer (``utils/trainer.py:125``;
    third-party ``salesforce-lavis``, unpinned): move every tensor of the batch to the device."""
    if device is None:
        return samples
    return {k: (v.to(device, non_blocking=True) if isinstance(v, torch.Tensor) else v) for k, v in samples.items()}
