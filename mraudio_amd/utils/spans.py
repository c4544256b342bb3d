"""Integer span handling on the host, behaviour-compatible with the reference.

``evaluate.py:48`` turns the model's output string into spans with
``moment_str_to_list(post_process(raw))`` (``utils/utils.py:66-132,364-415``); ``generate`` of this
build emits strings those two functions accept.  They are restated here (own code, same observable
behaviour, pinned by ``tests/golden/integer_kats.json`` captured from the reference's functions)
so a caller does not need the reference's ``utils`` package (which imports wandb) to parse them.
Also the temporal IoU of ``eval/mr_utils.py:16-75`` used by the metrics.
"""
from __future__ import annotations

import ast
import re
from typing import List

import numpy as np

_NESTED = re.compile(r"\[\[.*\]\]")
ERROR_SPAN = "[[-1, -1]]"


def post_process(pred: str) -> str:
    """Repair a generated span string (reference ``utils/utils.py:66-132``): cut at ``</s>``, drop
    newlines, require the ``[[...]]`` shape (else ``"[[-1, -1]]"``), split windows at whitespace
    before ``[``, strip trailing commas, insert a missing comma between two digits, collapse comma
    runs, and swap a reversed pair of integers."""
    pred = pred.split("</s>")[0].replace("\n", "").replace("\r", "")
    if not _NESTED.match(pred):
        return ERROR_SPAN
    body = pred[1:-1]
    fixed = []
    for win in re.split(r"\s+(?=\[)", body):
        win = re.sub(r",+$", "", win)
        win = re.sub(r"(\d) (\d)", r"\1, \2", win)
        win = re.sub(r",+", ",", win)
        nums = re.findall(r"\d+", win)
        if len(nums) == 2 and int(nums[0]) > int(nums[1]):
            win = "[" + nums[1] + ", " + nums[0] + "]"
        fixed.append(win)
    return "[" + ", ".join(fixed) + "]"


def moment_str_to_list(m: str) -> List[List[int]]:
    """``"[[0, 1], [4, 7]]"`` -> ``[[0, 1], [4, 7]]`` (reference ``utils/utils.py:364-415``): anything
    unparsable -> ``[[-1, -1]]``; a bare int entry -> ``[-1, -1]``; an entry whose length is not 2 ->
    ``[-len]``; non-integer members -> ``-1``."""
    if m == ERROR_SPAN or not _NESTED.match(m):
        return [[-1, -1]]
    try:
        val = ast.literal_eval(m)
    except Exception:
        return [[-1, -1]]
    if not isinstance(val, list):
        return [[-1, -1]]
    for i in range(len(val)):
        if isinstance(val[i], int):
            val[i] = [-1, -1]
        if len(val[i]) != 2:
            val[i] = [-len(val[i])]
        for j in range(len(val[i])):
            if not isinstance(val[i][j], int):
                val[i][j] = -1
    return val


def convert_percentages_to_second(percentages: str, duration: int) -> str:
    """Reference ``utils/utils.py:48-63``: every number in a ``[[...]]`` string becomes
    ``int(number * duration)``; other strings -> ``"[[-1, -1]]"``."""
    if not _NESTED.match(percentages):
        return ERROR_SPAN

    def repl(mt):
        try:
            return str(int(float(mt.group()) * duration))
        except Exception:
            return "-1"

    return re.sub(r"[-+]?\d*\.\d+|\d+", repl, percentages)


def temporal_iou_paired(pred: np.ndarray, gt: np.ndarray) -> np.ndarray:
    """Row-wise IoU of ``[start, end]`` windows with the reference's hull-as-union convention
    (``eval/mr_utils.py:16-39``); 0 where the hull is empty."""
    inter = np.maximum(0, np.minimum(pred[:, 1], gt[:, 1]) - np.maximum(pred[:, 0], gt[:, 0]))
    hull = np.maximum(pred[:, 1], gt[:, 1]) - np.minimum(pred[:, 0], gt[:, 0])
    return np.divide(inter, hull, out=np.zeros_like(inter, dtype=float), where=hull != 0)


def temporal_iou_cross(a: np.ndarray, b: np.ndarray) -> np.ndarray:
    """All-pairs IoU ``[N, M]`` with the true union (``eval/mr_utils.py:42-75``)."""
    la, lb = a[:, 1] - a[:, 0], b[:, 1] - b[:, 0]
    left = np.maximum(a[:, None, 0], b[None, :, 0])
    right = np.minimum(a[:, None, 1], b[None, :, 1])
    inter = np.clip(right - left, 0, None)
    union = la[:, None] + lb[None, :] - inter
    return inter / union
