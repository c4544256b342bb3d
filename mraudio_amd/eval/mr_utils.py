"""Name-compatible home of the helpers the reference keeps in ``eval/mr_utils.py``."""
import numpy as np

from ..utils.spans import temporal_iou_cross, temporal_iou_paired
from .mr_eval import compute_average_precision_detection, get_ap, interpolated_precision_recall, load_jsonl  # noqa: F401


def compute_temporal_iou_batch_paired(pred_windows, gt_windows):
    """``eval/mr_utils.py:16-39``."""
    return temporal_iou_paired(pred_windows, gt_windows)


def compute_temporal_iou_batch_cross(spans1, spans2):
    """``eval/mr_utils.py:40-68``: returns ``(iou, union)`` like the reference."""
    iou = temporal_iou_cross(spans1, spans2)
    inter = np.clip(np.minimum(spans1[:, None, 1], spans2[None, :, 1]) - np.maximum(spans1[:, None, 0], spans2[None, :, 0]), 0, None)
    return iou, (spans1[:, 1] - spans1[:, 0])[:, None] + (spans2[:, 1] - spans2[:, 0])[None, :] - inter
