"""Evaluation metrics for moment retrieval / highlight detection (counterpart of the reference's ``eval/``)."""
from .mr_eval import (compute_average_precision_detection, compute_hl_ap, compute_hl_hit1, compute_mr_ap,  # noqa: F401
                      compute_mr_r1, eval_highlight, eval_moment_retrieval, eval_submission, get_ap,
                      interpolated_precision_recall, load_jsonl, mk_gt_scores)
