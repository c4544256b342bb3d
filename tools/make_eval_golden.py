"""Generates tests/golden/mr_eval.json: seeded synthetic submissions / ground truths and the metrics the
REFERENCE's ``eval/mr_eval.py`` computes on them (imported from /root/reference in the build container;
``eval.mr_eval`` needs only numpy + scikit-learn).  The fixture holds data only -- inputs and expected
outputs -- and is what pins ``mraudio_amd/eval/mr_eval.py``.

    python tools/make_eval_golden.py            # rewrites tests/golden/mr_eval.json
"""
import contextlib
import io
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def synth_case(seed, n_q, duration=150, with_saliency=True, integer=True, invalid_every=0, multi_gt=True, ties=False):
    """QVHighlights-shaped records: windows on a 2-second clip grid, 3-annotator saliency in 0..4."""
    rng = np.random.default_rng(seed)
    sub, gt = [], []
    n_clips = duration // 2
    for q in range(n_q):
        n_gt = int(rng.integers(1, 4)) if multi_gt else 1
        wins = []
        for _ in range(n_gt):
            s = int(rng.integers(0, n_clips - 2)) * 2
            e = min(duration, s + 2 * int(rng.integers(1, 30)))
            wins.append([s, e])
        n_pred = int(rng.integers(1, 6))
        preds = []
        for k in range(n_pred):
            if ties and k and rng.random() < 0.3:
                preds.append(list(preds[-1]))                     # duplicated window -> the second copy is a false positive
                continue
            if rng.random() < 0.6:                                 # a jittered ground-truth window
                w = wins[int(rng.integers(0, n_gt))]
                s = max(0, w[0] + int(rng.integers(-6, 7)))
                e = min(duration, max(s + 1, w[1] + int(rng.integers(-6, 7))))
            else:
                s = int(rng.integers(0, duration - 2))
                e = min(duration, s + int(rng.integers(1, 60)))
            preds.append([s, e] if integer else [s + float(rng.random()), e + float(rng.random()) + 1.0])
        if invalid_every and q % invalid_every == 0:
            preds[0] = [-1, -1]                                     # what moment_str_to_list returns for an unparsable output
        rec = {"qid": q, "query": f"query {q}", "vid": f"v{q}", "pred_relevant_windows": preds}
        g = {"qid": q, "query": f"query {q}", "vid": f"v{q}", "duration": duration, "relevant_windows": wins}
        if with_saliency:
            rel = sorted({c for w in wins for c in range(w[0] // 2, min(n_clips, w[1] // 2))})
            g["relevant_clip_ids"] = rel
            g["saliency_scores"] = rng.integers(0, 5, size=(len(rel), 3)).tolist()
            n_scores = n_clips + int(rng.integers(-2, 3))        # longer / shorter than the clip count: cut / zero-padded
            scores = rng.standard_normal(n_scores)
            if rel:
                scores[[c for c in rel if c < n_scores]] += 1.0
            if ties:
                scores = np.round(scores, 1)
            rec["pred_saliency_scores"] = [float(x) for x in scores]
        sub.append(rec)
        gt.append(g)
    return sub, gt


CASES = {
    "qvh_like": dict(seed=0, n_q=40),
    "charades_like_single_gt": dict(seed=1, n_q=25, duration=40, multi_gt=False, with_saliency=False),
    "float_windows": dict(seed=2, n_q=30, integer=False, with_saliency=False),
    "invalid_and_ties": dict(seed=3, n_q=30, invalid_every=5, ties=True),
    "one_query": dict(seed=4, n_q=1),
}


def main():
    sys.path.insert(0, "/root/reference")
    from eval import mr_eval as R                      # the reference implementation (not shipped, not copied)
    from eval import mr_utils as RU

    out = {"generator": "tools/make_eval_golden.py", "reference": "eval/mr_eval.py eval_submission / compute_mr_ap / compute_mr_r1", "cases": {}}
    for name, kw in CASES.items():
        sub, gt = synth_case(**kw)
        with contextlib.redirect_stdout(io.StringIO()):
            res = R.eval_submission(json.loads(json.dumps(sub)), json.loads(json.dumps(gt)), verbose=False)
        out["cases"][name] = {"args": kw, "submission": sub, "ground_truth": gt, "expected": json.loads(json.dumps(res, default=float))}
    # direct known answers of the helpers
    rng = np.random.default_rng(9)
    kats = []
    for _ in range(6):
        g = [{"video-id": "a", "t-start": float(s), "t-end": float(s + d)} for s, d in zip(rng.integers(0, 50, 3), rng.integers(1, 30, 3))]
        p = [{"video-id": "a" if rng.random() < 0.85 else "b", "t-start": float(s), "t-end": float(s + d)}
             for s, d in zip(rng.integers(0, 50, 5), rng.integers(1, 30, 5))]
        ap = RU.compute_average_precision_detection(json.loads(json.dumps(g)), json.loads(json.dumps(p)))
        kats.append({"ground_truth": g, "prediction": p, "ap": [float(x) for x in ap]})
    out["detection_ap"] = kats
    gap = []
    for _ in range(8):
        n = int(rng.integers(3, 40))
        y = (rng.random(n) < 0.4).astype(float)
        s = np.round(rng.standard_normal(n), 1 if rng.random() < 0.5 else 6)
        gap.append({"y_true": y.tolist(), "y_predict": s.tolist(), "ap": float(RU.get_ap(y, s)),
                    "ap_raw": float(RU.get_ap(y, s, interpolate=False)) if len(set(y)) > 1 else None,
                    "ap_11": float(RU.get_ap(y, s, point_11=True)) if len(set(y)) > 1 else None})
    out["get_ap"] = gap
    with open(os.path.join(ROOT, "tests", "golden", "mr_eval.json"), "w") as f:
        json.dump(out, f)
    print("wrote tests/golden/mr_eval.json", os.path.getsize(os.path.join(ROOT, "tests", "golden", "mr_eval.json")), "bytes")


if __name__ == "__main__":
    main()
