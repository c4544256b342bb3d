"""Inference loop of the reference's ``evaluate.py:41-59``: batches -> ``model.generate`` -> span parsing ->
one JSON line per query, in the format ``eval/mr_eval.py`` (here ``mraudio_amd.eval.mr_eval``) scores.
On top of the reference's record this build also writes ``pred_saliency_scores`` -- the fused per-position
cosine logits the span was cut from -- so the highlight metrics can be computed as well.

    python -m mraudio_amd.evaluate --synthetic 8 --output-file out/pred.jsonl      # smoke run on the GPU
"""
from __future__ import annotations

import argparse
import json
import os
from typing import Iterable, List, Optional

import torch

from .utils.mr_dataset import prepare_sample
from .utils.spans import moment_str_to_list, post_process


@torch.no_grad()
def run_inference(model, dataloader: Iterable[dict], output_file: Optional[str] = None, with_saliency: bool = True,
                  device=None) -> List[dict]:
    records: List[dict] = []
    fh = None
    if output_file:
        os.makedirs(os.path.dirname(os.path.abspath(output_file)), exist_ok=True)
        fh = open(output_file, "w")
    for samples in dataloader:
        samples = prepare_sample(samples, device)
        scores = None
        if with_saliency and hasattr(model, "generate_with_scores"):
            outputs, scores = model.generate_with_scores(samples)
        else:
            outputs = model.generate(samples)
        for k, (qid, query, vid, raw) in enumerate(zip(samples["qid"], samples["query"], samples["vid"], outputs)):
            rec = {"qid": qid, "query": query, "vid": vid, "pred_relevant_windows": moment_str_to_list(post_process(raw)), "raw_out": raw}
            if scores is not None:
                rec["pred_saliency_scores"] = [float(x) for x in scores[k]]
            records.append(rec)
            if fh:
                fh.write(json.dumps(rec) + "\n")
    if fh:
        fh.close()
    return records


def main(argv=None) -> None:
    from torch.utils.data import DataLoader

    from .models.xinstructblip import XInstructBLIP
    from .utils.mr_dataset import MRDataset, SyntheticMRDataset, collate_fn

    ap = argparse.ArgumentParser()
    ap.add_argument("--model", default="X-InstructBLIP")
    ap.add_argument("--model-path", default=None)
    ap.add_argument("--audio-encoder", default=None)
    ap.add_argument("--checkpoint", default=None, help="state dict (.pth, reference key names) with the Q-Former / LN / projection weights")
    ap.add_argument("--partial-checkpoint", action="store_true", help="accept a checkpoint that holds only part of the parameters (e.g. the trainer's trainable-only files); what it lacks keeps the seeded init and is reported")
    ap.add_argument("--video-folder", default=None)
    ap.add_argument("--annotation-file", default=None)
    ap.add_argument("--embeds-folder", default=None, help="pre-extracted encoder outputs, <vid>.pt")
    ap.add_argument("--output-file", required=True)
    ap.add_argument("--device", default="cuda:0")
    ap.add_argument("--batch-size", type=int, default=2)
    ap.add_argument("--num-workers", type=int, default=0)
    ap.add_argument("--dataset", default="Charades_STA", choices=["QVH", "Charades_STA"])
    ap.add_argument("--synthetic", type=int, default=0, help="evaluate N seeded synthetic videos instead of a corpus")
    args = ap.parse_args(argv)
    n_frms = 60 if args.dataset == "QVH" else 20
    model = XInstructBLIP(args.model_path, args.audio_encoder, device=args.device, checkpoint=args.checkpoint, checkpoint_strict=not args.partial_checkpoint)
    print(f"weights: {model.weights_source}")
    if args.synthetic:
        ds = SyntheticMRDataset(args.synthetic, T=n_frms)
    else:
        ds = MRDataset(args.video_folder, args.annotation_file, None, None, model=args.model, embeds_root=args.embeds_folder)
    dl = DataLoader(ds, shuffle=False, batch_size=args.batch_size, num_workers=args.num_workers, collate_fn=collate_fn)
    recs = run_inference(model, dl, args.output_file, device=torch.device(args.device))
    print(f"wrote {len(recs)} predictions to {args.output_file}")


if __name__ == "__main__":
    main()
