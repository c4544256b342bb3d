"""Entry point with the reference's ``finetune.py`` flags (``:9-62``): one process per GPU under
``python -m torch.distributed.run --nproc-per-node N --master-addr 127.0.0.1 -m mraudio_amd.finetune ...``
(RCCL through backend "nccl"); a single process when RANK / WORLD_SIZE are not set.

    python -m mraudio_amd.finetune --dataset Charades_STA --synthetic 16 --max-epoch 2 --output-dir out/ft
"""
from __future__ import annotations

import argparse
import datetime
import logging
import os

import torch
import torch.distributed as dist

from .utils.trainer import Trainer


def init_distributed_mode(args) -> None:
    """Reference ``finetune.py:9-31`` (SLURM branch dropped: one node, torchrun-style env only)."""
    if "RANK" in os.environ and "WORLD_SIZE" in os.environ:
        args.rank, args.world_size, args.gpu = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"]), int(os.environ["LOCAL_RANK"])
    else:
        args.rank, args.world_size, args.gpu = 0, 1, 0
    torch.cuda.set_device(args.gpu)
    if args.world_size > 1:
        dist.init_process_group(backend="nccl", init_method="env://", world_size=args.world_size, rank=args.rank,
                                timeout=datetime.timedelta(minutes=30), device_id=torch.device("cuda", args.gpu))
        dist.barrier()


def run_train(args):
    init_distributed_mode(args)
    trainer = Trainer(args)
    out = trainer.train()
    if args.rank == 0:
        print({"history": trainer.history, **out})
    if dist.is_initialized():
        dist.destroy_process_group()
    return out


def main(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--model", default="X-InstructBLIP")
    ap.add_argument("--model-path", default=None)
    ap.add_argument("--audio-encoder", default=None)
    ap.add_argument("--checkpoint", default=None, help="initial weights: state dict (.pth, reference key names)")
    ap.add_argument("--partial-checkpoint", action="store_true", help="accept a checkpoint that holds only part of the parameters (e.g. the trainer's trainable-only files); what it lacks keeps the seeded init and is reported")
    ap.add_argument("--video-folder", default=None)
    ap.add_argument("--train-annotation-file", default=None)
    ap.add_argument("--val-annotation-file", default=None)
    ap.add_argument("--embeds-folder", default=None)
    ap.add_argument("--output-dir", required=True)
    ap.add_argument("--val-freq", type=int, default=1)
    ap.add_argument("--save-freq", type=int, default=1)
    ap.add_argument("--max-epoch", type=int, default=50)
    ap.add_argument("--batch-size", type=int, default=1)
    ap.add_argument("--num-workers", type=int, default=0)
    ap.add_argument("--dataset", required=True, choices=["QVH", "Charades_STA"])
    ap.add_argument("--synthetic", type=int, default=0)
    ap.add_argument("--lr", type=float, default=3e-4)
    ap.add_argument("--warmup-steps", type=int, default=1000)
    args = ap.parse_args(argv)
    logging.basicConfig(level=logging.INFO)
    return run_train(args)


if __name__ == "__main__":
    main()
