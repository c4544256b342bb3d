"""Training loop with the reference's shape (``utils/trainer.py:20-253``), SURVEY section 8(f) row N1.

What is kept from the reference, step by step (``train_epoch``, ``:108-154``): per iteration the LR schedule
is stepped with ``(cur_epoch, cur_step)``, the loss is divided by ``accum_grad_iters = 2`` before
``backward``, the optimizer steps and zeroes every second iteration, validation runs ``generate`` ->
``post_process`` -> ``moment_str_to_list`` -> ``eval_submission(results, results)`` and the best
``brief["MR-full-R1-avg"]`` decides ``checkpoint_best.pth``; checkpoints hold only parameters that require
grad plus optimizer / scaler / epoch (``:184-211``) and resume restores all of them (``:232-253``).

What is different, because the data-parallel hot path here is the HIP Q-Former and not an autograd graph:
  * no ``DistributedDataParallel`` wrapper -- ranks hold different samples, the Q-Former gradients live in
    one flat fp32 buffer per modality and ``model.all_reduce_grads()`` averages them with ONE RCCL
    all-reduce per buffer at the accumulation boundary (DDP would reduce after every backward; the sum is
    the same).  Any other trainable parameter is averaged in one coalesced all-reduce;
  * no ``GradScaler`` / autocast -- the kernels take f16/bf16 operands but accumulate, keep the residual
    stream and write gradients in fp32, so there is nothing to scale (``"scaler": None`` in checkpoints);
  * the Q-Formers are unfrozen (``model.enable_qformer_training()``; BASELINE config 5) -- with the
    reference's frozen Q-Formers and no LLM on this path nothing would be trainable;
  * validation results are gathered from all ranks before scoring (the reference scores rank 0's shard only).

``LinearWarmupCosineLRScheduler`` restates the scheduler of the third-party ``salesforce-lavis`` package
(``lavis/common/optims.py``; version unpinned by the reference, absent offline): linear warm-up over
``warmup_steps`` iterations of epoch 0, then a per-epoch cosine.  Parity for it is unpinned.
"""
from __future__ import annotations

import logging
import math
import os
from types import SimpleNamespace
from typing import Dict, List, Optional

import torch
import torch.distributed as dist
from torch.utils.data import DataLoader, DistributedSampler

from ..eval.mr_eval import eval_submission
from .mr_dataset import MRDataset, SyntheticMRDataset, collate_fn, prepare_sample
from .spans import moment_str_to_list, post_process

log = logging.getLogger("mraudio_amd.trainer")


class LinearWarmupCosineLRScheduler:
    def __init__(self, optimizer, max_epoch, min_lr, init_lr, warmup_steps=0, warmup_start_lr=-1, **unused):
        self.optimizer, self.max_epoch, self.min_lr, self.init_lr = optimizer, max_epoch, min_lr, init_lr
        self.warmup_steps = warmup_steps
        self.warmup_start_lr = warmup_start_lr if warmup_start_lr >= 0 else init_lr

    def lr_at(self, cur_epoch: int, cur_step: int) -> float:
        if cur_epoch == 0:
            return min(self.init_lr, self.warmup_start_lr + (self.init_lr - self.warmup_start_lr) * cur_step / max(self.warmup_steps, 1))
        return (self.init_lr - self.min_lr) * 0.5 * (1.0 + math.cos(math.pi * cur_epoch / self.max_epoch)) + self.min_lr

    def step(self, cur_epoch: int, cur_step: int) -> None:
        lr = self.lr_at(cur_epoch, cur_step)
        for g in self.optimizer.param_groups:
            g["lr"] = lr


def _world():
    return (dist.get_rank(), dist.get_world_size()) if dist.is_available() and dist.is_initialized() else (0, 1)


class Trainer:
    def __init__(self, args, model: Optional[torch.nn.Module] = None, train_dataset=None, val_dataset=None):
        self.val_freq, self.save_freq, self.max_epoch = args.val_freq, args.save_freq, args.max_epoch
        self.output_dir = args.output_dir
        self.resume_ckpt_path = getattr(args, "resume_ckpt_path", None)
        gpu = getattr(args, "gpu", None)
        self.device = torch.device("cuda", gpu) if isinstance(gpu, int) else torch.device(gpu if gpu is not None else "cpu")
        self.accum_grad_iters = getattr(args, "accum_grad_iters", 2)       # reference :31
        self.start_epoch = 0
        self.rank, self.world_size = _world()
        assert args.dataset in ["QVH", "Charades_STA"]
        n_frms = 60 if args.dataset == "QVH" else 20

        if model is None:
            if getattr(args, "model", "X-InstructBLIP") != "X-InstructBLIP":
                raise NotImplementedError("only X-InstructBLIP has a trainable path in this build (VideoLLaMA is a stub)")
            from ..models.xinstructblip import XInstructBLIP
            model = XInstructBLIP(getattr(args, "model_path", None), getattr(args, "audio_encoder", None), device=self.device,
                                  op_dtype=getattr(args, "op_dtype", torch.bfloat16), checkpoint=getattr(args, "checkpoint", None),
                                  checkpoint_strict=not getattr(args, "partial_checkpoint", False))
        self.model = model
        if hasattr(model, "clip_parallel"):
            model.clip_parallel = False            # ranks see different samples
        if getattr(args, "train_qformers", True) and hasattr(model, "enable_qformer_training"):
            model.enable_qformer_training()
        if hasattr(model, "flat_optimizer_params") and getattr(args, "flat_params", True) and getattr(model, "train_qformers", False):
            params = model.flat_optimizer_params()      # one flat parameter per Q-Former: one fused update launch
        else:
            params = [p for p in model.parameters() if p.requires_grad]
        if not params:
            raise RuntimeError("the model has no trainable parameter")
        lr = float(getattr(args, "lr", 3e-4))
        self.optimizer = torch.optim.Adam(params, lr=lr, **({"fused": True} if self.device.type == "cuda" else {}))   # reference :65
        self.lr_scheduler = LinearWarmupCosineLRScheduler(self.optimizer, self.max_epoch, min_lr=0, init_lr=lr,
                                                          warmup_steps=getattr(args, "warmup_steps", 1000), warmup_start_lr=1e-8)
        self.scaler = None

        if train_dataset is None:
            if getattr(args, "synthetic", 0):
                train_dataset = SyntheticMRDataset(args.synthetic, T=n_frms, seed=0, signal=1.0)
                val_dataset = SyntheticMRDataset(max(2, args.synthetic // 2), T=n_frms, seed=1, signal=1.0)
            else:
                from ..processors.alpro_processors import AlproVideoEvalProcessor_Stamps, AlproVideoTrainProcessor_Stamps
                emb = getattr(args, "embeds_folder", None)
                tp = None if emb else AlproVideoTrainProcessor_Stamps(n_frms=n_frms, image_size=224)
                vp = None if emb else AlproVideoEvalProcessor_Stamps(n_frms=n_frms, image_size=224)
                from ..processors.audio_processors import BeatsAudioProcessor
                ap_ = None if emb else BeatsAudioProcessor(model_name="iter3", sampling_rate=16000, n_frames=n_frms, is_eval=False, frame_length=512)
                train_dataset = MRDataset(args.video_folder, args.train_annotation_file, tp, ap_, model="X-InstructBLIP", embeds_root=emb)
                val_dataset = MRDataset(args.video_folder, args.val_annotation_file, vp, ap_, model="X-InstructBLIP", embeds_root=emb)
        bs, nw = getattr(args, "batch_size", 1), getattr(args, "num_workers", 0)
        self.train_sampler = DistributedSampler(train_dataset, shuffle=True, num_replicas=self.world_size, rank=self.rank)
        val_sampler = DistributedSampler(val_dataset, shuffle=False, num_replicas=self.world_size, rank=self.rank)
        self.train_dataloader = DataLoader(train_dataset, batch_size=bs, sampler=self.train_sampler, num_workers=nw, collate_fn=collate_fn)
        self.val_dataloader = DataLoader(val_dataset, batch_size=bs, sampler=val_sampler, num_workers=nw, collate_fn=collate_fn)
        self.history: List[dict] = []

    # ---- epochs ---------------------------------------------------------------------------------------
    def train(self):
        best_metric, best_epoch = 0, 0
        if self.resume_ckpt_path is not None:
            self._load_checkpoint(self.resume_ckpt_path)
        for cur_epoch in range(self.start_epoch, self.max_epoch):
            stats = self.train_epoch(cur_epoch)
            entry = {"epoch": cur_epoch, **stats}
            if cur_epoch % self.val_freq == 0:
                results = self.eval_epoch()
                agg = results["brief"]["MR-full-R1-avg"]
                entry["MR-full-R1-avg"] = agg
                if self.rank == 0:
                    log.info("MR performance at epoch %d: %s", cur_epoch, agg)
                    if agg > best_metric:
                        best_epoch, best_metric = cur_epoch, agg
                        self._save_checkpoint(cur_epoch, is_best=True)
            if self.save_freq > 0 and cur_epoch % self.save_freq == 0 and self.rank == 0:
                self._save_checkpoint(cur_epoch, is_best=False)
            self.history.append(entry)
        if self.world_size > 1:
            dist.barrier()
        return {"best_epoch": best_epoch, "best_metric": best_metric}

    def _sync_grads(self) -> None:
        if self.world_size == 1:
            return
        covered = set()
        if hasattr(self.model, "all_reduce_grads"):
            self.model.all_reduce_grads()
            for m in getattr(self.model, "modalities", []):
                qf = getattr(self.model, f"{m}_Qformer", None)
                if qf is not None:
                    covered.update(id(p) for p in qf.parameters())
                    covered.add(id(getattr(self.model, f"{m}_query_tokens")))
        rest = [p.grad for p in self.model.parameters() if p.requires_grad and p.grad is not None and id(p) not in covered]
        if rest:
            flat = torch.cat([g.reshape(-1).float() for g in rest])
            dist.all_reduce(flat)
            flat.div_(self.world_size)
            off = 0
            for g in rest:
                g.copy_(flat[off: off + g.numel()].view_as(g))
                off += g.numel()

    def train_epoch(self, cur_epoch: int) -> Dict[str, str]:
        self.model.train()
        self.train_sampler.set_epoch(cur_epoch)
        log.info("Start training epoch %d, %d iters per inner epoch.", cur_epoch, len(self.train_dataloader))
        loss_sum, n_it = 0.0, 0
        for i, samples in enumerate(self.train_dataloader):
            samples = prepare_sample(samples, self.device if self.device.type == "cuda" else None)
            self.lr_scheduler.step(cur_epoch=cur_epoch, cur_step=i)
            output = self.model(samples)
            loss = output["loss"] / self.accum_grad_iters
            if loss.requires_grad:
                loss.backward()
            if (i + 1) % self.accum_grad_iters == 0:        # reference :136-139
                self._sync_grads()
                self.optimizer.step()
                self.optimizer.zero_grad()
            loss_sum += float(output["loss"].detach())
            n_it += 1
        stat = torch.tensor([loss_sum, float(n_it)], dtype=torch.float64, device=self.device if self.device.type == "cuda" else "cpu")
        if self.world_size > 1:
            dist.all_reduce(stat)
        avg = (stat[0] / stat[1].clamp(min=1)).item()
        log.info("Averaged stats: loss %.4f lr %.6f", avg, self.optimizer.param_groups[0]["lr"])
        return {"loss": "{:.3f}".format(avg), "lr": "{:.3f}".format(self.optimizer.param_groups[0]["lr"]), "loss_value": avg}

    @torch.no_grad()
    def eval_epoch(self):
        self.model.eval()
        results = []
        for samples in self.val_dataloader:
            samples = prepare_sample(samples, self.device if self.device.type == "cuda" else None)
            outputs = self.model.generate(samples)
            for qid, query, vid, target, output in zip(samples["qid"], samples["query"], samples["vid"], samples["text_output"], outputs):
                results.append({"qid": qid, "query": query, "vid": vid, "relevant_windows": moment_str_to_list(post_process(target)),
                                "pred_relevant_windows": moment_str_to_list(post_process(output))})
        if self.world_size > 1:
            parts: List[Optional[list]] = [None] * self.world_size
            dist.all_gather_object(parts, results)
            seen, results = set(), []
            for part in parts:                              # the sampler pads the last shard with repeats
                for r in part:
                    if r["qid"] not in seen:
                        seen.add(r["qid"])
                        results.append(r)
        return eval_submission(results, results, verbose=False)

    # ---- checkpoints ----------------------------------------------------------------------------------
    def _save_checkpoint(self, cur_epoch: int, is_best: bool = False) -> str:
        trainable = {k for k, v in self.model.named_parameters() if v.requires_grad}
        named = {k for k, _ in self.model.named_parameters()}
        state = {k: v.detach().cpu() for k, v in self.model.state_dict().items() if k in trainable or k not in named}
        os.makedirs(self.output_dir, exist_ok=True)
        save_to = os.path.join(self.output_dir, "checkpoint_{}.pth".format("best" if is_best else cur_epoch))
        log.info("Saving checkpoint at epoch %d to %s.", cur_epoch, save_to)
        torch.save({"model": state, "optimizer": self.optimizer.state_dict(), "scaler": None, "epoch": cur_epoch}, save_to)
        return save_to

    def _reload_best_model(self, model):
        path = os.path.join(self.output_dir, "checkpoint_best.pth")
        log.info("Loading checkpoint from %s.", path)
        ckpt = torch.load(path, map_location="cpu", weights_only=True)
        try:
            model.load_state_dict(ckpt["model"])
        except RuntimeError:
            log.warning("Key mismatch when loading checkpoint (expected when only part of the model was saved); retrying with strict=False.")
            model.load_state_dict(ckpt["model"], strict=False)
        return model

    def _load_checkpoint(self, filename: str) -> None:
        if not os.path.isfile(filename):
            raise RuntimeError("checkpoint path is invalid (URLs are not fetched: this build runs offline)")
        ckpt = torch.load(filename, map_location="cpu", weights_only=True)
        self.model.load_state_dict(ckpt["model"], strict=False)
        self.optimizer.load_state_dict(ckpt["optimizer"])
        self.start_epoch = ckpt["epoch"] + 1
        log.info("Resume checkpoint from %s", filename)


def default_args(**kw) -> SimpleNamespace:
    """The reference's ``finetune.py`` defaults (``:44-62``) as a namespace, for programmatic use."""
    base = dict(model="X-InstructBLIP", model_path=None, audio_encoder=None, video_folder=None, train_annotation_file=None,
                val_annotation_file=None, output_dir="out", val_freq=1, save_freq=1, max_epoch=50, batch_size=1, num_workers=0,
                dataset="Charades_STA", gpu=0)
    base.update(kw)
    return SimpleNamespace(**base)
