"""Row N1: the training loop's host logic on CPU with a small stand-in model --
accumulation, LR schedule, checkpoint contents / resume (``utils/trainer.py:108-253``), validation through
the metrics module, and the world_size-2 gloo gradient averaging."""
import math
import os
import socket

import torch
import torch.distributed as dist
import torch.multiprocessing as mp
from torch import nn

from mraudio_amd.utils.mr_dataset import MRDataset, SyntheticMRDataset, build_prompt, collate_fn
from mraudio_amd.utils.trainer import LinearWarmupCosineLRScheduler, Trainer, default_args


class TinyScorer(nn.Module):
    """Same surface as the model (forward -> {"loss"}, generate -> list[str]) on 16-wide features."""

    def __init__(self):
        super().__init__()
        torch.manual_seed(0)
        self.w = nn.Linear(16, 1)
        self.frozen = nn.Linear(4, 4)
        for p in self.frozen.parameters():
            p.requires_grad_(False)
        self.calls = 0

    def _logits(self, samples):
        return self.w(samples["video_embeds"].mean(2)).squeeze(-1)            # [B, T]

    def forward(self, samples):
        self.calls += 1
        x = self._logits(samples)
        tgt = torch.zeros_like(x)
        for r, txt in enumerate(samples["text_output"]):
            s, e = [int(v) for v in txt.strip("[]").split(",")]
            ts = torch.tensor(samples["timestamps"][r])
            tgt[r] = ((ts >= s) & (ts <= e)).float()
        return {"loss": nn.functional.binary_cross_entropy_with_logits(x, tgt)}

    @torch.no_grad()
    def generate(self, samples):
        out = []
        for r, row in enumerate(self._logits(samples)):
            on = (row > 0).nonzero().flatten().tolist() or [int(row.argmax())]
            ts = samples["timestamps"][r]
            out.append(f"[[{ts[on[0]]}, {ts[on[-1]]}]]")
        return out


class Narrow(SyntheticMRDataset):
    def __getitem__(self, i):
        rec = super().__getitem__(i)
        rec["video_embeds"] = rec["video_embeds"][..., :16].contiguous()
        return rec


def _trainer(tmp_path, n=6, **kw):
    args = default_args(output_dir=str(tmp_path), gpu="cpu", max_epoch=kw.pop("max_epoch", 2), warmup_steps=4, **kw)
    tr = Trainer(args, model=TinyScorer(), train_dataset=Narrow(n, T=20, seed=0, kv_video=2, modalities=("video",), signal=2.0),
                 val_dataset=Narrow(4, T=20, seed=1, kv_video=2, modalities=("video",), signal=2.0))
    return tr


def test_scheduler_restates_lavis_formula():
    opt = torch.optim.SGD([nn.Parameter(torch.zeros(1))], lr=1.0)
    s = LinearWarmupCosineLRScheduler(opt, max_epoch=10, min_lr=0, init_lr=3e-4, warmup_steps=1000, warmup_start_lr=1e-8)
    s.step(0, 0)
    assert opt.param_groups[0]["lr"] == 1e-8
    s.step(0, 500)
    assert abs(opt.param_groups[0]["lr"] - (1e-8 + (3e-4 - 1e-8) * 0.5)) < 1e-15
    s.step(0, 5000)
    assert opt.param_groups[0]["lr"] == 3e-4                      # capped
    s.step(5, 3)
    assert abs(opt.param_groups[0]["lr"] - 3e-4 * 0.5 * (1 + math.cos(math.pi * 0.5))) < 1e-15
    s.step(9, 0)
    assert 0 < opt.param_groups[0]["lr"] < 3e-4 * 0.03


def test_accumulation_steps_every_second_iteration(tmp_path):
    tr = _trainer(tmp_path, n=5, max_epoch=1)
    steps = []
    orig = tr.optimizer.step
    tr.optimizer.step = lambda *a, **k: (steps.append(tr.model.calls), orig(*a, **k))[1]
    stats = tr.train_epoch(0)
    assert steps == [2, 4]                                       # 5 iterations, accum 2: the 5th gradient stays pending
    assert tr.model.w.weight.grad is not None                    # ... in .grad, as in the reference
    assert set(stats) >= {"loss", "lr"} and stats["loss"] == "{:.3f}".format(stats["loss_value"])


def test_train_eval_checkpoint_resume(tmp_path):
    tr = _trainer(tmp_path, n=8, max_epoch=3, lr=0.05)
    out = tr.train()
    assert [h["epoch"] for h in tr.history] == [0, 1, 2] and all("MR-full-R1-avg" in h for h in tr.history)
    assert tr.history[-1]["loss_value"] < tr.history[0]["loss_value"]
    files = sorted(os.listdir(tmp_path))
    assert "checkpoint_0.pth" in files and "checkpoint_2.pth" in files
    ck = torch.load(os.path.join(tmp_path, "checkpoint_2.pth"), weights_only=True)
    assert set(ck) == {"model", "optimizer", "scaler", "epoch"} and ck["epoch"] == 2 and ck["scaler"] is None
    assert set(ck["model"]) == {"w.weight", "w.bias"}            # frozen parameters are not saved (reference :188-196)
    # resume: a fresh trainer continues after epoch 2 with the saved weights and optimizer state
    tr2 = _trainer(tmp_path, n=8, max_epoch=4, lr=0.05)
    tr2.resume_ckpt_path = os.path.join(tmp_path, "checkpoint_2.pth")
    tr2.train()
    assert [h["epoch"] for h in tr2.history] == [3]
    fresh = TinyScorer()
    assert not torch.equal(tr2.model.w.weight, fresh.w.weight)
    if out["best_metric"] > 0:
        m = tr._reload_best_model(TinyScorer())
        assert isinstance(m, TinyScorer)


def test_eval_epoch_returns_reference_shaped_metrics(tmp_path):
    tr = _trainer(tmp_path)
    res = tr.eval_epoch()
    assert list(res.keys())[0] == "brief" and "MR-full-R1@0.5" in res["brief"] and "MR-full-mIoU" in res["brief"]
    assert res["brief"]["MR-full-mAP"] == res["full"]["MR-mAP"]["average"]


def test_dataset_record_layout(tmp_path):
    import json
    ann = tmp_path / "ann.jsonl"
    ann.write_text(json.dumps({"qid": 7, "query": "a man walks", "vid": "v0", "duration": 30, "relevant_windows": [[2, 9]]}) + "\n")
    frames = torch.rand(3, 4, 8, 8)
    ds = MRDataset(str(tmp_path), str(ann), lambda p: (frames, [0, 30, 60, 95], 30.0), lambda p: torch.zeros(4, 16, 128))
    rec = ds[0]
    assert rec["text_input"] == "Query: a man walks\nGiven the video and the query, find the relevant windows.\nRelevant windows: "
    assert rec["text_input"] == build_prompt("a man walks")
    assert rec["text_output"] == "[[2, 9]]" and rec["timestamps"] == [0, 1, 2, 3] and rec["qid"] == 7
    b = collate_fn([rec, rec])
    assert b["video"].shape == (2, 3, 4, 8, 8) and b["audio"].shape == (2, 4, 16, 128) and b["duration"] == [30, 30]
    torch.save({"video_embeds": torch.zeros(4, 257, 8), "timestamps": [0, 2, 4, 6]}, tmp_path / "v0.pt")
    rec = MRDataset(str(tmp_path), str(ann), None, None, embeds_root=str(tmp_path))[0]
    assert rec["video_embeds"].shape == (4, 257, 8) and rec["timestamps"] == [0, 2, 4, 6] and "video" not in rec


# ---- world_size 2 ---------------------------------------------------------------------------------------
def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, ws, port, outdir, q):
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=ws)
    try:
        tr = _trainer(outdir, n=8, max_epoch=2, lr=0.05)
        tr.train()
        res = tr.eval_epoch()
        n_q = len(tr.val_dataloader.dataset)
        q.put((rank, tr.model.w.weight.detach().flatten().tolist(), res["brief"]["MR-full-mIoU"], n_q, tr.history[-1]["loss_value"]))
    finally:
        dist.destroy_process_group()


def test_two_ranks_average_gradients_and_pool_validation(tmp_path):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, str(tmp_path), q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=180) for _ in procs], key=lambda t: t[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    (_, w0, miou0, n0, l0), (_, w1, miou1, n1, l1) = res
    assert w0 == w1                                              # same averaged gradients -> identical replicas
    assert miou0 == miou1 and l0 == l1                           # pooled validation / averaged loss on every rank
    assert w0 != TinyScorer().w.weight.detach().flatten().tolist()
