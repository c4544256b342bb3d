"""``-m gpu``: BASELINE config 4 on the HIP path at world_size 2 -- the clips of one video sharded in contiguous
blocks over two ranks (both on ``cuda:0`` of the one-GPU box, ``gloo`` rendezvous on 127.0.0.1), Q-Former per shard,
ONE packed all-gather of the query embeddings and [CLS] vectors, then score / fuse / span on every rank -- must
reproduce the unsharded single-process result: same fused logits, same integer spans, on every rank, for equal
(num = 8) and ragged (num = 7) shards and for a raw-input run that goes through the sharded batched encoder.

This file sorts first among the GPU tests on purpose: the three children are fresh ``spawn`` processes started
while the parent pytest process has not touched the GPU yet (it never does in this file), and only two of them use
the card at any time.  RCCL itself needs one GPU per rank: the 8-GPU run is the driver's (``bench.py --gpus N``).
"""
import os
import socket

import pytest
import torch
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu

PROMPT = "Query: a person opens the door.\nGiven the video and the query, find the relevant windows.\nRelevant windows: "


class _TinyEncoder(torch.nn.Module):
    """Deterministic stand-in for ViT-g / BEATs (random projections of pooled pixels): frames [n, ...] -> [n, Kv, E]."""

    def __init__(self, kv, width, seed):
        super().__init__()
        g = torch.Generator().manual_seed(seed)
        self.register_buffer("w", torch.randn(48, kv * 8, generator=g) * 0.3)
        self.register_buffer("u", torch.randn(8, width, generator=g))
        self.kv = kv

    def forward(self, x):
        v = torch.nn.functional.adaptive_avg_pool1d(x.flatten(1).float()[:, None, :], 48)[:, 0]
        return ((v @ self.w).view(x.shape[0], self.kv, 8) @ self.u).to(torch.float16)


def _samples(num, raw):
    g = torch.Generator().manual_seed(100 + num)
    s = {"text_input": [PROMPT], "timestamps": [list(range(0, 2 * num, 2))], "duration": [2 * num]}
    if raw:      # through the encoders: video [B, 3, T, H, W], audio [B, T, F, 128]
        s["video"] = torch.rand(1, 3, num, 28, 28, generator=g)
        s["audio"] = torch.randn(1, num, 16, 128, generator=g)
    else:
        s["video_embeds"] = torch.randn(1, num, 300, 1408, generator=g).half()
        s["audio_embeds"] = torch.randn(1, num, 64, 768, generator=g).half()
    return s


def _worker(rank, ws, port, out_path):
    import torch.distributed as dist

    from mraudio_amd.models.xinstructblip import XInstructBLIP

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    torch.cuda.set_device(0)
    if ws > 1:
        dist.init_process_group("gloo", rank=rank, world_size=ws)
    try:
        dev = torch.device("cuda:0")
        model = XInstructBLIP(seed=0, perturb=True, device=dev, video_encoder=_TinyEncoder(40, 1408, 1).to(dev),
                              audio_encoder=_TinyEncoder(24, 768, 2).to(dev))
        res = {}
        for name, num, raw in (("equal8", 8, False), ("ragged7", 7, False), ("raw_encode7", 7, True)):
            out = model.encode_fuse(_samples(num, raw))
            strings = model.generate(_samples(num, raw))
            torch.cuda.synchronize()
            res[name] = {"fused": out["fused"].cpu(), "spans": out["spans"].cpu(), "strings": strings,
                         "z_video": out["z"]["video"].cpu(), "logit_audio": out["logit"]["audio"].cpu()}
        torch.save(res, out_path)
    finally:
        if ws > 1:
            dist.destroy_process_group()


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _launch(ws, tmp_path):
    ctx = mp.get_context("spawn")
    port = _free_port()
    paths = [str(tmp_path / f"ws{ws}_rank{r}.pt") for r in range(ws)]
    procs = [ctx.Process(target=_worker, args=(r, ws, port, paths[r])) for r in range(ws)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(timeout=900)
        assert p.exitcode == 0, f"world {ws}: a rank exited with {p.exitcode}"
    return [torch.load(p, weights_only=True) for p in paths]


def test_sharded_scores_and_spans_equal_the_single_process_run(tmp_path):
    (single,) = _launch(1, tmp_path)
    ranks = _launch(2, tmp_path)
    for name in ("equal8", "ragged7", "raw_encode7"):
        ref = single[name]
        for r, rk in enumerate(ranks):
            got = rk[name]
            # integers and strings: exact on every rank
            assert torch.equal(got["spans"], ref["spans"]), (name, r)
            assert got["strings"] == ref["strings"], (name, r)
            # every element is accumulated in the same order whatever the batch size (tiles only regroup rows), so the
            # gathered embeddings and the logits are bit-identical to the unsharded run; 1e-6 leaves room for nothing
            # but a different split of the attention's KV range
            assert (got["z_video"] - ref["z_video"]).abs().max().item() <= 1e-6, (name, r)
            assert (got["fused"] - ref["fused"]).abs().max().item() <= 1e-6, (name, r)
            assert (got["logit_audio"] - ref["logit_audio"]).abs().max().item() <= 1e-6, (name, r)
        assert torch.equal(ranks[0][name]["fused"], ranks[1][name]["fused"])   # the ranks agree bit for bit
