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


def _config4_worker(rank, ws, port, out_path):
    """BASELINE config 4 at its per-rank shape: 32 clips per rank x (8224 + 496) tokens (the bench's weak-scaling unit), folded path,
    side streams, kv_first, ONE packed all-gather -- exactly ``bench.py``'s step with ``world`` ranks."""
    import torch.distributed as dist

    from mraudio_amd.models.xinstructblip import ENC_WIDTH, XInstructBLIP

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    torch.cuda.set_device(0)
    if ws > 1:
        dist.init_process_group("gloo", rank=rank, world_size=ws)
    try:
        dev = torch.device("cuda:0")
        model = XInstructBLIP(seed=0, perturb=False, op_dtype=torch.float16, device=dev)
        n_total, per, L = 64, 64 // ws, 32
        kv = {"video": 32 * 257, "audio": 496}
        g = torch.Generator().manual_seed(4321)
        ids = torch.randint(1000, 30000, (n_total, L), generator=g)
        feats = {}
        for m in ("video", "audio"):      # the SAME 64 clips in every process; a rank holds its contiguous block only
            base = torch.randn(4, kv[m], ENC_WIDTH[m], generator=g).half()
            scale = 1.0 + 0.01 * torch.arange(n_total).view(-1, 1, 1)
            feats[m] = ((base.repeat(16, 1, 1).float() * scale).half())[rank * per:(rank + 1) * per].to(dev)
        tmask = torch.ones(per, L, dtype=torch.long, device=dev)
        calls = {"n": 0}
        if ws > 1:
            from mraudio_amd import parallel
            real = parallel.all_gather_packed

            def counted(*a, **k):
                calls["n"] += 1
                return real(*a, **k)
            parallel.all_gather_packed = counted
        out = model.fuse_score(feats, ids[rank * per:(rank + 1) * per].to(dev), tmask, bs=1, num=n_total)
        torch.cuda.synchronize()
        torch.save({"fused": out["fused"].cpu(), "spans": out["spans"].cpu(), "z_video": out["z"]["video"].cpu(),
                    "logit_audio": out["logit"]["audio"].cpu(), "gathers": calls["n"]}, out_path)
    finally:
        if ws > 1:
            dist.destroy_process_group()


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


def _launch(ws, tmp_path, target=None):
    ctx = mp.get_context("spawn")
    port = _free_port()
    tag = "cfg4" if target is not None else "ws"
    paths = [str(tmp_path / f"{tag}{ws}_rank{r}.pt") for r in range(ws)]
    procs = [ctx.Process(target=target or _worker, args=(r, ws, port, paths[r])) for r in range(ws)]
    for p in procs:
        p.start()
    try:
        for p in procs:
            p.join(timeout=900)
    finally:
        # a rank that is still alive (timeout, or a peer died and it waits in a collective) holds the GPU and the gloo rendezvous: end it
        for p in procs:
            if p.is_alive():
                p.terminate()
                p.join(10)
                if p.is_alive():
                    p.kill()
                    p.join()
    codes = [p.exitcode for p in procs]
    assert codes == [0] * ws, f"world {ws}: exit codes {codes}"
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


def test_config4_per_rank_shape_two_ranks_equal_one_process(tmp_path):
    """2 ranks x 32 clips x Kv 8224 + 496 (folded path, side streams, kv_first, one packed all-gather on the critical
    path once per step) equal the 1-process 64-clip run: integer spans exact, embeddings and logits to fp32 summation order."""
    (single,) = _launch(1, tmp_path, _config4_worker)
    ranks = _launch(2, tmp_path, _config4_worker)
    for r, got in enumerate(ranks):
        assert got["gathers"] == 1, (r, got["gathers"])                        # ONE collective per step
        assert torch.equal(got["spans"], single["spans"]), r
        assert got["fused"].shape == (64,) and torch.isfinite(got["fused"]).all()
        assert (got["z_video"] - single["z_video"]).abs().max().item() <= 1e-4, r
        assert (got["fused"] - single["fused"]).abs().max().item() <= 1e-5, r
        assert (got["logit_audio"] - single["logit_audio"]).abs().max().item() <= 1e-5, r
    assert torch.equal(ranks[0]["fused"], ranks[1]["fused"])


def test_bench_multi_rank_branch_runs_under_gloo_on_one_device():
    """``bench.py``'s N > 1 branch (barrier, max-over-ranks timing, rank-0 JSON line, weak scaling) rehearsed with 2 ranks on the one GPU:
    ``BENCH_SINGLE_DEVICE=1 BENCH_BACKEND=gloo`` (RCCL needs one GPU per rank; the 8-GPU run is the driver's)."""
    import json
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, BENCH_SINGLE_DEVICE="1", BENCH_BACKEND="gloo", MASTER_ADDR="127.0.0.1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--no-encode", "--cpu-clips", "0"]
    res = subprocess.run(cmd, capture_output=True, text=True, timeout=900, cwd=root, env=env)
    assert res.returncode == 0, res.stderr[-2000:]
    lines = [l for l in res.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, res.stdout[-2000:]                                     # rank 0 only
    line = json.loads(lines[0])
    assert line["n_gpus"] == 2 and line["scaling"] == "weak" and line["config"]["global_clips"] == 64
    assert line["value"] > 0 and abs(line["value"] - 64 / (line["ms_per_step"] * 1e-3)) / line["value"] < 1e-2
