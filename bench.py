"""bench.py -- clips/s of the encode/fuse/score hot path on 1..8 MI355X (one process per GPU).

    python bench.py --gpus 1 --steps 5 --warmup 2
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

A step = one pass of the hot path over one batch of synthetic input that is already resident in
HBM: per GPU 32 clips, each clip = 32 video frames (32 x 257 = 8224 ViT-g tokens x 1408, the
BASELINE "32-clip x 32-frame" shape) + 10 s of audio (496 BEATs tokens x 768), a
32-token prompt.  Timed: modality LayerNorm -> video Q-Former -> audio Q-Former -> (N > 1: RCCL
all-gather of the query embeddings) -> cosine scores -> fusion -> integer spans.  Nothing is skipped
or cached between steps.  The ViT-g / BEATs encoders are NOT in the timed region of `value`: the features
are the synthetic input (BASELINE configs 1-4).  The ViT-g encode of the same 1024 frames -- on this build's own
kernels (mra_vit_forward) and, beside it, stock PyTorch -- is timed separately and reported under `encode_stage`.
Weak scaling: every GPU holds its own 32 clips of a 32*N-clip video; value = all clips / time.

Also on the JSON line: `roofline` of the dominant kernels -- with the folded cross-attention (Kv >= 2048, the
headline shape) the block of one cross layer (per-head Q' GEMM, scores + split-softmax statistics, P.enc, per-head
context GEMM), timed inside the steps by an event pair the library records on the launch stream; `frac` is EXECUTED
flops / dense f16 MFMA peak, the reference formulation's algorithmic flops are reported beside it -- and
`cpu_baseline` (the CPU oracle on a bounded sample, median of five passes, rank 0, N = 1).
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

PEAK_F16_TFLOPS = 2500.0  # dense f16/bf16 MFMA peak, /opt/skills/guides/MI355X_MICROARCH.md


def log(msg):
    print(f"[bench {time.strftime('%H:%M:%S')}] {msg}", file=sys.stderr, flush=True)


def host_threads() -> int:
    """Cores this process may actually use: the affinity mask, capped by the cgroup CPU quota (a GPU
    box hands one GPU's share of a large host; os.cpu_count() would oversubscribe it)."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except Exception:
        pass
    return max(1, min(n, 64))


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--workload", default="clip32x32", choices=["clip32x32", "ref", "finetune"],
                    help="clip32x32: Kv=8224 video / 496 audio tokens per clip (headline); ref: the reference's one-frame items (257 / 256); "
                         "finetune: BASELINE config 5 (Q-Former fwd + bwd + optimizer step, bf16, B = 1 x T = 20 per GPU; tools/bench_finetune.py's line)")
    ap.add_argument("--clips", type=int, default=32, help="clips per GPU")
    ap.add_argument("--text-len", type=int, default=32)
    ap.add_argument("--cpu-clips", type=int, default=-1, help="clips in the CPU-baseline sample (0 = skip, -1 = auto)")
    ap.add_argument("--dtype", default="f16", choices=["f16", "bf16"])
    ap.add_argument("--cross-mode", default="auto", choices=["auto", "kv_cache", "fold", "fold_stream", "fold384", "fold_rescale_pass"],
                    help="cross-attention formulation (auto = folded from Kv >= 2048; fold_stream / fold384: A/B variants)")
    ap.add_argument("--cross-precision", default="op", choices=["op", "split"],
                    help="precision of the cross-attention score chain: op = f16 / bf16 operands (default); split = hi + lo pairs (~22 bits) for sharply attending weights")
    ap.add_argument("--chain-ring", type=int, default=-1, help="A/B: mask of chain GEMMs on the ring kernel's tiles (1 QKV, 2 FFN-up, 4 residual projections; -1 = library default)")
    ap.add_argument("--pair", action="store_true", help="A/B: the two modality Q-Formers as ONE grouped launch sequence (mra_qformer_forward_pair) instead of two forwards on two streams; "
                                                        "measured slower (6.72-6.87 vs 6.57 ms per step), although the folded block runs at 0.65 instead of 0.79 ms")
    ap.add_argument("--no-priority", action="store_true", help="A/B: same stream priority for both modalities")
    ap.add_argument("--no-kv-first", action="store_true", help="A/B: let the light modality start beside the heavy K/V projection")
    ap.add_argument("--no-encode", action="store_true", help="skip the separately timed ViT-g encode stage")
    return ap.parse_args()


def main():
    args = parse()
    if args.workload == "finetune":     # BASELINE config 5 has its own step; same launch contract (one process per GPU, rank 0 prints one JSON line)
        from tools import bench_finetune
        return bench_finetune.main(["--steps", str(args.steps), "--warmup", str(args.warmup), "--dtype", "bf16" if args.dtype == "f16" else args.dtype])
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("launch with torch.distributed.run --nproc-per-node N for --gpus N")
        args.gpus = world
    assert torch.cuda.is_available(), "bench.py needs a GPU (there is no CPU path)"
    if os.environ.get("BENCH_SINGLE_DEVICE"):  # rehearsal of the N > 1 path on a one-GPU box (with BENCH_BACKEND=gloo)
        local = 0
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    import torch.distributed as dist

    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        backend = os.environ.get("BENCH_BACKEND", "nccl")  # "nccl" is RCCL on ROCm
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group(backend)

    from mraudio_amd import _lib
    from mraudio_amd.models.xinstructblip import ENC_WIDTH, XInstructBLIP

    kv = {"video": 32 * 257, "audio": 496} if args.workload == "clip32x32" else {"video": 257, "audio": 256}
    n_local, L = args.clips, args.text_len
    n_total = n_local * world
    op_dtype = torch.float16 if args.dtype == "f16" else torch.bfloat16
    # BERT-style synthetic weights, seed 0: N(0, 0.02) matrices, zero biases, unit LayerNorms
    model = XInstructBLIP(seed=0, perturb=False, op_dtype=op_dtype, device=dev)
    model.pair_forward = args.pair
    model.kv_first = not args.no_kv_first
    model.prioritize_heavy = not args.no_priority
    for m in ("video", "audio"):
        getattr(model, f"{m}_Qformer").set_cross_mode(args.cross_mode)
        if args.cross_precision != "op":
            getattr(model, f"{m}_Qformer").set_cross_precision(args.cross_precision)
        if args.chain_ring >= 0:
            getattr(model, f"{m}_Qformer").set_option("chain_ring", args.chain_ring)
    g = torch.Generator(device=dev).manual_seed(1234 + rank)
    feats = {m: torch.randn(n_local, kv[m], ENC_WIDTH[m], generator=g, device=dev, dtype=torch.float16) for m in ("video", "audio")}
    ids = torch.randint(1000, 30000, (n_local, L), generator=g, device=dev)
    tmask = torch.ones(n_local, L, dtype=torch.long, device=dev)

    def step():
        # one video of n_total clips (bs = 1); this rank owns clips [rank*n_local, (rank+1)*n_local)
        return model.fuse_score(feats, ids, tmask, bs=1, num=n_total)

    def fence():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    # lazy one-time work (workspace growth in the caching allocator, first-launch kernel attributes, per-head
    # weight regrouping of the folded path, event creation, clock ramp) belongs to set-up, not to the W warm-up steps
    for _ in range(3):
        step()
    fence()
    log(f"rank {rank}/{world}: model ready, {n_local} clips/GPU, kv {kv}")
    for _ in range(args.warmup):
        out = step()
    fence()
    log("warm-up done")
    # events around the dominant kernels (folded path: the block of cross layer 0; K/V-cache path: the K/V projection GEMM) inside
    # every timed step, recorded by the library on the launch stream (= torch's current stream)
    lib = _lib.lib()
    qf = model.video_Qformer
    evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(args.steps)]
    for a, b in evs:   # create the underlying hipEvents (torch creates them lazily on first record)
        a.record(); b.record()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(args.steps):
        model.roofline_events = {"video": evs[i]}
        out = step()
    fence()
    dt = time.perf_counter() - t0
    model.roofline_events = None

    kv_step_ms = sum(a.elapsed_time(b) for a, b in evs) / len(evs)
    if world > 1:
        tmax = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dt = float(tmax.item())
    assert torch.isfinite(out["fused"]).all()
    value = n_total * args.steps / dt
    log(f"timed region done: {dt / args.steps * 1e3:.2f} ms/step")

    # ---- stand-alone timing of the K/V projection GEMM of the video Q-Former (the dominant kernel of the K/V-cache path; with the
    #      folded path it is reported beside the block's roofline as kv_cache_mode_gemm) ----
    enc = qf.modality_ln(feats["video"])
    nb = int(lib.mra_kv_cache_bytes(qf._handle, n_local, kv["video"]))
    cache = torch.empty(nb, dtype=torch.uint8, device=dev)
    reps = 5
    for _ in range(2):
        _lib.check(lib.mra_kv_project(qf._handle, _lib.ptr(enc), n_local, kv["video"], _lib.ptr(cache), _lib.current_stream()), "kv_project")
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()  # torch's current stream == the stream the launches go to
    for _ in range(reps):
        _lib.check(lib.mra_kv_project(qf._handle, _lib.ptr(enc), n_local, kv["video"], _lib.ptr(cache), _lib.current_stream()), "kv_project")
    e1.record()
    torch.cuda.synchronize()
    kv_ms = e0.elapsed_time(e1) / reps
    ncross, H, E = 6, 768, ENC_WIDTH["video"]
    kv_flops = 2.0 * n_local * kv["video"] * E * (ncross * 2 * H)
    Q, R = 32, 12 * 32
    folded = args.cross_precision == "split" or args.cross_mode in ("fold", "fold_stream", "fold384", "fold_rescale_pass") or (args.cross_mode == "auto" and kv["video"] >= 2048)
    if folded:
        # folded cross-attention: the library's event pair brackets the launches of cross layer 0.  Executed work = the
        # re-associated products actually run (that is what `frac` prices); algorithmic work = what the reference
        # formulation does for one layer (K and V projection of every token + the attention core).
        block_alg = 4.0 * n_local * kv["video"] * E * H + 4.0 * n_local * Q * kv["video"] * H
        block_exec = 4.0 * n_local * R * kv["video"] * E + 4.0 * n_local * Q * H * E
        if args.cross_precision == "split":   # Q' as (hi | lo): the scores product walks the slab twice; the two small projections run three-fold
            block_exec += 2.0 * n_local * R * kv["video"] * E + 4.0 * n_local * Q * H * E
        achieved = block_exec / (kv_step_ms * 1e-3) / 1e12
    else:
        block_alg = block_exec = kv_flops
        achieved = kv_flops / (kv_step_ms * 1e-3) / 1e12  # priced on the in-step launches (timed region)
    del cache, enc

    traffic = None
    pmc_file = os.path.join(ROOT, "profiles", "r02c_pmc_kvproj_p8.json")   # the eight-phase kernel; r01_pmc_kvproj_ws.json holds the loader-wave kernel's
    if not folded and args.workload == "clip32x32" and args.dtype == "f16" and n_local == 32 and os.path.exists(pmc_file):
        # HBM-side bytes per launch from a separate rocprofv3 --pmc pass of the same kernel and shape
        # (PMC cannot be collected from inside this process); see the file's _note for the gfx950 correction
        traffic = json.load(open(pmc_file)).get("hbm_bytes_per_launch")

    # PMC passes of the build in force: the in-register rescale (default) or the separate rescale pass (cross mode 5 = the r02 mid-round build)
    fold_pmc = os.path.join(ROOT, "profiles", "r02_pmc_fold.json" if args.cross_mode == "fold_rescale_pass" else "r03_pmc_fold.json")
    if not os.path.exists(fold_pmc):
        fold_pmc = os.path.join(ROOT, "profiles", "r02c_pmc_fold.json")
    traffic_src = "profiles/r02c_pmc_kvproj_p8.json (rocprofv3 --pmc FETCH_SIZE x2 + WRITE_SIZE, separate pass)" if traffic else None
    if folded and args.workload == "clip32x32" and args.dtype == "f16" and n_local == 32 and args.cross_precision == "op" and os.path.exists(fold_pmc):
        traffic = json.load(open(fold_pmc)).get("hbm_bytes_per_block")
        traffic_src = f"profiles/{os.path.basename(fold_pmc)} (rocprofv3 --pmc FETCH_SIZE x2 + WRITE_SIZE per kernel, separate passes; summed over the block)"

    flops_step = sum(getattr(model, f"{m}_Qformer").flops(n_local, L, kv[m], True) for m in ("video", "audio"))   # reference formulation
    flops_exec = flops_step - (ncross * (block_alg - block_exec) if folded else 0.0)

    # ---- the encode stage (row A1), timed SEPARATELY: EVA ViT-g over every frame of the step (mra_vit_forward, and stock PyTorch beside it) ----
    encode = None
    if rank == 0 and world == 1 and not args.no_encode and args.workload == "clip32x32":
        encode = time_encode(dev, n_local, 32, dt / args.steps)

    # ---- CPU baseline: the oracle (a CPU port of the same path) on a bounded sample ----
    cpu = None
    if rank == 0 and world == 1 and args.cpu_clips != 0:
        cpu = cpu_baseline(args, kv, L)

    if rank == 0:
        line = {
            "metric": "clips/s encode+fuse+score, 32-clip x 32-frame synthetic",
            "value": round(value, 2), "unit": "clips/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(dt / args.steps * 1e3, 3), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": args.dtype, "data": "synthetic",
            "config": {
                "workload": ("32 clips/GPU x 32 frames: video Kv=8224 x1408 + 10 s audio Kv=496 x768, L=32; "
                             "features resident in HBM (f16); LN + Q-Former(video) + Q-Former(audio) + cosine score + span"
                             if args.workload == "clip32x32" else
                             "32 items/GPU, reference item shape: video Kv=257 x1408 + audio Kv=256 x768, L=32; LN + 2 Q-Formers + score + span"),
                "clips_per_gpu": n_local, "global_clips": n_total, "text_len": L, "kv_video": kv["video"], "kv_audio": kv["audio"],
                "parallelism": f"clip-shard x{world}" + (" + RCCL all-gather of query embeddings" if world > 1 else ""),
                "encoders": "not part of value (synthetic features stand in for ViT-g / BEATs outputs); the ViT-g encode of the same frames is timed separately under encode_stage",
                "weights": "synthetic BERT init, seed 0",
                "cross_attention": "folded" if folded else "kv_cache", "cross_precision": args.cross_precision,
                "modalities": "one grouped launch sequence (mra_qformer_forward_pair)" if (args.pair and args.cross_precision == "op" and args.cross_mode != "fold_stream") else "two forwards on two streams",
            },
            "executed_tflops_per_gpu": round(flops_exec * args.steps / dt / 1e12, 1),
            "algorithmic_tflops_per_gpu": round(flops_step * args.steps / dt / 1e12, 1),
            "roofline": ({"bound": "mfma",
                          "kernel": ("folded cross-attention of one layer, video, the launches between the library's event pair (cross layer 0 of 6): "
                                     + ("per-head Q' GEMM (gemm_kernel<128,128>, K = 64) + Q' re-pack + fold_stream_kernel<scores> + row statistics + fold_stream_kernel<pv> "
                                        "+ per-head context GEMM" if args.cross_mode == "fold_stream" else
                                        "per-head Q' GEMM (gemm_kernel<128,128>, K = 64) + batched scores GEMM (gemm_ws_kernel<176x384, EPI_SOFTPART>: exp2(s - tile max) in f16 "
                                        "+ tile statistics) + " + ("softmax_rescale_kernel + batched P.enc GEMM (gemm_ws_kernel<176x384>, encoder tokens as the K-major operand)"
                                                                    if args.cross_mode == "fold_rescale_pass" else
                                                                    "fold_rowfactor_kernel + batched P.enc GEMM (gemm_ws_kernel<176x384, PSC>: encoder tokens as the K-major "
                                                                    "operand, row factors applied to the P~ fragments in registers)")
                                        + " + per-head context GEMM (gemm_k128_kernel<64,128>)")),
                          "achieved": round(achieved, 1), "peak": PEAK_F16_TFLOPS, "unit": "TFLOP/s",
                          "frac": round(achieved / PEAK_F16_TFLOPS, 4), "traffic": traffic, "traffic_source": traffic_src,
                          "avg_launch_ms": round(kv_step_ms, 4), "flops_per_launch": block_exec,
                          "note": "achieved / frac = EXECUTED flops of the block (the re-associated products) / its duration measured inside the timed steps; "
                                  "algorithmic_tflops = what the reference formulation (K/V projection of every token + attention core) would need for the same layer",
                          "algorithmic_flops_per_launch": block_alg, "algorithmic_tflops": round(block_alg / (kv_step_ms * 1e-3) / 1e12, 1),
                          "hbm": None if not traffic else {"achieved": round(traffic / (kv_step_ms * 1e-3) / 1e9, 1), "peak": 8000.0, "unit": "GB/s",
                                                          "frac": round(traffic / (kv_step_ms * 1e-3) / 1e9 / 8000.0, 4)},
                          "kv_cache_mode_gemm": {"kernel": "gemm_p8_kernel<EPI_KV> (eight-phase 256x256x64; what --cross-mode kv_cache runs instead)",
                                                 "standalone_launch_ms": round(kv_ms, 4),
                                                 "standalone_tflops": round(kv_flops / (kv_ms * 1e-3) / 1e12, 1)}}
                         if folded else
                         {"bound": "mfma", "kernel": "gemm_p8_kernel<EPI_KV> (eight-phase 256x256x64; K/V projection of all cross layers, video)",
                          "achieved": round(achieved, 1), "peak": PEAK_F16_TFLOPS, "unit": "TFLOP/s",
                          "frac": round(achieved / PEAK_F16_TFLOPS, 4), "traffic": traffic,
                          "traffic_source": traffic_src,
                          "avg_launch_ms": round(kv_step_ms, 4), "flops_per_launch": kv_flops,
                          "standalone_launch_ms": round(kv_ms, 4), "standalone_tflops": round(kv_flops / (kv_ms * 1e-3) / 1e12, 1)}),
            "cpu_baseline": cpu,
            "encode_stage": encode,
        }
        print(json.dumps(line), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


def time_encode(dev, clips, frames_per_clip, fuse_s):
    """The encode stage (row A1 / N4), timed SEPARATELY from `value`: EVA ViT-g/14 over the clips x frames 224 x 224 frames
    that one step's video features stand for -- on this build's kernels (mra_vit_forward: one batched pass, GEMMs with fused
    epilogues, 96-padded attention core) and, beside it, as stock PyTorch f16 (SDPA + hipBLASLt).  Random weights; there is
    no BEATs source in this image, audio features stay synthetic."""
    from mraudio_amd.models.eva_vit import create_eva_vit_g

    out = {}
    nframes, chunk = clips * frames_per_clip, clips * frames_per_clip   # every backend takes all frames of the step in one batched pass (its fastest setting)
    for backend in ("hip", "hip_separate_ln", "hip_f16_residual", "torch"):
        try:
            if backend.startswith("hip"):
                vit = create_eva_vit_g(224, 0, False, "fp16", backend="hip", device=dev, residual="op" if backend.endswith("residual") else "fp32",
                                       ln_fold=backend != "hip_separate_ln").eval().init_seeded_(0)
            else:
                with torch.device(dev):
                    vit = create_eva_vit_g(224, 0, False, "fp16").eval()
            x = torch.randn(chunk, 3, 224, 224, device=dev, dtype=torch.float16)
            with torch.no_grad():
                vit(x)
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                for _ in range(nframes // chunk):
                    y = vit(x)
                torch.cuda.synchronize()
                t = time.perf_counter() - t0
            assert y.shape == (chunk, 257, 1408)
            if not bool(torch.isfinite(y).all()):
                raise FloatingPointError(f"{backend}: non-finite encoder output")
            # spot check of the timed batch: frames 0 / 1 / middle / last re-encoded as a 4-frame batch (other tile kernels, other
            # positions in the batch) must reproduce their rows of the big batch
            pick = [0, 1, chunk // 2, chunk - 1]
            with torch.no_grad():
                small = vit(x[pick])
            spot = (small.float() - y[pick].float()).abs().max().item()
            spot_bar = 5e-2 if backend in ("hip", "hip_separate_ln") else 5e-1     # 39 blocks: f16-rounded intermediates on |y| ~ 30 (f16 residual / stock f16: coarser)
            if not spot < spot_bar:
                raise FloatingPointError(f"{backend}: 4-frame spot check differs from the timed batch by {spot:.3e}")
            fl = vit.flops_per_frame() * nframes
            out[backend] = {"what": ("EVA ViT-g/14 on mra_vit_forward (hand-written gfx950 kernels, f16 operands, fp32 residual, LayerNorms folded into the GEMMs around them)" if backend == "hip" else
                                     "the same with the LayerNorms as separate launches (round 2's form; mra_vit_set_option ln_fold 0)" if backend == "hip_separate_ln" else
                                     "the same with the residual stream in f16 (every add rounds to 16 bits, as LAVIS' precision=\"fp16\" encoder does)" if backend == "hip_f16_residual" else
                                     "EVA ViT-g/14, stock PyTorch f16 (SDPA + hipBLASLt)") + f", random weights, {nframes} frames in chunks of {chunk}",
                            "ms": round(t * 1e3, 1), "tflops": round(fl / t / 1e12, 1), "gflop_per_frame": round(vit_gf(fl, nframes), 1),
                            "finite": True, "spot_check_max_abs_diff_4_frames": round(spot, 5),
                            "clips_per_s_encode_only": round(clips / t, 2), "clips_per_s_encode_plus_fuse_score": round(clips / (t + fuse_s), 2)}
            del vit, x, y, small
            torch.cuda.empty_cache()
        except Exception as e:  # the encode stage is context, never a reason to lose the bench line
            out[backend] = {"error": repr(e)[:300]}
    return out


def vit_gf(total_flops, nframes):
    return total_flops / nframes / 1e9


def cpu_baseline(args, kv, L):
    """The oracle (kind "port": a CPU restatement in torch fp32, all host threads) on the same workload, bounded to a few
    clips so the default run stays within minutes.  Protocol (SURVEY 8d / BASELINE.md section 3): one warm-up pass, then the MEDIAN
    of FIVE timed passes over the sample; the reference item shape (Kv 257 / 256) is timed the same way beside it."""
    from mraudio_amd.models.xinstructblip import ENC_WIDTH
    from oracle import qformer_ref as O

    threads = host_threads()
    torch.set_num_threads(threads)
    log(f"cpu baseline: {threads} threads")
    cfgs = {m: O.QFormerCfg(enc_width=ENC_WIDTH[m]) for m in ("video", "audio")}
    ws = {"video": O.init_weights(cfgs["video"], seed=0), "audio": O.init_weights(cfgs["audio"], seed=1)}

    def timed(n, kvs, passes=5):
        g = torch.Generator().manual_seed(1234)
        feats = {m: torch.randn(n, kvs[m], ENC_WIDTH[m], generator=g) for m in ("audio", "video")}
        ids = torch.randint(1000, 30000, (n, L), generator=g)
        tm = torch.ones(n, L, dtype=torch.long)
        ts = []
        with torch.no_grad():
            O.encode_fuse_score(ws, cfgs, {m: feats[m][:2] for m in feats}, ids[:2], tm[:2], 1, 2)  # warm-up
            for _ in range(passes):
                t0 = time.perf_counter()
                O.encode_fuse_score(ws, cfgs, feats, ids, tm, 1, n)
                ts.append(time.perf_counter() - t0)
        return sorted(ts)[len(ts) // 2], ts

    # bounded samples: ~0.47 s per 32-frame clip on 16 threads -> 10 clips x 5 passes = ~24 s; reference items: 0.05 s each
    n = args.cpu_clips if args.cpu_clips > 0 else (10 if args.workload == "clip32x32" else 96)
    med, ts = timed(n, kv)
    out = {"value": round(n / med, 3), "unit": "clips/s", "cores": threads, "kind": "port",
           "sample": f"{n} clips of the same workload, torch fp32 oracle: 1 warm-up + 5 timed passes, median {med:.1f} s (passes {[round(t, 1) for t in ts]})"}
    if args.workload == "clip32x32":
        n_ref = 64
        med_r, ts_r = timed(n_ref, {"video": 257, "audio": 256})
        out["reference_item_shape"] = {"value": round(n_ref / med_r, 2), "unit": "items/s",
                                       "sample": f"{n_ref} items at the reference's own item shape (Kv 257 video + 256 audio), same protocol, median {med_r:.1f} s"}
    return out


if __name__ == "__main__":
    main()
