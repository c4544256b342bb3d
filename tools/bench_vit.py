"""Times the EVA ViT-g encode stage (row A1 / N4) on the HIP extension; prints one JSON line.  For rocprofv3:
    rocprofv3 --kernel-trace --stats -d gpurun_out/prof_vit -- python3 tools/bench_vit.py --frames 256 --reps 2"""
import argparse
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mraudio_amd.models.eva_vit import create_eva_vit_g  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--frames", type=int, default=256)
ap.add_argument("--reps", type=int, default=4)
ap.add_argument("--backend", default="hip")
ap.add_argument("--residual", default="fp32", choices=["fp32", "op"])
ap.add_argument("--attn-persist", type=int, default=0, help="1: the attention core as persistent workgroups that prefetch the next unit (mra_vit_set_option attn_persist 1; measured slower)")
ap.add_argument("--gemm-persist", type=int, nargs="*", default=None, help="A/B: mra_vit_set_option('gemm_persist', v) values timed one after the other in this process")
ap.add_argument("--ln-fold", type=int, default=1, help="0: the blocks' LayerNorms as separate launches (mra_vit_set_option ln_fold 0)")
a = ap.parse_args()
dev = torch.device("cuda:0")
if a.backend == "hip":
    vit = create_eva_vit_g(224, 0, False, "fp16", backend="hip", device=dev, residual=a.residual, ln_fold=bool(a.ln_fold)).eval().init_seeded_(0)
else:
    with torch.device(dev):
        vit = create_eva_vit_g(224, 0, False, "fp16").eval()
if a.backend == "hip" and a.attn_persist:
    vit.set_option("attn_persist", 1)
x = torch.randn(a.frames, 3, 224, 224, device=dev, dtype=torch.float16)
if a.gemm_persist:
    with torch.no_grad():
        vit(x)
        ref = None
        for rnd in range(2):
            for v in a.gemm_persist:
                vit.set_option("gemm_persist", v)
                vit(x)
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                for _ in range(a.reps):
                    y = vit(x)
                torch.cuda.synchronize()
                t = (time.perf_counter() - t0) / a.reps
                ref = y.clone() if ref is None else ref
                print(json.dumps({"gemm_persist": v, "frames": a.frames, "ms": round(t * 1e3, 3), "tflops": round(vit.flops_per_frame() * a.frames / t / 1e12, 1),
                                  "max_abs_diff_vs_first": float((y - ref).abs().max())}), flush=True)
    sys.exit(0)
with torch.no_grad():
    vit(x)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(a.reps):
        vit(x)
    torch.cuda.synchronize()
    t = (time.perf_counter() - t0) / a.reps
fl = vit.flops_per_frame() * a.frames
print(json.dumps({"backend": a.backend, "residual": a.residual if a.backend == "hip" else "f16 (torch autocast-free half model)", "ln_fold": bool(a.ln_fold) if a.backend == "hip" else None, "frames": a.frames, "ms": round(t * 1e3, 2), "tflops": round(fl / t / 1e12, 1), "frames_per_s": round(a.frames / t, 1)}))
