"""Times BASELINE config 5: a finetune.py-shaped step with the Q-Formers unfrozen -- forward with tape,
backward, gradient all-reduce, Adam, weight re-upload -- on a synthetic Charades-STA-shaped batch
(B = 1 video, T = 20 positions, 257 ViT-g + 256 BEATs tokens per position, L = 32-ish prompt tokens).
Beyond the reference (its Q-Formers are frozen); not the bench.py headline.

    python tools/bench_finetune.py [--steps 10] [--dtype bf16]
    python -m torch.distributed.run --nproc-per-node N --master-addr 127.0.0.1 tools/bench_finetune.py
"""
import argparse
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--dtype", default="bf16", choices=["bf16", "f16"])
    ap.add_argument("--per-tensor-adam", action="store_true", help="A/B: ~800 per-tensor parameters instead of the two flat ones")
    ap.add_argument("--fused-adam", action="store_true", help="A/B: the one-pass mra_qformer_adam_step (update + device-copy refresh + transposed copies + gradient "
                                                              "clearing) instead of torch.optim.Adam(fused=True) on the flat parameters; measured r03f: 17.6 vs 17.3 ms per step")
    ap.add_argument("--streams", action="store_true", help="A/B: each modality's Q-Former forward + backward on its own stream (model.train_streams)")
    ap.add_argument("--train-ring", type=int, default=None, help="A/B: mra_qformer_set_option('train_ring', mask) on both Q-Formers")
    args = ap.parse_args(argv)
    world, rank, local = int(os.environ.get("WORLD_SIZE", "1")), int(os.environ.get("RANK", "0")), int(os.environ.get("LOCAL_RANK", "0"))
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    import torch.distributed as dist
    if world > 1 and not dist.is_initialized():
        dist.init_process_group("nccl", device_id=dev)
    from mraudio_amd.models.xinstructblip import XInstructBLIP

    model = XInstructBLIP(seed=0, perturb=False, op_dtype=torch.bfloat16 if args.dtype == "bf16" else torch.float16, device=dev)
    model.enable_qformer_training()
    model.train_streams = args.streams
    if args.train_ring is not None:
        for m in model.modalities:
            getattr(model, f"{m}_Qformer").set_option("train_ring", args.train_ring)
    g = torch.Generator().manual_seed(100 + rank)
    samples = {"video_embeds": torch.randn(1, 20, 257, 1408, generator=g).to(dev), "audio_embeds": torch.randn(1, 20, 256, 768, generator=g).to(dev),
               "text_input": ["Query: a person opens the door and walks in.\nGiven the video and the query, find the relevant windows.\nRelevant windows: "],
               "text_output": ["[[6, 12]]"], "timestamps": [list(range(0, 40, 2))], "duration": [40]}
    if args.per_tensor_adam:
        params = [p for d in model.get_optimizer_params(0.05) for p in d["params"] if p.shape[0] != 30523]
        opt = torch.optim.Adam(params, lr=1e-4, fused=True)
    elif not args.fused_adam:
        params = model.flat_optimizer_params()       # one flat fp32 parameter per Q-Former
        opt = torch.optim.Adam(params, lr=1e-4, fused=True)
    else:
        from mraudio_amd.utils.optim import FusedQFormerAdam
        opt = FusedQFormerAdam(model, lr=1e-4)       # update + device-copy refresh + gradient clearing in one pass per Q-Former
    t = {"fwd": 0.0, "bwd": 0.0, "allreduce": 0.0, "adam": 0.0}

    def tick():
        torch.cuda.synchronize()
        return time.perf_counter()

    for it in range(args.warmup + args.steps):
        if it == args.warmup:
            for k in t:
                t[k] = 0.0
            if world > 1:
                dist.barrier()
            t_all = tick()
        opt.zero_grad(set_to_none=True)
        a = tick(); loss = model(samples)["loss"]
        b = tick(); loss.backward()
        c = tick(); model.all_reduce_grads()
        d = tick(); opt.step()
        e = tick()
        t["fwd"] += b - a; t["bwd"] += c - b; t["allreduce"] += d - c; t["adam"] += e - d
    if world > 1:
        dist.barrier()
    total = tick() - t_all
    if rank == 0:
        print(json.dumps({"config": "BASELINE config 5: Q-Former fwd+bwd finetune step, B=1 x T=20 per GPU, both modalities", "n_gpus": world,
                          "dtype": args.dtype, "steps": args.steps, "ms_per_step": round(total / args.steps * 1e3, 2),
                          "steps_per_s_per_gpu": round(args.steps / total, 2), "clips_per_s": round(20 * world * args.steps / total, 1),
                          "optimizer": "torch.optim.Adam per tensor" if args.per_tensor_adam else ("mra_qformer_adam_step (FusedQFormerAdam)" if args.fused_adam else "torch.optim.Adam(fused) on flat parameters"),
                          "ms": {k: round(v / args.steps * 1e3, 2) for k, v in t.items()}, "loss": round(loss.item(), 4)}))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
