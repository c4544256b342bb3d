"""Workload for tools/pmc_fold.sh: the video Q-Former forward at the headline shape (32 clips, Kv 8224) with the
folded cross-attention, a few times, so rocprofv3 --pmc can attribute HBM-side bytes to its kernels."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mraudio_amd.qformer import QFormer, QFormerConfig, draw_seeded  # noqa: E402

dev = torch.device("cuda:0")
qf = QFormer(QFormerConfig(enc_width=1408), device=dev)
g = qf.init_seeded_(seed=0)
qf.push("query_tokens", draw_seeded(g, (1, 32, 768), "w", False))
qf.push("ln.weight", torch.ones(1408))
qf.push("ln.bias", torch.zeros(1408))
n, L, kv = 32, 32, 8224
gen = torch.Generator(device=dev).manual_seed(1)
ids = torch.randint(1000, 30000, (n, L), device=dev, generator=gen)
att = torch.ones(n, 32 + L, dtype=torch.long, device=dev)
enc = torch.randn(n, kv, 1408, device=dev, dtype=torch.float16, generator=gen)
qf.set_cross_mode("fold")
for _ in range(3):
    qf.forward_fused(ids, att, enc, want_query=True, want_cls=True)
torch.cuda.synchronize()
