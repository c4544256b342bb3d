"""Optimizer for BASELINE config 5 (the Q-Formers unfrozen): ``torch.optim.Adam`` semantics with the Q-Former part of the step on the
HIP extension.

The reference's trainer builds ``torch.optim.AdamW`` over the parameters that require grad and steps it through a GradScaler
(``utils/trainer.py:60-69,124-140``); there the Q-Formers are frozen.  With them trainable, 372 M fp32 parameters make the optimizer
step ~20 % of the step time if it runs as ``torch.optim``'s multi-tensor kernels followed by the library's weight refresh and the
transposed-copy rebuild (2.7 + 0.4 + 1.0 ms measured in round 2).  ``FusedQFormerAdam`` keeps ``torch.optim.Adam``'s arithmetic
(bit-compatible up to fp32 rounding order, checked in ``tests/test_gpu_backward.py``) and runs it as ONE pass per Q-Former
(``mra_qformer_adam_step``): update + device-copy refresh + gradient clearing; parameters outside the Q-Formers (none by default) go to
a stock ``torch.optim.Adam``.
"""
from __future__ import annotations

from typing import Dict, Tuple

import torch


class FusedQFormerAdam:
    """``opt = FusedQFormerAdam(model, lr=1e-4); loss.backward(); opt.step(); opt.zero_grad()`` (``zero_grad`` is free: the step clears
    the flat gradient buffers on the way).  ``param_groups`` exposes ``lr`` so LR schedulers written against ``torch.optim`` work."""

    def __init__(self, model, lr: float = 1e-4, betas: Tuple[float, float] = (0.9, 0.999), eps: float = 1e-8, weight_decay: float = 0.0):
        if not getattr(model, "train_qformers", False):
            model.enable_qformer_training()
        self.model = model
        self.defaults = dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay)
        self.param_groups = [dict(self.defaults, params=[])]
        covered = set()
        self._qformers = []
        for m in model.modalities:
            qf = getattr(model, f"{m}_Qformer")
            self._qformers.append(qf)
            covered.update(id(p) for p in qf.parameters())
            covered.add(id(getattr(model, f"{m}_query_tokens")))
        rest = [p for p in model.parameters() if p.requires_grad and id(p) not in covered]
        self._rest = torch.optim.Adam(rest, lr=lr, betas=betas, eps=eps, weight_decay=weight_decay) if rest else None

    @torch.no_grad()
    def step(self) -> None:
        g = self.param_groups[0]
        for qf in self._qformers:
            qf.adam_step(g["lr"], g["betas"], g["eps"], g["weight_decay"], zero_grad=True)
        self.model._extras_dirty = True          # the query tokens (views of the master buffers) moved
        if self._rest is not None:
            for pg in self._rest.param_groups:
                pg["lr"] = g["lr"]
            self._rest.step()

    def zero_grad(self, set_to_none: bool = False) -> None:
        # the Q-Former gradient buffers were cleared by step(); before the first step (or after a backward without a step) clear them here
        for qf in self._qformers:
            if not getattr(qf, "_grads_zeroed", False) and getattr(qf, "_grad_flat", None) is not None:
                qf._grad_flat.zero_()
            qf._grads_zeroed = False
        if self._rest is not None:
            self._rest.zero_grad(set_to_none=set_to_none)

    def state_dict(self) -> Dict[str, object]:
        return {"t": [getattr(qf, "_adam_t", 0) for qf in self._qformers],
                "exp_avg": [getattr(qf, "_exp_avg", None) for qf in self._qformers],
                "exp_avg_sq": [getattr(qf, "_exp_avg_sq", None) for qf in self._qformers],
                "rest": self._rest.state_dict() if self._rest is not None else None, "param_groups": [{k: v for k, v in self.param_groups[0].items() if k != "params"}]}

    def load_state_dict(self, sd) -> None:
        for qf, t, m, v in zip(self._qformers, sd["t"], sd["exp_avg"], sd["exp_avg_sq"]):
            qf.enable_training()
            if m is not None:
                qf._exp_avg = m.to(qf._master_flat.device).clone()
                qf._exp_avg_sq = v.to(qf._master_flat.device).clone()
                qf._adam_t = int(t)
        if self._rest is not None and sd.get("rest") is not None:
            self._rest.load_state_dict(sd["rest"])
        self.param_groups[0].update(sd["param_groups"][0])
