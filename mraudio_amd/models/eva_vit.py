"""Stock-PyTorch EVA ViT-g/14 visual encoder (row A1 callee).

The reference builds it with LAVIS ``create_eva_vit_g(224, 0, False, "fp16")``
(``models/xinstructblip.py:658-666``) and calls it once per sampled frame (``:262-266``).  LAVIS is not
in this image, so this is a plain ``torch.nn`` restatement of the published EVA-CLIP-g geometry as LAVIS
instantiates it: 14x14 patches of a 224x224 frame (256 + [CLS] = 257 tokens), width 1408, 16 heads,
MLP 6144, **39** blocks (LAVIS drops the 40th), q/v bias without k bias, pre-LN blocks, no final
norm (the reference's separate ``video_ln`` plays that role, ``:664``).  Its arithmetic stays
PyTorch-ROCm (``F.scaled_dot_product_attention`` + rocBLAS/hipBLASLt GEMMs): the north star names no
ViT kernels.  It exists so that an ``evaluate.py``-shaped caller has a ``video_encoder`` and so that
``bench.py`` can report the encode stage beside the fused path.  Random weights only (no download).
"""
from __future__ import annotations

import torch
import torch.nn as nn
import torch.nn.functional as F


class _Attention(nn.Module):
    def __init__(self, dim: int, heads: int):
        super().__init__()
        self.heads = heads
        self.qkv = nn.Linear(dim, dim * 3, bias=False)
        self.q_bias = nn.Parameter(torch.zeros(dim))
        self.v_bias = nn.Parameter(torch.zeros(dim))
        self.proj = nn.Linear(dim, dim)

    def forward(self, x):
        b, n, c = x.shape
        bias = torch.cat((self.q_bias, torch.zeros_like(self.v_bias), self.v_bias))
        qkv = F.linear(x, self.qkv.weight, bias).reshape(b, n, 3, self.heads, c // self.heads).permute(2, 0, 3, 1, 4)
        o = F.scaled_dot_product_attention(qkv[0], qkv[1], qkv[2])
        return self.proj(o.transpose(1, 2).reshape(b, n, c))


class _Block(nn.Module):
    def __init__(self, dim: int, heads: int, mlp: int):
        super().__init__()
        self.norm1 = nn.LayerNorm(dim, eps=1e-6)
        self.attn = _Attention(dim, heads)
        self.norm2 = nn.LayerNorm(dim, eps=1e-6)
        self.fc1 = nn.Linear(dim, mlp)
        self.fc2 = nn.Linear(mlp, dim)

    def forward(self, x):
        x = x + self.attn(self.norm1(x))
        return x + self.fc2(F.gelu(self.fc1(self.norm2(x))))


class EvaViTg(nn.Module):
    """``encoder(frames [B, 3, 224, 224]) -> [B, 257, 1408]``; ``num_features = 1408`` (reference ``:123``)."""

    def __init__(self, img_size: int = 224, patch: int = 14, dim: int = 1408, depth: int = 39, heads: int = 16, mlp: int = 6144):
        super().__init__()
        self.num_features = dim
        self.patch_embed = nn.Conv2d(3, dim, kernel_size=patch, stride=patch)
        n = (img_size // patch) ** 2
        self.cls_token = nn.Parameter(torch.zeros(1, 1, dim))
        self.pos_embed = nn.Parameter(torch.zeros(1, n + 1, dim))
        self.blocks = nn.ModuleList([_Block(dim, heads, mlp) for _ in range(depth)])
        nn.init.trunc_normal_(self.pos_embed, std=0.02)
        nn.init.trunc_normal_(self.cls_token, std=0.02)

    def forward(self, x):
        x = self.patch_embed(x).flatten(2).transpose(1, 2)
        x = torch.cat((self.cls_token.expand(x.shape[0], -1, -1), x), dim=1) + self.pos_embed
        for blk in self.blocks:
            x = blk(x)
        return x

    # ---- weights ---------------------------------------------------------------------------------------------
    @torch.no_grad()
    def init_seeded_(self, seed: int = 0) -> "EvaViTg":
        """Seeded synthetic weights (no EVA checkpoint exists offline), drawn on the CPU in ``state_dict()`` order from one
        generator so every machine derives the same tensors: matrices / embeddings N(0, 0.02), biases N(0, 0.02),
        LayerNorm gains 1 + N(0, 0.1), LayerNorm biases N(0, 0.05) -- every parameter influences the output."""
        g = torch.Generator().manual_seed(seed)
        for k, p in self.state_dict().items():
            if "norm" in k:
                v = (1.0 + torch.randn(p.shape, generator=g) * 0.1) if k.endswith("weight") else torch.randn(p.shape, generator=g) * 0.05
            else:
                v = torch.randn(p.shape, generator=g) * 0.02
            p.copy_(v.to(p.dtype))
        return self

    def hf_state_dict(self) -> dict:
        """This module's weights under the key names of ``transformers.InstructBlipVisionModel`` (the in-image structural
        stand-in for LAVIS' EVA ViT-g).  Name map::

            cls_token                      embeddings.class_embedding            [1, 1, D]
            pos_embed                      embeddings.position_embedding         [1, 257, D]
            patch_embed.{weight,bias}      embeddings.patch_embedding.{weight,bias}
            blocks.i.norm1 / norm2         encoder.layers.i.layer_norm1 / layer_norm2
            blocks.i.attn.qkv.weight       encoder.layers.i.self_attn.qkv.weight [3D, D]
            blocks.i.attn.{q,v}_bias       encoder.layers.i.self_attn.qkv.bias = cat(q_bias, 0, v_bias)   (no key bias)
            blocks.i.attn.proj             encoder.layers.i.self_attn.projection
            blocks.i.fc1 / fc2             encoder.layers.i.mlp.fc1 / fc2
            (none)                         post_layernorm   -- the reference's separate ``video_ln`` plays that role
        """
        sd = {"embeddings.class_embedding": self.cls_token, "embeddings.position_embedding": self.pos_embed,
              "embeddings.patch_embedding.weight": self.patch_embed.weight, "embeddings.patch_embedding.bias": self.patch_embed.bias}
        for i, b in enumerate(self.blocks):
            p = f"encoder.layers.{i}."
            sd[p + "self_attn.qkv.weight"] = b.attn.qkv.weight
            sd[p + "self_attn.qkv.bias"] = torch.cat((b.attn.q_bias, torch.zeros_like(b.attn.v_bias), b.attn.v_bias))
            for a, c in (("self_attn.projection", b.attn.proj), ("layer_norm1", b.norm1), ("layer_norm2", b.norm2), ("mlp.fc1", b.fc1), ("mlp.fc2", b.fc2)):
                sd[p + a + ".weight"], sd[p + a + ".bias"] = c.weight, c.bias
        return {k: v.detach() for k, v in sd.items()}

    @torch.no_grad()
    def load_hf_state_dict(self, sd: dict) -> None:
        """Inverse of ``hf_state_dict``: accepts an ``InstructBlipVisionModel`` state dict (``post_layernorm.*`` is skipped: it
        belongs to ``video_ln``).  The HF key-bias slice must be zero, as EVA has no key bias."""
        D = self.num_features
        self.cls_token.copy_(sd["embeddings.class_embedding"]); self.pos_embed.copy_(sd["embeddings.position_embedding"])
        self.patch_embed.weight.copy_(sd["embeddings.patch_embedding.weight"]); self.patch_embed.bias.copy_(sd["embeddings.patch_embedding.bias"])
        for i, b in enumerate(self.blocks):
            p = f"encoder.layers.{i}."
            b.attn.qkv.weight.copy_(sd[p + "self_attn.qkv.weight"])
            bias = sd[p + "self_attn.qkv.bias"]
            if bool(bias[D:2 * D].abs().max() > 0):
                raise ValueError(f"layer {i}: non-zero key bias cannot be represented (EVA ViT-g has q / v bias only)")
            b.attn.q_bias.copy_(bias[:D]); b.attn.v_bias.copy_(bias[2 * D:])
            for a, c in (("self_attn.projection", b.attn.proj), ("layer_norm1", b.norm1), ("layer_norm2", b.norm2), ("mlp.fc1", b.fc1), ("mlp.fc2", b.fc2)):
                c.weight.copy_(sd[p + a + ".weight"]); c.bias.copy_(sd[p + a + ".bias"])

    def flops_per_frame(self) -> float:
        n, d = self.pos_embed.shape[1], self.num_features
        mlp = self.blocks[0].fc1.out_features
        per_block = 2 * n * d * 3 * d + 4 * n * n * d + 2 * n * d * d + 4 * n * d * mlp
        return float(len(self.blocks) * per_block + 2 * (n - 1) * 3 * 14 * 14 * d)


class HipEvaViTg(EvaViTg):
    """The same encoder on the HIP extension (``mra_vit_*``, ``mraudio_amd/csrc/vit.hip``): this module is the parameter
    container (state_dict keys unchanged); ``forward`` runs ALL given frames as one batched pass of hand-written gfx950
    kernels -- the four GEMMs per block on the eight-phase MFMA kernels with bias / GELU / residual fused, a 96-padded
    attention core -- and returns ``[n, 257, 1408]``: fp32 with the default fp32 residual stream, the operand dtype with
    ``residual="op"`` (every residual add rounds to 16 bits, as LAVIS' ``precision="fp16"`` encoder does).  The two
    LayerNorms of a block are folded into the GEMMs around them (``ln_fold``, ``mra_vit_set_option``; ``ln_fold=False`` runs them
    as separate launches).  No CPU path."""

    def __init__(self, *args, op_dtype: torch.dtype = torch.float16, residual: str = "fp32", device=None, ln_fold: bool = True, **kw):
        super().__init__(*args, **kw)
        import ctypes as C

        from .. import _lib
        self._lib, self._C = _lib, C
        self._device = torch.device(device) if device is not None else torch.device("cuda", torch.cuda.current_device())
        self._handle = C.c_void_p()
        blk = self.blocks[0]
        cfg = _lib.mra_vit_cfg(self.num_features, blk.attn.heads, blk.fc1.out_features, len(self.blocks), self.patch_embed.kernel_size[0],
                               self.patch_embed.kernel_size[0] * int(round((self.pos_embed.shape[1] - 1) ** 0.5)), 1e-6,
                               _lib.MRA_BF16 if op_dtype == torch.bfloat16 else _lib.MRA_F16,
                               _lib.MRA_F32 if residual == "fp32" else (_lib.MRA_BF16 if op_dtype == torch.bfloat16 else _lib.MRA_F16))
        if residual not in ("fp32", "op"):
            raise ValueError("residual must be 'fp32' (default) or 'op' (the operand dtype: the reference's precision='fp16' semantics)")
        self._out_dtype = torch.float32 if residual == "fp32" else op_dtype
        with torch.cuda.device(self._device):
            _lib.check(_lib.lib().mra_vit_create(C.byref(cfg), C.byref(self._handle)), "mra_vit_create")
        self._dirty, self._ws = True, None
        self.to(self._device)
        if not ln_fold:
            self.set_option("ln_fold", 0)

    def set_option(self, name: str, value: int) -> None:
        """Per-handle switch of the HIP encoder (``mra_vit_set_option``): ``"ln_fold"`` 0 / 1."""
        self._lib.check(self._lib.lib().mra_vit_set_option(self._handle, name.encode(), int(value)), f"mra_vit_set_option({name})")

    def _apply(self, fn, recurse=True):
        out = super()._apply(fn, recurse)
        self._dirty = True
        return out

    def load_state_dict(self, *a, **kw):
        res = super().load_state_dict(*a, **kw)
        self._dirty = True
        return res

    def __del__(self):
        try:
            if self._handle:
                self._lib.lib().mra_vit_destroy(self._handle)
                self._handle = self._C.c_void_p()
        except Exception:
            pass

    def sync_weights(self) -> None:
        ver = sum(p._version for p in self.parameters())
        if ver != getattr(self, "_ver", None):
            self._ver, self._dirty = ver, True
        if not self._dirty:
            return
        lib, C = self._lib, self._C
        with torch.cuda.device(self._device):
            for k, v in self.state_dict().items():
                t = v.detach().to(self._device)
                t = (t if t.dtype in (torch.float32, torch.float16, torch.bfloat16) else t.float()).contiguous()
                shape = (C.c_int64 * max(t.dim(), 1))(*t.shape)
                lib.check(lib.lib().mra_vit_load(self._handle, k.encode(), lib.ptr(t), lib.mra_dtype(t.dtype), shape, t.dim(), lib.current_stream()),
                          f"mra_vit_load({k})")
        self._dirty = False

    @torch.no_grad()
    def forward(self, x):
        lib = self._lib
        self.sync_weights()
        x = x.to(self._device)
        if x.dtype not in (torch.float32, torch.float16):
            x = x.float()
        x = x.contiguous()
        n = int(x.shape[0])
        out = torch.empty(n, self.pos_embed.shape[1], self.num_features, dtype=self._out_dtype, device=self._device)
        if n == 0:
            return out
        with torch.cuda.device(self._device):
            nbytes = (int(lib.lib().mra_vit_workspace_bytes(self._handle, n)) + 255) // 256 * 256
            if self._ws is None or self._ws.numel() < nbytes:
                self._ws = None
                self._ws = torch.empty(nbytes, dtype=torch.uint8, device=self._device)
            lib.check(lib.lib().mra_vit_forward(self._handle, lib.ptr(x), lib.mra_dtype(x.dtype), n, lib.ptr(out), lib.ptr(self._ws), self._ws.numel(),
                                                lib.current_stream()), "mra_vit_forward")
        return out

    def flops(self, frames: int) -> float:
        return float(self._lib.lib().mra_vit_flops(self._handle, frames))


def create_eva_vit_g(img_size=224, drop_path_rate=0.0, use_checkpoint=False, precision="fp16", backend: str = "torch", **kw) -> EvaViTg:
    """Same call shape as LAVIS' factory (the reference passes exactly the first four, ``:660-662``).  ``backend="hip"``
    gives the encoder on this build's kernels (f16 MFMA operands = the reference's ``precision="fp16"``, fp32 residual)."""
    if backend == "hip":
        return HipEvaViTg(img_size=img_size, op_dtype=torch.float16 if precision == "fp16" else torch.bfloat16, **kw)
    m = EvaViTg(img_size=img_size, **kw)
    return m.half() if precision == "fp16" else m
