"""Host-side mirror of the LAVIS Q-Former the reference drives, executing on the HIP extension.

Reference interface kept (``models/xinstructblip.py:286-293``)::

    out = model.video_Qformer.bert(input_ids, attention_mask=..., query_embeds=...,
                                   encoder_hidden_states=..., encoder_attention_mask=...,
                                   return_dict=True)
    out.last_hidden_state            # [N, 32 + L, 768]

``QFormer`` is an ``nn.Module`` whose parameter tree reproduces the LAVIS checkpoint key names
(``bert.embeddings.word_embeddings.weight``, ``bert.encoder.layer.{i}.attention.self.query.weight``
... reference loader ``models/xinstructblip.py:644-653``) so ``state_dict()`` /
``load_state_dict()`` / ``.to()`` behave as the reference's callers expect.  The modules are
parameter containers only: all arithmetic happens in ``libmra_hip.so``; torch provides device
memory and the stream.  There is no CPU path.
"""
from __future__ import annotations

import ctypes as C
import weakref
from dataclasses import dataclass
from typing import Dict, Iterable, List, Optional, Tuple

import torch
import torch.nn as nn

from . import _lib
from ._lib import MRA_BF16, MRA_F16, MRA_F32, MraError, check, current_stream, lib, mra_cfg, ptr


@dataclass
class QFormerConfig:
    """BERT-base + cross-attention every 2nd layer (reference ``models/xinstructblip.py:614-627``)."""

    hidden: int = 768
    heads: int = 12
    inter: int = 3072
    layers: int = 12
    cross_freq: int = 2
    enc_width: int = 1408
    n_query: int = 32
    vocab: int = 30523
    max_pos: int = 512
    ln_eps: float = 1e-12
    enc_ln_eps: float = 1e-5
    llm_hidden: int = 4096
    op_dtype: torch.dtype = torch.float16  # MFMA operand type; accumulation/residual/LN stay fp32

    def to_c(self) -> mra_cfg:
        return mra_cfg(self.hidden, self.heads, self.inter, self.layers, self.cross_freq, self.enc_width, self.n_query,
                       self.vocab, self.max_pos, self.ln_eps, self.enc_ln_eps, self.llm_hidden,
                       MRA_BF16 if self.op_dtype == torch.bfloat16 else MRA_F16)


class _Lin(nn.Module):
    def __init__(self, out_f: int, in_f: int):
        super().__init__()
        self.weight = nn.Parameter(torch.empty(out_f, in_f), requires_grad=False)
        self.bias = nn.Parameter(torch.empty(out_f), requires_grad=False)


class _LN(nn.Module):
    def __init__(self, n: int):
        super().__init__()
        self.weight = nn.Parameter(torch.empty(n), requires_grad=False)
        self.bias = nn.Parameter(torch.empty(n), requires_grad=False)


class _Emb(nn.Module):
    def __init__(self, n: int, d: int):
        super().__init__()
        self.weight = nn.Parameter(torch.empty(n, d), requires_grad=False)


class _Wrap(nn.Module):
    def __init__(self, **children):
        super().__init__()
        for k, v in children.items():
            self.add_module(k, v)


def _attention(h: int, kv_width: int) -> nn.Module:
    self_ = _Wrap(query=_Lin(h, h), key=_Lin(h, kv_width), value=_Lin(h, kv_width))
    out = _Wrap(dense=_Lin(h, h), LayerNorm=_LN(h))
    m = nn.Module()
    m.add_module("self", self_)
    m.add_module("output", out)
    return m


class _Bert(nn.Module):
    """``qformer.bert``: parameter container named like LAVIS' ``BertModel`` whose call signature is
    the one the reference uses (``models/xinstructblip.py:286-293``)."""

    def __init__(self, embeddings: nn.Module, encoder: nn.Module):
        super().__init__()
        self.embeddings = embeddings
        self.encoder = encoder
        object.__setattr__(self, "_owner_ref", None)

    def forward(self, input_ids=None, attention_mask=None, position_ids=None, query_embeds=None,
                encoder_hidden_states=None, encoder_attention_mask=None, return_dict=True, **unused):
        o = self._owner_ref()
        if encoder_hidden_states is None:
            raise ValueError("encoder_hidden_states must be given for cross-attention layers")
        if query_embeds is None:
            raise ValueError("You have to specify query_embeds")
        if position_ids is not None:
            raise NotImplementedError("custom position_ids are not supported (the reference never passes them)")
        if encoder_attention_mask is not None and not bool(encoder_attention_mask.to(torch.bool).all()):
            raise NotImplementedError("encoder_attention_mask with zeros is not supported: the reference always "
                                      "passes all ones (models/xinstructblip.py:266,275)")
        res = o.forward_fused(input_ids, attention_mask, encoder_hidden_states, query_embeds=query_embeds,
                              want_query=False, want_full=True)
        out = QFormerOutput(res["full"])
        return out if return_dict else (out.last_hidden_state,)


def _bert_tree(cfg: QFormerConfig) -> "_Bert":
    emb = _Wrap(word_embeddings=_Emb(cfg.vocab, cfg.hidden), position_embeddings=_Emb(cfg.max_pos, cfg.hidden),
                LayerNorm=_LN(cfg.hidden))
    layers = nn.ModuleList()
    for i in range(cfg.layers):
        lay = nn.Module()
        lay.add_module("attention", _attention(cfg.hidden, cfg.hidden))
        if i % cfg.cross_freq == 0:
            lay.add_module("crossattention", _attention(cfg.hidden, cfg.enc_width))
        lay.add_module("intermediate", _Wrap(dense=_Lin(cfg.inter, cfg.hidden)))
        lay.add_module("output", _Wrap(dense=_Lin(cfg.hidden, cfg.inter), LayerNorm=_LN(cfg.hidden)))
        lay.add_module("intermediate_query", _Wrap(dense=_Lin(cfg.inter, cfg.hidden)))
        lay.add_module("output_query", _Wrap(dense=_Lin(cfg.hidden, cfg.inter), LayerNorm=_LN(cfg.hidden)))
        layers.append(lay)
    return _Bert(emb, _Wrap(layer=layers))


def seeded_parameter_order(cfg: QFormerConfig) -> List[Tuple[str, str]]:
    """(key, kind) in the order the seeded synthetic init draws tensors; kind in w/b/g/z
    (matrix, bias, LayerNorm gain, LayerNorm bias).  Same order as the checker's recipe."""
    out = [("bert.embeddings.word_embeddings.weight", "w"), ("bert.embeddings.position_embeddings.weight", "w"),
           ("bert.embeddings.LayerNorm.weight", "g"), ("bert.embeddings.LayerNorm.bias", "z")]
    for i in range(cfg.layers):
        p = f"bert.encoder.layer.{i}."
        blocks = ["attention"] + (["crossattention"] if i % cfg.cross_freq == 0 else [])
        for blk in blocks:
            for n in ("query", "key", "value"):
                out += [(p + f"{blk}.self.{n}.weight", "w"), (p + f"{blk}.self.{n}.bias", "b")]
            out += [(p + f"{blk}.output.dense.weight", "w"), (p + f"{blk}.output.dense.bias", "b"),
                    (p + f"{blk}.output.LayerNorm.weight", "g"), (p + f"{blk}.output.LayerNorm.bias", "z")]
        for suf in ("", "_query"):
            out += [(p + f"intermediate{suf}.dense.weight", "w"), (p + f"intermediate{suf}.dense.bias", "b"),
                    (p + f"output{suf}.dense.weight", "w"), (p + f"output{suf}.dense.bias", "b"),
                    (p + f"output{suf}.LayerNorm.weight", "g"), (p + f"output{suf}.LayerNorm.bias", "z")]
    return out


def draw_seeded(gen: torch.Generator, shape, kind: str, perturb: bool) -> torch.Tensor:
    """One tensor of the synthetic init: matrices N(0, 0.02) (``initializer_range``, reference
    ``models/xinstructblip.py:627``); biases 0, LayerNorm 1/0 -- or, with ``perturb``, biases
    N(0, 0.02), gains 1 + N(0, 0.1), LayerNorm biases N(0, 0.05) so every parameter matters."""
    if kind == "w":
        return torch.randn(shape, generator=gen, dtype=torch.float32) * 0.02
    if kind == "b":
        return torch.randn(shape, generator=gen) * 0.02 if perturb else torch.zeros(shape)
    if kind == "g":
        return 1.0 + torch.randn(shape, generator=gen) * 0.1 if perturb else torch.ones(shape)
    return torch.randn(shape, generator=gen) * 0.05 if perturb else torch.zeros(shape)


class QFormerOutput:
    """What the reference reads from the LAVIS model output (``.last_hidden_state``, ``:303``)."""

    def __init__(self, last_hidden_state: torch.Tensor):
        self.last_hidden_state = last_hidden_state
        self.pooler_output = None

    def __getitem__(self, i):
        return (self.last_hidden_state,)[i]


class QFormer(nn.Module):
    """One modality Q-Former on one GPU (one ``mra_qformer`` handle)."""

    def __init__(self, cfg: QFormerConfig, device: Optional[torch.device] = None):
        super().__init__()
        self.cfg = cfg
        self.bert = _bert_tree(cfg)
        object.__setattr__(self.bert, "_owner_ref", weakref.ref(self))
        self._handle = C.c_void_p()
        self._device = torch.device(device) if device is not None else torch.device("cuda", torch.cuda.current_device())
        self._ws: Optional[torch.Tensor] = None
        self._dirty = True
        with torch.cuda.device(self._device):
            c = cfg.to_c()
            check(lib().mra_qformer_create(C.byref(c), C.byref(self._handle)), "mra_qformer_create")
        self.bert.to(self._device)

    def load_state_dict(self, state_dict, strict: bool = True, assign: bool = False):
        # LAVIS checkpoints also carry the position_ids buffer and the LM head the reference deletes
        # (``modality_qformer.cls = None``, models/xinstructblip.py:135)
        sd = {k: v for k, v in state_dict.items() if k != "bert.embeddings.position_ids" and not k.startswith("cls.")}
        res = super().load_state_dict(sd, strict=strict, assign=assign)
        self._dirty = True
        return res

    def _apply(self, fn, recurse=True):
        out = super()._apply(fn, recurse)
        self._dirty = True
        return out

    def __del__(self):
        try:
            if self._handle:
                lib().mra_qformer_destroy(self._handle)
                self._handle = C.c_void_p()
        except Exception:
            pass

    # ---- weights -> handle ---------------------------------------------------------------------
    def push(self, name: str, t: torch.Tensor) -> None:
        """Copy one tensor into the handle under its ABI name (``include/mra.h`` mra_qformer_load)."""
        t = t.detach()
        if t.device != self._device:
            t = t.to(self._device)
        if t.dtype not in (torch.float32, torch.float16, torch.bfloat16):
            t = t.float()
        t = t.contiguous()
        shape = (C.c_int64 * max(t.dim(), 1))(*t.shape)
        with torch.cuda.device(self._device):
            check(lib().mra_qformer_load(self._handle, name.encode(), ptr(t), _lib.mra_dtype(t.dtype), shape, t.dim(),
                                         current_stream()), f"mra_qformer_load({name})")

    def sync_weights(self) -> None:
        # in-place updates (an optimizer step) bump the tensors' version counters
        ver = sum(p._version for p in self.bert.parameters())
        if ver != getattr(self, "_param_version", None):
            self._param_version = ver
            self._dirty = True
        if not self._dirty:
            return
        if getattr(self, "_master_flat", None) is not None:
            # training: the parameters are views of one flat f32 master buffer -> one launch refreshes all
            self._bind_master()
            with torch.cuda.device(self._device):
                check(lib().mra_qformer_load_flat(self._handle, ptr(self._master_flat), self._master_flat.numel() * 4, current_stream()),
                      "mra_qformer_load_flat")
        else:
            for k, v in self.bert.state_dict(prefix="bert.").items():
                self.push(k, v)
        self._dirty = False

    def missing(self) -> List[str]:
        buf = C.create_string_buffer(1 << 16)
        n = lib().mra_qformer_missing(self._handle, buf, len(buf))
        return [s for s in buf.value.decode().split(",") if s] if n else []

    @torch.no_grad()
    def init_seeded_(self, seed: int = 0, perturb: bool = False, gen: Optional[torch.Generator] = None) -> torch.Generator:
        """Synthetic weights (no pretrained checkpoints exist offline).  Drawn on the CPU from one
        generator in ``seeded_parameter_order`` so that every machine derives the same tensors."""
        g = gen if gen is not None else torch.Generator().manual_seed(seed)
        sd = self.bert.state_dict(prefix="bert.", keep_vars=True)
        for key, kind in seeded_parameter_order(self.cfg):
            p = sd[key]
            p.data.copy_(draw_seeded(g, tuple(p.shape), kind, perturb))
        self._dirty = True
        return g

    # ---- ops -----------------------------------------------------------------------------------
    def _workspace(self, nbytes: int) -> torch.Tensor:
        if self._ws is None or self._ws.numel() < nbytes:
            self._ws = None
            self._ws = torch.empty(nbytes, dtype=torch.uint8, device=self._device)
        return self._ws

    def modality_ln(self, x: torch.Tensor, item_index: Optional[torch.Tensor] = None, items: Optional[int] = None) -> torch.Tensor:
        """A2 + A3: ``ln(encoder(frame))`` then ``cat(embeds)[indices]`` (reference ``:265,281-285``).
        x [src_items, tokens, E] f32/f16/bf16 -> [items, tokens, E] operand dtype."""
        if x.dim() != 3 or x.shape[-1] != self.cfg.enc_width:
            raise MraError(f"modality_ln expects [items, tokens, {self.cfg.enc_width}], got {tuple(x.shape)}")
        x = x.contiguous()
        n = int(items if items is not None else (item_index.numel() if item_index is not None else x.shape[0]))
        if item_index is not None:
            item_index = item_index.to(device=x.device, dtype=torch.int64).contiguous()
        out = torch.empty((n, x.shape[1], x.shape[2]), dtype=self.cfg.op_dtype, device=x.device)
        with torch.cuda.device(self._device):
            check(lib().mra_modality_ln(self._handle, ptr(x), _lib.mra_dtype(x.dtype), ptr(item_index), n, x.shape[1],
                                        ptr(out), current_stream()), "mra_modality_ln")
        return out

    def forward_fused(self, input_ids: Optional[torch.Tensor], attention_mask: Optional[torch.Tensor], enc: torch.Tensor,
                      query_embeds: Optional[torch.Tensor] = None, want_query: bool = True, want_full: bool = False,
                      want_cls: bool = False, kv_events=None) -> Dict[str, torch.Tensor]:
        """A4 on the extension.  enc [N, Kv, E] in the operand dtype (``modality_ln`` output).
        Returns a dict with ``query`` [N,32,H], ``full`` [N,32+L,H], ``cls`` [N,H] (fp32) as requested.
        ``kv_events``: optional (start, stop) ``torch.cuda.Event`` pair for ``mra_qformer_set_kv_events`` (bench
        instrumentation of the dominant kernel, recorded by the library on the launch stream)."""
        self.sync_weights()
        cfg = self.cfg
        if enc.dim() != 3 or enc.shape[-1] != cfg.enc_width:
            raise MraError(f"encoder_hidden_states must be [N, Kv, {cfg.enc_width}], got {tuple(enc.shape)}")
        if enc.dtype != cfg.op_dtype:
            enc = enc.to(cfg.op_dtype)
        enc = enc.contiguous()
        N, Kv = int(enc.shape[0]), int(enc.shape[1])
        L = 0 if input_ids is None else int(input_ids.shape[1])
        dev = enc.device
        if input_ids is not None:
            if input_ids.shape[0] != N:
                raise MraError(f"input_ids has {input_ids.shape[0]} rows, encoder_hidden_states {N}")
            input_ids = input_ids.to(device=dev, dtype=torch.int64).contiguous()
        if attention_mask is not None:
            if tuple(attention_mask.shape) != (N, cfg.n_query + L):
                raise MraError(f"attention_mask must be [{N}, {cfg.n_query + L}], got {tuple(attention_mask.shape)}")
            attention_mask = attention_mask.to(device=dev, dtype=torch.int64).contiguous()
        q_items = 0
        if query_embeds is not None:
            if query_embeds.dim() != 3 or tuple(query_embeds.shape[1:]) != (cfg.n_query, cfg.hidden) or query_embeds.shape[0] not in (1, N):
                raise MraError(f"query_embeds must be [1 or {N}, {cfg.n_query}, {cfg.hidden}]")
            query_embeds = query_embeds.to(device=dev, dtype=torch.float32).contiguous()
            q_items = int(query_embeds.shape[0])
        out: Dict[str, torch.Tensor] = {}
        if N == 0:
            return {"query": torch.empty(0, cfg.n_query, cfg.hidden, device=dev), "full": torch.empty(0, cfg.n_query + L, cfg.hidden, device=dev),
                    "cls": torch.empty(0, cfg.hidden, device=dev)}
        if want_query:
            out["query"] = torch.empty(N, cfg.n_query, cfg.hidden, dtype=torch.float32, device=dev)
        if want_full:
            out["full"] = torch.empty(N, cfg.n_query + L, cfg.hidden, dtype=torch.float32, device=dev)
        if want_cls:
            out["cls"] = torch.empty(N, cfg.hidden, dtype=torch.float32, device=dev)
        with torch.cuda.device(self._device):
            nbytes = (int(lib().mra_qformer_workspace_bytes(self._handle, N, L, Kv)) + 255) // 256 * 256
            ws = self._workspace(nbytes)
            if kv_events is not None:
                e0, e1 = kv_events
                check(lib().mra_qformer_set_kv_events(self._handle, e0.cuda_event, e1.cuda_event), "set_kv_events")
            try:
                check(lib().mra_qformer_forward(
                    self._handle, ptr(input_ids), ptr(attention_mask), ptr(query_embeds), q_items, ptr(enc), N, L, Kv,
                    ptr(out.get("query")), ptr(out.get("full")), ptr(out.get("cls")), ptr(ws), nbytes, current_stream()),
                    "mra_qformer_forward")
            finally:
                if kv_events is not None:
                    check(lib().mra_qformer_set_kv_events(self._handle, None, None), "set_kv_events")
        return out

    @staticmethod
    def forward_pair(qf0: "QFormer", qf1: "QFormer", input_ids: Optional[torch.Tensor], attention_mask: Optional[torch.Tensor], enc0: torch.Tensor,
                     enc1: torch.Tensor, want_cls: bool = True, kv_events=None):
        """Both modality Q-Formers of a step in ONE launch sequence (``mra_qformer_forward_pair``): the layer chains as grouped launches,
        each lane its own cross-attention.  enc0 / enc1 [N, Kv_l, E_l] in the operand dtype, the same prompt rows for both lanes.  Returns
        ``[(query0, cls0), (query1, cls1)]`` (fp32; cls ``None`` without ``want_cls``).  ``kv_events``: (start, stop) pair for lane 0's
        cross-layer-0 block (bench instrumentation)."""
        qfs, encs = (qf0, qf1), []
        for qf, enc in zip(qfs, (enc0, enc1)):
            qf.sync_weights()
            if enc.dim() != 3 or enc.shape[-1] != qf.cfg.enc_width:
                raise MraError(f"encoder_hidden_states must be [N, Kv, {qf.cfg.enc_width}], got {tuple(enc.shape)}")
            encs.append(enc.to(qf.cfg.op_dtype).contiguous())
        N = int(encs[0].shape[0])
        if int(encs[1].shape[0]) != N:
            raise MraError("forward_pair: both lanes need the same number of items")
        dev = encs[0].device
        L = 0 if input_ids is None else int(input_ids.shape[1])
        if input_ids is not None:
            input_ids = input_ids.to(device=dev, dtype=torch.int64).contiguous()
        if attention_mask is not None:
            attention_mask = attention_mask.to(device=dev, dtype=torch.int64).contiguous()
        H, Q = qf0.cfg.hidden, qf0.cfg.n_query
        outs = [(torch.empty(N, Q, H, dtype=torch.float32, device=dev), torch.empty(N, H, dtype=torch.float32, device=dev) if want_cls else None) for _ in qfs]
        if N == 0:
            return outs
        with torch.cuda.device(qf0._device):
            nbytes = (int(lib().mra_qformer_pair_workspace_bytes(qf0._handle, qf1._handle, N, L, int(encs[0].shape[1]), int(encs[1].shape[1]))) + 255) // 256 * 256
            ws = qf0._workspace(nbytes)
            if kv_events is not None:
                check(lib().mra_qformer_set_kv_events(qf0._handle, kv_events[0].cuda_event, kv_events[1].cuda_event), "set_kv_events")
            try:
                check(lib().mra_qformer_forward_pair(qf0._handle, qf1._handle, ptr(input_ids), ptr(attention_mask), ptr(encs[0]), ptr(encs[1]), N, L,
                                                     int(encs[0].shape[1]), int(encs[1].shape[1]), ptr(outs[0][0]), ptr(outs[0][1]), ptr(outs[1][0]), ptr(outs[1][1]),
                                                     ptr(ws), nbytes, current_stream()), "mra_qformer_forward_pair")
            finally:
                if kv_events is not None:
                    check(lib().mra_qformer_set_kv_events(qf0._handle, None, None), "set_kv_events")
        return outs

    def llm_proj(self, z: torch.Tensor, out_dtype: torch.dtype = torch.float32) -> torch.Tensor:
        """A5: ``{modality}_llm_proj(last_hidden_state[:, :32, :])`` (reference ``:303``)."""
        cfg = self.cfg
        zz = z.reshape(-1, cfg.hidden).to(torch.float32).contiguous()
        rows = int(zz.shape[0])
        out = torch.empty(rows, cfg.llm_hidden, dtype=out_dtype, device=zz.device)
        ws = self._workspace(max(256, rows * cfg.hidden * 2 + 256))
        with torch.cuda.device(self._device):
            check(lib().mra_llm_proj(self._handle, ptr(zz), rows, ptr(out), _lib.mra_dtype(out_dtype), ptr(ws), ws.numel(),
                                     current_stream()), "mra_llm_proj")
        return out.reshape(*z.shape[:-1], cfg.llm_hidden)

    def set_cross_mode(self, mode) -> None:
        """How the cross-attention layers run: ``"auto"`` (folded from Kv >= 2048), ``"kv_cache"``, ``"fold"``, or the
        A/B formulations ``"fold384"`` (128 x 384 tiles) / ``"fold_stream"`` (streaming kernels, f16 only)
        (``mra_qformer_set_cross_mode``).  Same arithmetic, re-associated; the workspace size follows the mode."""
        code = {"auto": 0, "kv_cache": 1, "fold": 2, "fold384": 3, "fold_stream": 4, "fold_rescale_pass": 5}.get(mode, mode)
        check(lib().mra_qformer_set_cross_mode(self._handle, int(code)), "mra_qformer_set_cross_mode")
        self._cross_mode = mode if isinstance(mode, str) else {v: k for k, v in {"auto": 0, "kv_cache": 1, "fold": 2, "fold384": 3, "fold_stream": 4, "fold_rescale_pass": 5}.items()}.get(int(mode), "auto")

    def set_option(self, name: str, value: int) -> None:
        """Per-handle tuning option (``mra_qformer_set_option``), e.g. ``("chain_ring", mask)``."""
        check(lib().mra_qformer_set_option(self._handle, name.encode(), int(value)), f"mra_qformer_set_option({name})")

    def set_cross_precision(self, mode) -> None:
        """Precision of the cross-attention score chain: ``"op"`` (default: f16 / bf16 operands) or ``"split"`` (hidden state, W_cq, Q,
        W_k and Q' as hi + lo pairs, ~22 bits; folded form at any Kv; ``mra_qformer_set_cross_precision``).  For sharply attending
        (trained) weights, where the f16 rounding of the score operands is amplified by the softmax."""
        code = {"op": 0, "f16": 0, "split": 1}.get(mode, mode)
        with torch.cuda.device(self._device):
            check(lib().mra_qformer_set_cross_precision(self._handle, int(code)), "mra_qformer_set_cross_precision")
        self._cross_precision = "split" if int(code) == 1 else "op"

    def flops(self, items: int, L: int, kv: int, with_last_text: bool) -> float:
        return float(lib().mra_qformer_flops(self._handle, items, L, kv, int(with_last_text)))

    # ---- training (BASELINE config 5): forward with an activation tape + HIP backward behind torch.autograd ----
    def enable_training(self) -> None:
        """Allocate the flat gradient buffer, bind every ``bert.*`` parameter's ``.grad`` to its slice and build
        the transposed weight copies (call again after changing weights; ``forward_train`` does it lazily)."""
        self.sync_weights()
        with torch.cuda.device(self._device):
            check(lib().mra_qformer_enable_training(self._handle, current_stream()), "mra_qformer_enable_training")
        if getattr(self, "_grad_flat", None) is None:
            n = int(lib().mra_qformer_grad_bytes(self._handle)) // 4
            self._grad_flat = torch.zeros(n, dtype=torch.float32, device=self._device)
            self._anchor = torch.zeros((), dtype=torch.float32, device=self._device, requires_grad=True)
            self._train_ws = None
            self._slices = {}
            for k, p in self.bert.state_dict(prefix="bert.", keep_vars=True).items():
                self._slices[k] = (*self._slice_of(k), p)
                p.requires_grad_(True)
            # master weights: one flat f32 buffer in the gradient buffer's layout; every parameter becomes a view of
            # it, so any optimizer updates it in place and mra_qformer_load_flat refreshes the device copies at once
            self._master_flat = torch.zeros(n, dtype=torch.float32, device=self._device)
            self._bind_master()


    def _bind_master(self) -> None:
        """(Re-)point every ``bert.*`` parameter at its slice of the master buffer.  ``module.to()`` /
        ``load_state_dict(assign=True)`` replace parameter storage; the current values are carried over."""
        base = self._master_flat.data_ptr()
        first = next(iter(self._slices.values()))
        if first[2].data_ptr() == base + first[0] * 4 and getattr(self, "_master_bound", False):
            return
        with torch.no_grad():
            for off, numel, p in self._slices.values():
                view = self._master_flat[off: off + numel].view(p.shape)
                if p.data_ptr() != view.data_ptr():
                    view.copy_(p.data)
                    p.data = view
        self._master_bound = True


    def grad_of(self, name: str) -> torch.Tensor:
        """View of parameter ``name``'s gradient inside the flat buffer (ABI names, e.g. ``query_tokens``)."""
        off, numel = C.c_size_t(), C.c_int64()
        check(lib().mra_qformer_grad_offset(self._handle, name.encode(), C.byref(off), C.byref(numel)), f"grad_offset({name})")
        return self._grad_flat[off.value // 4: off.value // 4 + numel.value]


    def _bind_grads(self) -> None:
        probe = self.bert.embeddings.LayerNorm.weight
        if probe.grad is not None and probe.grad.data_ptr() == self.grad_of("bert.embeddings.LayerNorm.weight").data_ptr():
            return                                   # still bound from the previous backward
        for off, numel, p in self._slices.values():
            p.grad = self._grad_flat[off: off + numel].view(p.shape)


    def forward_train(self, input_ids, attention_mask, enc, want_cls: bool = True):
        """Training forward: ``(out_query [N,32,H], out_cls [N,H])`` connected to autograd.  ``loss.backward()``
        accumulates parameter gradients (``p.grad`` of every ``bert.*`` parameter, ``grad_of('query_tokens')``)."""
        self.enable_training()
        cfg = self.cfg
        enc = enc.to(cfg.op_dtype).contiguous()
        if input_ids is not None:
            input_ids = input_ids.to(device=enc.device, dtype=torch.int64).contiguous()
        if attention_mask is not None:
            attention_mask = attention_mask.to(device=enc.device, dtype=torch.int64).contiguous()
        q, c = _QFormerTrainFn.apply(self._anchor, self, input_ids, attention_mask, enc, want_cls)
        return (q, c) if want_cls else (q, None)


    def _slice_of(self, name: str):
        off, numel = C.c_size_t(), C.c_int64()
        check(lib().mra_qformer_grad_offset(self._handle, name.encode(), C.byref(off), C.byref(numel)), f"grad_offset({name})")
        return off.value // 4, int(numel.value)


    def flat_parameter(self) -> torch.nn.Parameter:
        """ONE ``nn.Parameter`` over the whole master buffer with the whole gradient buffer as its ``.grad``: an
        optimizer given this instead of the ~400 per-tensor parameters updates the Q-Former in one fused launch
        (fused Adam over 186 M elements: 0.8 ms instead of 2.3 ms for the per-tensor lists).  The per-tensor
        parameters stay valid views of the same memory (state_dict, checkpoints)."""
        self.enable_training()
        fp = getattr(self, "_flat_param", None)
        if fp is None or fp.data_ptr() != self._master_flat.data_ptr():
            fp = torch.nn.Parameter(self._master_flat, requires_grad=True)
            object.__setattr__(self, "_flat_param", fp)      # not registered: it aliases the per-tensor parameters
        fp.grad = self._grad_flat
        return fp


    def adam_step(self, lr: float, betas=(0.9, 0.999), eps: float = 1e-8, weight_decay: float = 0.0, zero_grad: bool = True) -> None:
        """One ``torch.optim.Adam`` step on this Q-Former's flat master / gradient buffers on the HIP extension
        (``mra_qformer_adam_step``): update, refresh of every device weight copy (operand dtype + transposed training copies) and, with
        ``zero_grad``, the clearing of the gradient buffer in ONE pass.  The moments live in two flat fp32 buffers created here."""
        self.enable_training()
        if getattr(self, "_exp_avg", None) is None:
            self._exp_avg = torch.zeros_like(self._master_flat)
            self._exp_avg_sq = torch.zeros_like(self._master_flat)
            self._adam_t = 0
        self._bind_master()
        self._adam_t += 1
        with torch.cuda.device(self._device):
            check(lib().mra_qformer_adam_step(self._handle, ptr(self._master_flat), ptr(self._grad_flat), ptr(self._exp_avg), ptr(self._exp_avg_sq),
                                              self._master_flat.numel() * 4, float(lr), float(betas[0]), float(betas[1]), float(eps),
                                              float(weight_decay), int(self._adam_t), int(bool(zero_grad)), current_stream()), "mra_qformer_adam_step")
        # the pass refreshed the device copies itself: nothing is stale (the version counters were not bumped either)
        self._param_version = sum(p._version for p in self.bert.parameters())
        self._dirty = False
        self._grads_zeroed = bool(zero_grad)

    def _run_backward(self, input_ids, attention_mask, enc, N, L, Kv, d_q, d_c) -> None:
        # optimizer.zero_grad(set_to_none=True) drops the views: start from a clean buffer in that case
        probe = self.bert.embeddings.LayerNorm.weight
        fp = getattr(self, "_flat_param", None)
        if (fp is not None and fp.grad is None) or (fp is None and probe.grad is None):
            self._grad_flat.zero_()
        if fp is not None:
            fp.grad = self._grad_flat
        with torch.cuda.device(self._device):
            check(lib().mra_qformer_backward(self._handle, ptr(input_ids), ptr(attention_mask), ptr(enc), N, L, Kv, ptr(d_q), ptr(d_c),
                                             ptr(self._grad_flat), ptr(self._train_ws), self._train_ws.numel(), current_stream()),
                  "mra_qformer_backward")
        self._bind_grads()
        binder = getattr(self, "_extra_grad_binder", None)
        if binder is not None:
            binder()
        # A backward is followed by an optimizer step sooner or later, and fused optimizers (torch._fused_adam_)
        # update parameters WITHOUT bumping their version counters: presume the device copies stale from here on.
        self._dirty = True


# --------------------------------------------------------------------------------------------------
# training (BASELINE config 5): forward with an activation tape + HIP backward behind torch.autograd
# --------------------------------------------------------------------------------------------------
class _QFormerTrainFn(torch.autograd.Function):
    """Autograd node around ``mra_qformer_forward_train`` / ``mra_qformer_backward``.  The parameters
    live inside the handle, so the node takes an anchor tensor that requires grad; its backward runs the
    HIP backward, which ADDS into the owner's flat f32 gradient buffer (``QFormer.grad_of``)."""

    @staticmethod
    def forward(ctx, anchor, owner, input_ids, attention_mask, enc, want_cls):
        cfg = owner.cfg
        N, Kv = int(enc.shape[0]), int(enc.shape[1])
        L = 0 if input_ids is None else int(input_ids.shape[1])
        dev = enc.device
        out_q = torch.empty(N, cfg.n_query, cfg.hidden, dtype=torch.float32, device=dev)
        out_c = torch.empty(N, cfg.hidden, dtype=torch.float32, device=dev) if want_cls else None
        with torch.cuda.device(dev):
            nbytes = int(lib().mra_qformer_train_workspace_bytes(owner._handle, N, L, Kv))
            if owner._train_ws is None or owner._train_ws.numel() < nbytes:
                owner._train_ws = None
                owner._train_ws = torch.empty(nbytes, dtype=torch.uint8, device=dev)
            check(lib().mra_qformer_forward_train(owner._handle, ptr(input_ids), ptr(attention_mask), ptr(enc), N, L, Kv, ptr(out_q),
                                                  ptr(out_c), ptr(owner._train_ws), owner._train_ws.numel(), current_stream()),
                  "mra_qformer_forward_train")
        ctx.owner, ctx.shape = owner, (N, L, Kv)
        ctx.save_for_backward(input_ids, attention_mask, enc)
        ctx.want_cls = want_cls
        return (out_q, out_c) if want_cls else (out_q, torch.empty(0, device=dev))

    @staticmethod
    def backward(ctx, d_q, d_c):
        owner = ctx.owner
        input_ids, attention_mask, enc = ctx.saved_tensors
        N, L, Kv = ctx.shape
        d_q = None if d_q is None else d_q.to(torch.float32).contiguous()
        d_c = d_c.to(torch.float32).contiguous() if (ctx.want_cls and d_c is not None) else None
        owner._run_backward(input_ids, attention_mask, enc, N, L, Kv, d_q, d_c)
        return torch.zeros_like(owner._anchor), None, None, None, None, None
