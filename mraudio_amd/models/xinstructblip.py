"""``XInstructBLIP`` with the reference's Python surface, its encode/fuse hot path on MI355X kernels.

Mirrors ``models/xinstructblip.py`` of the reference (class at ``:51``):

* ``XInstructBLIP(model_path, audio_path)``; attributes ``{m}_encoder``, ``{m}_ln``, ``{m}_Qformer``,
  ``{m}_query_tokens``, ``{m}_llm_proj`` with the checkpoint key names the reference's loader routes
  (``:769-816``); ``.generate(samples) -> list[str]`` (``:222``), ``.forward(samples) -> {"loss"}``
  (``:399``), ``.load_state_dict`` returning ``_IncompatibleKeys``, ``.get_optimizer_params``.
* ``samples`` schema as produced by ``utils/mr_dataset.py`` + ``collate_fn``:
  ``video [B,3,T,224,224]``, ``audio [B,T,F,128]``, ``text_input``, ``timestamps``, ``duration``.

What differs, on purpose: the reference hands the projected query embeddings to an
8-bit Vicuna-7B and parses generated text; here ``generate`` scores clips with the cosine scorer and
formats the span as the same ``"[[start, end]]"`` string, so ``evaluate.py``-shaped callers
(``moment_str_to_list(post_process(out))``, ``evaluate.py:48``) run unchanged.  The ViT-g / BEATs
encoders (row A1) are pluggable stock PyTorch modules; pre-computed encoder outputs can be passed
as ``samples["video_embeds"] [B,T,Kv,1408]`` / ``samples["audio_embeds"] [B,T,Kv,768]``.

All arithmetic between the encoder output and the span runs in ``libmra_hip.so``: modality
LayerNorm + reorder (``:265,281-285``), Q-Former (``:286-293``), slice + llm_proj (``:303-306``),
scorer.  No CPU fallback exists.
"""
from __future__ import annotations

import logging
import re
import zlib
from typing import Dict, List, Optional, Sequence

import torch
import torch.nn as nn
from torch.nn.modules.module import _IncompatibleKeys

from .. import parallel, scorer
from .._lib import MraError
from ..qformer import QFormer, QFormerConfig, draw_seeded


class HashTokenizer:
    """Offline stand-in for ``BertTokenizer("bert-base-uncased")`` (reference ``:609-612``), whose
    vocabulary file is a network download.  Lower-cased words and punctuation map to stable ids in
    [1000, 30521] by CRC32; [CLS]=101, [SEP]=102, [PAD]=0; same call signature and outputs
    (``input_ids``, ``attention_mask``, right padding to the longest, left truncation).
    Token ids are synthetic: use the real tokenizer with real checkpoints."""

    cls_id, sep_id, pad_id = 101, 102, 0

    def __init__(self, truncation_side: str = "left"):
        self.truncation_side = truncation_side

    def __len__(self):
        return 30523

    def _encode(self, text: str, max_length: int) -> List[int]:
        words = re.findall(r"[a-z0-9]+|[^\sa-z0-9]", text.lower())
        ids = [1000 + zlib.crc32(w.encode()) % 29522 for w in words]
        room = max_length - 2
        if len(ids) > room:
            ids = ids[-room:] if self.truncation_side == "left" else ids[:room]
        return [self.cls_id] + ids + [self.sep_id]

    def __call__(self, text, padding="longest", truncation=True, max_length=128, return_tensors="pt", **unused):
        texts = [text] if isinstance(text, str) else list(text)
        enc = [self._encode(t, max_length) for t in texts]
        width = max(len(e) for e in enc)
        ids = torch.full((len(enc), width), self.pad_id, dtype=torch.long)
        att = torch.zeros((len(enc), width), dtype=torch.long)
        for i, e in enumerate(enc):
            ids[i, : len(e)] = torch.tensor(e)
            att[i, : len(e)] = 1
        return _Encoding(ids, att)


class _Encoding:
    def __init__(self, input_ids, attention_mask):
        self.input_ids, self.attention_mask = input_ids, attention_mask

    def to(self, device):
        return _Encoding(self.input_ids.to(device), self.attention_mask.to(device))


class LayerNorm(nn.Module):
    """``{modality}_ln`` (reference ``:822-828``): fp32 LayerNorm, eps 1e-5, result in the input dtype.
    Parameter container; called directly it runs the HIP modality-LN kernel and returns the operand
    dtype tensor the Q-Former consumes."""

    def __init__(self, num_features: int):
        super().__init__()
        self.weight = nn.Parameter(torch.ones(num_features), requires_grad=False)
        self.bias = nn.Parameter(torch.zeros(num_features), requires_grad=False)
        object.__setattr__(self, "_qformer_ref", None)

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        qf = self._qformer_ref()
        shape = x.shape
        return qf.modality_ln(x.reshape(-1, shape[-2], shape[-1])).reshape(shape)


class _Proj(nn.Module):
    def __init__(self, in_f: int, out_f: int):
        super().__init__()
        self.weight = nn.Parameter(torch.empty(out_f, in_f), requires_grad=False)
        self.bias = nn.Parameter(torch.empty(out_f), requires_grad=False)
        object.__setattr__(self, "_qformer_ref", None)

    def forward(self, z: torch.Tensor) -> torch.Tensor:
        return self._qformer_ref().llm_proj(z)


ENC_WIDTH = {"video": 1408, "audio": 768}  # EVA ViT-g / BEATs num_features (reference :123)


class XInstructBLIP(nn.Module):
    def __init__(self, model_path: Optional[str] = None, audio_path: Optional[str] = None, *,
                 modalities: Optional[Sequence[str]] = None, video_encoder: Optional[nn.Module] = None,
                 audio_encoder: Optional[nn.Module] = None, tokenizer=None, seed: Optional[int] = 0,
                 perturb: bool = False, op_dtype: torch.dtype = torch.float16, device=None,
                 compat_repeat: bool = True, score_alpha: float = 0.5, fuse_weights: Optional[Sequence[float]] = None,
                 process_group=None, qformer_overrides: Optional[dict] = None, overlap_modalities: bool = True,
                 llm_hidden_size: int = 4096, checkpoint: Optional[str] = None, checkpoint_strict: bool = True):
        super().__init__()
        self.model_path, self.audio_path = model_path, audio_path
        self.modalities = list(modalities) if modalities is not None else ["audio", "video"]  # reference :71
        self.max_txt_len = 128           # reference :75
        self.max_output_txt_len = 64
        self.num_query_token = 32        # reference :120
        self.compat_repeat = compat_repeat
        self.score_alpha = score_alpha
        self.fuse_weights = fuse_weights
        self.process_group = process_group
        # True: the clips of one batch are sharded over the ranks (inference, every rank is given the same
        # samples).  False: ranks hold different samples (data-parallel training, utils/trainer.py) and only
        # gradients are exchanged.
        self.clip_parallel = True
        self.overlap_modalities = overlap_modalities
        self.pair_forward = False        # see fuse_score: two modalities of equal Q-Former shape as ONE launch sequence (mra_qformer_forward_pair); measured slower, opt-in
        self.kv_first = True             # see fuse_score: light modalities wait for the heavy K/V projection
        self.prioritize_heavy = True     # see fuse_score
        self.encode_chunk = 64           # frames per encoder call of the batched [B*T] encode (row A1)
        self.roofline_events = None      # bench instrumentation: {modality: (start, stop) events}
        self._streams: Dict[str, torch.cuda.Stream] = {}
        self._device = torch.device(device) if device is not None else torch.device("cuda", torch.cuda.current_device())
        self.tokenizer = tokenizer if tokenizer is not None else self.init_tokenizer(truncation_side="left")
        self.video_encoder = video_encoder
        self.audio_encoder = audio_encoder
        self.llm_hidden_size = llm_hidden_size   # Vicuna-7B: 4096 (reference :167)
        self.llm_model = None                    # stock causal LM, attached by attach_llm (row N2)
        self.llm_tokenizer = None
        self.prompt_assembler = None
        gen = torch.Generator().manual_seed(seed) if seed is not None else None
        # synthetic init draws modalities in sorted order so the tensors do not depend on list order
        for m in sorted(self.modalities):
            cfg = QFormerConfig(enc_width=ENC_WIDTH[m], op_dtype=op_dtype, llm_hidden=self.llm_hidden_size,
                                **(qformer_overrides or {}))
            qf, qt = self.init_Qformer(self.num_query_token, ENC_WIDTH[m], cfg=cfg, device=self._device)
            setattr(self, f"{m}_Qformer", qf)
            setattr(self, f"{m}_query_tokens", qt)
            ln = self.init_ln(ENC_WIDTH[m])
            proj = self.init_vicuna_projection(cfg.hidden, self.llm_hidden_size)
            setattr(self, f"{m}_ln", ln)
            setattr(self, f"{m}_llm_proj", proj)
            object.__setattr__(ln, "_qformer_ref", _ref(qf))
            object.__setattr__(proj, "_qformer_ref", _ref(qf))
            if gen is not None:
                mg = torch.Generator().manual_seed(int(seed) + (0 if m == "video" else 1))
                qf.init_seeded_(perturb=perturb, gen=mg)
                qt.data.copy_(draw_seeded(mg, (1, cfg.n_query, cfg.hidden), "w", perturb))
                ln.weight.data.copy_(draw_seeded(mg, (cfg.enc_width,), "g", perturb))
                ln.bias.data.copy_(draw_seeded(mg, (cfg.enc_width,), "z", perturb))
                proj.weight.data.copy_(draw_seeded(mg, (self.llm_hidden_size, cfg.hidden), "w", perturb))
                proj.bias.data.copy_(draw_seeded(mg, (self.llm_hidden_size,), "b", perturb))
        self.to(self._device)
        self._extras_dirty = True
        # Where the Q-Former / LN / projection weights come from.  The reference downloads them in its constructor
        # (``:79-194``; ``model_path`` there is the Vicuna directory, ``audio_path`` the BEATs checkpoint); offline they
        # come from ``checkpoint`` (a state dict with the reference's key names) or stay the seeded synthetic init.
        self.weights_source = f"synthetic (seed {seed})"
        if checkpoint is not None:   # strict like the reference's load_checkpoint; checkpoint_strict=False = its load_from_pretrained (a trainer checkpoint holds trainable parameters only)
            self.load_checkpoint(checkpoint) if checkpoint_strict else self.load_from_pretrained(checkpoint)
        elif model_path is not None or audio_path is not None:
            logging.warning("XInstructBLIP(model_path=%r, audio_path=%r): no checkpoint was given, the Q-Former / LayerNorm / "
                            "projection weights are the SYNTHETIC seeded init -- predictions are not meaningful.  Pass "
                            "checkpoint=<state dict .pth> (evaluate.py / finetune.py: --checkpoint).", model_path, audio_path)

    # ---- construction helpers with the reference's names (:609-735) --------------------------------
    @classmethod
    def init_tokenizer(cls, truncation_side="right"):
        try:  # the real vocabulary when it is on disk; never a download
            from transformers import BertTokenizer
            tok = BertTokenizer.from_pretrained("bert-base-uncased", truncation_side=truncation_side, local_files_only=True)
            tok.add_special_tokens({"bos_token": "[DEC]"})
            return tok
        except Exception:
            logging.info("bert-base-uncased vocabulary not on disk: using the offline HashTokenizer")
            return HashTokenizer(truncation_side=truncation_side)

    @classmethod
    def init_Qformer(cls, num_query_token, modality_width, cross_attention_freq=2, cfg: Optional[QFormerConfig] = None,
                     device=None):
        cfg = cfg or QFormerConfig(enc_width=modality_width, cross_freq=cross_attention_freq, n_query=num_query_token)
        qformer = QFormer(cfg, device=device)
        query_tokens = nn.Parameter(torch.zeros(1, num_query_token, cfg.hidden), requires_grad=False)
        return qformer, query_tokens

    @classmethod
    def init_ln(cls, num_features, load_ln_path=False, load_ln_type=""):
        return LayerNorm(num_features)

    @classmethod
    def init_vicuna_projection(cls, input_size, output_size, load_projection_path=False, load_projection_type="", projection_key=None):
        return _Proj(input_size, output_size)

    @property
    def device(self):
        return self._device

    def _apply(self, fn, recurse=True):
        out = super()._apply(fn, recurse)
        self._extras_dirty = True
        return out

    # ---- checkpoints (reference :737-816) --------------------------------------------------------------
    def load_state_dict(self, state_dict, strict=True, assign=False):
        """Routes ``{m}_Qformer.*``, ``{m}_query_tokens``, ``{m}_ln.*``, ``{m}_llm_proj.*`` as the
        reference does (``:769-816``); ``llm_model.*`` and encoder keys are ignored here (no LLM on
        this path)."""
        missing, unexpected = [], []
        for m in self.modalities:
            sub = {".".join(k.split(".")[1:]): v for k, v in state_dict.items() if k.split(".")[0] == f"{m}_Qformer"}
            msg = getattr(self, f"{m}_Qformer").load_state_dict(sub, strict=False)
            missing += [f"{m}_Qformer.{k}" for k in msg.missing_keys]
            unexpected += [f"{m}_Qformer.{k}" for k in msg.unexpected_keys]
            if f"{m}_query_tokens" not in state_dict:
                missing.append(f"{m}_query_tokens")
            else:
                getattr(self, f"{m}_query_tokens").data.copy_(state_dict[f"{m}_query_tokens"])
            for part in ("ln", "llm_proj"):
                sub = {".".join(k.split(".")[1:]): v for k, v in state_dict.items() if k.split(".")[0] == f"{m}_{part}"}
                msg = nn.Module.load_state_dict(getattr(self, f"{m}_{part}"), sub, strict=False)
                missing += [f"{m}_{part}.{k}" for k in msg.missing_keys]
                unexpected += [f"{m}_{part}.{k}" for k in msg.unexpected_keys]
        known = tuple(f"{m}_" for m in self.modalities)
        unexpected += [k for k in state_dict if not k.startswith(known) and k.split(".")[0] != "llm_model"
                       and "encoder" not in k.split(".")[0]]
        self._extras_dirty = True
        if strict and (missing or unexpected):
            raise RuntimeError(f"Error(s) in loading state_dict for XInstructBLIP: missing {missing}, unexpected {unexpected}")
        return _IncompatibleKeys(missing, unexpected)

    def _load_file(self, filename, strict: bool):
        ckpt = torch.load(filename, map_location="cpu", weights_only=True)
        sd = ckpt["model"] if "model" in ckpt else ckpt
        mine = [k for k in sd if k.startswith(tuple(f"{m}_" for m in self.modalities))]
        if not mine:
            raise RuntimeError(f"{filename}: no key of this model's modalities {self.modalities} found")
        msg = self.load_state_dict(sd, strict=strict)
        self.weights_source = f"checkpoint {filename}"
        if msg.missing_keys:
            self.weights_source += f" ({len(msg.missing_keys)} parameters NOT in the file keep their previous values)"
            logging.warning("%s: %d parameters keep their previous values (first: %s)", filename, len(msg.missing_keys), msg.missing_keys[:3])
        if isinstance(self.tokenizer, HashTokenizer):
            logging.warning("checkpoint weights are used with the offline HashTokenizer: token ids do NOT match the "
                            "bert-base-uncased vocabulary the checkpoint was trained with")
        return msg

    def load_checkpoint(self, filename, strict: bool = True, **kwargs):
        """Weights-only load of a finetuned ``.pth`` holding the reference's key names (``{m}_Qformer.*``, ``{m}_query_tokens``,
        ``{m}_ln.*``, ``{m}_llm_proj.*``; optionally under ``"model"``).  STRICT, like the reference's ``load_checkpoint``
        (``:759-767``: "this should expect no mismatch in the model keys and the checkpoint keys"): a file that lacks any
        Q-Former / LayerNorm / projection parameter of this model raises instead of silently keeping the synthetic init
        (``llm_model.*`` and encoder keys are not part of this path and are ignored).  ``strict=False`` is an explicit opt-in
        (or use ``load_from_pretrained``); the number of missing parameters then shows in ``weights_source``."""
        return self._load_file(filename, strict)

    def load_from_pretrained(self, filename, **kwargs):
        """The reference's non-strict loader (``:749-757``, ``load_state_dict(strict=False)``): partial files are accepted, what is
        missing keeps its previous values and is counted in ``weights_source``."""
        return self._load_file(filename, False)

    def get_optimizer_params(self, weight_decay, lr_scale=1):
        """LAVIS ``BaseModel.get_optimizer_params`` semantics (reference ``:818-820``): parameters that
        require grad, split into decay / no-decay (ndim < 2, bias, ln, bn) groups."""
        decay, no_decay = [], []
        for n, p in self.named_parameters():
            if not p.requires_grad:
                continue
            (no_decay if p.ndim < 2 or "bias" in n or "ln" in n or "bn" in n else decay).append(p)
        return [{"params": decay, "weight_decay": weight_decay, "lr_scale": lr_scale},
                {"params": no_decay, "weight_decay": 0, "lr_scale": lr_scale}]

    def _clip_world(self):
        return parallel.world(self.process_group) if self.clip_parallel else (0, 1)

    def _gather(self, rows: torch.Tensor, n_total: int) -> torch.Tensor:
        return parallel.all_gather_rows(rows, n_total, self.process_group) if self.clip_parallel else rows

    def _kv_done_event(self, modality: str) -> torch.cuda.Event:
        """The event the library records after ``modality``'s K/V projection (``mra_qformer_set_kv_done_event``)."""
        qf: QFormer = getattr(self, f"{modality}_Qformer")
        ev = getattr(qf, "_kv_done_event", None)
        if ev is None:
            from .. import _lib
            ev = torch.cuda.Event()
            with torch.cuda.device(self._device):
                ev.record()                       # torch creates the hipEvent lazily on the first record
            _lib.check(_lib.lib().mra_qformer_set_kv_done_event(qf._handle, ev.cuda_event), "set_kv_done_event")
            qf._kv_done_event = ev
        return ev

    def _side_stream(self, modality: str, high_priority: bool = False) -> torch.cuda.Stream:
        key = (modality, high_priority)
        if key not in self._streams:
            self._streams[key] = torch.cuda.Stream(device=self._device, priority=-1 if high_priority else 0)
        return self._streams[key]

    def _sync(self):
        ver = sum(getattr(self, f"{m}_{n}")._version if n == "query_tokens" else sum(p._version for p in getattr(self, f"{m}_{n}").parameters())
                  for m in self.modalities for n in ("query_tokens", "ln", "llm_proj"))
        if ver != getattr(self, "_extras_version", None):
            self._extras_version = ver
            self._extras_dirty = True
        for m in self.modalities:
            qf: QFormer = getattr(self, f"{m}_Qformer")
            qf.sync_weights()
            if self._extras_dirty:
                qf.push("query_tokens", getattr(self, f"{m}_query_tokens"))
                qf.push("ln.weight", getattr(self, f"{m}_ln").weight)
                qf.push("ln.bias", getattr(self, f"{m}_ln").bias)
                qf.push("llm_proj.weight", getattr(self, f"{m}_llm_proj").weight)
                qf.push("llm_proj.bias", getattr(self, f"{m}_llm_proj").bias)
        self._extras_dirty = False

    # ---- the hot path -------------------------------------------------------------------------------------
    def _encode(self, samples, modality: str, lo: Optional[int] = None, hi: Optional[int] = None):
        """Row A1: the encoder calls of the reference's per-position loop (``:262-275``: ``T`` sequential
        calls at batch ``B``), collapsed to ONE sample-major ``[B*T]`` batch fed to the encoder in chunks of
        ``encode_chunk`` frames.  Item ``k = r * T + i`` of the order the reference builds afterwards with
        ``cat(embeds)[indices]`` (``:281-285``) is frame ``i`` of sample ``r``, so no gather index is needed:
        the modality LayerNorm consumes the encoder output in place.  ``[lo, hi)`` restricts the work to one
        rank's contiguous block of items (clip-sharded inference): the encoder -- 100x the Q-Former's cost --
        is sharded too, not replicated.  Returns ``(raw [hi - lo, Kv, E], None, bs, num)``."""
        key = f"{modality}_embeds"
        if key in samples:  # pre-computed encoder outputs, already sample-major [B, T, Kv, E]
            e = samples[key]
            bs, num = int(e.shape[0]), int(e.shape[1])
            e = e.reshape(bs * num, e.shape[2], e.shape[3])
            if lo is not None:
                e = e[lo:hi]
            return e.to(self._device), None, bs, num
        encoder = getattr(self, f"{modality}_encoder")
        if encoder is None:
            raise MraError(f"no {modality}_encoder was given and samples has no '{key}'")
        data = samples[modality]
        if modality == "video":      # [B, 3, T, H, W] -> sample-major frames [B*T, 3, H, W]
            bs, num = int(data.shape[0]), int(data.shape[2])
            frames = data.permute(0, 2, 1, 3, 4).reshape(bs * num, data.shape[1], data.shape[3], data.shape[4])
        else:                        # [B, T, F, 128] -> [B*T, F, 128]
            bs, num = int(data.shape[0]), int(data.shape[1])
            frames = data.reshape(bs * num, data.shape[2], data.shape[3])
        if lo is not None:
            frames = frames[lo:hi]
        outs = []
        with torch.no_grad():
            for c0 in range(0, int(frames.shape[0]), max(1, int(self.encode_chunk))):
                outs.append(encoder(frames[c0: c0 + self.encode_chunk].to(self._device)))
        raw = outs[0] if len(outs) == 1 else torch.cat(outs)
        return raw, None, bs, num

    @torch.no_grad()
    def fuse_score(self, embeds: Dict[str, torch.Tensor], input_ids: torch.Tensor, text_mask: torch.Tensor, bs: int, num: int,
                   index: Optional[Dict[str, torch.Tensor]] = None, want_llm: bool = False, want_full: bool = False) -> Dict[str, object]:
        """Everything after the encoders and the tokenizer, on THIS rank's block of items.

        ``embeds[m]`` raw encoder outputs; the rank's items are ``embeds[m][index[m]]`` when an index is
        given, else ``embeds[m]`` itself (``[n_local, Kv, E]``).  ``input_ids`` / ``text_mask``
        ``[n_local, L]``.  The rank's block is rows ``shard_range(bs*num, rank, world)`` of the
        sample-major item order.  LN + reorder -> Q-Former -> all-gather of the query embeddings and
        [CLS] vectors over the process group -> cosine score -> fuse -> span (identical on all ranks)."""
        self._sync()
        n = bs * num
        rank, ws = self._clip_world()
        lo, hi = parallel.shard_range(n, rank, ws)
        n_local = hi - lo
        out: Dict[str, object] = {"z": {}, "cls": {}, "sim": {}, "logit": {}, "full": {}}
        ids = input_ids.to(self._device)
        tmask = text_mask.to(self._device)
        att = torch.cat([torch.ones(n_local, self.num_query_token, dtype=torch.long, device=self._device), tmask], dim=1)
        # The modality Q-Formers are independent until the fusion: each runs on its own HIP stream so
        # the short launches of one chain fill the gaps of the other (and of its K/V projection).
        cur = torch.cuda.current_stream(self._device)
        live = [m for m in self.modalities if m in embeds]
        use_streams = self.overlap_modalities and len(live) > 1
        heavy_done = None
        if use_streams:
            live.sort(key=lambda m: int(embeds[m].shape[-2]) * int(embeds[m].shape[-1]), reverse=True)
        if use_streams and self.kv_first:
            # The modality with the largest K/V projection goes first and the others start when that GEMM has
            # finished: a chip-filling GEMM gains nothing from sharing CUs with a latency-bound layer chain (it
            # ran 20 % longer beside one), whereas two layer chains overlap each other almost for free.
            live.sort(key=lambda m: int(embeds[m].shape[-2]) * int(embeds[m].shape[-1]), reverse=True)
            heavy = live[0]
            kv_flops = 2.0 * n_local * embeds[heavy].shape[-2] * embeds[heavy].shape[-1] * 9216
            if kv_flops >= 1e12:                  # ~1 ms of GEMM; below that the wait costs more than the contention
                heavy_done = self._kv_done_event(heavy)
        used_streams = []
        sharded = ws > 1
        local: Dict[str, tuple] = {}
        if self.pair_forward and len(live) == 2 and not want_full and not want_llm and self._pair_ok(live):
            # Both Q-Formers in ONE launch sequence on the caller's stream: their layer chains have identical shapes, so every chain GEMM /
            # attention core / LayerNorm takes both modalities in one launch (0.7 ms less kernel time per headline step, folded blocks free
            # of the other stream: 0.65 instead of 0.79 ms each) -- but the two-stream form hides ~0.65 ms of the light modality behind the heavy
            # one's latency-bound chain phases, and one sequence hides nothing: 6.72-6.87 vs 6.57 ms per step, reference item shape 2.65 vs
            # 2.55 ms (r03p, same session).  Opt-in.  The heavier lane first (its cross-layer-0 block carries the roofline events).
            qfs = [getattr(self, f"{m}_Qformer") for m in live]
            encs = [qf.modality_ln(embeds[m].to(self._device), item_index=None if index is None else index.get(m), items=n_local) for qf, m in zip(qfs, live)]
            res = QFormer.forward_pair(qfs[0], qfs[1], ids, att, encs[0], encs[1], want_cls=True, kv_events=(self.roofline_events or {}).get(live[0]))
            for m, (z, cls) in zip(live, res):
                if sharded:
                    local[m] = (z, cls)
                else:
                    sim, logit = scorer.cosine_scores(z, cls)
                    out["z"][m], out["cls"][m], out["sim"][m], out["logit"][m] = z, cls, sim, logit
            live = []
        for pos, m in enumerate(live):
            qf: QFormer = getattr(self, f"{m}_Qformer")
            idx = None if index is None else index.get(m)
            # the modality with the most work is the step's critical path: its launches go first when both streams are ready
            side = self._side_stream(m, high_priority=self.prioritize_heavy and pos == 0) if use_streams else cur
            if use_streams:
                used_streams.append(side)
                side.wait_stream(cur)
                if heavy_done is not None and pos > 0:
                    side.wait_event(heavy_done)
            with torch.cuda.stream(side):
                enc = qf.modality_ln(embeds[m].to(self._device), item_index=idx, items=n_local)
                res = qf.forward_fused(ids, att, enc, want_query=True, want_full=want_full, want_cls=True,
                                       kv_events=(self.roofline_events or {}).get(m))
                z, cls = res["query"], res["cls"]
                if sharded:      # scored after ONE packed all-gather of every modality's rows, below
                    local[m] = (z, cls)
                    if want_full:
                        out["full"][m] = self._gather(res["full"], n)
                    if use_streams:
                        for t in (enc, z, cls, out["full"].get(m)):
                            if t is not None:
                                t.record_stream(cur)
                    continue
                sim, logit = scorer.cosine_scores(z, cls)
                out["z"][m], out["cls"][m], out["sim"][m], out["logit"][m] = z, cls, sim, logit
                if want_full:
                    out["full"][m] = res["full"]
                if want_llm:  # reference :303-306
                    y = qf.llm_proj(z)
                    out.setdefault("inputs_llm", {})[m] = y.reshape(bs, num, self.num_query_token, -1).view(bs, num * self.num_query_token, -1)
                    out.setdefault("atts_llm", {})[m] = torch.ones(bs, num * self.num_query_token, dtype=torch.long, device=self._device)
                if use_streams:  # the results are consumed on the caller's stream
                    for t in (enc, z, cls, sim, logit, out["full"].get(m), out.get("inputs_llm", {}).get(m), out.get("atts_llm", {}).get(m)):
                        if t is not None:
                            t.record_stream(cur)
        for side in used_streams:          # join exactly the streams that were forked
            cur.wait_stream(side)
        if sharded and local:
            # the only exchange of the path: query embeddings and [CLS] vectors of all modalities in one RCCL all-gather
            mods_l = list(local)
            gathered = parallel.all_gather_packed([t for m in mods_l for t in local[m]], n, self.process_group)
            for k, m in enumerate(mods_l):
                z, cls = gathered[2 * k], gathered[2 * k + 1]
                sim, logit = scorer.cosine_scores(z, cls)
                out["z"][m], out["cls"][m], out["sim"][m], out["logit"][m] = z, cls, sim, logit
                if want_llm:
                    y = getattr(self, f"{m}_Qformer").llm_proj(z)
                    out.setdefault("inputs_llm", {})[m] = y.reshape(bs, num, self.num_query_token, -1).view(bs, num * self.num_query_token, -1)
                    out.setdefault("atts_llm", {})[m] = torch.ones(bs, num * self.num_query_token, dtype=torch.long, device=self._device)
        mods = [m for m in self.modalities if m in out["logit"]]
        if not mods:
            raise MraError("no features for any of the model's modalities")
        out["fused"] = scorer.fuse_logits([out["logit"][m] for m in mods], self.fuse_weights)
        out["spans"] = scorer.spans_from_logits(out["fused"], bs, num, self.score_alpha)
        out["bs"], out["num"] = bs, num
        return out

    def _pair_ok(self, live) -> bool:
        """The two live modalities can share one launch sequence: equal Q-Former shapes, operand-dtype score chain, no streaming fold."""
        a, b = (getattr(self, f"{m}_Qformer").cfg for m in live)
        same = all(getattr(a, k) == getattr(b, k) for k in ("hidden", "heads", "inter", "layers", "cross_freq", "n_query", "op_dtype"))
        return same and not any(getattr(getattr(self, f"{m}_Qformer"), "_cross_precision", "op") != "op" or getattr(getattr(self, f"{m}_Qformer"), "_cross_mode", "auto") == "fold_stream"
                                for m in live)

    @torch.no_grad()
    def encode_fuse(self, samples, want_llm: bool = False, want_full: bool = False) -> Dict[str, object]:
        """The seam ``generate`` and ``forward`` share (reference ``:228-306`` / ``:409-478``).

        Returns ``z[m]`` [N,32,768] (= ``last_hidden_state[:, :32]``), ``cls[m]`` [N,768], ``sim[m]``
        [N,32], ``logit[m]`` [N], ``fused`` [N], ``spans`` int32 [B,2]; with ``want_llm`` also
        ``inputs_llm[m]`` [B, T*32, 4096] and ``atts_llm[m]`` (``:303-306``).  With a process group the
        items are sharded in contiguous blocks over the ranks (every rank is given the same samples)."""
        prompt = samples["text_input"]
        text = self.tokenizer(prompt, padding="longest", truncation=True, max_length=self.max_txt_len, return_tensors="pt")
        ids, tmask = text.input_ids.to(self._device), text.attention_mask.to(self._device)
        rank, ws = self._clip_world()
        embeds, index = {}, {}
        bs = num = None
        for m in self.modalities:
            if m not in samples and f"{m}_embeds" not in samples:
                continue
            src = samples.get(f"{m}_embeds", samples.get(m))
            n_items = int(src.shape[0]) * int(src.shape[2] if (m == "video" and f"{m}_embeds" not in samples) else src.shape[1])
            lo, hi = parallel.shard_range(n_items, rank, ws)
            embeds[m], _, bs, num = self._encode(samples, m, lo, hi)     # this rank's block only: the encoder is sharded too
        if bs is None:
            raise MraError("samples holds none of the model's modalities")
        if self.compat_repeat:   # reference :287-288  ids.repeat(num, 1): row k carries prompt k % bs
            ids_n, tm_n = ids.repeat(num, 1), tmask.repeat(num, 1)
        else:                    # aligned: row k carries the prompt of its own sample k // num
            ids_n, tm_n = ids.repeat_interleave(num, 0), tmask.repeat_interleave(num, 0)
        lo, hi = parallel.shard_range(bs * num, rank, ws)
        return self.fuse_score(embeds, ids_n[lo:hi], tm_n[lo:hi], bs, num, index=index or None, want_llm=want_llm, want_full=want_full)

    @torch.no_grad()
    def generate(self, samples) -> List[str]:
        """Reference ``:221-397`` with the LLM decode replaced by the scorer: one ``"[[start, end]]"``
        string (seconds) per sample."""
        out = self.encode_fuse(samples)
        spans = out["spans"].cpu().tolist()
        ts = samples.get("timestamps")
        if ts is None:
            ts = [list(range(out["num"]))] * out["bs"]
        ts = [t.tolist() if torch.is_tensor(t) else list(t) for t in ts]
        return [o.strip() for o in scorer.spans_to_text(spans, ts)]

    @torch.no_grad()
    def generate_with_scores(self, samples):
        """``generate`` plus the fused per-position logits ``[B][T]`` the spans were cut from (the
        ``pred_saliency_scores`` of the highlight metrics, ``eval/mr_eval.py:291-325``)."""
        out = self.encode_fuse(samples)
        spans = out["spans"].cpu().tolist()
        ts = samples.get("timestamps")
        if ts is None:
            ts = [list(range(out["num"]))] * out["bs"]
        ts = [t.tolist() if torch.is_tensor(t) else list(t) for t in ts]
        scores = out["fused"].view(out["bs"], out["num"]).float().cpu().tolist()
        return [o.strip() for o in scorer.spans_to_text(spans, ts)], scores

    # ---- row N2: the reference's own decode, for callers that attach a stock LLM ----------------------------
    def attach_llm(self, llm_model: nn.Module, llm_tokenizer, enumerate_inputs: bool = False, interleave_seconds: bool = True) -> None:
        """Attach a stock causal LM (``LlamaForCausalLM`` in the reference, ``:146-172``) and its tokenizer.  The hot
        path then feeds it exactly as the reference does (``models/llm_prompt.py``); the LM runs as ordinary
        PyTorch-ROCm.  Kept outside ``state_dict`` (the reference checkpoints hold the LLM separately)."""
        from .llm_prompt import PromptAssembler

        object.__setattr__(self, "llm_model", llm_model)
        self.llm_tokenizer = llm_tokenizer
        emb = llm_model.get_input_embeddings()
        if emb.weight.shape[1] != self.llm_hidden_size:
            raise MraError(f"LLM hidden size {emb.weight.shape[1]} != llm_hidden_size {self.llm_hidden_size} of the projections")
        self.prompt_assembler = PromptAssembler(llm_tokenizer, emb, modalities=self.modalities, num_query_token=self.num_query_token,
                                                enumerate_inputs=enumerate_inputs, interleave_seconds=interleave_seconds,
                                                max_txt_len=self.max_txt_len, max_output_txt_len=self.max_output_txt_len, device=self._device)

    def _llm_inputs(self, samples):
        if self.prompt_assembler is None:
            raise MraError("no LLM attached: call attach_llm(llm_model, llm_tokenizer) first")
        with torch.no_grad():
            out = self.encode_fuse(samples, want_llm=True)
        dt = self.llm_model.get_input_embeddings().weight.dtype
        return {m: v.to(dt) for m, v in out["inputs_llm"].items()}, out["atts_llm"]

    @torch.no_grad()
    def generate_llm(self, samples, max_new_tokens: int = 64) -> List[str]:
        """Reference ``generate`` end to end (``:221-397``): hot path -> prompt assembly -> ``llm_model.generate``."""
        inputs_llm, atts_llm = self._llm_inputs(samples)
        embeds, mask = self.prompt_assembler.assemble_generate(samples, inputs_llm, atts_llm)
        outputs = self.llm_model.generate(inputs_embeds=embeds, attention_mask=mask, max_new_tokens=max_new_tokens)
        outputs[outputs == 0] = 2                                      # :392
        return [o.strip() for o in self.llm_tokenizer.batch_decode(outputs, skip_special_tokens=True)]

    def forward_llm(self, samples):
        """Reference ``forward`` end to end (``:399-606``): the LM's cross-entropy on the answer tokens.  The Q-Former
        side is computed without grad (frozen, as in the reference); the loss trains whatever the LM leaves trainable."""
        inputs_llm, atts_llm = self._llm_inputs(samples)
        embeds, mask, targets = self.prompt_assembler.assemble_forward(samples, inputs_llm, atts_llm)
        return {"loss": self.llm_model(inputs_embeds=embeds, attention_mask=mask, return_dict=True, labels=targets).loss}

    _WINDOW = re.compile(r"\[\s*(-?\d+(?:\.\d+)?)\s*,\s*(-?\d+(?:\.\d+)?)\s*\]")

    @classmethod
    def parse_windows(cls, txt: str) -> List[List[float]]:
        """``text_output`` of the reference's datasets (``utils/mr_dataset.py:100-106``): ``"[[s, e]]"`` or several
        windows ``"[[s0, e0], [s1, e1]]"`` (QVHighlights), integer or float seconds (``save_float=True`` in the
        reference's preprocessing).  Returns every ``[start, end]`` pair; ``[[-1, -1]]`` (no window) gives ``[]``.
        Raises on text with no window at all, so a malformed annotation cannot silently train on an empty target."""
        wins = [[float(a), float(b)] for a, b in cls._WINDOW.findall(txt)]
        if not wins:
            raise ValueError(f"text_output holds no [start, end] window: {txt!r}")
        return [[min(a, b), max(a, b)] for a, b in wins if a >= 0 and b >= 0]

    def _targets(self, samples, bs, num):
        """Clip-membership targets: position ``i`` of sample ``r`` is positive when its timestamp lies inside ANY of
        the windows of ``samples["text_output"][r]`` (union over windows, float seconds honoured)."""
        target = torch.zeros(bs, num, dtype=torch.float32, device=self._device)
        ts = samples.get("timestamps") or [list(range(num))] * bs
        for r, txt in enumerate(samples.get("text_output", ["[[-1, -1]]"] * bs)):
            t = torch.as_tensor(ts[r], dtype=torch.float32, device=self._device)
            for s0, e0 in self.parse_windows(txt):
                target[r] = torch.maximum(target[r], ((t >= s0) & (t <= e0)).float())
        return target

    def enable_qformer_training(self) -> None:
        """Unfreeze the Q-Formers (the reference keeps them frozen, ``:196-204``; BASELINE config 5 trains
        them): parameters get ``requires_grad`` and gradients from the HIP backward."""
        self.train_qformers = True
        for m in self.modalities:
            qf: QFormer = getattr(self, f"{m}_Qformer")
            qf.enable_training()
            qt = getattr(self, f"{m}_query_tokens")
            qt.requires_grad_(True)

            # the query tokens live in the Q-Former's flat master / gradient buffers too (slot "query_tokens")
            off, numel = qf._slice_of("query_tokens")
            with torch.no_grad():
                view = qf._master_flat[off: off + numel].view_as(qt)
                view.copy_(qt.data)
                qt.data = view

            def binder(qf=qf, qt=qt):
                qt.grad = qf.grad_of("query_tokens").view_as(qt)
                self._extras_dirty = True      # fused optimizers do not bump version counters (see QFormer._run_backward)

            qf._extra_grad_binder = binder

    def forward(self, samples):
        """Reference ``:399-606`` returns the LLM's cross-entropy.  Without an LLM on this path the
        training signal is build-defined: binary cross-entropy between sigmoid(20 * fused logit) and clip
        membership of the target span parsed from ``text_output``.  After ``enable_qformer_training()`` the
        loss is differentiable w.r.t. every Q-Former parameter (forward + backward on the HIP extension; the
        few-KB scorer and the loss run as torch ops so autograd can seed the backward); otherwise it is a
        forward-only value, as the reference's frozen Q-Formers would give."""
        if samples is None or samples == {} or not any(m in samples or f"{m}_embeds" in samples for m in self.modalities):
            return {"loss": torch.tensor(0.0)}
        if not getattr(self, "train_qformers", False):
            out = self.encode_fuse(samples)
            bs, num = out["bs"], out["num"]
            logits = out["fused"].view(bs, num) * 20.0
            return {"loss": nn.functional.binary_cross_entropy_with_logits(logits, self._targets(samples, bs, num))}
        self._sync()
        text = self.tokenizer(samples["text_input"], padding="longest", truncation=True, max_length=self.max_txt_len, return_tensors="pt")
        ids, tmask = text.input_ids.to(self._device), text.attention_mask.to(self._device)
        per_mod, bs, num = [], None, None
        # ``train_streams``: each modality's forward (and, since autograd runs a node's backward on its forward's stream, its backward)
        # on its own stream.  Round 2 measured nothing from it (17.4 vs 17.0 ms: the step was bound by the GPU time of its ~1300 launches);
        # measured again after the launch merges of round 3 -- see DESIGN.md section 8.
        cur = torch.cuda.current_stream(self._device)
        use_streams = bool(getattr(self, "train_streams", False))
        for m in self.modalities:
            if m not in samples and f"{m}_embeds" not in samples:
                continue
            qf: QFormer = getattr(self, f"{m}_Qformer")
            side = self._side_stream(m) if use_streams else cur
            if use_streams:
                side.wait_stream(cur)
            with torch.cuda.stream(side):
                with torch.no_grad():
                    raw, idx, bs, num = self._encode(samples, m)
                    enc = qf.modality_ln(raw, item_index=idx, items=bs * num)
                n = bs * num
                ids_n, tm_n = (ids.repeat(num, 1), tmask.repeat(num, 1)) if self.compat_repeat else \
                              (ids.repeat_interleave(num, 0), tmask.repeat_interleave(num, 0))
                att = torch.cat([torch.ones(n, self.num_query_token, dtype=torch.long, device=self._device), tm_n], dim=1)
                z, cls = qf.forward_train(ids_n, att, enc)
                sim = nn.functional.cosine_similarity(z, cls[:, None, :], dim=-1, eps=1e-8)
                per_mod.append(sim.max(dim=1).values)
        if use_streams:
            for m in self.modalities:
                if (m, False) in self._streams:
                    cur.wait_stream(self._streams[(m, False)])
        w = self.fuse_weights or [1.0 / len(per_mod)] * len(per_mod)
        fused = sum(x * wt for x, wt in zip(per_mod, w))
        return {"loss": nn.functional.binary_cross_entropy_with_logits(fused.view(bs, num) * 20.0, self._targets(samples, bs, num))}

    def flat_optimizer_params(self) -> List[nn.Parameter]:
        """Parameters for an optimizer after ``enable_qformer_training()``: one flat parameter per Q-Former (bert.* and
        the query tokens are views of it) plus whatever else requires grad.  Same update as the per-tensor list."""
        covered, out = set(), []
        for m in self.modalities:
            qf: QFormer = getattr(self, f"{m}_Qformer")
            out.append(qf.flat_parameter())
            covered.update(id(p) for p in qf.parameters())
            covered.add(id(getattr(self, f"{m}_query_tokens")))
        out.extend(p for p in self.parameters() if p.requires_grad and id(p) not in covered)
        return out

    def all_reduce_grads(self) -> None:
        """Data-parallel gradient averaging over the process group: one all-reduce of each Q-Former's flat f32
        gradient buffer (what DDP does bucket by bucket in the reference's trainer, ``utils/trainer.py:69,133``)."""
        rank, ws = parallel.world(self.process_group)
        if ws == 1:
            return
        import torch.distributed as dist
        for m in self.modalities:
            flat = getattr(self, f"{m}_Qformer")._grad_flat
            dist.all_reduce(flat, group=self.process_group)
            flat.div_(ws)


def _ref(obj):
    import weakref
    return weakref.ref(obj)
