"""LLM prompt assembly: what the reference does between the hot path's output
(``inputs_llm[m]`` = ``{m}_llm_proj(last_hidden_state[:, :32])``, ``[B, T*32, D]``) and the Llama call.

Layout of the embedded sequence, per sample (reference ``models/xinstructblip.py:341-381`` for ``generate``,
``:541-595`` for ``forward``), with T temporal positions and both modalities::

    for pos in 0..T-1:   [ "(a) " enumeration ]                       only if enumerate_inputs (default off, :70)
                         cue(" video: ")  video queries [pos]  (32)
                         cue(" audio: ")  audio queries [pos]  (32)
                         " {timestamp[pos]} "                          if interleave_seconds (default on, :73)
    "{duration} "
    prompt tokens                                                      generate: stripped text, no special tokens
                                                                       forward : input ++ output[1:] ++ input padding

``forward`` additionally builds the labels: -100 on everything before the prompt, on the instruction part and
on padding; the answer tokens (``text_output + eos``) carry the loss (``:502-516,584-595``).

The LLM itself is a stock module and stays out of this package: the assembler takes a tokenizer with the
HuggingFace call signature and an ``embed(ids) -> [.., D]`` callable (``llm_model.get_input_embeddings()``).
Neither Vicuna weights nor the Llama sentencepiece model exist offline, so ``SimpleLlmTokenizer`` is the
stand-in vocabulary for tests and smoke runs; with real checkpoints pass ``LlamaTokenizer`` instead.
Parity: the reference's class cannot be imported here (LAVIS / peft / Vicuna absent) and holds no fixtures for
this step -> **unpinned**; the tests re-derive the layout index by index from the lines cited above.
"""
from __future__ import annotations

from typing import Callable, Dict, List, Optional, Sequence, Tuple

import torch

MODALITY_TO_CUE = {"video": " video: ", "audio": " audio: "}       # reference :206-209


class _Enc:
    def __init__(self, input_ids: torch.Tensor, attention_mask: torch.Tensor):
        self.input_ids, self.attention_mask = input_ids, attention_mask

    def to(self, device):
        return _Enc(self.input_ids.to(device), self.attention_mask.to(device))


class SimpleLlmTokenizer:
    """Offline stand-in for ``LlamaTokenizer`` with the special tokens the reference installs (``:141-145``:
    pad ``[PAD]``, bos = eos = unk = ``</s>``).  Bytes map to ids 3..258; ``</s>`` = 2, ``[PAD]`` = 259.
    Implements what the assembler and ``generate`` use: the call (padding to the longest on ``padding_side``,
    truncation on ``truncation_side``, ``add_special_tokens`` = one leading bos), ``eos_token``,
    ``pad_token_id``, ``batch_decode``."""

    eos_token, eos_token_id, bos_token_id, pad_token_id = "</s>", 2, 2, 259

    def __init__(self, padding_side: str = "right", truncation_side: str = "left"):
        self.padding_side, self.truncation_side = padding_side, truncation_side

    def __len__(self) -> int:
        return 260

    def _ids(self, text: str, add_special_tokens: bool) -> List[int]:
        out: List[int] = [self.bos_token_id] if add_special_tokens else []
        while text:
            if text.startswith(self.eos_token):
                out.append(self.eos_token_id)
                text = text[len(self.eos_token):]
            else:
                out.extend(3 + b for b in text[0].encode("utf-8"))
                text = text[1:]
        return out

    def __call__(self, text, padding=False, truncation=False, max_length=None, return_tensors="pt", add_special_tokens=True, **unused):
        texts = [text] if isinstance(text, str) else list(text)
        rows = [self._ids(t, add_special_tokens) for t in texts]
        if truncation and max_length is not None:
            rows = [(r[-max_length:] if self.truncation_side == "left" else r[:max_length]) if len(r) > max_length else r for r in rows]
        width = max((len(r) for r in rows), default=0)
        ids = torch.full((len(rows), width), self.pad_token_id, dtype=torch.long)
        att = torch.zeros((len(rows), width), dtype=torch.long)
        for i, r in enumerate(rows):
            if not r:
                continue
            sl = slice(width - len(r), width) if self.padding_side == "left" else slice(0, len(r))
            ids[i, sl] = torch.tensor(r)
            att[i, sl] = 1
        return _Enc(ids, att)

    def batch_decode(self, ids, skip_special_tokens: bool = True) -> List[str]:
        out = []
        for row in torch.as_tensor(ids).tolist():
            text = b"".join(bytes([t - 3]) if 3 <= t <= 258 else (b"" if skip_special_tokens or t != self.eos_token_id else b"</s>")
                            for t in row)
            out.append(text.decode("utf-8", errors="ignore"))
        return out


def concat_text_input_output(input_ids, input_atts, output_ids, output_atts):
    """Per row: instruction tokens, then the answer without its leading bos, then the instruction's padding
    (reference ``:26-48``).  Returns the merged ``{"input_ids", "attention_mask"}`` and the instruction lengths."""
    lens, ids, atts = [], [], []
    for i in range(input_ids.size(0)):
        n = int(input_atts[i].sum())
        lens.append(n)
        ids.append(torch.cat([input_ids[i][:n], output_ids[i][1:], input_ids[i][n:]]))
        atts.append(torch.cat([input_atts[i][:n], output_atts[i][1:], input_atts[i][n:]]))
    return {"input_ids": torch.stack(ids), "attention_mask": torch.stack(atts)}, lens


class PromptAssembler:
    def __init__(self, llm_tokenizer, embed: Callable[[torch.Tensor], torch.Tensor], modalities: Sequence[str] = ("video", "audio"),
                 num_query_token: int = 32, enumerate_inputs: bool = False, interleave_seconds: bool = True, max_txt_len: int = 128,
                 max_output_txt_len: int = 64, device=None):
        self.tok, self.embed = llm_tokenizer, embed
        self.modalities = [m for m in ("video", "audio") if m in modalities]     # the reference interleaves in this order (:359)
        self.num_query_token = num_query_token
        self.enumerate_inputs, self.interleave_seconds = enumerate_inputs, interleave_seconds
        self.max_txt_len, self.max_output_txt_len = max_txt_len, max_output_txt_len
        self.device = torch.device(device) if device is not None else torch.device("cpu")
        # cues are tokenised WITH special tokens (reference :215 uses the tokenizer's default), i.e. each carries a bos
        self.tokenized_cue: Dict[str, _Enc] = {}
        self.emb_cue: Dict[str, torch.Tensor] = {}
        for m in self.modalities:
            enc = self.tok(MODALITY_TO_CUE[m], return_tensors="pt")
            self.tokenized_cue[m] = enc
            with torch.no_grad():
                self.emb_cue[m] = self.embed(enc.input_ids.to(self.device))

    # ---- shared prefix: positions, duration ---------------------------------------------------------------
    def _prefix(self, samples, inputs_llm: Dict[str, torch.Tensor], atts_llm: Dict[str, torch.Tensor], n_prompts: int
                ) -> Tuple[List[torch.Tensor], List[torch.Tensor]]:
        dev, Q = self.device, self.num_query_token
        first = self.modalities[0]
        bs = inputs_llm[first].shape[0]
        num = {m: inputs_llm[m].shape[1] // Q for m in self.modalities}
        ts_emb = ts_att = None
        if self.interleave_seconds:
            flat = [f" {t} " for row in samples["timestamps"] for t in (row.tolist() if torch.is_tensor(row) else row)]
            tt = self.tok(flat, padding="longest", truncation=True, return_tensors="pt", add_special_tokens=False).to(dev)
            n_rows, per_row = len(samples["timestamps"]), len(samples["timestamps"][0])
            e = self.embed(tt.input_ids)
            ts_emb = e.view(n_rows, per_row, *e.shape[1:])
            ts_att = tt.attention_mask.view(n_rows, per_row, -1)
        inp: List[torch.Tensor] = []
        att: List[torch.Tensor] = []
        for pos in range(num[first]):
            if self.enumerate_inputs:
                en = self.tok([f"{'' if pos == 0 else ' '}({chr(97 + pos)}) " for _ in range(n_prompts)], return_tensors="pt",
                              add_special_tokens=pos == 0).to(dev)
                inp.append(self.embed(en.input_ids))
                att.append(en.attention_mask)
            for m in self.modalities:
                att.append(self.tokenized_cue[m].attention_mask.to(dev).repeat(bs, 1))
                att.append(atts_llm[m].view(bs, num[m], Q)[:, pos, :])
                inp.append(self.emb_cue[m].to(dev).repeat(bs, 1, 1))
                inp.append(inputs_llm[m].view(bs, num[m], Q, -1)[:, pos, :, :])
            if self.interleave_seconds:
                inp.append(ts_emb[:, pos, :, :])
                att.append(ts_att[:, pos, :])
        dur = self.tok([f"{d} " for d in samples["duration"]], padding="longest", truncation=True, return_tensors="pt",
                       add_special_tokens=False).to(dev)
        att.append(dur.attention_mask)
        inp.append(self.embed(dur.input_ids))
        return inp, att

    def _cat(self, inp: List[torch.Tensor], att: List[torch.Tensor]):
        dt = inp[-1].dtype
        return torch.cat([x.to(dt) for x in inp], dim=1), torch.cat(att, dim=1)

    # ---- generate (:309-381) --------------------------------------------------------------------------------
    def assemble_generate(self, samples, inputs_llm, atts_llm) -> Tuple[torch.Tensor, torch.Tensor]:
        self.tok.padding_side = "left"                               # :223
        prompt = [p.strip() for p in samples["text_input"]]          # :309
        llm = self.tok(prompt, padding="longest", return_tensors="pt", add_special_tokens=False).to(self.device)
        inp, att = self._prefix(samples, inputs_llm, atts_llm, len(prompt))
        att.append(llm.attention_mask)
        inp.append(self.embed(llm.input_ids))
        return self._cat(inp, att)

    # ---- forward (:481-595) -----------------------------------------------------------------------------------
    def assemble_forward(self, samples, inputs_llm, atts_llm) -> Tuple[torch.Tensor, torch.Tensor, torch.Tensor]:
        tok, dev = self.tok, self.device
        tok.padding_side, tok.truncation_side = "right", "left"
        tin = tok(samples["text_input"], return_tensors="pt", padding="longest", truncation=True, max_length=self.max_txt_len,
                  add_special_tokens=True).to(dev)
        tok.truncation_side = "right"
        tout = tok([t + tok.eos_token for t in samples["text_output"]], return_tensors="pt", padding="longest", truncation=True,
                   max_length=self.max_output_txt_len).to(dev)
        llm, in_len = concat_text_input_output(tin.input_ids, tin.attention_mask, tout.input_ids, tout.attention_mask)
        targets = llm["input_ids"].masked_fill(llm["input_ids"] == tok.pad_token_id, -100)
        for i, n in enumerate(in_len):
            targets[i][:n] = -100
        inp, att = self._prefix(samples, inputs_llm, atts_llm, len(samples["text_input"]))
        empty = torch.full(torch.cat(att, dim=1).size(), -100, dtype=torch.long, device=dev)
        att.append(llm["attention_mask"])
        inp.append(self.embed(llm["input_ids"]))
        embeds, mask = self._cat(inp, att)
        return embeds, mask, torch.cat([empty, targets], dim=1)
