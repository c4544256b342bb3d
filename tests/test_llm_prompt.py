"""Row N2: the embedded LLM sequence the reference builds after the hot path
(``models/xinstructblip.py:309-381`` generate, ``:481-595`` forward), re-derived here segment by segment.
Parity with the reference class itself is unpinned (not importable, no fixtures); this pins the layout rules."""
import torch
from torch import nn

from mraudio_amd.models.llm_prompt import MODALITY_TO_CUE, PromptAssembler, SimpleLlmTokenizer, concat_text_input_output

D, Q = 16, 32


def _setup(T=3, B=2, **kw):
    torch.manual_seed(0)
    tok = SimpleLlmTokenizer()
    emb = nn.Embedding(len(tok), D)
    asm = PromptAssembler(tok, emb, num_query_token=Q, **kw)
    inputs = {m: torch.randn(B, T * Q, D) for m in ("video", "audio")}
    atts = {m: torch.ones(B, T * Q, dtype=torch.long) for m in ("video", "audio")}
    samples = {"text_input": ["Query: a dog barks.\nRelevant windows: ", " Query: rain \n"], "text_output": ["[[2, 9]]", "[[10, 31]]"],
               "timestamps": [[0, 7, 15], [3, 10, 120]], "duration": [20, 150]}
    return tok, emb, asm, inputs, atts, samples


def test_tokenizer_contract():
    tok = SimpleLlmTokenizer()
    e = tok(["ab", "abcd"], padding="longest", return_tensors="pt", add_special_tokens=False)
    assert e.input_ids.tolist() == [[100, 101, 259, 259], [100, 101, 102, 103]] and e.attention_mask.tolist() == [[1, 1, 0, 0], [1, 1, 1, 1]]
    tok.padding_side = "left"
    e = tok(["ab", "abcd"], padding="longest", add_special_tokens=True)
    assert e.input_ids.tolist() == [[259, 259, 2, 100, 101], [2, 100, 101, 102, 103]]
    assert tok(["x</s>"], add_special_tokens=False).input_ids.tolist() == [[123, 2]]
    tok.truncation_side = "left"
    assert tok(["abcdef"], truncation=True, max_length=3, add_special_tokens=False).input_ids.tolist() == [[103, 104, 105]]
    assert tok.batch_decode(torch.tensor([[2, 100, 101, 259, 0]])) == ["ab"]


def test_generate_layout_segment_by_segment():
    tok, emb, asm, inputs, atts, samples = _setup()
    embeds, mask = asm.assemble_generate(samples, inputs, atts)
    cue = {m: tok(MODALITY_TO_CUE[m]).input_ids for m in ("video", "audio")}          # with bos, as the reference tokenises them
    assert cue["video"][0, 0].item() == tok.bos_token_id
    tok.padding_side = "left"        # generate sets it before anything is tokenised (:223): timestamps and durations are left-padded too
    ts = tok([" 0 ", " 7 ", " 15 ", " 3 ", " 10 ", " 120 "], padding="longest", add_special_tokens=False)
    ts_ids, ts_att = ts.input_ids.view(2, 3, -1), ts.attention_mask.view(2, 3, -1)
    dur = tok(["20 ", "150 "], padding="longest", add_special_tokens=False)
    assert dur.attention_mask.tolist() == [[0, 1, 1, 1], [1, 1, 1, 1]]
    prm = tok([p.strip() for p in samples["text_input"]], padding="longest", add_special_tokens=False)
    per_pos = cue["video"].shape[1] + Q + cue["audio"].shape[1] + Q + ts_ids.shape[-1]
    assert embeds.shape == (2, 3 * per_pos + dur.input_ids.shape[1] + prm.input_ids.shape[1], D) and mask.shape == embeds.shape[:2]
    o = 0
    for pos in range(3):
        for m in ("video", "audio"):
            n = cue[m].shape[1]
            assert torch.equal(embeds[:, o:o + n], emb(cue[m]).repeat(2, 1, 1)) and mask[:, o:o + n].all()
            o += n
            assert torch.equal(embeds[:, o:o + Q], inputs[m][:, pos * Q:(pos + 1) * Q]) and mask[:, o:o + Q].all()
            o += Q
        n = ts_ids.shape[-1]
        assert torch.equal(embeds[:, o:o + n], emb(ts_ids[:, pos])) and torch.equal(mask[:, o:o + n], ts_att[:, pos])
        o += n
    n = dur.input_ids.shape[1]
    assert torch.equal(embeds[:, o:o + n], emb(dur.input_ids)) and torch.equal(mask[:, o:o + n], dur.attention_mask)
    o += n
    assert torch.equal(embeds[:, o:], emb(prm.input_ids)) and torch.equal(mask[:, o:], prm.attention_mask)
    assert mask[1, o] == 0 and mask[0, o] == 1                      # the shorter (stripped) prompt is LEFT-padded for generation


def test_enumeration_and_no_seconds_variants():
    tok, emb, asm, inputs, atts, samples = _setup(enumerate_inputs=True, interleave_seconds=False)
    embeds, mask = asm.assemble_generate(samples, inputs, atts)
    e0 = tok(["(a) "], add_special_tokens=True).input_ids            # bos only on the first enumeration token (:350)
    e1 = tok([" (b) "], add_special_tokens=False).input_ids
    assert torch.equal(embeds[0, : e0.shape[1]], emb(e0)[0])
    cue_v, cue_a = tok(MODALITY_TO_CUE["video"]).input_ids.shape[1], tok(MODALITY_TO_CUE["audio"]).input_ids.shape[1]
    o = e0.shape[1] + cue_v + Q + cue_a + Q
    assert torch.equal(embeds[1, o: o + e1.shape[1]], emb(e1)[0])


def test_forward_targets_only_on_the_answer():
    tok, emb, asm, inputs, atts, samples = _setup()
    embeds, mask, targets = asm.assemble_forward(samples, inputs, atts)
    assert embeds.shape[:2] == mask.shape == targets.shape
    tok.padding_side, tok.truncation_side = "right", "left"
    tin = tok(samples["text_input"], padding="longest", truncation=True, max_length=128, add_special_tokens=True)
    tout = tok([t + "</s>" for t in samples["text_output"]], padding="longest", truncation=True, max_length=64)
    tail = tin.input_ids.shape[1] + tout.input_ids.shape[1] - 1
    prefix = targets.shape[1] - tail
    assert (targets[:, :prefix] == -100).all()
    for i in range(2):
        n_in, n_out = int(tin.attention_mask[i].sum()), int(tout.attention_mask[i].sum())
        row = targets[i, prefix:]
        assert (row[:n_in] == -100).all()
        assert row[n_in: n_in + n_out - 1].tolist() == tout.input_ids[i, 1:n_out].tolist()     # answer without its bos, ending in eos
        assert row[n_in + n_out - 2].item() == tok.eos_token_id
        assert (row[n_in + n_out - 1:] == -100).all()                # answer padding + instruction padding
        assert mask[i, prefix: prefix + n_in + n_out - 1].all()


def test_concat_text_input_output():
    ids, att = torch.tensor([[5, 6, 0], [7, 8, 9]]), torch.tensor([[1, 1, 0], [1, 1, 1]])
    oid, oat = torch.tensor([[2, 11, 12], [2, 13, 0]]), torch.tensor([[1, 1, 1], [1, 1, 0]])
    out, lens = concat_text_input_output(ids, att, oid, oat)
    assert lens == [2, 3] and out["input_ids"].tolist() == [[5, 6, 11, 12, 0], [7, 8, 9, 13, 0]]
    assert out["attention_mask"].tolist() == [[1, 1, 1, 1, 0], [1, 1, 1, 1, 0]]


def test_feeds_a_stock_llama_end_to_end():
    """The assembled tensors drive an unmodified HF ``LlamaForCausalLM`` (random tiny config): greedy decode
    and the label-masked LM loss, as ``generate`` (:383-396) / ``forward`` (:598-606) call it."""
    tf = __import__("pytest").importorskip("transformers")
    torch.manual_seed(0)
    tok = SimpleLlmTokenizer()
    cfg = tf.LlamaConfig(vocab_size=len(tok), hidden_size=64, intermediate_size=128, num_hidden_layers=2, num_attention_heads=4,
                         num_key_value_heads=4, max_position_embeddings=2048, pad_token_id=tok.pad_token_id, bos_token_id=2, eos_token_id=2)
    llm = tf.LlamaForCausalLM(cfg).eval()
    asm = PromptAssembler(tok, llm.get_input_embeddings(), num_query_token=Q)
    inputs = {m: torch.randn(2, 3 * Q, 64) for m in ("video", "audio")}
    atts = {m: torch.ones(2, 3 * Q, dtype=torch.long) for m in ("video", "audio")}
    samples = {"text_input": ["Query: a dog.\nRelevant windows: ", "Query: b\n"], "text_output": ["[[1, 2]]", "[[3, 4]]"],
               "timestamps": [[0, 1, 2], [3, 4, 5]], "duration": [10, 20]}
    with torch.no_grad():
        e, m = asm.assemble_generate(samples, inputs, atts)
        out = llm.generate(inputs_embeds=e, attention_mask=m, max_new_tokens=6, do_sample=False)
    assert out.shape == (2, 6) and len(tok.batch_decode(out)) == 2
    e, m, t = asm.assemble_forward(samples, inputs, atts)
    loss = llm(inputs_embeds=e, attention_mask=m, labels=t, return_dict=True).loss
    loss.backward()
    assert torch.isfinite(loss) and llm.lm_head.weight.grad.abs().sum() > 0
