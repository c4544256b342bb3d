"""Rows N1 / N3 on the device (``-m gpu``): the training loop and the inference loop around the HIP path, on
the seeded synthetic corpus (no dataset exists offline), scored by the reference-pinned metrics module."""
import ast
import json
import os

import pytest
import torch
from torch.utils.data import DataLoader

from mraudio_amd.eval.mr_eval import eval_submission
from mraudio_amd.utils.mr_dataset import SyntheticMRDataset, collate_fn

pytestmark = pytest.mark.gpu


def test_inference_loop_writes_scorable_predictions(tmp_path):
    from mraudio_amd.evaluate import run_inference
    from mraudio_amd.models.xinstructblip import XInstructBLIP

    dev = torch.device("cuda:0")
    model = XInstructBLIP(seed=0, device=dev)
    ds = SyntheticMRDataset(5, T=20, seed=3, duration=40)
    dl = DataLoader(ds, batch_size=2, shuffle=False, collate_fn=collate_fn)
    out = tmp_path / "pred" / "charades.jsonl"
    recs = run_inference(model, dl, str(out), device=dev)
    lines = [json.loads(x) for x in out.read_text().splitlines()]
    assert len(recs) == len(lines) == 5 and [r["qid"] for r in lines] == list(range(5))
    gt = []
    for i in range(5):
        rec = ds[i]
        wins = ast.literal_eval(rec["text_output"])
        ids = sorted({c for w in wins for c in range(w[0] // 2, min(20, w[1] // 2 + 1))})
        gt.append({"qid": i, "query": rec["query"], "vid": rec["vid"], "duration": 40, "relevant_windows": wins,
                   "relevant_clip_ids": ids, "saliency_scores": [[3, 3, 3]] * len(ids)})
    for r in lines:
        (s, e), = r["pred_relevant_windows"]
        assert 0 <= s <= e <= 40 and len(r["pred_saliency_scores"]) == 20
        assert r["raw_out"] == f"[[{s}, {e}]]"
    m = eval_submission(lines, gt, verbose=False)
    assert m["brief"]["MR-full-invalid_pred_num"] == 0 and 0.0 <= m["brief"]["MR-full-mIoU"] <= 1.0
    assert "HL-min-Good-mAP" in m["brief"]
    # a batch of one video gives the same prediction as a batch of two (items are independent)
    solo = run_inference(model, DataLoader(ds, batch_size=1, shuffle=False, collate_fn=collate_fn), None, device=dev)
    assert [r["pred_relevant_windows"] for r in solo] == [r["pred_relevant_windows"] for r in lines]


def test_trainer_epoch_on_the_hip_path(tmp_path):
    from mraudio_amd.utils.trainer import Trainer, default_args

    args = default_args(output_dir=str(tmp_path), gpu=0, max_epoch=2, synthetic=4, dataset="Charades_STA", lr=2e-5, warmup_steps=1)
    tr = Trainer(args)
    res = tr.train()
    assert len(tr.history) == 2 and all(torch.isfinite(torch.tensor(h["loss_value"])) for h in tr.history)
    assert tr.history[1]["loss_value"] < tr.history[0]["loss_value"]          # same four videos twice: the loss must drop
    assert "MR-full-R1-avg" in tr.history[0] and set(res) == {"best_epoch", "best_metric"}
    ck = torch.load(os.path.join(tmp_path, "checkpoint_1.pth"), weights_only=True)
    keys = set(ck["model"])
    assert "video_Qformer.bert.encoder.layer.0.crossattention.self.key.weight" in keys and "audio_query_tokens" in keys
    assert not any(k.startswith("video_ln") or k.startswith("audio_llm_proj") for k in keys)   # frozen -> not saved
    # resume continues at epoch 2 with the trained weights
    args2 = default_args(output_dir=str(tmp_path), gpu=0, max_epoch=3, synthetic=4, dataset="Charades_STA", lr=2e-5, warmup_steps=1,
                         resume_ckpt_path=os.path.join(tmp_path, "checkpoint_1.pth"))
    tr2 = Trainer(args2)
    tr2.train()
    assert [h["epoch"] for h in tr2.history] == [2]
    w = "video_Qformer.bert.encoder.layer.0.crossattention.self.key.weight"
    assert tr2.history[0]["loss_value"] < tr.history[0]["loss_value"]


def test_llm_decode_path_with_a_stock_llama():
    """Row N2 on the device: hot path -> llm_proj -> prompt assembly -> an unmodified HF Llama (tiny random
    config, hidden 256 = the projections' output width here) -> strings / LM loss, as reference generate / forward."""
    tf = pytest.importorskip("transformers")
    from mraudio_amd.models.llm_prompt import SimpleLlmTokenizer
    from mraudio_amd.models.xinstructblip import XInstructBLIP

    dev = torch.device("cuda:0")
    torch.manual_seed(0)
    tok = SimpleLlmTokenizer()
    cfg = tf.LlamaConfig(vocab_size=len(tok), hidden_size=256, intermediate_size=512, num_hidden_layers=2, num_attention_heads=4,
                         num_key_value_heads=4, max_position_embeddings=4096, pad_token_id=tok.pad_token_id, bos_token_id=2, eos_token_id=2)
    llm = tf.LlamaForCausalLM(cfg).to(dev).half().eval()
    model = XInstructBLIP(seed=0, device=dev, llm_hidden_size=256)
    model.attach_llm(llm, tok)
    ds = SyntheticMRDataset(2, T=6, seed=5, duration=12)
    batch = collate_fn([ds[0], ds[1]])
    batch = {k: (v.to(dev) if torch.is_tensor(v) else v) for k, v in batch.items()}
    out = model.generate_llm(batch, max_new_tokens=8)
    assert isinstance(out, list) and len(out) == 2 and all(isinstance(o, str) for o in out)
    # the assembled sequence: 6 positions x (cue + 32 + cue + 32 + seconds) + duration + prompt, all from the HIP path's projections
    inputs_llm, atts_llm = model._llm_inputs(batch)
    assert inputs_llm["video"].shape == (2, 6 * 32, 256) and inputs_llm["video"].dtype == torch.float16
    ref = model.encode_fuse(batch, want_llm=True)["inputs_llm"]["audio"]
    assert torch.equal(inputs_llm["audio"], ref.to(torch.float16))
    loss = model.forward_llm(batch)["loss"]
    assert torch.isfinite(loss)
    assert "llm_model.lm_head.weight" not in model.state_dict()     # the LLM is not part of this model's checkpoint


def test_checkpoint_argument_loads_the_reference_key_names(tmp_path):
    """``XInstructBLIP(model_path, audio_path)`` used to ignore both arguments and always run on the synthetic init.
    ``checkpoint=`` (CLI ``--checkpoint``) loads a weights-only ``.pth`` with the reference's key names (models/xinstructblip.py:
    759-816 routing); a file with none of the model's keys is an error, not a silent no-op."""
    from mraudio_amd.models.xinstructblip import XInstructBLIP

    dev = torch.device("cuda:0")
    src = XInstructBLIP(seed=7, perturb=True, device=dev)
    g = torch.Generator().manual_seed(2)
    samples = {"video_embeds": torch.randn(1, 4, 40, 1408, generator=g), "audio_embeds": torch.randn(1, 4, 24, 768, generator=g),
               "text_input": ["Query: a cat jumps.\nRelevant windows: "], "timestamps": [[0, 2, 4, 6]], "duration": [8]}
    want = src.encode_fuse(samples)["fused"].clone()
    path = tmp_path / "weights.pth"
    torch.save({"model": {k: v.cpu() for k, v in src.state_dict().items()}}, path)
    other = XInstructBLIP("/path/to/vicuna", "/path/to/beats.pt", seed=0, device=dev, checkpoint=str(path))
    assert "checkpoint" in other.weights_source
    assert torch.equal(other.encode_fuse(samples)["fused"], want)
    fresh = XInstructBLIP(seed=0, device=dev)
    assert fresh.weights_source.startswith("synthetic") and not torch.equal(fresh.encode_fuse(samples)["fused"], want)
    bogus = tmp_path / "bogus.pth"
    torch.save({"llm_model.weight": torch.zeros(2)}, bogus)
    with pytest.raises(RuntimeError):
        XInstructBLIP(seed=0, device=dev, checkpoint=str(bogus))
    # load_checkpoint is STRICT like the reference's (models/xinstructblip.py:759-767); a partial file is only accepted
    # through the explicit non-strict loader (reference :749-757), and what it lacks is counted in weights_source
    partial = tmp_path / "partial.pth"
    torch.save({k: v.cpu() for k, v in src.state_dict().items() if k.startswith("video_ln")}, partial)
    with pytest.raises(RuntimeError):
        XInstructBLIP(seed=0, device=dev, checkpoint=str(partial))
    loose = XInstructBLIP(seed=0, device=dev)
    msg = loose.load_from_pretrained(str(partial))
    assert msg.missing_keys and "NOT in the file" in loose.weights_source
    assert torch.equal(loose.video_ln.weight.cpu(), src.video_ln.weight.cpu())
