"""The C-ABI library loads on a CPU-only box and exports every symbol include/mra.h declares.
No compute call is made here (there is no GPU)."""
import ctypes
import os
import re

from mraudio_amd import _lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_functions():
    text = open(os.path.join(ROOT, "include", "mra.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(mra_[a-z_0-9]+)\s*\(", text)))


def test_header_symbols_are_exported_and_bound():
    names = declared_functions()
    assert len(names) >= 15
    handle = ctypes.CDLL(_lib.LIB_PATH)
    for n in names:
        assert hasattr(handle, n), f"{n} declared in include/mra.h but not exported"
        assert n in _lib.PROTOTYPES, f"{n} has no ctypes prototype"
    assert sorted(_lib.PROTOTYPES) == names


def test_version_and_defaults_without_gpu():
    lib = _lib.lib()
    assert b"gfx950" in lib.mra_version()
    cfg = _lib.mra_cfg()
    lib.mra_cfg_default(ctypes.byref(cfg), 1408)
    assert (cfg.hidden, cfg.heads, cfg.inter, cfg.layers, cfg.cross_freq) == (768, 12, 3072, 12, 2)
    assert (cfg.enc_width, cfg.n_query, cfg.vocab, cfg.max_pos, cfg.llm_hidden) == (1408, 32, 30523, 512, 4096)
    assert abs(cfg.ln_eps - 1e-12) < 1e-18 and abs(cfg.enc_ln_eps - 1e-5) < 1e-11
    assert cfg.op_dtype == _lib.MRA_F16


def test_bad_arguments_are_rejected_before_touching_the_gpu():
    lib = _lib.lib()
    assert lib.mra_qformer_create(None, None) == -1
    assert b"null" in lib.mra_last_error()
    assert lib.mra_qformer_workspace_bytes(None, 4, 4, 4) == 0
    assert lib.mra_cosine_score(None, None, 1, 4, 32, 768, None, None, None) == -1
    assert lib.mra_span_from_logits(None, 0, 4, 0.5, None, None) == 0  # empty input is a no-op
