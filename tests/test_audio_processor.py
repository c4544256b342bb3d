"""Input side of row A1: the BEATs filterbank front end (``BeatsAudioProcessor`` call signature of
``evaluate.py:24`` / ``utils/trainer.py:46``).  torchaudio and LAVIS are absent, the reference holds no audio
fixtures -> parity unpinned; checked against a by-definition restatement and analytic properties."""
import math

import numpy as np
import torch

from mraudio_amd.processors.audio_processors import FBANK_MEAN, FBANK_STD, BeatsAudioProcessor, kaldi_fbank, mel_banks


def _fbank_by_definition(x, sr=16000, nmel=128):
    """Frame by frame, DFT as an explicit matrix product, filters from the textbook triangle formula."""
    x = np.asarray(x, dtype=np.float64)
    win, shift, nfft = 400, 160, 512
    frames = 1 + (len(x) - win) // shift
    k = np.arange(nfft // 2)
    dft = np.exp(-2j * np.pi * np.outer(np.arange(nfft), k) / nfft)
    mel = lambda f: 1127.0 * np.log(1.0 + f / 700.0)
    lo, hi = mel(20.0), mel(sr / 2)
    d = (hi - lo) / (nmel + 1)
    fm = mel(k * sr / nfft)
    out = np.zeros((frames, nmel))
    w = (0.5 - 0.5 * np.cos(2 * np.pi * np.arange(win) / (win - 1))) ** 0.85
    for t in range(frames):
        f = x[t * shift: t * shift + win].copy()
        f -= f.mean()
        g = f.copy()
        g[1:] -= 0.97 * f[:-1]
        g[0] -= 0.97 * f[0]
        g *= w
        p = np.abs(np.concatenate([g, np.zeros(nfft - win)]) @ dft) ** 2
        for b in range(nmel):
            l, c, r = lo + b * d, lo + (b + 1) * d, lo + (b + 2) * d
            wt = np.where((fm > l) & (fm <= c), (fm - l) / (c - l), np.where((fm > c) & (fm < r), (r - fm) / (r - c), 0.0))
            out[t, b] = np.log(max(float(p @ wt), np.finfo(np.float32).eps))
    return out


def test_fbank_matches_the_definition():
    g = torch.Generator().manual_seed(0)
    x = (torch.randn(16000 // 4, generator=g) * 3000.0)
    got = kaldi_fbank(x).numpy()
    want = _fbank_by_definition(x.numpy())
    assert got.shape == want.shape == (1 + (4000 - 400) // 160, 128)
    assert np.abs(got - want).max() < 2e-4


def test_sine_lands_in_its_mel_bin_and_filters_are_triangles():
    sr = 16000
    t = torch.arange(sr, dtype=torch.float64) / sr
    banks = mel_banks(128, 512, sr)
    assert banks.shape == (128, 256) and (banks >= 0).all() and banks.max() <= 1.0 + 1e-12
    assert ((banks > 0).sum(1) >= 1).float().mean() > 0.95       # a few sub-bin-width filters at the low end are empty, as in Kaldi
    for f in (440.0, 1000.0, 3000.0):
        fb = kaldi_fbank(torch.sin(2 * math.pi * f * t) * 8000.0)
        peak = int(fb.mean(0).argmax())
        centre = 1127.0 * math.log(1 + f / 700.0)
        lo, hi = 1127.0 * math.log(1 + 20 / 700.0), 1127.0 * math.log(1 + 8000 / 700.0)
        expect = (centre - lo) / ((hi - lo) / 129) - 1.0        # filter whose centre is nearest to f on the mel axis
        assert abs(peak - expect) <= 1.0, (f, peak, expect)
    assert kaldi_fbank(torch.zeros(100)).shape == (0, 128)        # shorter than one window: no frame (snip_edges)


def test_processor_signature_and_shapes():
    wave = torch.sin(torch.arange(16000 * 6, dtype=torch.float32) * 0.05) * 0.3
    proc = BeatsAudioProcessor(model_name="iter3", sampling_rate=16000, n_frames=4, is_eval=True, frame_length=512,
                               reader=lambda path: (wave, 16000))
    out = proc("clip.mp4")
    assert out.shape == (4, 512, 128) and out.dtype == torch.float32
    # 1.5 s per position = 148 frames, the rest of the 512 is zero padding
    assert (out[:, 148:] == 0).all() and out[:, :148].abs().sum() > 0
    seg = proc.features(wave[:24000])
    raw = kaldi_fbank(wave[:24000] * 32768.0)
    assert torch.allclose(seg[: raw.shape[0]], (raw - FBANK_MEAN) / (2 * FBANK_STD))
    # other sample rates are resampled to the model's; a position longer than frame_length frames is cut
    proc2 = BeatsAudioProcessor(n_frames=1, frame_length=64, reader=lambda p: (wave[::2], 8000))
    assert proc2("x").shape == (1, 64, 128)
    try:
        BeatsAudioProcessor()("x")
        raise AssertionError("must demand a reader")
    except RuntimeError:
        pass
