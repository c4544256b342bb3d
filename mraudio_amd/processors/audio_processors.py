"""Audio processor with the call signature the reference uses (``evaluate.py:24``, ``utils/trainer.py:46``):
``BeatsAudioProcessor(model_name, sampling_rate, n_frames, is_eval, frame_length)(path) -> Tensor[T, F, 128]``
-- T temporal positions of F = ``frame_length`` filterbank frames with 128 mel bins, the input the BEATs encoder
(row A1) patches into 16 x 16 tokens (512 x 128 -> 256 tokens per position).

The reference takes this class from the third-party ``salesforce-lavis`` package (unpinned, absent offline), which in
turn calls ``torchaudio.compliance.kaldi.fbank`` (absent too).  What is restated here is the PUBLISHED algorithm of that
call with the arguments BEATs uses (``num_mel_bins=128, sample_frequency=16000, frame_length=25, frame_shift=10``;
Kaldi defaults otherwise: no dither, DC-offset removal, 0.97 pre-emphasis, povey window, 512-point FFT, power
spectrum, mel banks from 20 Hz to Nyquist, log with a float-epsilon floor, ``snip_edges``) followed by BEATs'
normalisation ``(x - 15.41663) / (2 * 6.55582)``.  Neither package can be imported here and the reference holds no
audio fixtures: **parity unpinned**; ``tests/test_audio_processor.py`` checks the implementation against a direct
(DFT-by-definition) restatement and against analytic properties.  Decoding the audio track of an ``.mp4`` is IO outside
the path: pass ``reader(path) -> (waveform float tensor [samples] in [-1, 1], sample_rate)``.
"""
from __future__ import annotations

import math
from typing import Callable, Optional, Tuple

import torch

FBANK_MEAN, FBANK_STD = 15.41663, 6.55582      # BEATs' dataset statistics (its preprocess())


def _mel(f):
    return 1127.0 * math.log(1.0 + f / 700.0)


def mel_banks(num_bins: int, padded_window: int, sample_rate: float, low_freq: float = 20.0, high_freq: float = 0.0) -> torch.Tensor:
    """Kaldi's triangular mel filters: ``[num_bins, padded_window // 2]`` (the Nyquist bin carries no weight)."""
    nyquist = 0.5 * sample_rate
    if high_freq <= 0.0:
        high_freq += nyquist
    n_fft_bins = padded_window // 2
    fft_bin_width = sample_rate / padded_window
    mel_lo, mel_hi = _mel(low_freq), _mel(high_freq)
    delta = (mel_hi - mel_lo) / (num_bins + 1)
    b = torch.arange(num_bins, dtype=torch.float64).unsqueeze(1)
    left, center, right = mel_lo + b * delta, mel_lo + (b + 1.0) * delta, mel_lo + (b + 2.0) * delta
    mel = 1127.0 * torch.log(1.0 + fft_bin_width * torch.arange(n_fft_bins, dtype=torch.float64) / 700.0).unsqueeze(0)
    up, down = (mel - left) / (center - left), (right - mel) / (right - center)
    return torch.clamp(torch.minimum(up, down), min=0.0)


def kaldi_fbank(waveform: torch.Tensor, num_mel_bins: int = 128, sample_frequency: float = 16000.0, frame_length: float = 25.0,
                frame_shift: float = 10.0, preemphasis: float = 0.97, low_freq: float = 20.0, high_freq: float = 0.0) -> torch.Tensor:
    """Log mel filterbank energies ``[frames, num_mel_bins]`` of a mono waveform ``[samples]`` (Kaldi ``compute-fbank-feats``
    conventions as exposed by ``torchaudio.compliance.kaldi.fbank`` with its defaults)."""
    x = waveform.reshape(-1).to(torch.float64)
    win = int(sample_frequency * frame_length * 0.001)
    shift = int(sample_frequency * frame_shift * 0.001)
    padded = 1 << (win - 1).bit_length()                      # round_to_power_of_two
    if x.numel() < win:
        return torch.zeros(0, num_mel_bins)
    n_frames = 1 + (x.numel() - win) // shift                 # snip_edges
    frames = x.unfold(0, win, shift)[:n_frames].clone()       # [frames, win]
    frames = frames - frames.mean(dim=1, keepdim=True)        # remove_dc_offset
    if preemphasis != 0.0:                                    # x[i] -= c * x[i - 1], the first sample against itself
        prev = torch.cat([frames[:, :1], frames[:, :-1]], dim=1)
        frames = frames - preemphasis * prev
    n = torch.arange(win, dtype=torch.float64)
    window = (0.5 - 0.5 * torch.cos(2.0 * math.pi * n / (win - 1))) ** 0.85      # povey
    frames = frames * window
    spec = torch.fft.rfft(frames, n=padded, dim=1)
    power = spec.real ** 2 + spec.imag ** 2                   # [frames, padded / 2 + 1]
    banks = mel_banks(num_mel_bins, padded, sample_frequency, low_freq, high_freq)
    energies = power[:, : padded // 2] @ banks.t()
    return torch.log(torch.clamp(energies, min=torch.finfo(torch.float32).eps)).to(torch.float32)


class BeatsAudioProcessor:
    def __init__(self, model_name: str = "iter3", sampling_rate: int = 16000, n_frames: int = 2, is_eval: bool = False,
                 frame_length: int = 512, reader: Optional[Callable[[str], Tuple[torch.Tensor, int]]] = None):
        self.model_name, self.sampling_rate, self.n_frames, self.is_eval, self.frame_length = model_name, sampling_rate, n_frames, is_eval, frame_length
        self.reader = reader

    def _resample(self, wav: torch.Tensor, sr: int) -> torch.Tensor:
        if sr == self.sampling_rate:
            return wav
        n_out = int(round(wav.numel() * self.sampling_rate / sr))             # linear interpolation; IO-side convenience
        return torch.nn.functional.interpolate(wav.view(1, 1, -1), size=n_out, mode="linear", align_corners=False).view(-1)

    def features(self, segment: torch.Tensor) -> torch.Tensor:
        """One temporal position: ``[frame_length, 128]`` normalised filterbank frames (zero-padded or cut at the end)."""
        fb = kaldi_fbank(segment * (1 << 15), num_mel_bins=128, sample_frequency=self.sampling_rate)
        fb = (fb - FBANK_MEAN) / (2.0 * FBANK_STD)
        out = torch.zeros(self.frame_length, 128)
        k = min(self.frame_length, fb.shape[0])
        out[:k] = fb[:k]
        return out

    def __call__(self, path) -> torch.Tensor:
        if self.reader is None:
            raise RuntimeError("no audio reader configured (decoding is not part of this build); pass reader=")
        wav, sr = self.reader(path)
        wav = self._resample(torch.as_tensor(wav, dtype=torch.float32).reshape(-1), int(sr))
        # n_frames equal windows over the clip, one per temporal position (the video processor samples its frames the same way)
        edges = torch.linspace(0, wav.numel(), self.n_frames + 1).long().tolist()
        return torch.stack([self.features(wav[a:b]) for a, b in zip(edges[:-1], edges[1:])])
