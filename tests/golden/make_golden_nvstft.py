"""Golden log-mel from the REAL vdecoder/nsf_hifigan/nvSTFT.STFT.get_mel (build container only).  librosa / soundfile are
not installed: `librosa.filters.mel` is stubbed with oracle.mel.mel_filterbank (the Slaney basis restated from librosa
0.9.1), so this pins the STFT / padding / magnitude / log chain, not the basis (UNPINNED, as for modules/mel_processing).
usage: python tests/golden/make_golden_nvstft.py"""
import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)


def main():
    from oracle import nsf_hifigan_oracle as NO
    from oracle.mel import mel_filterbank
    for name in ("librosa", "librosa.filters", "librosa.core", "soundfile"):
        sys.modules[name] = types.ModuleType(name)
    sys.modules["librosa.filters"].mel = lambda sr, n_fft, n_mels, fmin, fmax: mel_filterbank(sr, n_fft, n_mels, fmin, fmax)
    sys.modules["librosa"].filters = sys.modules["librosa.filters"]
    sys.path.insert(0, "/root/reference")
    from vdecoder.nsf_hifigan.nvSTFT import STFT
    g = torch.Generator().manual_seed(77)
    y = 0.8 * (2 * torch.rand(2, 512 * 24, generator=g) - 1)
    y[1, 3000:5000] *= 0.01
    st = STFT(44100, 128, 2048, 2048, 512, 40, 16000)
    ref = st.get_mel(y)
    mine = NO.get_mel(y)
    d = (ref - mine).abs().max().item()
    print(f"oracle vs reference nvSTFT.get_mel: max|diff| {d:.3e}, shape {tuple(ref.shape)}, range [{ref.min():.2f}, {ref.max():.2f}]")
    assert d < 1e-4
    np.savez_compressed(os.path.join(HERE, "nvstft_mel.npz"), y=y.numpy(), mel=ref.numpy())
    print("wrote nvstft_mel.npz")


if __name__ == "__main__":
    main()
