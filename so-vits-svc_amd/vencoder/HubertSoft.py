"""Mirror of vencoder/HubertSoft.py: the `hubertsoft` speech encoder (256-d soft units) on the MI355X engine."""
import torch

from vencoder.encoder import SpeechEncoder, batch_padded
from vencoder.hubert import hubert_model


class HubertSoft(SpeechEncoder):
    def __init__(self, vec_path="pretrain/hubert-soft-0d54a1f4.pt", device=None, model=None):
        super().__init__()
        if device is None:
            if not torch.cuda.is_available():
                raise RuntimeError("HubertSoft: no GPU visible and the MI355X engine has no CPU fallback")
            self.dev = torch.device("cuda")
        else:
            self.dev = torch.device(device)
        self.hidden_dim = 256
        if model is None:
            print("load model(s) from {}".format(vec_path))
            model = hubert_model.hubert_soft(vec_path)
        self.model = model.to(self.dev).eval()

    def encoder(self, wav):
        feats = wav
        if feats.dim() == 2:  # double channels
            feats = feats.mean(-1)
        assert feats.dim() == 1, feats.dim()
        feats = feats[None, None, :]
        with torch.no_grad():
            units = self.model.units(feats.to(self.dev))
            return units.transpose(1, 2)

    def encoder_batch(self, wavs):
        def run(x, lengths):
            u = self.model.units(x.to(self.dev), lengths=lengths).transpose(1, 2)
            return u, self.model.last_frames
        with torch.no_grad():
            return batch_padded(wavs, run)
