"""Mirror of vencoder/ContentVec768L12.py: the default `vec768l12` speech encoder (768-d output of transformer layer 12 of the
ContentVec HuBERT-base model) on the MI355X engine.  The reference loads the checkpoint through fairseq
(`checkpoint_utils.load_model_ensemble_and_task`, :12-15) and calls `model.extract_features(source, padding_mask,
output_layer=12)` (:28-36); here the same checkpoint file is mapped onto vencoder/hubert/hubert_model.Hubert without fairseq
(`load_fairseq_hubert`) and `Hubert.encode(wav, layer=12)` runs the stack in libsvc_hip.so."""
import torch

from vencoder.encoder import SpeechEncoder, batch_padded
from vencoder.hubert import hubert_model


class ContentVec768L12(SpeechEncoder):
    OUTPUT_LAYER = 12
    USE_FINAL_PROJ = False

    def __init__(self, vec_path="pretrain/checkpoint_best_legacy_500.pt", device=None, model=None):
        super().__init__()
        self.hidden_dim = 256 if self.USE_FINAL_PROJ else 768
        if device is None:
            if not torch.cuda.is_available():
                raise RuntimeError(f"{type(self).__name__}: no GPU visible and the MI355X engine has no CPU fallback")
            self.dev = torch.device("cuda")
        else:
            self.dev = torch.device(device)
        if model is None:
            print("load model(s) from {}".format(vec_path))
            model = hubert_model.load_fairseq_hubert(vec_path)
        self.model = model.to(self.dev).eval()

    def encoder(self, wav):
        """wav: 16 kHz mono [n] (or [n, channels]) -> [1, hidden_dim, n_frames]."""
        feats = wav
        if feats.dim() == 2:  # double channels
            feats = feats.mean(-1)
        assert feats.dim() == 1, feats.dim()
        with torch.no_grad():
            x, _ = self.model.encode(feats.view(1, 1, -1).to(self.dev), layer=self.OUTPUT_LAYER)      # [1, 768, T]
            if self.USE_FINAL_PROJ:
                x = self.model.project(x)
        return x

    def encoder_batch(self, wavs):
        def run(x, lengths):
            y, _ = self.model.encode(x.to(self.dev), layer=self.OUTPUT_LAYER, lengths=lengths)
            return (self.model.project(y) if self.USE_FINAL_PROJ else y), self.model.last_frames
        with torch.no_grad():
            return batch_padded(wavs, run)
