"""Mirror of vencoder/encoder.py: the speech-unit encoder interface Svc uses (`hubert_model.encoder(wav16k)`)."""


class SpeechEncoder(object):
    def __init__(self, vec_path="pretrain/checkpoint_best_legacy_500.pt", device=None):
        self.model = None
        self.hidden_dim = 768

    def encoder(self, wav):
        """wav: 16 kHz mono [n] -> units [1, hidden_dim, n_frames] (50 fps)."""
        raise NotImplementedError
