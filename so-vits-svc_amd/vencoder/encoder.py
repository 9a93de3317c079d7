"""Mirror of vencoder/encoder.py: the speech-unit encoder interface Svc uses (`hubert_model.encoder(wav16k)`)."""


class SpeechEncoder(object):
    def __init__(self, vec_path="pretrain/checkpoint_best_legacy_500.pt", device=None):
        self.model = None
        self.hidden_dim = 768

    def encoder(self, wav):
        """wav: 16 kHz mono [n] -> units [1, hidden_dim, n_frames] (50 fps)."""
        raise NotImplementedError

    def encoder_batch(self, wavs):
        """Engine extension (`Svc.slice_inference(batch_chunks=True)`): units of several 16 kHz waves, [1, hidden_dim, n_frames]
        each.  Default: one `encoder` call per wave (encoders loaded from the reference tree); the engine's HuBERT-based encoders
        run waves of EQUAL length as one batch — every op of that stack is per item (GroupNorm(512, 512) normalises each item
        over its own time axis), so equal lengths batch exactly; unequal lengths cannot (zero padding would enter that norm)."""
        return [self.encoder(w) for w in wavs]


def batch_equal_lengths(wavs, run):
    """wavs: list of [n] tensors; run([B, 1, n]) -> [B, C, T].  Groups equal lengths, keeps the order."""
    import torch
    out = [None] * len(wavs)
    groups = {}
    for i, w in enumerate(wavs):
        w = w.mean(-1) if w.dim() == 2 else w
        groups.setdefault(int(w.shape[0]), []).append((i, w))
    for items in groups.values():
        x = torch.stack([w for _, w in items], 0)[:, None, :]
        y = run(x)
        for b, (i, _) in enumerate(items):
            out[i] = y[b:b + 1]
    return out
