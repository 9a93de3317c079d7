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
        run waves of ANY lengths as padded batches with per-item lengths (`batch_padded`, Hubert.encode(lengths=...): masked
        GroupNorm statistics, masked positional-conv input, padding-mask attention) — every item exactly as if encoded alone."""
        return [self.encoder(w) for w in wavs]


def batch_padded(wavs, run, max_waste=0.25):
    """wavs: list of [n] tensors of ANY lengths; run([B, 1, n_max], lengths) -> ([B, C, T_max], frames per item).  Waves are grouped
    by length so that a group's zero padding stays below `max_waste` of its longest item (sorted greedy), each group runs as one
    padded batch with per-item lengths, and every item comes back cut to its own frames, in the input order."""
    import torch
    out = [None] * len(wavs)
    flat = [(i, (w.mean(-1) if w.dim() == 2 else w)) for i, w in enumerate(wavs)]
    flat.sort(key=lambda t: -int(t[1].shape[0]))
    k = 0
    while k < len(flat):
        n_max = int(flat[k][1].shape[0])
        grp = [flat[k]]
        k += 1
        while k < len(flat) and int(flat[k][1].shape[0]) >= (1.0 - max_waste) * n_max:
            grp.append(flat[k])
            k += 1
        x = torch.zeros((len(grp), 1, n_max), dtype=grp[0][1].dtype, device=grp[0][1].device)
        for b, (_, w) in enumerate(grp):
            x[b, 0, :w.shape[0]] = w
        y, frames = run(x, [int(w.shape[0]) for _, w in grp])
        for b, (i, _) in enumerate(grp):
            out[i] = y[b:b + 1, :, :frames[b]].contiguous()
    return out


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
