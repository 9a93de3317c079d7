"""MI355X-native mirror of vencoder/hubert/hubert_model.py (HuBERT-base "soft" unit encoder).

Same classes, constructor signatures and `state_dict` keys as the reference (the torch modules below are used ONLY as
parameter containers, so a reference checkpoint — `hubert-soft-0d54a1f4.pt` — loads key-for-key); every forward op runs
on libsvc_hip.so in the engine's [B, C, T] layout (the reference's [B,T,C] Linear / LayerNorm are 1x1 convs / channel
LayerNorms there, so the two `transpose(1, 2)` of :42,:124-127 never materialise):

  FeatureExtractor (:71-94)   conv0 (1->512, k10, s5)            svc_conv1d_direct_f32
                              GroupNorm(512,512) + GELU          svc_channel_norm_gelu_f32
                              conv1..6 (512->512, k3|k2, s2)     svc_decimate_f32 + svc_conv1d_f32 (MFMA, GELU epilogue)
  FeatureProjection (:97-108) LayerNorm(512) + Linear(512,768)   svc_add_layernorm_f32 + 1x1 MFMA conv
  PositionalConvEmbedding     weight-normed (dim=2) grouped conv k128, groups 16, drop last, GELU   svc_gconv1d_fwd_f32
  (:111-127)
  TransformerEncoder (:130-152) 12 x post-norm layers: fused qkv 1x1 conv, flash attention (12 heads x 64),
                              out_proj, add+LayerNorm, FFN 768->3072 (GELU epilogue) ->768, add+LayerNorm
  proj (:26,:68)              Linear(768,256)                    1x1 MFMA conv

The fairseq ContentVec encoders (vencoder/ContentVec768L12.py, ContentVec256L9.py: fairseq `HubertModel`,
`extract_features(source, padding_mask, output_layer=L)`) are the same network under fairseq's parameter names;
`hubert_from_fairseq_state_dict` / `load_fairseq_hubert` below map a `checkpoint_best_legacy_500.pt` onto this module
(q_proj|k_proj|v_proj -> in_proj, fc1/fc2 -> linear1/2, post_extract_proj -> feature_projection.projection, ...) without
importing fairseq.  fairseq and its checkpoint are absent from this image, so that mapping is UNPINNED against fairseq itself;
the arithmetic is pinned through this in-tree variant (tests/golden/hubert_soft_1s.npz from the real module).  Inference only.
"""
import copy
from typing import Optional

import torch
import torch.nn as nn

import svc_hip as S


def _pack(w):
    return S.pack_conv1d_weight(w.detach().contiguous())


class Hubert(nn.Module):
    def __init__(self, num_label_embeddings: int = 100, mask: bool = True):
        super().__init__()
        self._mask = mask
        self.feature_extractor = FeatureExtractor()
        self.feature_projection = FeatureProjection()
        self.positional_embedding = PositionalConvEmbedding()
        self.norm = nn.LayerNorm(768)
        self.dropout = nn.Dropout(0.1)
        self.encoder = TransformerEncoder(nn.TransformerEncoderLayer(768, 12, 3072, activation="gelu", batch_first=True), 12)
        self.proj = nn.Linear(768, 256)
        self.masked_spec_embed = nn.Parameter(torch.FloatTensor(768).uniform_())
        self.label_embedding = nn.Embedding(num_label_embeddings, 256)
        self._cache = {}

    def _packed(self, name, fn):
        """Packed weights keyed by the parameter versions they were built from."""
        params = [p for p in self.parameters()]
        key = (name, sum(p._version for p in params), str(params[0].device))
        hit = self._cache.get(name)
        if hit is None or hit[0] != key:
            self._cache[name] = (key, fn())
        return self._cache[name][1]

    def encode(self, x: torch.Tensor, layer: Optional[int] = None, lengths=None):
        """x: [B, 1, n] 16 kHz waveform -> ([B, 768, T] features in the engine's channel-major layout, None).

        `lengths` (engine extension; list of B sample counts): the batch holds waves of DIFFERENT lengths, zero-padded to n
        (Svc.slice_inference's chunks, inference/infer_tool.py:446-495, which the reference encodes one by one).  Every item then
        comes out exactly as if it had been encoded alone, on its frames [:, :, :frames(lengths[b])]: the only ops of the stack
        that look along time are (1) GroupNorm(512, 512) — per-item statistics over the item's own conv0 output length
        (svc_channel_norm_gelu_len_f32), (2) the positional conv, whose padding must be zeros — the projected features are
        masked to zero beyond each item's frames first — and (3) attention, which gets the padding mask (keys beyond an item's
        frames weigh exp(-1e4 - max) == 0 in fp32).  The strided convs are local: a valid output never reads beyond its item's
        valid input.  Returns (features, None) and leaves the per-item frame counts in `self.last_frames`."""
        if self.training and self._mask:
            raise NotImplementedError("masked training of the unit encoder is out of scope (inference only)")
        if not x.is_cuda:
            raise S.SvcError("Hubert.encode needs CUDA/ROCm tensors: the MI355X engine has no CPU fallback")
        with torch.no_grad():
            if lengths is None:
                x = self.feature_extractor(x.float().contiguous(), self)
                x = self.feature_projection(x, self)
                x = self.positional_embedding(x, self)
                x = S.add_layernorm(x, None, self.norm.weight, self.norm.bias, eps=self.norm.eps)
                x = self.encoder(x, self, output_layer=layer)
                self.last_frames = [x.shape[2]] * x.shape[0]
                return x, None
            lengths = [int(v) for v in lengths]
            if len(lengths) != x.shape[0] or max(lengths) > x.shape[2] or min(lengths) < 400:
                raise S.SvcError(f"Hubert.encode: lengths {lengths} do not fit a batch of shape {tuple(x.shape)} (>= 400 samples each)")
            x, frames = self.feature_extractor(x.float().contiguous(), self, lengths=lengths)
            T = x.shape[2]
            fl = torch.tensor(frames, device=x.device, dtype=torch.int64)
            mask = (torch.arange(T, device=x.device).view(1, 1, T) < fl.view(-1, 1, 1)).to(torch.float32)      # [B, 1, T]
            x = self.feature_projection(x, self, mask=mask)
            x = self.positional_embedding(x, self)
            x = S.add_layernorm(x, None, self.norm.weight, self.norm.bias, eps=self.norm.eps)
            x = self.encoder(x, self, output_layer=layer, src_key_padding_mask=mask)
            self.last_frames = frames
        return x, None

    def project(self, x):
        """`proj` / fairseq `final_proj` on channel-major features: [B, 768, T] -> [B, P, T] (1x1 MFMA conv)."""
        wp = self._packed("proj", lambda: _pack(self.proj.weight.unsqueeze(-1)))
        return S.conv1d(x, wp, self.proj.out_features, 1, bias=self.proj.bias)

    def logits(self, x):
        raise NotImplementedError("label logits are a training-time head of HuBERT (out of scope)")

    def forward(self, x):
        raise NotImplementedError("Hubert.forward (masked prediction) is out of scope; use HubertSoft.units")


class HubertSoft(Hubert):
    def __init__(self):
        super().__init__()

    @torch.no_grad()
    def units(self, wav: torch.Tensor, lengths=None) -> torch.Tensor:
        """wav [B, 1, n] -> soft units [B, T, 256] (reference :63-68: zero-pad 40 samples each side, encode, proj).  `lengths`:
        see Hubert.encode (item b's units are rows [:self.last_frames[b]])."""
        x, _ = self._encode_padded(wav, lengths)
        return self.project(x).transpose(1, 2)

    def _encode_padded(self, wav, lengths=None):
        self.feature_extractor.pad = (400 - 320) // 2
        try:
            return self.encode(wav, lengths=lengths)
        finally:
            self.feature_extractor.pad = 0


class FeatureExtractor(nn.Module):
    def __init__(self):
        super().__init__()
        self.conv0 = nn.Conv1d(1, 512, 10, 5, bias=False)
        self.norm0 = nn.GroupNorm(512, 512)
        self.conv1 = nn.Conv1d(512, 512, 3, 2, bias=False)
        self.conv2 = nn.Conv1d(512, 512, 3, 2, bias=False)
        self.conv3 = nn.Conv1d(512, 512, 3, 2, bias=False)
        self.conv4 = nn.Conv1d(512, 512, 3, 2, bias=False)
        self.conv5 = nn.Conv1d(512, 512, 2, 2, bias=False)
        self.conv6 = nn.Conv1d(512, 512, 2, 2, bias=False)
        self.pad = 0

    @staticmethod
    def _strided_pack(w, s):
        """Weight of a stride-s conv re-indexed for decimate + dense conv (svc_autograd.conv1d's lowering, padding 0):
        input position t*s + k = (t + m)*s + r  ->  channel r*Cin + ci, tap m."""
        Cout, Cin, KS = w.shape
        KSd = (KS - 1) // s + 1
        wpad = torch.nn.functional.pad(w.detach(), (0, s * KSd - KS))
        wd = wpad.view(Cout, Cin, KSd, s).permute(0, 3, 1, 2).reshape(Cout, s * Cin, KSd).contiguous()
        return _pack(wd), KSd

    def forward(self, x, owner, lengths=None):
        """lengths (list of sample counts, batch of unequal waves): returns (h, frames per item) — see Hubert.encode."""
        B, _, n = x.shape
        w0 = owner._packed("conv0", lambda: _pack(self.conv0.weight))
        T = (n + 2 * self.pad - 10) // 5 + 1
        h = S.conv1d_direct(x, w0, 512, 10, stride=5, pad_left=self.pad, Tout=T)
        lens = None
        if lengths is not None:
            lens = [(m + 2 * self.pad - 10) // 5 + 1 for m in lengths]
            h = S.channel_norm_gelu(h, self.norm0.weight, self.norm0.bias, eps=self.norm0.eps,
                                    lengths=torch.tensor(lens, device=x.device, dtype=torch.int32))
        else:
            h = S.channel_norm_gelu(h, self.norm0.weight, self.norm0.bias, eps=self.norm0.eps)
        for i in range(1, 7):
            conv = getattr(self, f"conv{i}")
            KS = conv.kernel_size[0]
            wp, KSd = owner._packed(f"conv{i}", lambda conv=conv: self._strided_pack(conv.weight, 2))
            Tin = h.shape[2]
            Tout = (Tin - KS) // 2 + 1
            hd = S.decimate(h, 2, 0, (Tin + 1) // 2)                                  # [B, 1024, Q]
            h = S.conv1d(hd, wp, 512, KSd, pad_left=0, Tout=Tout, post_act=S.ACT_GELU)
            if lens is not None:
                lens = [(m - KS) // 2 + 1 for m in lens]
        return h if lengths is None else (h, lens)


class FeatureProjection(nn.Module):
    def __init__(self):
        super().__init__()
        self.norm = nn.LayerNorm(512)
        self.projection = nn.Linear(512, 768)
        self.dropout = nn.Dropout(0.1)

    def forward(self, x, owner, mask=None):
        x = S.add_layernorm(x, None, self.norm.weight, self.norm.bias, eps=self.norm.eps)
        wp = owner._packed("projection", lambda: _pack(self.projection.weight.unsqueeze(-1)))
        return S.conv1d(x, wp, 768, 1, bias=self.projection.bias, mask=mask)


class PositionalConvEmbedding(nn.Module):
    def __init__(self):
        super().__init__()
        self.conv = nn.Conv1d(768, 768, kernel_size=128, padding=128 // 2, groups=16)
        self.conv = nn.utils.weight_norm(self.conv, name="weight", dim=2)

    def _weight(self):
        """weight_norm over dim=2 (reference :121): w[:, :, k] = g[k] * v[:, :, k] / ||v[:, :, k]||_F, folded and packed for the
        MFMA kernel ([groups][48][128][64]: svc_posconv_pack_f32)."""
        return S.posconv_pack(self.conv.weight_v.detach(), self.conv.weight_g.detach(), groups=self.conv.groups)

    def forward(self, x, owner):
        """x + gelu(conv(x)[..., :-1]) (reference :125-129) in ONE launch (csrc/posconv.hip): round 2 ran the grouped k = 128
        conv on the generic one-thread-per-output kernel — 1.88 of the unit encoder's 5.5 ms per 10 s clip."""
        w = owner._packed("pos_conv", self._weight)
        return S.posconv(x, w, self.conv.bias, pad=self.conv.padding[0])


class TransformerEncoder(nn.Module):
    def __init__(self, encoder_layer: nn.TransformerEncoderLayer, num_layers: int) -> None:
        super().__init__()
        self.layers = nn.ModuleList([copy.deepcopy(encoder_layer) for _ in range(num_layers)])
        self.num_layers = num_layers

    def forward(self, src, owner, mask=None, src_key_padding_mask=None, output_layer: Optional[int] = None):
        if mask is not None:
            raise NotImplementedError("attention masks are not used by the so-vits-svc unit encoders")
        # src_key_padding_mask here: the engine's [B, 1, T] float mask of VALID frames (1 = frame of the item, 0 = padding) of a
        # batch of unequal items (Hubert.encode(lengths=...)) — the opposite polarity of torch's boolean argument of that name
        pm = src_key_padding_mask
        x = src
        for li, layer in enumerate(self.layers[:output_layer]):
            sa = layer.self_attn
            E, H = sa.embed_dim, sa.num_heads
            wqkv = owner._packed(f"l{li}.qkv", lambda sa=sa: _pack(sa.in_proj_weight.unsqueeze(-1)))
            wo = owner._packed(f"l{li}.o", lambda sa=sa: _pack(sa.out_proj.weight.unsqueeze(-1)))
            w1 = owner._packed(f"l{li}.w1", lambda layer=layer: _pack(layer.linear1.weight.unsqueeze(-1)))
            w2 = owner._packed(f"l{li}.w2", lambda layer=layer: _pack(layer.linear2.weight.unsqueeze(-1)))
            qkv = S.conv1d(x, wqkv, 3 * E, 1, bias=sa.in_proj_bias)
            att = S.attention(qkv[:, :E], qkv[:, E:2 * E], qkv[:, 2 * E:], H, mask=pm, mask_mode=1 if pm is not None else 0)
            y = S.conv1d(att, wo, E, 1, bias=sa.out_proj.bias)
            x = S.add_layernorm(x, y, layer.norm1.weight, layer.norm1.bias, eps=layer.norm1.eps)
            h = S.conv1d(x, w1, layer.linear1.out_features, 1, bias=layer.linear1.bias, post_act=S.ACT_GELU)
            y = S.conv1d(h, w2, E, 1, bias=layer.linear2.bias)
            x = S.add_layernorm(x, y, layer.norm2.weight, layer.norm2.bias, eps=layer.norm2.eps)
        return x


def hubert_soft(path: str) -> HubertSoft:
    """Reference :222-232: build HubertSoft and load a checkpoint (keys may carry a "module." prefix)."""
    from torch.nn.modules.utils import consume_prefix_in_state_dict_if_present
    hubert = HubertSoft()
    checkpoint = torch.load(path, map_location="cpu")
    consume_prefix_in_state_dict_if_present(checkpoint, "module.")
    hubert.load_state_dict(checkpoint)
    hubert.eval()
    return hubert


# ------------------------------------------------------------------------------------------------------------
# fairseq HubertModel checkpoints (vencoder/ContentVec768L12.py:12-15: checkpoint_utils.load_model_ensemble_and_task)
# ------------------------------------------------------------------------------------------------------------
_FAIRSEQ_DIRECT = {
    "feature_extractor.conv_layers.0.2.weight": "feature_extractor.norm0.weight",
    "feature_extractor.conv_layers.0.2.bias": "feature_extractor.norm0.bias",
    "layer_norm.weight": "feature_projection.norm.weight", "layer_norm.bias": "feature_projection.norm.bias",
    "post_extract_proj.weight": "feature_projection.projection.weight",
    "post_extract_proj.bias": "feature_projection.projection.bias",
    "encoder.pos_conv.0.bias": "positional_embedding.conv.bias",
    "encoder.pos_conv.0.weight_g": "positional_embedding.conv.weight_g",
    "encoder.pos_conv.0.weight_v": "positional_embedding.conv.weight_v",
    "encoder.layer_norm.weight": "norm.weight", "encoder.layer_norm.bias": "norm.bias",
    "final_proj.weight": "proj.weight", "final_proj.bias": "proj.bias", "mask_emb": "masked_spec_embed",
}
_FAIRSEQ_LAYER = {"self_attn.out_proj": "self_attn.out_proj", "self_attn_layer_norm": "norm1", "fc1": "linear1",
                  "fc2": "linear2", "final_layer_norm": "norm2"}


def hubert_from_fairseq_state_dict(fsd):
    """fairseq `HubertModel.state_dict()` (HuBERT-base: extractor_mode 'default', conv_bias False, layer_norm_first False)
    -> the key layout of `Hubert` above.  Tensors are re-used, q/k/v projections are concatenated into in_proj_*;
    `label_embs_concat` (training head) is dropped."""
    out = {}
    for k, v in fsd.items():
        if k in _FAIRSEQ_DIRECT:
            out[_FAIRSEQ_DIRECT[k]] = v
            continue
        parts = k.split(".")
        if k.startswith("feature_extractor.conv_layers.") and parts[3] == "0" and parts[4] == "weight":
            out[f"feature_extractor.conv{parts[2]}.weight"] = v
            continue
        if k.startswith("encoder.layers."):
            li, rest = parts[2], ".".join(parts[3:-1])
            leaf = parts[-1]
            if rest in ("self_attn.q_proj", "self_attn.k_proj", "self_attn.v_proj"):
                continue                                          # gathered below
            if rest in _FAIRSEQ_LAYER:
                out[f"encoder.layers.{li}.{_FAIRSEQ_LAYER[rest]}.{leaf}"] = v
                continue
        if k == "label_embs_concat":
            continue
        raise KeyError(f"unexpected key in a fairseq HuBERT checkpoint: {k}")
    n_layers = 1 + max(int(k.split(".")[2]) for k in fsd if k.startswith("encoder.layers."))
    for li in range(n_layers):
        for leaf in ("weight", "bias"):
            out[f"encoder.layers.{li}.self_attn.in_proj_{leaf}"] = torch.cat(
                [fsd[f"encoder.layers.{li}.self_attn.{p}_proj.{leaf}"] for p in "qkv"], 0)
    return out


def _load_fairseq_pickle(path):
    """torch.load of a fairseq checkpoint without fairseq installed: its `cfg` / `args` entries pickle fairseq / omegaconf
    classes, which are replaced by inert placeholders (only the `model` tensors are used)."""
    import pickle
    import types

    class _Stub:
        def __init__(self, *a, **k):
            pass

        def __setstate__(self, state):
            self.__dict__["state"] = state

    class _Unpickler(pickle.Unpickler):
        def find_class(self, module, name):
            try:
                return super().find_class(module, name)
            except (ImportError, AttributeError):
                return type(name, (_Stub,), {"__module__": module})

    shim = types.ModuleType("svc_fairseq_pickle")
    shim.Unpickler = _Unpickler
    shim.load = lambda f, **kw: _Unpickler(f, **kw).load()
    shim.__name__ = "pickle"
    for attr in ("PickleError", "UnpicklingError", "dump", "dumps", "loads", "HIGHEST_PROTOCOL"):
        setattr(shim, attr, getattr(pickle, attr))
    return torch.load(path, map_location="cpu", pickle_module=shim, weights_only=False)


def load_fairseq_hubert(path, num_label_embeddings=100):
    """`checkpoint_best_legacy_500.pt` (ContentVec, fairseq format) -> `Hubert` in eval mode."""
    ckpt = _load_fairseq_pickle(path)
    fsd = ckpt["model"] if isinstance(ckpt, dict) and "model" in ckpt else ckpt
    sd = hubert_from_fairseq_state_dict(fsd)
    model = Hubert(num_label_embeddings=num_label_embeddings)
    own = model.state_dict()
    if "proj.weight" in sd and tuple(sd["proj.weight"].shape) != tuple(own["proj.weight"].shape):
        model.proj = nn.Linear(sd["proj.weight"].shape[1], sd["proj.weight"].shape[0])
    missing, unexpected = model.load_state_dict(sd, strict=False)
    allowed = {"label_embedding.weight", "masked_spec_embed", "proj.weight", "proj.bias"}
    bad = [k for k in missing if k not in allowed]
    if bad or unexpected:
        raise RuntimeError(f"fairseq HuBERT checkpoint does not fit the HuBERT-base layout: missing {bad}, unexpected {unexpected}")
    return model.eval()
