"""CPU oracle of the HuBERT-soft unit encoder.  TEST INFRASTRUCTURE ONLY (see svc_oracle.py).

torch-CPU fp32 restatement of vencoder/hubert/hubert_model.py (`HubertSoft.units`, :63-68) on a plain state_dict:
FeatureExtractor :71-94, FeatureProjection :97-108, PositionalConvEmbedding :111-127 (weight_norm over dim=2),
12 x nn.TransformerEncoderLayer(768, 12, 3072, gelu, batch_first, post-norm) :20-25,130-152, proj :26.
Pinned by tests/golden/hubert_soft_1s.npz, generated from the REAL in-tree module (tests/golden/make_golden_hubert.py).
"""
import math
import zlib

import torch
import torch.nn.functional as F


def param_shapes():
    """state_dict layout of hubert_model.HubertSoft() (166 tensors, 94,594,176 values)."""
    P = {"masked_spec_embed": (768,), "feature_extractor.conv0.weight": (512, 1, 10),
         "feature_extractor.norm0.weight": (512,), "feature_extractor.norm0.bias": (512,)}
    for i, k in zip(range(1, 7), (3, 3, 3, 3, 2, 2)):
        P[f"feature_extractor.conv{i}.weight"] = (512, 512, k)
    P.update({"feature_projection.norm.weight": (512,), "feature_projection.norm.bias": (512,),
              "feature_projection.projection.weight": (768, 512), "feature_projection.projection.bias": (768,),
              "positional_embedding.conv.bias": (768,), "positional_embedding.conv.weight_g": (1, 1, 128),
              "positional_embedding.conv.weight_v": (768, 48, 128), "norm.weight": (768,), "norm.bias": (768,)})
    for l in range(12):
        p = f"encoder.layers.{l}"
        P.update({p + ".self_attn.in_proj_weight": (2304, 768), p + ".self_attn.in_proj_bias": (2304,),
                  p + ".self_attn.out_proj.weight": (768, 768), p + ".self_attn.out_proj.bias": (768,),
                  p + ".linear1.weight": (3072, 768), p + ".linear1.bias": (3072,),
                  p + ".linear2.weight": (768, 3072), p + ".linear2.bias": (768,),
                  p + ".norm1.weight": (768,), p + ".norm1.bias": (768,), p + ".norm2.weight": (768,), p + ".norm2.bias": (768,)})
    P.update({"proj.weight": (256, 768), "proj.bias": (256,), "label_embedding.weight": (100, 256)})
    return P


def make_state_dict(seed=1234):
    """Deterministic synthetic checkpoint: every tensor from its own generator (crc32(name) ^ seed), fan-in scaled."""
    sd = {}
    for name, shape in param_shapes().items():
        g = torch.Generator()
        g.manual_seed((zlib.crc32(name.encode()) ^ (seed * 0x9E3779B1)) & 0x7FFFFFFF)
        r = torch.randn(*shape, generator=g)
        leaf = name.rsplit(".", 1)[-1]
        if leaf == "bias" or name.endswith("in_proj_bias"):
            t = 0.05 * r
        elif leaf == "weight" and len(shape) == 1:                       # norm gains
            t = 1.0 + 0.1 * r
        elif leaf == "weight_g":
            t = 0.05 * (1.0 + 0.1 * r).abs() * math.sqrt(768 * 48) / math.sqrt(48 * 128)
        elif leaf == "weight_v":
            t = 0.1 * r
        elif len(shape) >= 2:
            fan_in = 1
            for s_ in shape[1:]:
                fan_in *= s_
            t = r * (1.6 if "feature_extractor" in name else 1.0) / math.sqrt(fan_in)
        else:
            t = r
        sd[name] = t
    return sd


def units(sd, wav):
    """HubertSoft.units (:63-68): wav [B, 1, n] (16 kHz) -> [B, T, 256]: zero-pad 40 samples each side, encode, proj."""
    return F.linear(encode(sd, wav, pad=40), sd["proj.weight"], sd["proj.bias"])


def encode(sd, wav, layer=None, pad=0):
    """Hubert.encode (:37-51) in eval mode: wav [B, 1, n] -> [B, T, 768] after `layer` transformer layers (None = all 12).
    With pad=0 this is also fairseq HubertModel.extract_features(source, padding_mask=all-False, output_layer=layer)[0],
    the call of vencoder/ContentVec768L12.py:28-36 (layer 12) and ContentVec256L9.py:28-37 (layer 9, then final_proj = `proj`):
    the in-tree module is a re-keyed copy of that network."""
    x = F.pad(wav, (pad, pad))
    x = F.conv1d(x, sd["feature_extractor.conv0.weight"], None, 5)
    x = F.gelu(F.group_norm(x, 512, sd["feature_extractor.norm0.weight"], sd["feature_extractor.norm0.bias"]))
    for i in range(1, 7):
        x = F.gelu(F.conv1d(x, sd[f"feature_extractor.conv{i}.weight"], None, 2))
    x = x.transpose(1, 2)
    x = F.layer_norm(x, (512,), sd["feature_projection.norm.weight"], sd["feature_projection.norm.bias"])
    x = F.linear(x, sd["feature_projection.projection.weight"], sd["feature_projection.projection.bias"])
    v, g = sd["positional_embedding.conv.weight_v"], sd["positional_embedding.conv.weight_g"]
    w = v * (g / v.pow(2).sum((0, 1), keepdim=True).sqrt())              # weight_norm(dim=2)
    p = F.conv1d(x.transpose(1, 2), w, sd["positional_embedding.conv.bias"], padding=64, groups=16)
    x = x + F.gelu(p[:, :, :-1]).transpose(1, 2)
    x = F.layer_norm(x, (768,), sd["norm.weight"], sd["norm.bias"])
    B, T, E = x.shape
    H, dk = 12, 64
    for l in range(12 if layer is None else layer):
        p_ = f"encoder.layers.{l}"
        qkv = F.linear(x, sd[p_ + ".self_attn.in_proj_weight"], sd[p_ + ".self_attn.in_proj_bias"])
        q, k, v_ = [t.view(B, T, H, dk).transpose(1, 2) for t in qkv.split(E, dim=-1)]
        a = torch.softmax((q * dk ** -0.5) @ k.transpose(-1, -2), dim=-1) @ v_
        a = a.transpose(1, 2).reshape(B, T, E)
        a = F.linear(a, sd[p_ + ".self_attn.out_proj.weight"], sd[p_ + ".self_attn.out_proj.bias"])
        x = F.layer_norm(x + a, (E,), sd[p_ + ".norm1.weight"], sd[p_ + ".norm1.bias"])
        f = F.linear(F.gelu(F.linear(x, sd[p_ + ".linear1.weight"], sd[p_ + ".linear1.bias"])),
                     sd[p_ + ".linear2.weight"], sd[p_ + ".linear2.bias"])
        x = F.layer_norm(x + f, (E,), sd[p_ + ".norm2.weight"], sd[p_ + ".norm2.bias"])
    return x


def to_fairseq_state_dict(sd):
    """The same tensors under fairseq HubertModel's parameter names (what `checkpoint_best_legacy_500.pt` holds): test
    fixture for the engine's fairseq-free checkpoint loader."""
    fs = {}
    for k, v in sd.items():
        p = k.split(".")
        if k.startswith("feature_extractor.conv"):
            fs[f"feature_extractor.conv_layers.{p[1][4:]}.0.weight"] = v
        elif k.startswith("feature_extractor.norm0"):
            fs[f"feature_extractor.conv_layers.0.2.{p[-1]}"] = v
        elif k.startswith("feature_projection.norm"):
            fs["layer_norm." + p[-1]] = v
        elif k.startswith("feature_projection.projection"):
            fs["post_extract_proj." + p[-1]] = v
        elif k.startswith("positional_embedding.conv"):
            fs["encoder.pos_conv.0." + p[-1]] = v
        elif k.startswith("norm."):
            fs["encoder.layer_norm." + p[-1]] = v
        elif k.startswith("proj."):
            fs["final_proj." + p[-1]] = v
        elif k == "masked_spec_embed":
            fs["mask_emb"] = v
        elif k == "label_embedding.weight":
            fs["label_embs_concat"] = torch.zeros(504, 256)
        elif "in_proj_" in k:
            for n, t in zip("qkv", v.chunk(3, 0)):
                fs[f"encoder.layers.{p[2]}.self_attn.{n}_proj.{p[-1].split('_')[-1]}"] = t.clone()
        else:
            m = {"self_attn.out_proj": "self_attn.out_proj", "norm1": "self_attn_layer_norm", "linear1": "fc1",
                 "linear2": "fc2", "norm2": "final_layer_norm"}[".".join(p[3:-1])]
            fs[f"encoder.layers.{p[2]}.{m}.{p[-1]}"] = v
    return fs
