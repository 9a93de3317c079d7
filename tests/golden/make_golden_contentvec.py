"""Golden vectors for the ContentVec speech encoders (`vec768l12`, `vec256l9`) from an INDEPENDENT third-party HuBERT-base:
`transformers.HubertModel` (transformers 5.15, installed in the build container).

Why third party: the reference computes ContentVec through fairseq 0.12.2 (`vencoder/ContentVec768L12.py:12-15,28-36`:
`model.extract_features(source, padding_mask=all-False, output_layer=12)`; `ContentVec256L9.py:28-37`: layer 9 + `final_proj`),
and neither fairseq nor `checkpoint_best_legacy_500.pt` exists here.  Hugging Face's HubertModel is a separate
implementation of the same network (hubert-base: GroupNorm feature extractor, post-LN encoder, weight-normed k=128 g=16
positional conv) whose mapping FROM fairseq parameter names is public — `FAIRSEQ_TO_HF` below restates the MAPPING table of
transformers' `convert_hubert_original_pytorch_checkpoint_to_pytorch.py`.  So a synthetic state dict under FAIRSEQ names
(`oracle.hubert_oracle.to_fairseq_state_dict`) is loaded (a) into HubertModel through that public table -> this golden, and
(b) into the engine through its own fairseq-free loader (`vencoder.hubert.hubert_model.hubert_from_fairseq_state_dict`) ->
tests/test_contentvec.py compares.  A wrong key mapping, a pre- instead of post-LN layer, a missing `layer_norm` in front of
`post_extract_proj`, or a wrong output-layer index all show up as O(1) differences.

Not covered (stated, not hidden): fairseq's own forward can still differ from HF's in ways both tests would miss only if HF
shared the bug; a padded batch with a real padding mask is NOT a call the reference makes (B = 1, `padding_mask` all False,
ContentVec768L12.py:30-31), so the batched case here is two equal-length items.

usage: python tests/golden/make_golden_contentvec.py
"""
import json
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)

# fairseq name -> transformers name (convert_hubert_original_pytorch_checkpoint_to_pytorch.py: MAPPING + load_conv_layer);
# "*" is the layer index
FAIRSEQ_TO_HF = {
    "post_extract_proj": "feature_projection.projection",
    "encoder.pos_conv.0": "encoder.pos_conv_embed.conv",
    "self_attn.k_proj": "encoder.layers.*.attention.k_proj",
    "self_attn.v_proj": "encoder.layers.*.attention.v_proj",
    "self_attn.q_proj": "encoder.layers.*.attention.q_proj",
    "self_attn.out_proj": "encoder.layers.*.attention.out_proj",
    "self_attn_layer_norm": "encoder.layers.*.layer_norm",
    "fc1": "encoder.layers.*.feed_forward.intermediate_dense",
    "fc2": "encoder.layers.*.feed_forward.output_dense",
    "final_layer_norm": "encoder.layers.*.final_layer_norm",
    "encoder.layer_norm": "encoder.layer_norm",
    "mask_emb": "masked_spec_embed",
}


def fairseq_to_hf(fs):
    """-> (HubertModel state dict, final_proj (weight, bias))."""
    hf, final_proj = {}, {}
    for name, v in fs.items():
        p = name.split(".")
        if name.startswith("feature_extractor.conv_layers."):                    # load_conv_layer: type 0 = conv, 2 = norm
            layer, typ = int(p[2]), int(p[3])
            hf[f"feature_extractor.conv_layers.{layer}.{'conv' if typ == 0 else 'layer_norm'}.{p[-1]}"] = v
        elif name.startswith("final_proj."):
            final_proj[p[-1]] = v
        elif name in ("label_embs_concat",):
            continue
        elif name.startswith("layer_norm."):                                       # the extractor-side LN ("layer_norm" in fairseq)
            hf["feature_projection.layer_norm." + p[-1]] = v
        else:
            for key, mapped in FAIRSEQ_TO_HF.items():
                if key in name and not (key == "encoder.layer_norm" and "layers" in name):
                    tgt = mapped.replace("*", p[2]) if "*" in mapped else mapped
                    leaf = name[name.index(key) + len(key):].lstrip(".")
                    # torch >= 2.1 weight_norm parametrization: original0 = g, original1 = v
                    leaf = {"weight_g": "parametrizations.weight.original0", "weight_v": "parametrizations.weight.original1"}.get(leaf, leaf)
                    hf[tgt + ("." + leaf if leaf else "")] = v
                    break
            else:
                raise KeyError(f"no public mapping for fairseq parameter {name}")
    return hf, final_proj


def hf_model():
    from transformers import HubertConfig, HubertModel
    cfg = HubertConfig(hidden_size=768, num_hidden_layers=12, num_attention_heads=12, intermediate_size=3072, hidden_act="gelu",
                       hidden_dropout=0.0, activation_dropout=0.0, attention_dropout=0.0, feat_proj_dropout=0.0,
                       feat_proj_layer_norm=True, final_dropout=0.0, layerdrop=0.0, feat_extract_norm="group",
                       feat_extract_activation="gelu", conv_dim=(512,) * 7, conv_stride=(5, 2, 2, 2, 2, 2, 2),
                       conv_kernel=(10, 3, 3, 3, 3, 2, 2), conv_bias=False, num_conv_pos_embeddings=128,
                       num_conv_pos_embedding_groups=16, do_stable_layer_norm=False, apply_spec_augment=False)
    return HubertModel(cfg).eval()


def main():
    torch.manual_seed(0)
    torch.set_num_threads(8)
    from oracle import hubert_oracle as HO
    seed = 91
    sd = HO.make_state_dict(seed)
    fs = HO.to_fairseq_state_dict(sd)
    hf, fproj = fairseq_to_hf(fs)
    net = hf_model()
    missing, unexpected = net.load_state_dict(hf, strict=False)
    assert not unexpected and not missing, (missing, unexpected)
    g = torch.Generator().manual_seed(seed)
    n = 16000
    t = torch.arange(n) / 16000.0
    wav = (0.3 * torch.sin(2 * torch.pi * 180 * t) + 0.1 * torch.randn(n, generator=g)).view(1, n)
    wav2 = torch.stack([wav[0, :9000], 0.2 * torch.randn(9000, generator=g)])             # B = 2, equal lengths
    out = {}
    with torch.no_grad():
        for tag, w in (("a", wav), ("b", wav2)):
            hs = net(w, output_hidden_states=True).hidden_states                          # hs[i] = after i transformer layers
            out[f"{tag}_l12"] = hs[12].numpy()
            out[f"{tag}_l9proj"] = torch.nn.functional.linear(hs[9], fproj["weight"], fproj["bias"]).numpy()
            o12 = HO.encode(sd, w.unsqueeze(1), layer=12)
            o9 = torch.nn.functional.linear(HO.encode(sd, w.unsqueeze(1), layer=9), sd["proj.weight"], sd["proj.bias"])
            for name, mine, ref in ((f"{tag}_l12", o12, out[f"{tag}_l12"]), (f"{tag}_l9proj", o9, out[f"{tag}_l9proj"])):
                d = float((mine - torch.from_numpy(ref)).abs().max())
                print(f"{name}: shape {ref.shape} max|ref| {np.abs(ref).max():.3f}  oracle vs transformers max|diff| {d:.3e}")
                assert d <= 5e-5 * max(1.0, float(np.abs(ref).max())), name
    np.savez_compressed(os.path.join(HERE, "contentvec_hf.npz"), wav_a=wav.numpy(), wav_b=wav2.numpy(),
                        meta=json.dumps(dict(seed=seed, transformers=__import__("transformers").__version__)), **out)
    print("wrote contentvec_hf.npz")


if __name__ == "__main__":
    main()
