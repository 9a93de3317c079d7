"""ContentVec (`vec768l12`, `vec256l9`) pinned against an independent third-party HuBERT-base.

The reference computes these through fairseq (`vencoder/ContentVec768L12.py:12-15,28-36`, `ContentVec256L9.py:28-37`); fairseq
is absent, so `tests/golden/contentvec_hf.npz` holds the outputs of `transformers.HubertModel` on a synthetic checkpoint that
was written under FAIRSEQ parameter names and loaded through the public fairseq -> transformers key table
(`tests/golden/make_golden_contentvec.py`).  The engine reads the SAME fairseq-named checkpoint through its own loader.
CPU: the oracle (oracle/hubert_oracle.encode — the restatement the HuBERT-soft golden already pins to the in-tree module) equals
the transformers vectors; the engine's loader maps every fairseq key.  GPU: ContentVec768L12 / ContentVec256L9 `.encoder()`,
loaded from a fairseq-format checkpoint FILE, equal the vectors; B = 2 equal-length items through `Hubert.encode`.
Tolerance 2e-4 of max|ref| (as tests/test_hubert.py: 12 post-norm layers of fp32 MFMA reductions)."""
import json
import os
import sys

import numpy as np
import pytest
import torch

from oracle import hubert_oracle as HO

G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def _golden():
    z = np.load(os.path.join(G, "contentvec_hf.npz"))
    return z, json.loads(str(z["meta"]))


def test_oracle_equals_transformers_hubert_base():
    z, meta = _golden()
    sd = HO.make_state_dict(meta["seed"])
    with torch.no_grad():
        for tag in ("a", "b"):
            w = torch.from_numpy(z["wav_" + tag]).unsqueeze(1)
            l12 = HO.encode(sd, w, layer=12)
            l9 = torch.nn.functional.linear(HO.encode(sd, w, layer=9), sd["proj.weight"], sd["proj.bias"])
            for got, name in ((l12, f"{tag}_l12"), (l9, f"{tag}_l9proj")):
                ref = z[name]
                assert got.shape == ref.shape
                assert np.abs(got.numpy() - ref).max() <= 5e-5 * max(1.0, np.abs(ref).max()), name
    # the layers are distinguishable at this tolerance: a wrong output-layer index cannot pass
    assert np.abs(HO.encode(sd, torch.from_numpy(z["wav_a"]).unsqueeze(1), layer=11).numpy() - z["a_l12"]).max() > 0.1


def test_engine_loader_and_public_key_table_agree_on_every_fairseq_key():
    """Both routes out of the fairseq-named state dict — the engine's loader and the public fairseq -> transformers table
    the golden was made through — must place every tensor; cross-checked name by name through the oracle's own layout."""
    sys.path.insert(0, G)
    import make_golden_contentvec as MG
    from vencoder.hubert import hubert_model as HM
    sd = HO.make_state_dict(3)
    fs = HO.to_fairseq_state_dict(sd)
    mine = HM.hubert_from_fairseq_state_dict(fs)
    for k, v in sd.items():
        if k != "label_embedding.weight":
            assert torch.equal(mine[k], v), k
    hf, fproj = MG.fairseq_to_hf(fs)
    assert torch.equal(fproj["weight"], sd["proj.weight"]) and len(hf) == 211          # HubertModel(hubert-base).state_dict()
    l = 7
    E = 768
    assert torch.equal(hf[f"encoder.layers.{l}.attention.k_proj.weight"], sd[f"encoder.layers.{l}.self_attn.in_proj_weight"][E:2 * E])
    assert torch.equal(hf[f"encoder.layers.{l}.layer_norm.weight"], sd[f"encoder.layers.{l}.norm1.weight"])
    assert torch.equal(hf[f"encoder.layers.{l}.final_layer_norm.bias"], sd[f"encoder.layers.{l}.norm2.bias"])
    assert torch.equal(hf["feature_projection.layer_norm.weight"], sd["feature_projection.norm.weight"])
    assert torch.equal(hf["encoder.pos_conv_embed.conv.parametrizations.weight.original1"], sd["positional_embedding.conv.weight_v"])


def _fairseq_checkpoint(tmp_path, seed):
    path = str(tmp_path / "checkpoint_best_legacy_500.pt")
    torch.save({"model": HO.to_fairseq_state_dict(HO.make_state_dict(seed)), "cfg": None, "args": None}, path)
    return path


@pytest.mark.gpu
def test_contentvec_encoders_match_transformers_golden(dev, tmp_path):
    from vencoder.ContentVec256L9 import ContentVec256L9
    from vencoder.ContentVec768L12 import ContentVec768L12
    z, meta = _golden()
    path = _fairseq_checkpoint(tmp_path, meta["seed"])
    wav = torch.from_numpy(z["wav_a"][0]).to(dev)
    for cls, name, dim in ((ContentVec768L12, "a_l12", 768), (ContentVec256L9, "a_l9proj", 256)):
        enc = cls(vec_path=path, device=dev)
        c = enc.encoder(wav)                                                  # [1, dim, T] as Svc consumes it
        ref = torch.from_numpy(z[name]).transpose(1, 2)
        assert c.shape == ref.shape == (1, dim, 49) and enc.hidden_dim == dim
        err = (c.cpu() - ref).abs().max().item()
        assert err <= 2e-4 * max(1.0, ref.abs().max().item()), (name, err)
    # stereo input is averaged (ContentVec768L12.py:25-26)
    c2 = enc.encoder(torch.stack([wav, wav], dim=-1))
    assert torch.equal(c2, c)


@pytest.mark.gpu
def test_contentvec_batched_equal_lengths_match_transformers_golden(dev, tmp_path):
    from vencoder.hubert import hubert_model as HM
    z, meta = _golden()
    net = HM.load_fairseq_hubert(_fairseq_checkpoint(tmp_path, meta["seed"])).to(dev)
    w = torch.from_numpy(z["wav_b"]).unsqueeze(1).to(dev)
    x12, _ = net.encode(w, layer=12)
    x9, _ = net.encode(w, layer=9)
    for got, name in ((x12, "b_l12"), (net.project(x9), "b_l9proj")):
        ref = torch.from_numpy(z[name]).transpose(1, 2)
        assert got.shape == ref.shape
        err = (got.cpu() - ref).abs().max().item()
        assert err <= 2e-4 * max(1.0, ref.abs().max().item()), (name, err)
