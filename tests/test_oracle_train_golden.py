"""CPU suite: pins oracle/train_oracle.py (training graph, losses, gradients) and oracle/mel.py to the vectors the REAL
reference produced (tests/golden/train_small.npz)."""
import numpy as np
import torch

from oracle import mel as OM
from oracle import train_oracle as TO
from oracle import weights as W
from train_common import LOSS_KEYS, load_case, load_dropout_case, load_transflow_case


def test_mpd_layout_matches_reference():
    sh = W.mpd_param_shapes()
    assert len(sh) == 111                                   # 111 D parameter tensors
    assert sum(int(np.prod(s)) for s in sh.values()) == 46747132


def test_mel_basis_matches_independent_implementation():
    from transformers.audio_utils import mel_filter_bank
    a = OM.mel_filterbank(44100, 2048, 80, 0, 22050)
    b = mel_filter_bank(1025, 80, 0, 22050, 44100, norm="slaney", mel_scale="slaney").T
    assert a.shape == (80, 1025) and np.abs(a - b).max() < 1e-7


def test_mel_scale_matches_librosa_published_known_answers():
    """librosa==0.9.1 (requirements.txt:23; call site modules/mel_processing.py:72) cannot be installed here — the container has
    no network (`pip download librosa==0.9.1`: "no matching distribution") and no wheel is in the offline wheelhouse — so the
    basis cannot be compared with librosa's OUTPUT.  What librosa does publish are known answers in its own docstrings
    (librosa/core/convert.py: `hz_to_mel`, `mel_to_hz`, `mel_frequencies`; librosa/filters.py: `mel`), unchanged between 0.8 and
    0.10; the restatement reproduces every one of them to the printed precision:
      hz_to_mel(60) = 0.9; hz_to_mel([110, 220, 440]) = [1.65, 3.3, 6.6]; mel_to_hz(3) = 200.;
      mel_to_hz([1..5]) = [66.667, 133.333, 200., 266.667, 333.333]; mel_frequencies(n_mels=40) (fmin 0, fmax 11025): the 40
      values below; filters.mel(sr=22050, n_fft=2048)[0, :2] = [0., 0.016].
    Together with the element-wise agreement with transformers' independent implementation (test above) this is as far as
    the basis can be pinned without librosa itself."""
    assert abs(float(OM.hz_to_mel(60)) - 0.9) < 1e-12
    assert np.allclose(OM.hz_to_mel([110, 220, 440]), [1.65, 3.3, 6.6], atol=1e-12)
    assert float(OM.mel_to_hz(3)) == 200.0
    assert np.array_equal(np.round(OM.mel_to_hz([1, 2, 3, 4, 5]), 3), [66.667, 133.333, 200.0, 266.667, 333.333])
    doc = [0., 85.317, 170.635, 255.952, 341.269, 426.586, 511.904, 597.221, 682.538, 767.855, 853.173, 938.49, 1024.856,
           1119.114, 1222.042, 1334.436, 1457.167, 1591.187, 1737.532, 1897.337, 2071.84, 2262.393, 2470.47, 2697.686, 2945.799,
           3216.731, 3512.582, 3835.643, 4188.417, 4573.636, 4994.285, 5453.621, 5955.205, 6502.92, 7101.009, 7754.107, 8467.272,
           9246.028, 10096.408, 11025.]
    mine = OM.mel_to_hz(np.linspace(OM.hz_to_mel(0.0), OM.hz_to_mel(11025.0), 40))
    assert np.array_equal(np.round(mine, 3), np.array(doc))
    w = OM.mel_filterbank(22050, 2048, 128)
    assert w.shape == (128, 1025) and abs(w[0, 0]) == 0.0 and round(float(w[0, 1]), 3) == 0.016
    # Slaney area normalisation: every filter integrates to ~1 over frequency (2 / (f[i+2] - f[i]) * triangle area)
    b = OM.mel_filterbank(44100, 2048, 80, 0, 22050).astype(np.float64)
    assert np.abs(b.sum(1) * (22050 / 1024) - 1.0).max() < 0.06


def test_oracle_reproduces_reference_training_step():
    cs = load_case()
    z = cs["z"]
    sg = {k: v.clone().requires_grad_(True) for k, v in cs["sd_g"].items()}
    sd = {k: v.clone().requires_grad_(True) for k, v in cs["sd_d"].items()}
    out = TO.gan_step_losses(sg, sd, cs["cfg"], cs["data"], cs["batch"], cs["noise"], cs["mel_basis"])
    for k in LOSS_KEYS:
        ref = float(z["loss." + k])
        assert abs(float(out[k]) - ref) <= 2e-5 * max(1.0, abs(ref)), (k, float(out[k]), ref)
    assert np.abs(out["y_hat"].detach().numpy() - z["y_hat"]).max() <= 1e-5 * max(1.0, np.abs(z["y_hat"]).max())
    # gradient norms of every parameter + a few full gradients
    dk = [str(k) for k in z["gnorm_d_keys"]]
    gd = torch.autograd.grad(out["loss_disc"], [sd[k] for k in dk], retain_graph=True)
    for k, g, n in zip(dk, gd, z["gnorm_d"]):
        assert abs(g.norm().item() - n) <= 1e-4 * max(n, 1e-6), ("D", k)
    gk = [str(k) for k in z["gnorm_g_keys"]]
    gg = torch.autograd.grad(out["loss_gen_all"], [sg[k] for k in gk], allow_unused=True)
    for k, g, n in zip(gk, gg, z["gnorm_g"]):
        if g is None:
            assert n == 0, k
        elif not k.endswith("conv_k.bias"):      # exactly-zero gradient (softmax shift invariance): round-off only
            assert abs(g.norm().item() - n) <= 2e-4 * max(n, 1e-5), ("G", k, g.norm().item(), n)
    for name in z.files:
        if name.startswith("grad_g."):
            g = gg[gk.index(name[7:])]
            assert np.abs(g.numpy() - z[name]).max() <= 2e-4 * max(np.abs(z[name]).max(), 1e-6), name


def test_oracle_loop_reproduces_reference_training_loop():
    """3 iterations incl. AdamW in the reference's order (tests/golden/train_loop_small.npz, from the REAL reference)."""
    import json
    import os
    from train_common import G
    cs = load_case()
    z = np.load(os.path.join(G, "train_loop_small.npz"), allow_pickle=False)
    meta = json.loads(str(z["meta"]))
    hist, sg, sd = TO.gan_train_loop(cs["sd_g"], cs["sd_d"], cs["cfg"], cs["data"], cs["batch"], cs["noise"],
                                     cs["mel_basis"], meta["n_iter"], lr=meta["lr"], betas=tuple(meta["betas"]),
                                     eps=meta["eps"])
    for it in range(meta["n_iter"]):
        for k in LOSS_KEYS:
            ref = float(z[f"it{it}.{k}"])
            assert abs(hist[it][k] - ref) <= 1e-4 * max(1.0, abs(ref)), (it, k, hist[it][k], ref)
    for name in z.files:
        if name.startswith("param_g."):
            assert np.abs(sg[name[8:]].numpy() - z[name]).max() <= 2e-5, name
        if name.startswith("param_d."):
            assert np.abs(sd[name[8:]].numpy() - z[name]).max() <= 2e-5, name


def test_oracle_reproduces_reference_training_forward_with_dropout():
    """p_dropout = 0.1 (the shipped configs' value): attention-probability / attention-output / FFN dropouts with injected
    draws, prior statistics + pred_lf0 + loss_kl + loss_lf0 + their gradients vs the REAL reference."""
    cs = load_dropout_case()
    z = cs["z"]
    sg = {k: v.clone().requires_grad_(True) for k, v in cs["sd_g"].items()}
    c, f0, uv, spec, y, sid, lengths = cs["batch"]
    o = TO.synth_forward(sg, cs["cfg"], c, f0, uv, spec, sid, lengths, lengths, cs["noise"])
    m_p, logs_p, pred = o[3][2], o[3][3], o[4]
    assert np.abs(m_p.detach().numpy() - z["m_p"]).max() <= 2e-5 * max(1.0, np.abs(z["m_p"]).max())
    assert np.abs(pred.detach().numpy() - z["pred_lf0"]).max() <= 2e-5 * max(1.0, np.abs(z["pred_lf0"]).max())
    kl = TO.kl_loss(o[3][1], o[3][5], m_p, logs_p, o[2])
    lf0 = torch.nn.functional.mse_loss(pred, o[6])
    assert abs(float(kl) - float(z["loss_kl"])) <= 2e-5 * max(1.0, abs(float(z["loss_kl"])))
    assert abs(float(lf0) - float(z["loss_lf0"])) <= 2e-5 * max(1.0, abs(float(z["loss_lf0"])))
    keys = [str(k) for k in z["gnorm_keys"]]
    gs = torch.autograd.grad(kl + lf0, [sg[k] for k in keys], allow_unused=True)
    for k, g, n in zip(keys, gs, z["gnorm"]):
        if g is None:
            assert n == 0, k
        elif not k.endswith("conv_k.bias"):
            assert abs(g.norm().item() - n) <= 2e-4 * max(n, 1e-5), (k, g.norm().item(), n)
    # and dropout is really active in this case: the eval-mode statistics differ
    o0 = TO.synth_forward(cs["sd_g"], dict(cs["cfg"], p_dropout=0.0), c, f0, uv, spec, sid, lengths, lengths, cs["noise"])
    assert (o0[3][2] - m_p).abs().max().item() > 1e-3


def test_oracle_reproduces_reference_transformer_flow_training():
    """use_transformer_flow (models.py:438-439) in train() mode with p_dropout = 0.1: z_p = flow(z), the losses that depend
    on it and their gradients (incl. the flow's cond_pre / cond_layer / FFT parameters) vs the REAL reference."""
    cs = load_transflow_case()
    z = cs["z"]
    sg = {k: v.clone().requires_grad_(True) for k, v in cs["sd_g"].items()}
    c, f0, uv, spec, y, sid, lengths = cs["batch"]
    o = TO.synth_forward(sg, cs["cfg"], c, f0, uv, spec, sid, lengths, lengths, cs["noise"])
    z_p, m_p, pred = o[3][1], o[3][2], o[4]
    assert np.abs(z_p.detach().numpy() - z["z_p"]).max() <= 2e-5 * max(1.0, np.abs(z["z_p"]).max())
    assert np.abs(m_p.detach().numpy() - z["m_p"]).max() <= 2e-5 * max(1.0, np.abs(z["m_p"]).max())
    kl = TO.kl_loss(z_p, o[3][5], m_p, o[3][3], o[2])
    lf0 = torch.nn.functional.mse_loss(pred, o[6])
    assert abs(float(kl) - float(z["loss_kl"])) <= 2e-5 * max(1.0, abs(float(z["loss_kl"])))
    assert abs(float(lf0) - float(z["loss_lf0"])) <= 2e-5 * max(1.0, abs(float(z["loss_lf0"])))
    keys = [str(k) for k in z["gnorm_keys"]]
    assert any(k.startswith("flow.flows.0.enc.cond_pre") for k in keys)
    gs = dict(zip(keys, torch.autograd.grad(kl + lf0, [sg[k] for k in keys], allow_unused=True)))
    for k, n in zip(keys, z["gnorm"]):
        if gs[k] is None:
            assert n == 0, k
        elif not k.endswith("conv_k.bias"):
            assert abs(gs[k].norm().item() - n) <= 2e-4 * max(n, 1e-5), (k, gs[k].norm().item(), n)
    for name in z.files:
        if name.startswith("grad."):
            g = gs[name[5:]].numpy()
            assert np.abs(g - z[name]).max() <= 1e-4 * max(np.abs(z[name]).max(), 1e-6), name
