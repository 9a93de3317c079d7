"""Model-variant configurations behind the committed inference goldens (tests/golden/make_golden*.py)."""
from oracle import weights as W


def variant_config(name):
    if name == "full":
        return W.full_config()
    if name == "tiny":
        return W.small_tiny_config()
    if name == "tinyfull":
        return W.tiny_config()                          # config_tiny_template.json:42-71 at its real widths
    cfg = W.small_config()
    if name == "snake":
        cfg["vocoder_name"] = "nsf-snake-hifigan"      # vdecoder/hifiganwithsnake (SnakeAlias activations)
    elif name == "transflow":
        cfg["use_transformer_flow"] = True              # models.py:438-439 TransformerCouplingBlock
    elif name == "transflow_shared":
        cfg.update(use_transformer_flow=True, flow_share_parameter=True, n_flow_layer=3, n_layers_trans_flow=2)
    elif name != "small":
        raise KeyError(name)
    return cfg


INFER_GOLDENS = [("infer_small_T40.npz", "small"), ("infer_small_T40_predf0.npz", "small"), ("infer_full_T24.npz", "full"),
                 ("infer_snake_T40.npz", "snake"), ("infer_tiny_T40.npz", "tiny"),
                 ("infer_tinyfull_T24.npz", "tinyfull"), ("infer_transflow_T40.npz", "transflow"),
                 ("infer_transflow_shared_T40.npz", "transflow_shared")]
