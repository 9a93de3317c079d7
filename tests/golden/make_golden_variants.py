"""Golden vectors for the model VARIANTS from the REAL reference (build container only; see make_golden.py):
  snake_T40   small config with vocoder_name="nsf-snake-hifigan" (vdecoder/hifiganwithsnake, SnakeAlias activations)
  tiny_T40    small config with the tiny template's switches (configs_template/config_tiny_template.json:
              use_depthwise_conv, flow_share_parameter, odd decoder widths 100/50/25/12/6)

usage: python tests/golden/make_golden_variants.py
"""
import os
import sys

import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
from make_golden import import_reference, run_case  # noqa: E402


def main():
    torch.manual_seed(0)
    torch.set_num_threads(8)
    models, utils = import_reference()
    from oracle import weights as W
    snake = W.small_config()
    snake["vocoder_name"] = "nsf-snake-hifigan"
    run_case(models, "snake_T40", snake, B=2, T=40, seed=13)
    run_case(models, "tiny_T40", W.small_tiny_config(), B=2, T=40, seed=14)


if __name__ == "__main__":
    main()
