from modules.commons import get_padding, init_weights  # noqa: F401  (same helpers, vdecoder/hifigan/utils.py)
