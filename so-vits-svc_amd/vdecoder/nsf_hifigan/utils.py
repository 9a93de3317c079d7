from vdecoder.hifigan.utils import get_padding, init_weights  # noqa: F401
