from vdecoder.hifigan.env import AttrDict  # noqa: F401
