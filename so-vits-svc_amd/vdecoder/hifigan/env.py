class AttrDict(dict):
    """dict with attribute access (vdecoder/hifigan/env.py in the reference)."""

    def __init__(self, *args, **kwargs):
        super().__init__(*args, **kwargs)
        self.__dict__ = self
