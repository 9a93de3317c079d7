"""`oracle.weights` = the repo-level synthetic-data generator (synthetic_data.py): deterministic (config, seed) -> state_dict /
inputs / noise.  Kept under this name because every oracle module and test imports it as `W`.  TEST INFRASTRUCTURE."""
import os
import sys

_ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if _ROOT not in sys.path:
    sys.path.insert(0, _ROOT)

import synthetic_data as _sd  # noqa: E402
from synthetic_data import *  # noqa: E402,F401,F403

globals().update({k: v for k, v in vars(_sd).items() if not k.startswith("__")})
