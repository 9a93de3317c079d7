from .act import *       # noqa: F401,F403
from .filter import *    # noqa: F401,F403
from .resample import *  # noqa: F401,F403
