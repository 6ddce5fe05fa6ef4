from .pytorch import *  # noqa: F401,F403
from . import pytorch  # noqa: F401
