from .detectors import *  # noqa: F401,F403
from .necks import *  # noqa: F401,F403
from .dense_heads import *  # noqa: F401,F403
