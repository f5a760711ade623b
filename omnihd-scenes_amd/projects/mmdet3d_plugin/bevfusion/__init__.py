from .detectors import *  # noqa: F401,F403
from .necks import *  # noqa: F401,F403
