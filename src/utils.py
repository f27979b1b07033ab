from whvi_amd.utils import *  # noqa: F401,F403
from whvi_amd.utils import build_H_recursive  # noqa: F401
