from whvi_amd.layers import *  # noqa: F401,F403
