from whvi_amd.activations import *  # noqa: F401,F403
