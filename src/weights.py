from whvi_amd.weights import *  # noqa: F401,F403
