from whvi_amd.networks import *  # noqa: F401,F403
