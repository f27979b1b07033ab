from whvi_amd.evaluation import *  # noqa: F401,F403
