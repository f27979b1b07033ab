from whvi_amd.likelihoods import *  # noqa: F401,F403
