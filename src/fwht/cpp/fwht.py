from whvi_amd.fwht.cpp import *  # noqa: F401,F403
