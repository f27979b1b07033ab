from whvi_amd.fwht.python import *  # noqa: F401,F403
