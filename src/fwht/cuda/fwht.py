from whvi_amd.fwht.cuda import *  # noqa: F401,F403
