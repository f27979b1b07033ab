"""tools/profile_f16.py [launches] -- BASELINE config 5 on one GPU: D = 4096 fp16 (and bf16), 2^20 rows = 8 GiB in place, FINITE
data (an exact 2^-6 rescale after every transform keeps the magnitude constant), for rocprofv3 passes."""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from whvi_amd import _hip

n = int(sys.argv[1]) if len(sys.argv) > 1 else 16
dev = torch.device("cuda", 0)
for dtype in (torch.float16, torch.bfloat16):
    x = torch.empty(1 << 20, 4096, device=dev, dtype=dtype)
    x.view(256, 4096, 4096).copy_((torch.randn(4096, 4096, device=dev) * 2.0 ** -8).to(dtype))
    for _ in range(n):
        _hip.fwht_rows(x, out=x)
        x.mul_(2.0 ** -6)
    print(dtype, _hip.last_kernel(), "finite:", bool(torch.isfinite(x[::4099].float()).all()), flush=True)
    del x
torch.cuda.synchronize()
