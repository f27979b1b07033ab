"""f32 vs i32 at D = 512 (4 GiB in place), a few launches each, for rocprofv3 --pmc (why is the i32 instance 2 % faster?)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from whvi_amd import _hip
for dt in (torch.float32, torch.int32):
    x = torch.zeros((1 << 30) // 512, 512, dtype=dt, device="cuda")
    for _ in range(12):
        _hip.fwht_rows(x, out=x)
    torch.cuda.synchronize()
    del x
