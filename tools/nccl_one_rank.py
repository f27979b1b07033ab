"""tools/nccl_one_rank.py -- exercise the RCCL code paths of whvi_amd.parallel and bench.py with a
one-rank "nccl" process group on the single-GPU box (two ranks cannot share one GPU under RCCL)."""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
os.environ.setdefault("MASTER_PORT", "29533")
import torch
import torch.distributed as dist
import torch.nn as nn
from whvi_amd import parallel
from whvi_amd.layers import WHVILinear
from whvi_amd.networks import WHVIRegression

dev = torch.device("cuda", 0)
torch.cuda.set_device(dev)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
dist.barrier()
t = torch.tensor([1.5], dtype=torch.float64, device=dev)
dist.all_reduce(t, op=dist.ReduceOp.MAX)
local = torch.randn(5, 2, 3, device=dev)
buf = torch.empty(5, 2, 3, device=dev)
dist.all_gather_into_tensor(buf.view(5, 2, 3), local)           # same call gather_predictions makes
assert torch.equal(buf, local)
net = WHVIRegression([WHVILinear(3, 16), nn.ReLU(), WHVILinear(16, 1)]).to(dev)
pred = parallel.mc_sharded_forward(net, torch.randn(7, 3, device=dev), n_samples=6, base_seed=1)
assert pred.shape == (7, 1, 6)
net.loss(torch.randn(7, 3, device=dev), torch.randn(7, 1, device=dev), 70).backward()
parallel.all_reduce_grads(net)
torch.cuda.synchronize()
# bench.py's N > 1 extras (fp16 row shards, MC-sharded network pass with one all-gather) at their real sizes
import json
import bench
print(json.dumps(bench.multi_gpu_extras(dev, 0, 1)), flush=True)
dist.destroy_process_group()
print("nccl one-rank paths ok", float(t))
