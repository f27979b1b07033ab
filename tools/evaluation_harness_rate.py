"""tools/evaluation_harness_rate.py -- steps per second of the reference's UCI regression protocol
(``evaluate_bayesian_regression_dnn``, src/evaluation.py:30-108) on a synthetic data set of the yacht data set's shape
(308 rows x 6 features, batches of 64 -> 5 steps per epoch, the last one 21 rows), reference flow (DataLoader, host
schedule, eager steps) against the fast path (packed, device-resident Adam + schedule, one hipGraph replay per step,
DeviceBatches).  Prints one JSON line; the full protocol is 50 500 epochs x 8 splits = 2.02 M steps."""
import json
import os
import sys
import tempfile
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from whvi_amd.evaluation import evaluate_bayesian_regression_dnn

rng = np.random.default_rng(0)
X = rng.normal(size=(308, 6)).astype(np.float32)
y = (X[:, :1] * X[:, 1:2] + 0.1 * rng.normal(size=(308, 1))).astype(np.float32)
out = {"rows": 308, "features": 6, "steps_per_epoch": 5, "protocol_steps": 50500 * 5 * 8}
for fast, epochs in ((False, 40), (True, 2000)):
    with tempfile.TemporaryDirectory() as where:
        np.random.seed(0)
        torch.manual_seed(0)
        evaluate_bayesian_regression_dnn(X, y, "cuda", where, epochs1=2, epochs2=2, n_splits=1, fast=fast)   # warm
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        result = evaluate_bayesian_regression_dnn(X, y, "cuda", where, epochs1=epochs // 10, epochs2=epochs - epochs // 10,
                                                  n_splits=1, fast=fast)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
    key = "fast_path" if fast else "reference_flow"
    out[key] = {"epochs_timed": epochs, "seconds": round(dt, 3), "steps_per_s": round(epochs * 5 / dt, 1),
                "ms_per_step": round(dt / (epochs * 5) * 1e3, 4), "full_protocol_hours": round(out["protocol_steps"] / (epochs * 5 / dt) / 3600, 2),
                "test_error": result[0], "test_mnll": result[2]}
out["speedup"] = round(out["fast_path"]["steps_per_s"] / out["reference_flow"]["steps_per_s"], 1)
print(json.dumps(out))
