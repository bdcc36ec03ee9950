#!/usr/bin/env python3
"""Is a configuration host-bound?  Time to ENQUEUE one training step (no synchronisation) against the time the GPU needs for it."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import CONFIGS, Workload  # noqa: E402

# "c2@64x96": the configuration's network at a small size - the same launch list on a GPU that finishes each launch at once, so
# the enqueue time is the host's own cost per step (at the full size the host also waits whenever the hardware queue is full)
for name in sys.argv[1:] or ["c1", "c2"]:
    cfg_name, _, size = name.partition("@")
    cfg = dict(CONFIGS[cfg_name])
    if size:
        cfg["H"], cfg["W"] = (int(v) for v in size.split("x"))
        cfg["batch"] = 1
    w = Workload(cfg, torch.device("cuda", 0), 0, False, False)
    for _ in range(5):
        w.step()
    torch.cuda.synchronize()
    n = 20
    t0 = time.perf_counter()
    enq = 0.0
    for _ in range(n):
        e0 = time.perf_counter()
        w.step()
        enq += time.perf_counter() - e0
    torch.cuda.synchronize()
    tot = time.perf_counter() - t0
    print(f"{name}: enqueue {enq / n * 1e3:.2f} ms/step, wall {tot / n * 1e3:.2f} ms/step", flush=True)
    del w
    torch.cuda.empty_cache()
