#!/usr/bin/env python3
"""Run-to-run reproducibility of the training forward at a BASELINE configuration: logits of repeated forwards on identical
inputs and weights in ONE fresh process (the process's first forward is the interesting one).
    python tools/determinism.py [config] [off-families joined by +, or -] [n forwards]"""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from bench import CONFIGS, make_config, synthetic_batch  # noqa: E402
from dmmfods_amd import _lib  # noqa: E402
from dmmfods_amd.graphs.models.Dense_U_Net_lidar import Dense_U_Net_lidar  # noqa: E402

cfg = sys.argv[1] if len(sys.argv) > 1 else "c2"
fam = sys.argv[2] if len(sys.argv) > 2 else "-"
nf = int(sys.argv[3]) if len(sys.argv) > 3 else 5
c = dict(CONFIGS[cfg])
dev = torch.device("cuda", 0)
L = _lib.lib()
for name in ([] if fam == "-" else fam.split("+")):
    _lib.check(L.dmm_set_option(name.encode(), 0))
torch.manual_seed(123)
model = Dense_U_Net_lidar(make_config(c), compute_dtype=c["dtype"]).to(dev).train()
rgb, lidar, tgt = synthetic_batch(c, dev, seed=0)


def fwd():
    with torch.no_grad():
        out = model(rgb, lidar).clone()
    torch.cuda.synchronize()
    return out


def describe(a, b):
    d = (a - b).abs()
    nz = d > 0
    n = int(nz.sum())
    if n == 0:
        return "identical"
    return (f"{n} of {d.numel()} differ, max |d| {float(d.max()):.3e} (max |logit| {float(b.abs().max()):.3e}), median |d| of differing "
            f"{float(d[nz].median()):.3e}, rel L2 {float((a - b).norm() / b.norm()):.3e}")


f = [fwd() for _ in range(nf)]
time.sleep(1.0)
f.append(fwd())          # after a pause (clocks down)
model.loss_backward(tgt)
torch.cuda.synchronize()
f.append(fwd())          # after a backward
print(f"[{cfg} off: {fam} env: " + " ".join(f"{k}={v}" for k, v in os.environ.items() if k.startswith("DMM_")) + "]", flush=True)
for i in range(len(f) - 1):
    tagx = {nf - 1: " (pause before the second)", nf: " (backward before the second)"}.get(i, "")
    print(f"   fwd{i + 1} vs fwd{i + 2}{tagx}: {describe(f[i], f[i + 1])}", flush=True)
if os.environ.get("REF32"):
    del model
    torch.cuda.empty_cache()
    c32 = dict(c, dtype="fp32")
    torch.manual_seed(123)
    m32 = Dense_U_Net_lidar(make_config(c32), compute_dtype="fp32").to(dev).train()
    with torch.no_grad():
        r = m32(rgb, lidar).clone()
    torch.cuda.synchronize()
    print(f"   vs fp32 storage: fwd1 rel L2 {float((f[0] - r).norm() / r.norm()):.4e}, fwd{nf} rel L2 {float((f[nf - 1] - r).norm() / r.norm()):.4e}", flush=True)
