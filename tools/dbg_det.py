import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from oracle import restatement as R
from tests.test_model_gpu import _arch, _model, TINY
variant = sys.argv[1] if len(sys.argv) > 1 else "early"
dtype = sys.argv[2] if len(sys.argv) > 2 else "fp32"
mfma = int(sys.argv[3]) if len(sys.argv) > 3 else 1
arch = _arch(R, TINY, variant)
model = _model(arch, dtype, use_mfma=bool(mfma)); model.load_state_dict(R.make_state(arch, seed=123)); model = model.to("cuda").train()
rgb, lidar, tgt = R.make_inputs(arch, 2, 64, 96, seed=0)
rgb, lidar, tgt = rgb.cuda(), lidar.cuda(), tgt.cuda()
names = [k for k, _ in model.named_parameters()]
sizes = [p.numel() for _, p in model.named_parameters()]
ref = None; refl = None
bad = {}
for run in range(30):
    with torch.no_grad(): lg = model(rgb, lidar)
    model.loss_backward(tgt); torch.cuda.synchronize()
    g = model.grad_arena.clone()
    if ref is None: ref, refl = g, lg.clone(); continue
    dl = (lg - refl).abs().max().item()
    if dl > 0: print("run", run, "logits differ by", dl)
    off = 0
    for k, n in zip(names, sizes):
        a, b = g[off:off+n], ref[off:off+n]
        s = b.abs().max().item() + 1e-30
        d = ((a - b).abs().max() / s).item()
        if d > 1e-5: bad.setdefault(k, []).append((run, d))
        off += n
print("variant", variant, dtype, "mfma", mfma, "nondeterministic tensors (>1e-5 rel):", len(bad))
for k, v in bad.items(): print("  ", k, ["%d:%.1e" % x for x in v][:8])
