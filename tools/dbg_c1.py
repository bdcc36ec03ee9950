import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from oracle import restatement as R
from tests.test_model_gpu import _model, _oracle_step
arch = R.densenet_arch(121, concat_before_block_num=1, stream_2_in_channels=0)
H, W = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (256, 384)
mfma = int(sys.argv[3]) if len(sys.argv) > 3 else 1
_, g64, _, _ = _oracle_step(R, arch, torch.float64, B=1, H=H, W=W)
_, g32, _, _ = _oracle_step(R, arch, torch.float32, B=1, H=H, W=W)
model = _model(arch, use_mfma=bool(mfma)); model.load_state_dict(R.make_state(arch, seed=123)); model = model.to("cuda").train()
rgb, lidar, tgt = R.make_inputs(arch, 1, H, W, seed=0)
with torch.no_grad(): model(rgb.cuda(), None)
model.loss_backward(tgt.cuda()); torch.cuda.synchronize()
rows = []
num = den = num32 = 0
for k, p in model.named_parameters():
    ref = g64[k]; s = ref.abs().max().clamp_min(1e-30)
    err = ((p.grad.cpu().double() - ref).abs().max() / s).item()
    noise = ((g32[k].double() - ref).abs().max() / s).item()
    rows.append((err / max(noise, 1e-12), err, noise, k))
    num += (p.grad.cpu().double() - ref).pow(2).sum().item(); num32 += (g32[k].double() - ref).pow(2).sum().item(); den += ref.pow(2).sum().item()
print("global L2: gpu %.3e cpu32 %.3e" % ((num/den)**.5, (num32/den)**.5))
import numpy as np
print("median ratio", np.median([r[0] for r in rows]), "median err", np.median([r[1] for r in rows]), "median noise", np.median([r[2] for r in rows]))
rows.sort(reverse=True)
for r in rows[:25]: print("ratio %.1f err %.2e noise %.2e %s" % r)
print("...")
for r in rows[-5:]: print("ratio %.1f err %.2e noise %.2e %s" % r)
