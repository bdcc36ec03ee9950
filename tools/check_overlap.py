"""Consistency soak: gradients with the two-stream backward vs one stream, and run-to-run, on the same weights/inputs."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from dmmfods_amd import _lib
from dmmfods_amd.graphs.models.Dense_U_Net_lidar import Dense_U_Net_lidar

c = dict(bench.CONFIGS["c2"]); c["batch"] = 2; c["height"], c["width"] = 640, 960
torch.manual_seed(123)
dev = torch.device("cuda")
model = Dense_U_Net_lidar(bench.make_config(c), compute_dtype=c["dtype"]).to(dev).train()
rgb, lidar, tgt = bench.synthetic_batch(c, dev, seed=0)
L = _lib.lib()
def grads(overlap):
    _lib.check(L.dmm_set_option(b"overlap_wgrad", overlap))
    with torch.no_grad():
        model(rgb, lidar)
    m = model.loss_backward(tgt)
    torch.cuda.synchronize()
    return model.grad_arena.clone().double(), m["loss_per_class"].clone()
ref, l0 = grads(0)
worst = 0.0
for it in range(30):
    g, l = grads(it % 2)
    d = ((g - ref).norm() / ref.norm()).item()
    worst = max(worst, d)
    assert torch.isfinite(g).all()
    assert torch.equal(l, l0) or (l - l0).abs().max() / l0.abs().max() < 1e-6
print("max relative L2 difference of the gradient arena over 30 runs (overlap on/off alternating): %.3e" % worst)
assert worst < 2e-3
print("ok")
