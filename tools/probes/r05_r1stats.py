"""Probe (round 5): norm1's BatchNorm-backward sums from the 5x5 weight gradient's factor correlations (wg5.hip PA = 3) against the
reductions-only data-gradient pass (DMM_NO_R1_STATS=1) - how far apart are the two paths, and how far apart are two runs of ONE path?"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from oracle import restatement as R
from tests.test_timed_kernels_gpu import _model, _rel, DEV

def run(model, rgb, lidar, tgt, off):
    if off: os.environ["DMM_NO_R1_STATS"] = "1"
    else: os.environ.pop("DMM_NO_R1_STATS", None)
    model.close()
    model(rgb, lidar); model.loss_backward(tgt); torch.cuda.synchronize()
    return {k: p.grad.detach().double().cpu().clone() for k, p in model.named_parameters()}

for dtype in ("fp16", "bf16"):
    arch = R.densenet_arch(121, concat_before_block_num=1, stream_2_in_channels=3)
    model = _model(arch, dtype)
    model.load_state_dict(R.make_state(arch, seed=29))
    model = model.to(DEV).train()
    rgb, lidar, tgt = (t.to(DEV) for t in R.make_inputs(arch, 2, 96, 160, seed=5))
    g = [run(model, rgb, lidar, tgt, o) for o in (0, 1, 0, 1)]
    keys = ["dec_out_to_heat_maps.norm1.bias", "dec_out_to_heat_maps.norm1.weight", "dec_out_to_heat_maps.refine1.weight",
            "dec_out_to_heat_maps.refine0.weight", "decoder.Transposed_Convolution_4.weight", "features.denseblock4.denselayer16.conv1.weight",
            "features.denseblock1.denselayer1.conv1.weight", "features.conv0.weight"]
    for k in keys:
        print(dtype, k, "new/old %.3e  new/new %.3e  old/old %.3e" % (_rel(g[0][k], g[1][k]), _rel(g[0][k], g[2][k]), _rel(g[1][k], g[3][k])))
    a, b = g[0]["dec_out_to_heat_maps.norm1.bias"], g[1]["dec_out_to_heat_maps.norm1.bias"]
    print(dtype, "norm1.bias per-channel max rel diff %.3e" % float(((a - b).abs() / b.abs().clamp_min(1e-30)).max()), "median %.3e" % float(((a - b).abs() / b.abs().clamp_min(1e-30)).median()))
    a, b = g[0]["dec_out_to_heat_maps.norm1.weight"], g[1]["dec_out_to_heat_maps.norm1.weight"]
    print(dtype, "norm1.weight per-channel max rel diff %.3e" % float(((a - b).abs() / b.abs().clamp_min(1e-30)).max()), "median %.3e" % float(((a - b).abs() / b.abs().clamp_min(1e-30)).median()))
    num = sum(float((g[0][k] - g[1][k]).pow(2).sum()) for k in g[0]); den = sum(float(g[1][k].pow(2).sum()) for k in g[0])
    print(dtype, "global relative L2 new/old %.3e" % (num / den) ** 0.5, " weights only: %.3e" % (
        sum(float((g[0][k] - g[1][k]).pow(2).sum()) for k in g[0] if g[0][k].dim() == 4) / sum(float(g[1][k].pow(2).sum()) for k in g[0] if g[0][k].dim() == 4)) ** 0.5)
    import collections
    rels = sorted(((_rel(g[0][k], g[1][k]), k) for k in g[0]), reverse=True)
    print(dtype, "tensors above 1e-2:", sum(r > 1e-2 for r, _ in rels), "of", len(rels), " top:", [(round(r, 4), k) for r, k in rels[:5]])
    worst = max(g[0], key=lambda k: _rel(g[0][k], g[1][k]))
    print(dtype, "worst tensor new/old:", worst, "%.3e" % _rel(g[0][worst], g[1][worst]), " old/old on it: %.3e" % _rel(g[1][worst], g[3][worst]))
    model.close()
