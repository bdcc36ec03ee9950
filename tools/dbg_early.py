import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from oracle import restatement as R
from tests.test_model_gpu import _arch, _model, _oracle_step, TINY
arch = _arch(R, TINY, sys.argv[1] if len(sys.argv) > 1 else "early")
o64, g64, P64, _ = _oracle_step(R, arch, torch.float64)
for run in range(3):
    model = _model(arch); model.load_state_dict(R.make_state(arch, seed=123)); model = model.to("cuda").train()
    rgb, lidar, tgt = R.make_inputs(arch, 2, 64, 96, seed=0)
    with torch.no_grad(): model(rgb.cuda(), lidar.cuda())
    model.loss_backward(tgt.cuda()); torch.cuda.synchronize()
    for k in ("dec_out_to_heat_maps.refine0.weight", "dec_out_to_heat_maps.refine1.weight", "dec_out_to_heat_maps.norm0.weight", "dec_out_to_heat_maps.norm0.bias", "dec_out_to_heat_maps.norm1.weight"):
        p = dict(model.named_parameters())[k]
        ref = g64[k]; d = (p.grad.cpu().double() - ref).abs(); s = ref.abs().max()
        idx = torch.nonzero(d > 1e-3 * s)
        print(run, k, "max rel", (d.max()/s).item(), "n_bad", len(idx), "of", d.numel(), "first bad idx", idx[:6].tolist())
        if k.endswith("refine0.weight") and len(idx):
            i = idx[0]; print("   got", p.grad.cpu()[tuple(i)].item(), "ref", ref[tuple(i)].item(), " bad cin set", sorted(set(idx[:,1].tolist())))
