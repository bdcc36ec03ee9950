"""tools/grad_dump.py <variant> <dtype> <out.pt>: one training step of the tiny test architecture under the library DMM_LIB_PATH names
(default: in-tree); writes logits, metrics and every parameter gradient.  tools/grad_dump.py --diff a.pt b.pt prints per-key differences."""
import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
import torch


def main():
    if sys.argv[1] == "--diff":
        a, b = torch.load(sys.argv[2]), torch.load(sys.argv[3])
        for k in a:
            x, y = a[k].double(), b[k].double()
            d = (x - y).abs().max().item() / max(y.abs().max().item(), 1e-30)
            if d > 1e-6:
                print(f"{k:60s} rel diff {d:.3e}")
            if len(sys.argv) > 4 and sys.argv[4] in k:   # per-channel detail of the named tensors
                print("   ", [f"{v:.2e}" for v in ((x - y).abs() / max(y.abs().max().item(), 1e-30)).flatten().tolist()[:64]])
        return
    variant, dtype, out = sys.argv[1:4]
    from oracle import restatement as R
    from tests.test_model_gpu import _arch, _model, TINY
    arch = _arch(R, TINY, variant)
    model = _model(arch, dtype)
    model.load_state_dict(R.make_state(arch, seed=123))
    model = model.to("cuda").train()
    rgb, lidar, tgt = R.make_inputs(arch, 2, 64, 96, seed=0)
    logits = model(rgb.cuda(), lidar.cuda())
    met = model.loss_backward(tgt.cuda())
    torch.cuda.synchronize()
    d = {"logits": logits.detach().cpu()}
    for k, p in model.named_parameters():
        d["grad/" + k] = p.grad.detach().cpu()
    for k, v in model.state_dict().items():      # the batch statistics of the forward, as the running statistics show them
        if k.endswith(("running_mean", "running_var")):
            d["buf/" + k] = v.detach().cpu()
    torch.save(d, out)


main()
