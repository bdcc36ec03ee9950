"""Is test_thin_logits_kernel_matches_generic_kernels' difference a property of the two kernels or run-to-run variation of the forward?
Runs the test's model / inputs: thin on twice, off twice, and prints max |diff| of every pair (bit-equal pairs print 0.0)."""
import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", ".."))
import torch
from dmmfods_amd import _lib
from dmmfods_amd.graphs.models.Dense_U_Net_lidar import densenet121_u_lidar
from dmmfods_amd.utils.Dense_U_Net_lidar_helper import get_config

for seed in (0, 1, 2):
    torch.manual_seed(seed)
    cfg = get_config("/tmp/none")
    model = densenet121_u_lidar(config=cfg, compute_dtype="fp16").cuda().train()
    g = torch.Generator().manual_seed(5)
    for (H, W) in ((64, 96), (160, 288)):
        rgb = (torch.rand(2, 3, H, W, generator=g) * 255).cuda()
        lidar = (torch.rand(2, 1, H, W, generator=g) * 80).cuda()
        outs = []
        for on in (1, 1, 0, 0, 1):
            _lib.check(_lib.lib().dmm_set_option(b"thin_logits", on))
            model.close()
            with torch.no_grad():
                outs.append((on, model(rgb, lidar).clone()))
        _lib.check(_lib.lib().dmm_set_option(b"thin_logits", 1))
        model.close()
        scale = float(outs[0][1].abs().max())
        line = []
        for i in range(len(outs)):
            for j in range(i + 1, len(outs)):
                d = (outs[i][1] - outs[j][1]).abs()
                line.append(f"{outs[i][0]}{outs[j][0]}[{i}{j}]:{float(d.max()) / scale:.2e}({int((d > 0).sum())})")
        print(f"seed {seed} {H}x{W} scale {scale:.3f}  " + " ".join(line), flush=True)
