import torch, sys
a, b = torch.load(sys.argv[1]), torch.load(sys.argv[2])
for k in a:
    if "Sequence_4.norm0" in k:
        x, y = a[k].double(), b[k].double()
        print(k, "ch4 main", repr(float(x[4])), "variant", repr(float(y[4])), "rel", float(((x - y).abs() / x.abs().clamp_min(1e-300))[4]),
              "| worst per-channel rel over all channels", float(((x - y).abs() / x.abs().clamp_min(1e-300)).max()))
worst = (0, None)
for k in a:
    if k.startswith("buf/"):
        x, y = a[k].double(), b[k].double()
        r = float(((x - y).abs() / x.abs().clamp_min(1e-30)).max())
        worst = max(worst, (r, k))
print("worst per-channel relative difference of any running statistic:", worst)
