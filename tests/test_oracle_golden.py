"""The CPU oracle (oracle/restatement.py) against the golden vectors produced by the reference's
own Dense_U_Net_lidar module (oracle/make_golden.py).  This is what pins the oracle."""
import gzip
import hashlib
import json
import os

import numpy as np
import pytest
import torch

from oracle import restatement as R

VARIANTS = {"no": (1, 0), "early": (1, 3), "mid2": (2, 3), "mid3": (3, 3), "mid4": (4, 3)}
TINY = dict(growth_rate=8, block_config=(2, 2, 2, 2), num_init_features=16)
RTOL = 2e-5  # CPU restatement vs reference module: same ATen kernels, tiny reassociation only


def _arch(base, v):
    cbb, s2 = VARIANTS[v]
    return R.Arch(**base, concat_before_block_num=cbb, stream_2_in_channels=s2)


def _check_digest(g, name, t, rtol=RTOL, outlier_atol=0.0, outlier_frac=0.0):
    """``outlier_*``: Adam turns a gradient at noise level into a +-lr step of arbitrary sign, so after
    k steps a few elements may legitimately differ by up to 2*lr*k; allow that for a small fraction."""
    mom, sample = g[name + "/mom"], g[name + "/sample"]
    n, stride = int(mom[0]), int(mom[1])
    a = t.detach().to(torch.float64).flatten()
    assert a.numel() == n, name
    scale = max(mom[4], 1e-30)
    got = t.detach().flatten()[::stride][: len(sample)].float().numpy()
    if outlier_atol > 0:
        err = np.abs(got.astype(np.float64) - sample)
        tight = err <= rtol * np.abs(sample) + rtol * scale
        assert (~tight).sum() <= max(1, int(np.ceil(outlier_frac * len(sample)))), (name, int((~tight).sum()))
        assert err.max() <= outlier_atol + rtol * scale, (name, float(err.max()))
        assert abs(a.norm().item() - mom[3]) <= outlier_atol * np.sqrt(n) + rtol * max(mom[3], scale) * 4, name
        return
    np.testing.assert_allclose(got, sample, rtol=rtol, atol=rtol * scale, err_msg=name)
    assert abs(a.norm().item() - mom[3]) <= rtol * max(mom[3], scale) * 4 + 1e-30, name
    assert abs(a.abs().max().item() - mom[4]) <= rtol * scale * 4, name


def test_g1_topology(golden_dir):
    with gzip.open(os.path.join(golden_dir, "g1_topology.json.gz"), "rt") as f:
        g1 = json.load(f)
    for depth in (121, 161, 169, 201):
        for v in VARIANTS:
            ent = g1[f"d{depth}_{v}"]
            arch = _arch(R.DENSENETS[depth], v)
            tab = R.param_table(arch)
            blob = ";".join(f"{k}:{','.join(map(str, s))}" for k, s, _ in tab).encode()
            assert hashlib.sha256(blob).hexdigest() == ent["sha256"], (depth, v)
            assert len(tab) == ent["n_tensors"]
            assert R.num_params(arch) == ent["num_params"]
            assert arch.fusion == ent["fusion"]
            if "keys" in ent:
                assert [[k, list(s)] for k, s, _ in tab] == ent["keys"]
    # parameter totals of the BASELINE configs
    assert R.num_params(_arch(R.DENSENETS[121], "no")) == 22004102
    assert R.num_params(_arch(R.DENSENETS[121], "early")) == 22015244
    assert R.num_params(_arch(R.DENSENETS[121], "mid3")) == 23567564
    assert R.num_params(_arch(R.DENSENETS[169], "mid3")) == 35751628
    assert R.num_params(_arch(R.DENSENETS[201], "mid3")) == 57353932


@pytest.mark.parametrize("variant", list(VARIANTS))
def test_g2_tiny_training_steps(golden_dir, variant):
    g = np.load(os.path.join(golden_dir, f"g2_tiny_{variant}.npz"))
    arch = _arch(TINY, variant)
    P = R.make_state(arch, seed=123)
    tr = R.Trainer(arch, P)
    nsteps = 3
    for step in range(nsteps):
        rgb, lidar, tgt = R.make_inputs(arch, 2, 64, 96, seed=step)
        out = tr.step(rgb, lidar, tgt, do_update=False)
        if step == 0:
            ref = torch.from_numpy(g["logits_full"])
            torch.testing.assert_close(out["logits"], ref, rtol=RTOL, atol=RTOL * ref.abs().max().item())
            for k, t in tr.leaves:
                # fp32 accumulation-order noise: both this oracle and the reference module sit up to
                    # 5e-4*absmax from an fp64 run of the same step (BN gamma/beta grads cancel heavily)
                    _check_digest(g, f"grad0/{k}", t.grad, rtol=1.5e-3)
        tr.opt.step()
        # step 0 is tight.  Later steps follow Adam updates in which gradients at noise level become
        # +-lr steps of arbitrary sign, so trajectories of two fp32 runs separate by ~1e-3 (measured).
        loose = step > 0
        _check_digest(g, f"step{step}/logits", out["logits"], rtol=5e-3 if loose else RTOL)
        np.testing.assert_allclose(out["loss_per_class"].double().numpy(), g[f"step{step}/loss_per_class"],
                                   rtol=2e-3 if loose else 1e-5)
        np.testing.assert_allclose(out["iou"].numpy(), g[f"step{step}/iou"], rtol=1e-6, atol=5e-3 if loose else 0,
                                   equal_nan=True)
        np.testing.assert_allclose(out["acc"].numpy(), g[f"step{step}/acc"], rtol=1e-6, atol=5e-3 if loose else 0)
        if step in (0, nsteps - 1):
            for k, _, _ in R.param_table(arch):
                # Adam divides by sqrt(v)+eps: tiny grad differences are amplified where |g| ~ eps
                _check_digest(g, f"state{step}/{k}", P[k].float(), rtol=2e-4 if step == 0 else 5e-3,
                              outlier_atol=2e-3 * (step + 1), outlier_frac=0.02 if step == 0 else 0.25)
    rgb, lidar, tgt = R.make_inputs(arch, 2, 64, 96, seed=100)
    ev = tr.evaluate(rgb, lidar, tgt)
    ref = torch.from_numpy(g["eval/logits_full"])
    torch.testing.assert_close(ev["logits"], ref, rtol=1e-2, atol=1e-2 * ref.abs().max().item())
    np.testing.assert_allclose(ev["iou"].numpy(), g["eval/iou"], rtol=1e-6, atol=2e-2, equal_nan=True)
    np.testing.assert_allclose(ev["acc"].numpy(), g["eval/acc"], rtol=1e-6, atol=2e-2)


@pytest.mark.parametrize("variant", list(VARIANTS))
def test_g2_tiny_eval_mode_tight(golden_dir, variant):
    """BN running-stat update (momentum .1, unbiased var, num_batches_tracked) + eval-mode forward."""
    g = np.load(os.path.join(golden_dir, f"g2_tiny_{variant}.npz"))
    arch = _arch(TINY, variant)
    P = R.make_state(arch, seed=123)
    with torch.no_grad():
        rgb, lidar, _ = R.make_inputs(arch, 2, 64, 96, seed=0)
        R.forward(P, arch, rgb, lidar, training=True)
        rgb, lidar, _ = R.make_inputs(arch, 2, 64, 96, seed=50)
        out = R.forward(P, arch, rgb, lidar, training=False)
    ref = torch.from_numpy(g["eval1/logits_full"])
    torch.testing.assert_close(out, ref, rtol=RTOL, atol=RTOL * ref.abs().max().item())
    for k, _, kind in R.param_table(arch):
        if kind in ("bn_rm", "bn_rv", "bn_nbt"):
            _check_digest(g, f"eval1_state/{k}", P[k].float(), rtol=RTOL)


def test_g4_c1_densenet121(golden_dir):
    g = np.load(os.path.join(golden_dir, "g4_c1_d121_no.npz"))
    arch = _arch(R.DENSENETS[121], "no")
    P = R.make_state(arch, seed=123)
    tr = R.Trainer(arch, P)
    rgb, lidar, tgt = R.make_inputs(arch, 1, 256, 384, seed=0)
    out = tr.step(rgb, lidar, tgt, do_update=False)
    _check_digest(g, "logits", out["logits"], rtol=1e-4)
    np.testing.assert_allclose(out["loss_per_class"].double().numpy(), g["loss_per_class"], rtol=1e-5)
    grads = dict(tr.leaves)
    for k in ("features.conv0.weight", "features.denseblock3.denselayer24.conv2.weight",
              "decoder.Transposed_Convolution_2.weight", "dec_out_to_heat_maps.refine1.weight",
              "features.denseblock1.denselayer1.norm1.weight", "features.norm0.bias"):
        _check_digest(g, f"grad/{k}", grads[k].grad, rtol=2e-4)


def test_conv_flops_match_survey():
    # forward conv GFLOP per image
    a = _arch(R.DENSENETS[121], "no")
    assert abs(R.conv_flops_forward(a, 256, 384) / 1e9 - 36.75) < 0.05
    a = _arch(R.DENSENETS[121], "early")
    assert abs(R.conv_flops_forward(a, 1280, 1920) / 1e9 - 938.7) < 0.5
    a = _arch(R.DENSENETS[121], "mid3")
    assert abs(R.conv_flops_forward(a, 1280, 1920) / 1e9 - 1133.1) < 0.5


def test_size_constraint_raises_like_reference():
    arch = _arch(TINY, "no")
    P = R.make_state(arch)
    with pytest.raises(ValueError):
        R.forward(P, arch, torch.zeros(1, 3, 100, 100), None)


G3_ARCH = dict(growth_rate=24, block_config=(2, 2, 2, 2), num_init_features=48)


@pytest.mark.parametrize("variant", ["early", "mid3"])
def test_g3_layer_level_vectors(golden_dir, variant):
    """Forward-hook outputs of individual reference modules (dense layer / blocks, transitions, fusion module, decoder stages,
    ConvTranspose with output_size, head) and every parameter gradient, at K = 48 / 72 / 96 channels."""
    g = np.load(os.path.join(golden_dir, f"g3_layers_{variant}.npz"))
    arch = _arch(G3_ARCH, variant)
    P = R.make_state(arch, seed=321)
    for t in P.values():
        if t.is_floating_point():
            t.requires_grad_(False)
    leaves = R.leaf_params(P, arch)
    for _, t in leaves:
        t.requires_grad_(True)
    rgb, lidar, tgt = R.make_inputs(arch, 2, 64, 96, seed=7)
    cap = {}
    logits = R.forward(P, arch, rgb, lidar if arch.fusion != "no" else None, training=True, capture=cap)
    R.bce_with_logits(logits, tgt).sum().backward()
    np.testing.assert_allclose(logits.detach().numpy(), g["logits_full"], rtol=1e-4, atol=1e-4 * np.abs(g["logits_full"]).max())
    names = sorted({k.split("/")[1] for k in g.files if k.startswith("out/")})
    assert len(names) >= 15
    checked = 0
    for name in names:
        if name == "dec_out_to_heat_maps":          # the head module's output is the logits tensor
            _check_digest(g, f"out/{name}", logits, rtol=1e-4)
        elif name == "features.denseblock1.denselayer2":
            # a dense layer returns only its new growth channels: the last 24 channels of the block output
            _check_digest(g, f"out/{name}", cap["features.denseblock1"][:, -arch.growth_rate:], rtol=1e-4)
        else:
            assert name in cap, name
            _check_digest(g, f"out/{name}", cap[name], rtol=1e-4)
        checked += 1
    assert checked == len(names)
    for k, t in leaves:
        _check_digest(g, f"grad/{k}", t.grad, rtol=2e-3)  # BN-parameter gradients are cancelling sums: fp32 order noise


def test_g5_flop_trace_matches_restatement(golden_dir):
    """Per-config forward FLOPs and parameter counts traced on the reference module (meta device) = what the oracle counts."""
    with gzip.open(os.path.join(golden_dir, "g5_flop_trace.json.gz"), "rt") as f:
        g5 = json.load(f)
    want = {"c1": 36.75, "c2": 938.7, "c3": 1133.1, "c4": 1220.2, "c5": 348.9}
    for cname, ent in g5.items():
        arch = _arch(R.DENSENETS[ent["densenet"]], ent["variant"])
        assert R.num_params(arch) == ent["num_params"], cname
        fl = R.conv_flops_forward(arch, ent["H"], ent["W"]) / 1e9
        assert abs(fl - ent["fwd_gflop_per_img"]) < 1e-6 * ent["fwd_gflop_per_img"] + 1e-6, (cname, fl, ent["fwd_gflop_per_img"])
        assert abs(ent["fwd_gflop_per_img"] - want[cname]) < 0.06, cname
        assert len(ent["convs"]) == sum(1 for _, _, kind in R.param_table(arch) if kind in ("conv", "convT")), cname


@pytest.mark.parametrize("case", ["d121_mid3_64", "d201_mid3_64"])
def test_g7_baseline_architectures(golden_dir, case):
    """The C3 (DenseNet-121 mid-fusion) and C5 (DenseNet-201 mid-fusion) architectures as built by the reference's factories
    (M:335-347, M:377-388): logits, loss sums, metrics and every parameter gradient of one training step."""
    g = np.load(os.path.join(golden_dir, f"g7_{case}.npz"))
    meta = json.loads(bytes(g["meta/case"]).decode())
    arch = _arch(R.DENSENETS[meta["densenet"]], meta["variant"])
    P = R.make_state(arch, seed=meta["weight_seed"])
    tr = R.Trainer(arch, P)
    rgb, lidar, tgt = R.make_inputs(arch, meta["B"], meta["H"], meta["W"], seed=meta["data_seed"])
    out = tr.step(rgb, lidar, tgt, do_update=False)
    ref = g["logits_full"]
    np.testing.assert_allclose(out["logits"].numpy(), ref, rtol=1e-4, atol=1e-4 * np.abs(ref).max())
    np.testing.assert_allclose(out["loss_per_class"].double().numpy(), g["loss_per_class"], rtol=1e-5)
    np.testing.assert_allclose(out["iou"].numpy(), g["iou"], rtol=1e-6, equal_nan=True)
    np.testing.assert_allclose(out["acc"].numpy(), g["acc"], rtol=1e-6)
    for k, t in tr.leaves:
        _check_digest(g, f"grad/{k}", t.grad, rtol=5e-3)   # fp32 summation-order noise of cancelling BatchNorm sums


def test_g6_focal_dataset_and_agent_metrics(golden_dir, tmp_path):
    """Rows 8(f): the oracle's focal losses, batched-file slicing and per-batch metric aggregation against what the reference's
    FocalLoss / ClassWiseFocalLoss (L:9-91), WaymoDataset.get_batch (D:87-103) and agent block (A:247-260) produced."""
    g = np.load(os.path.join(golden_dir, "g6_frows.npz"))
    x, t = torch.from_numpy(g["focal/x"]), torch.from_numpy(g["focal/t"])
    cases = {"focal_a1_g2": (1.0, 2.0), "focal_a025_g15": (0.25, 1.5), "classwise_default": ([1, 1, 1], [2, 2, 2]),
             "classwise_mixed": ([1.0, 2.0, 0.5], [2.0, 1.0, 3.0])}
    for name, (al, ga) in cases.items():
        xi = x.clone().requires_grad_(True)
        out = R.focal_loss(xi, t, al, ga)
        out.sum().backward()
        np.testing.assert_allclose(out.detach().numpy(), g[f"focal/{name}/loss"], rtol=1e-5, atol=1e-7)
        np.testing.assert_allclose(xi.grad.numpy(), g[f"focal/{name}/dx"], rtol=1e-4, atol=1e-6)
    np.testing.assert_allclose(R.focal_loss(x, t, 1.0, 2.0).mean().item(), float(g["focal/focal_a1_g2/mean"]), rtol=1e-5)
    np.testing.assert_allclose(R.focal_loss(torch.sigmoid(x), t, 1.0, 2.0, logits=False).numpy(), g["focal/prob_a1_g2/loss"],
                               rtol=1e-5, atol=1e-7)
    for i in range(int(g["data/len"])):
        img, lid, hm = R.split_batch(torch.from_numpy(g[f"data/train/batch_{i}"]))
        files = json.loads(bytes(g["data/train_files"]).decode())
        assert files[i].endswith(f"batch_{i}.pt")
        assert np.array_equal(img.numpy(), g[f"data/get_batch/{i}/image"])
        assert np.array_equal(lid.numpy(), g[f"data/get_batch/{i}/lidar"])
        assert np.array_equal(hm.numpy(), g[f"data/get_batch/{i}/ht_map"])
    pred, gt = torch.from_numpy(g["agent/pred"]), torch.from_numpy(g["agent/gt"])
    np.testing.assert_allclose(R.bce_with_logits(pred, gt).double().sum(dim=(0, 2, 3)).numpy(), g["agent/loss_per_class"], rtol=1e-6)
    np.testing.assert_allclose(R.iou_whole_img_batch(pred, gt).numpy(), g["agent/iou_per_instance"], rtol=1e-6, equal_nan=True)
    iou_pc, nans, acc = R.batch_metrics(pred, gt)
    np.testing.assert_allclose(iou_pc.numpy(), g["agent/iou_per_class"], rtol=1e-6)
    assert np.array_equal(nans.numpy(), g["agent/iou_nans"])
    np.testing.assert_allclose(acc.numpy(), g["agent/acc_per_class"], rtol=1e-6)
    assert int(nans.sum()) >= 5 and float(iou_pc[1]) == 0.0    # the fixture does exercise the NaN rules
