"""BASELINE configurations that round 1 left without a GPU parity test (C3 = DenseNet-121 mid-fusion, C5 architecture =
DenseNet-201 mid-fusion), the fp16 training trajectory, shape fuzzing over H, W in 32*N and the call-order
contracts of the model's backward entry points."""
import json
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda"
VARIANTS = {"no": (1, 0), "early": (1, 3), "mid2": (2, 3), "mid3": (3, 3), "mid4": (4, 3)}
TINY = dict(growth_rate=8, block_config=(2, 2, 2, 2), num_init_features=16)
G3_ARCH = dict(growth_rate=24, block_config=(2, 2, 2, 2), num_init_features=48)


def _arch(R, base, v):
    cbb, s2 = VARIANTS[v]
    return R.Arch(**base, concat_before_block_num=cbb, stream_2_in_channels=s2)


def _model(arch, dtype="fp32", factory=None, **kw):
    from dmmfods_amd.graphs.models import Dense_U_Net_lidar as M
    from dmmfods_amd.utils.Dense_U_Net_lidar_helper import get_config
    cfg = get_config("/tmp/dmm_test")
    cfg.model.growth_rate, cfg.model.block_config, cfg.model.num_init_features = arch.growth_rate, arch.block_config, arch.num_init_features
    cfg.model.concat_before_block_num, cfg.model.stream_2_in_channels = arch.concat_before_block_num, arch.stream_2_in_channels
    if factory is not None:   # the reference's factories overwrite growth_rate / block_config / num_init_features (M:323-325)
        return getattr(M, factory)(pretrained=False, config=cfg, compute_dtype=dtype, **kw)
    return M.Dense_U_Net_lidar(cfg, compute_dtype=dtype, **kw)


def _rel(a, b):
    a, b = a.double().cpu(), b.double().cpu()
    return ((a - b).abs().max() / b.abs().max().clamp_min(1e-30)).item()


def _oracle_grads(R, arch, dt, B, H, W, seed, wseed, storage=None):
    P = {k: (t.to(dt) if t.is_floating_point() else t.clone()) for k, t in R.make_state(arch, seed=wseed).items()}
    tr = R.Trainer(arch, P, storage=storage)
    rgb, lidar, tgt = R.make_inputs(arch, B, H, W, seed=seed)
    out = tr.step(rgb.to(dt), lidar.to(dt), tgt.to(dt), do_update=False)
    return out, {k: t.grad.clone() for k, t in tr.leaves}


@pytest.mark.parametrize("case", ["d121_mid3_64", "d121_mid3_128", "d201_mid3_64"])
def test_baseline_architectures_against_reference_fixture(case, golden_dir):
    """C3 (DenseNet-121, fusion before block 3) and the C5 architecture (DenseNet-201, fusion before block 3) built by the
    factories, fp32: logits / loss sums / metrics / every parameter gradient against the fixture recorded from the reference's
    own factories (tests/golden/g7_*.npz) and, at 64x96, noise-aware against the fp64 oracle."""
    from oracle import restatement as R
    g = np.load(os.path.join(golden_dir, f"g7_{case}.npz"))
    meta = json.loads(bytes(g["meta/case"]).decode())
    arch = _arch(R, R.DENSENETS[meta["densenet"]], meta["variant"])
    model = _model(arch, factory=f"densenet{meta['densenet']}_u_lidar")
    assert model.fusion == "mid" and model.num_params == R.num_params(arch)
    model.load_state_dict(R.make_state(arch, seed=meta["weight_seed"]))
    model = model.to(DEV).train()
    B, H, W = meta["B"], meta["H"], meta["W"]
    rgb, lidar, tgt = R.make_inputs(arch, B, H, W, seed=meta["data_seed"])
    logits = model(rgb.to(DEV), lidar.to(DEV))
    met = model.loss_backward(tgt.to(DEV))
    torch.cuda.synchronize()
    assert _rel(logits.detach(), torch.from_numpy(g["logits_full"])) < 1e-3
    np.testing.assert_allclose(met["loss_per_class"].cpu().double().numpy(), g["loss_per_class"], rtol=1e-4)
    # thresholded counts: a logit within fp32 noise of the 0.7 threshold may fall on the other side (one pixel in 24 576 seen)
    px = 3.0 / (H * W)
    torch.testing.assert_close(met["iou_per_instance_per_class"].cpu(), torch.from_numpy(g["iou"]), rtol=1e-6, atol=20 * px, equal_nan=True)
    torch.testing.assert_close(met["acc_per_class"].cpu(), torch.from_numpy(g["acc"]).float(), rtol=1e-6, atol=px)
    # Gradients against the reference's own numbers.  Its fp32 CPU arithmetic is itself up to ~0.2 of a tensor's absmax away
    # from an fp64 run on the cancelling BatchNorm-parameter sums of a 121/201-layer net, so: convolution weights
    # per tensor at 5e-2 of absmax (two fp32 summation orders against each other: 3.0e-2 was the worst tensor with 128-column
    # tiles everywhere, 3.3e-2 with the narrower fp32 tiles of small launches - igemm.hip, DMM_MIN_WGS - while the global L2 and
    # the noise-aware fp64 comparison below did not move), BatchNorm parameters with a loose per-tensor backstop, and a global
    # relative L2 over all sampled elements.
    num = den = 0.0
    for k, p in model.named_parameters():
        mom, sample = g[f"grad/{k}/mom"], g[f"grad/{k}/sample"].astype(np.float64)
        assert p.numel() == int(mom[0]), k
        got = p.grad.detach().flatten()[:: int(mom[1])][: len(sample)].double().cpu().numpy()
        err = np.abs(got - sample).max() / max(mom[4], 1e-30)
        assert err < (5e-2 if p.dim() == 4 else 0.5), (k, err)
        num += float(((got - sample) ** 2).sum())
        den += float((sample ** 2).sum())
    print(f"{case}: sampled-gradient rel L2 vs the reference fixture {(num / den) ** 0.5:.3e}")
    assert (num / den) ** 0.5 < 2e-2
    if H == 64:
        _, g64 = _oracle_grads(R, arch, torch.float64, B, H, W, meta["data_seed"], meta["weight_seed"])
        _, g32 = _oracle_grads(R, arch, torch.float32, B, H, W, meta["data_seed"], meta["weight_seed"])
        num = den = num32 = 0.0
        for k, p in model.named_parameters():
            ref = g64[k]
            num += float((p.grad.detach().cpu().double() - ref).pow(2).sum())
            num32 += float((g32[k].double() - ref).pow(2).sum())
            den += float(ref.pow(2).sum())
        err, noise = (num / den) ** 0.5, (num32 / den) ** 0.5
        print(f"{case}: global grad rel L2 gpu {err:.3e} cpu-fp32 {noise:.3e}")
        assert err < max(2e-3, 3 * noise), (err, noise)


def _c2_full_size(monkeypatch=None):
    from oracle import restatement as R
    arch = R.densenet_arch(121, concat_before_block_num=1, stream_2_in_channels=3)
    model = _model(arch, "fp16").to(DEV).train()
    gen = torch.Generator(device=DEV).manual_seed(0)
    B = 4
    rgb = torch.rand(B, 3, 1280, 1920, device=DEV, generator=gen) * 255
    lidar = torch.rand(B, 3, 1280, 1920, device=DEV, generator=gen) * 255 * (torch.rand(B, 3, 1280, 1920, device=DEV, generator=gen) > 0.9)
    tgt = (torch.rand(B, 3, 1280, 1920, device=DEV, generator=gen) > 0.9).float()
    return model, rgb, lidar, tgt


def _host_bce_sums(logits, tgt):
    """Per-class sums of BCE-with-logits in fp64 on the HOST (numpy): independent of every GPU reduction."""
    x = logits.detach().cpu().double().numpy()
    t = tgt.detach().cpu().double().numpy()
    l = np.maximum(x, 0) - x * t + np.log1p(np.exp(-np.abs(x)))
    return torch.from_numpy(l.sum(axis=(0, 2, 3)))


def test_full_size_c2_batch4_step_properties(monkeypatch):
    """BASELINE configs[1] exactly (d121 early fusion, batch 4, 1280x1920, fp16 storage): size-independent properties.
    (a) the fused loss sums equal BCE summed in fp64 ON THE HOST from the returned logits (an independent third reference: round 2
    saw torch's one-step GPU reduction disagree with the kernel on one class, see test_torch_gpu_reduction_... below);
    (b) metric counts equal torch's on those logits; (c) every parameter tensor receives a finite, non-zero gradient;
    (d) a second identical step reproduces the loss; (e) nothing wrote outside the plan's workspace (guard bands)."""
    monkeypatch.setenv("DMM_GUARD_MB", "8")
    model, rgb, lidar, tgt = _c2_full_size()
    B = 4
    guard_before = model.grad_arena.new_full((1 << 20,), 7.0)          # a neighbouring torch allocation the product never sees
    with torch.no_grad():
        logits = model(rgb, lidar)
    met = model.loss_backward(tgt)
    torch.cuda.synchronize()
    assert torch.isfinite(logits).all()
    host = _host_bce_sums(logits, tgt)
    assert _rel(met["loss_per_class"], host) < 1e-6, (met["loss_per_class"].tolist(), host.tolist())
    assert model._last[0].check_guards(), "a kernel wrote outside the plan's workspace"
    assert bool((guard_before == 7.0).all())
    pred, gt = logits >= 0.7, tgt >= 0.7
    inter, union = (pred & gt).sum(dim=(2, 3)).double(), (pred | gt).sum(dim=(2, 3)).double()
    assert torch.equal(met["intersection"].cpu(), inter.cpu()) and torch.equal(met["union"].cpu(), union.cpu())
    # per-sample counts first (the reduction torch also uses for the IoU counts above), then over the batch
    eq = (pred == gt).sum(dim=(2, 3)).sum(dim=0).double() / float(B * 1280 * 1920)
    torch.testing.assert_close(met["acc_per_class"].double().cpu(), eq.cpu(), rtol=1e-6, atol=1e-7)
    ga = model.grad_arena
    assert torch.isfinite(ga).all() and float(ga.abs().max()) > 0
    nz = sum(1 for p in model.parameters() if float(p.grad.abs().max()) > 0)
    assert nz == sum(1 for _ in model.parameters())          # every tensor received a gradient
    loss1 = met["loss_per_class"].clone()
    model._tracked_arena.zero_()
    with torch.no_grad():
        model(rgb, lidar)
    met2 = model.loss_backward(tgt)
    assert _rel(met2["loss_per_class"], loss1) < 1e-6
    assert model._last[0].check_guards()


def test_full_size_c3_two_stream_schedules_agree(monkeypatch):
    """BASELINE configs[2] exactly (d121, second encoder fused before block 3, batch 4, 1280x1920, fp16 storage).  Round 3 runs the
    second stream's encoder on the side stream in forward and the weight gradients beside the data-gradient chain in backward:
    a missing dependency between the streams would show as a difference against the one-stream schedule of the same launches.
    Forward is bit-reproducible (no atomics on stored values), so the logits must be EQUAL; gradients up to the order of the
    fp32 weight-gradient atomics; the loss sums must match the host fp64 sum; nothing writes outside the workspace."""
    from oracle import restatement as R
    from dmmfods_amd import _lib
    monkeypatch.setenv("DMM_GUARD_MB", "8")
    L = _lib.lib()
    arch = R.densenet_arch(121, concat_before_block_num=3, stream_2_in_channels=3)
    model = _model(arch, "fp16").to(DEV).train()
    gen = torch.Generator(device=DEV).manual_seed(1)
    B = 4
    rgb = torch.rand(B, 3, 1280, 1920, device=DEV, generator=gen) * 255
    lidar = torch.rand(B, 3, 1280, 1920, device=DEV, generator=gen) * 255 * (torch.rand(B, 3, 1280, 1920, device=DEV, generator=gen) > 0.9)
    tgt = (torch.rand(B, 3, 1280, 1920, device=DEV, generator=gen) > 0.9).float()
    runs = {}
    try:
        for overlap in (1, 0, 1):
            _lib.check(L.dmm_set_option(b"overlap_wgrad", overlap))
            with torch.no_grad():
                logits = model(rgb, lidar)
            met = model.loss_backward(tgt)
            torch.cuda.synchronize()
            assert torch.isfinite(logits).all() and model._last[0].check_guards()
            runs.setdefault(overlap, []).append((logits.clone(), model.grad_arena.clone().double(), met["loss_per_class"].clone()))
    finally:
        _lib.check(L.dmm_set_option(b"overlap_wgrad", 1))
    (l1, g1, s1), (l1b, g1b, s1b) = runs[1]
    (l0, g0, s0), = runs[0]
    assert torch.equal(l1, l0) and torch.equal(l1, l1b), "the two-stream forward is not the one-stream forward"
    assert _rel(s1, _host_bce_sums(l1, tgt)) < 1e-6
    for a, b in ((g1, g0), (g1, g1b)):
        assert ((a - b).norm() / b.norm()).item() < 2e-3
    assert float(g1.abs().max()) > 0 and torch.isfinite(g1).all()


@pytest.mark.parametrize("depth,batch,H,W,dtype", [(169, 2, 1280, 1920, "fp16"), (201, 8, 640, 960, "bf16")], ids=["c4_d169", "c5_d201_bf16"])
def test_full_size_c4_c5_step_properties(depth, batch, H, W, dtype, monkeypatch):
    """BASELINE configs[3] and configs[4] at their own sizes (mid fusion before block 3): the fused loss sums against the host fp64 sum
    of the returned logits, metric counts against torch, a finite non-zero gradient for every tensor, reproducible loss, guard bands."""
    from oracle import restatement as R
    monkeypatch.setenv("DMM_GUARD_MB", "8")
    arch = R.densenet_arch(depth, concat_before_block_num=3, stream_2_in_channels=3)
    model = _model(arch, dtype).to(DEV).train()
    gen = torch.Generator(device=DEV).manual_seed(2)
    rgb = torch.rand(batch, 3, H, W, device=DEV, generator=gen) * 255
    lidar = torch.rand(batch, 3, H, W, device=DEV, generator=gen) * 255 * (torch.rand(batch, 3, H, W, device=DEV, generator=gen) > 0.9)
    tgt = (torch.rand(batch, 3, H, W, device=DEV, generator=gen) > 0.9).float()
    losses = []
    for _ in range(2):
        with torch.no_grad():
            logits = model(rgb, lidar)
        met = model.loss_backward(tgt)
        torch.cuda.synchronize()
        assert torch.isfinite(logits).all() and model._last[0].check_guards()
        assert _rel(met["loss_per_class"], _host_bce_sums(logits, tgt)) < 1e-6
        losses.append(met["loss_per_class"].clone())
    assert _rel(losses[1], losses[0]) < 1e-6
    pred, gt = logits >= 0.7, tgt >= 0.7
    assert torch.equal(met["intersection"].cpu(), (pred & gt).sum(dim=(2, 3)).double().cpu())
    assert torch.equal(met["union"].cpu(), (pred | gt).sum(dim=(2, 3)).double().cpu())
    ga = model.grad_arena
    assert torch.isfinite(ga).all()
    assert all(float(p.grad.abs().max()) > 0 for p in model.parameters())


@pytest.mark.xfail(strict=False, reason="round 2, gpurun_out/cvp_model.log: torch's one-step GPU reduction sum(dim=(0,2,3)) over the "
                                        "4x3x1280x1920 fp64 BCE tensor returned 3 874 303 for class 1 where the kernel, torch's two-step "
                                        "reduction and (this round) the host fp64 sum give 3 919 912; recorded here instead of printed")
def test_torch_gpu_reduction_agrees_with_host_on_the_c2_loss_tensor():
    """Settles who was wrong in round 2's full-size loss mismatch: the three GPU-side candidates (the product's fused loss sums,
    torch one-step, torch two-step) against the host fp64 sum of the same tensor.  The product is asserted in the test above; this
    one fails (xfail) exactly when torch's one-step reduction is the outlier and shows all four numbers."""
    model, rgb, lidar, tgt = _c2_full_size()
    with torch.no_grad():
        logits = model(rgb, lidar)
    met = model.loss_backward(tgt)
    torch.cuda.synchronize()
    host = _host_bce_sums(logits, tgt)
    bce = torch.nn.functional.binary_cross_entropy_with_logits(logits, tgt, reduction="none").double()
    two_step = bce.sum(dim=(2, 3)).sum(dim=0)
    one_step = bce.sum(dim=(0, 2, 3))
    digest = float(bce.view(-1)[:: 9973].sum())
    msg = (f"kernel {met['loss_per_class'].tolist()} host {host.tolist()} torch two-step {two_step.tolist()} one-step {one_step.tolist()} "
           f"bce digest {digest!r}")
    print(msg)
    # torch evaluates BCE in fp32 per element before the fp64 sum; the host evaluates it in fp64: 1e-8 apart on 1e7-element sums
    assert _rel(two_step, host) < 1e-7, msg
    assert _rel(one_step, host) < 1e-7, msg


def test_fp16_training_trajectory_tracks_fp32_oracle():
    """Five optimiser steps of the timed configuration's arithmetic (fp16 storage, fp32 accumulate) on the K = 48/72/96 net:
    the per-step loss sums stay within a stated bound of the fp32 CPU oracle's trajectory, and the first step's gradients are
    within a norm-wise bound of the oracle's fp16-storage emulation."""
    from oracle import restatement as R
    from dmmfods_amd.optim import FusedAdam
    arch = _arch(R, G3_ARCH, "mid3")
    P = R.make_state(arch, seed=321)
    tr = R.Trainer(arch, P)
    model = _model(arch, "fp16")
    model.load_state_dict(R.make_state(arch, seed=321))
    model = model.to(DEV).train()
    opt = FusedAdam(model)
    # first-step gradients against the fp16-storage emulation of the oracle (fp64 arithmetic)
    _, gh = _oracle_grads(R, arch, torch.float64, 2, 64, 96, 0, 321, storage=torch.float16)
    devs = []
    for step in range(5):
        rgb, lidar, tgt = R.make_inputs(arch, 2, 64, 96, seed=step)
        ref = tr.step(rgb, lidar, tgt)
        with torch.no_grad():
            model(rgb.to(DEV), lidar.to(DEV))
        met = model.loss_backward(tgt.to(DEV))
        if step == 0:
            num = den = 0.0
            for k, p in model.named_parameters():
                num += (p.grad.detach().cpu().double() - gh[k]).pow(2).sum().item()
                den += gh[k].pow(2).sum().item()
            g_err = (num / den) ** 0.5
        opt.step()
        got = met["loss_per_class"].double().cpu()
        want = ref["loss_per_class"].double()
        devs.append(((got - want).abs() / want.abs()).max().item())
    torch.cuda.synchronize()
    print("fp16 trajectory: per-step max rel loss deviation", ["%.2e" % d for d in devs], "step-0 grad rel L2 vs fp16 emulation %.3e" % g_err)
    assert max(devs) < 1e-2, devs          # measured 1e-4 .. 2e-3
    assert g_err < 0.08, g_err             # measured ~3e-2


def test_bf16_storage_d201_mid3_against_oracle_emulation():
    """BASELINE configs[4] arithmetic ("mixed bf16": bf16 storage and MFMA operands, fp32 accumulation) on its architecture
    (DenseNet-201, fusion before block 3) at a small size: logits, loss sums and the global gradient error against the oracle's
    bf16-storage emulation (fp64 arithmetic, rounding where the HIP path stores bf16), with the emulation's own distance from
    the unrounded fp64 run as the yardstick."""
    from oracle import restatement as R
    arch = _arch(R, R.DENSENETS[201], "mid3")
    B, H, W = 2, 64, 96
    ob, gb = _oracle_grads(R, arch, torch.float64, B, H, W, seed=13, wseed=77, storage=torch.bfloat16)
    o64, g64 = _oracle_grads(R, arch, torch.float64, B, H, W, seed=13, wseed=77)
    model = _model(arch, "bf16", factory="densenet201_u_lidar")
    model.load_state_dict(R.make_state(arch, seed=77))
    model = model.to(DEV).train()
    rgb, lidar, tgt = R.make_inputs(arch, B, H, W, seed=13)
    logits = model(rgb.to(DEV), lidar.to(DEV))
    met = model.loss_backward(tgt.to(DEV))
    torch.cuda.synchronize()
    assert torch.isfinite(logits).all() and torch.isfinite(model.grad_arena).all()
    e_log, y_log = _rel(logits.detach(), ob["logits"]), _rel(ob["logits"], o64["logits"])
    e_loss = _rel(met["loss_per_class"], ob["loss_per_class"])
    num = den = ynum = 0.0
    for k, p in model.named_parameters():
        num += float((p.grad.detach().cpu().double() - gb[k]).pow(2).sum())
        ynum += float((gb[k] - g64[k]).pow(2).sum())
        den += float(gb[k].pow(2).sum())
    e_g, y_g = (num / den) ** 0.5, (ynum / den) ** 0.5
    print(f"bf16 d201 mid3: logits {e_log:.3e} (emulation vs fp64 {y_log:.3e}), loss {e_loss:.3e}, grads rel L2 {e_g:.3e} (emulation vs fp64 {y_g:.3e})")
    # bf16 keeps 8 significant bits and this net runs 200 BatchNorms in series: the storage rounding itself moves the emulation
    # y_log / y_g away from fp64.  Two evaluations that meet the same rounding points in a different summation order sit about
    # that far apart; the HIP path must be CLOSER to the emulation than the emulation is to fp64 (measured 0.74x / 0.81x of the
    # yardsticks in round 2).  The per-kernel and layer-depth bf16 bounds are in tests/test_timed_kernels_gpu.py.
    assert e_log < y_log, (e_log, y_log)
    assert e_loss < 5e-3, e_loss          # measured 7.8e-4
    assert e_g < y_g, (e_g, y_g)


@pytest.mark.parametrize("variant,H,W", [("early", 32, 32), ("mid3", 32, 96), ("no", 96, 32), ("mid2", 160, 64), ("mid4", 224, 96),
                                         ("early", 96, 352)])
def test_shape_fuzz_multiples_of_32(variant, H, W):
    """sizes H, W in 32*N that are neither square nor powers of two (row tiles straddle image rows and batch
    borders; 1x1 ... 7x11 maps in block 4): logits, loss, and all gradients of the tiny net against the fp64 oracle."""
    from oracle import restatement as R
    arch = _arch(R, TINY, variant)
    B = 3
    o64, g64 = _oracle_grads(R, arch, torch.float64, B, H, W, seed=H + W, wseed=5)
    _, g32 = _oracle_grads(R, arch, torch.float32, B, H, W, seed=H + W, wseed=5)
    model = _model(arch)
    model.load_state_dict(R.make_state(arch, seed=5))
    model = model.to(DEV).train()
    rgb, lidar, tgt = R.make_inputs(arch, B, H, W, seed=H + W)
    logits = model(rgb.to(DEV), lidar.to(DEV))
    met = model.loss_backward(tgt.to(DEV))
    torch.cuda.synchronize()
    assert _rel(logits.detach(), o64["logits"]) < 1e-3
    assert _rel(met["loss_per_class"], o64["loss_per_class"]) < 1e-5
    for k, p in model.named_parameters():
        ref = g64[k]
        s = ref.abs().max().clamp_min(1e-30)
        err = ((p.grad.detach().cpu().double() - ref).abs().max() / s).item()
        noise = ((g32[k].double() - ref).abs().max() / s).item()
        assert err < max(3e-3, 4 * noise), (k, err, noise)
    with pytest.raises(ValueError):       # reference: ValueError from ConvTranspose2d(output_size=...) (M:261)
        model(torch.zeros(1, 3, H + 8, W, device=DEV), torch.zeros(1, max(arch.stream_2_in_channels, 1), H + 8, W, device=DEV))


def test_autograd_backward_after_fused_backward_does_not_accumulate():
    """dmm_plan_backward (the autograd bridge) must clear the gradient arena like the fused tail does: refine0's weight
    gradient is ADDED into the arena by the four output-parity phases, so a stale arena would leak into the result."""
    from oracle import restatement as R
    arch = _arch(R, TINY, "mid3")
    model = _model(arch)
    model.load_state_dict(R.make_state(arch, seed=123))
    model = model.to(DEV).train()
    rgb, lidar, tgt = R.make_inputs(arch, 2, 64, 96, seed=0)
    rgb, lidar, tgt = rgb.to(DEV), lidar.to(DEV), tgt.to(DEV)
    with torch.no_grad():
        model(rgb, lidar)
    model.loss_backward(tgt)
    want = model.grad_arena.clone()
    k = "dec_out_to_heat_maps.refine0.weight"
    want_r0 = dict(model.named_parameters())[k].grad.clone()
    for _ in range(2):                      # twice in a row: an accumulating path would keep growing
        pred = model(rgb, lidar)
        loss = torch.nn.BCEWithLogitsLoss(reduction="none")(pred, tgt)
        loss.backward(torch.ones_like(loss))
        torch.cuda.synchronize()
        assert _rel(dict(model.named_parameters())[k].grad, want_r0) < 1e-5
        assert ((model.grad_arena - want).norm() / want.norm()).item() < 1e-5


def test_loss_backward_requires_training_forward():
    from oracle import restatement as R
    arch = _arch(R, TINY, "no")
    model = _model(arch).to(DEV)
    rgb, lidar, tgt = R.make_inputs(arch, 1, 32, 32, seed=0)
    model.train()
    with torch.no_grad():
        model(rgb.to(DEV), None)
    model.loss_backward(tgt.to(DEV))              # fine
    model.eval()
    with torch.no_grad():
        model(rgb.to(DEV), None)
    with pytest.raises(RuntimeError):             # an eval forward leaves no batch statistics / saved activations to backprop through
        model.loss_backward(tgt.to(DEV))
