"""Whole-path parity on the GPU: Dense_U_Net_lidar forward, fused BCE + metrics, backward, Adam against the CPU oracle
(oracle/restatement.py) and the committed golden vectors produced by the reference's own module.

Tolerances.  Forward logits: 1e-3 relative to max|logit| (BASELINE north_star), measured ~1e-5.  Gradients: both the
reference module and the oracle sit up to 1e-2*absmax from an fp64 run on some tensors (accumulation-order noise of
heavily cancelling BatchNorm sums), so a gradient passes if its error vs the fp64 oracle is below
max(3e-3, 4 x the CPU-fp32 oracle's own error) relative to the tensor's absmax.  fp16 storage is judged against the
oracle's fp16-storage emulation with norm-wise bounds (it is intrinsically ~1e-2 on logits for this network)."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

VARIANTS = {"no": (1, 0), "early": (1, 3), "mid2": (2, 3), "mid3": (3, 3), "mid4": (4, 3)}
TINY = dict(growth_rate=8, block_config=(2, 2, 2, 2), num_init_features=16)
DEV = "cuda"


def _arch(R, base, v):
    cbb, s2 = VARIANTS[v]
    return R.Arch(**base, concat_before_block_num=cbb, stream_2_in_channels=s2)


def _model(arch, dtype="fp32", **kw):
    from dmmfods_amd.graphs.models.Dense_U_Net_lidar import Dense_U_Net_lidar
    from dmmfods_amd.utils.Dense_U_Net_lidar_helper import get_config
    cfg = get_config("/tmp/dmm_test")
    cfg.model.growth_rate, cfg.model.block_config, cfg.model.num_init_features = arch.growth_rate, arch.block_config, arch.num_init_features
    cfg.model.concat_before_block_num, cfg.model.stream_2_in_channels = arch.concat_before_block_num, arch.stream_2_in_channels
    return Dense_U_Net_lidar(cfg, compute_dtype=dtype, **kw)


def _rel(a, b):
    a, b = a.double().cpu(), b.double().cpu()
    return ((a - b).abs().max() / b.abs().max().clamp_min(1e-30)).item()


def _oracle_step(R, arch, dt, B=2, H=64, W=96, seed=0, storage=None, update=False, wseed=123):
    P = {k: (t.to(dt) if t.is_floating_point() else t.clone()) for k, t in R.make_state(arch, seed=wseed).items()}
    tr = R.Trainer(arch, P, storage=storage)
    rgb, lidar, tgt = R.make_inputs(arch, B, H, W, seed=seed)
    out = tr.step(rgb.to(dt), lidar.to(dt), tgt.to(dt), do_update=update)
    return out, {k: t.grad.clone() for k, t in tr.leaves}, P, tr


@pytest.mark.parametrize("variant", list(VARIANTS))
def test_tiny_training_step_fp32(variant, golden_dir):
    from oracle import restatement as R
    arch = _arch(R, TINY, variant)
    o64, g64, P64, _ = _oracle_step(R, arch, torch.float64)
    o32, g32, _, _ = _oracle_step(R, arch, torch.float32)
    model = _model(arch)
    model.load_state_dict(R.make_state(arch, seed=123))
    model = model.to(DEV).train()
    rgb, lidar, tgt = R.make_inputs(arch, 2, 64, 96, seed=0)
    logits = model(rgb.to(DEV), lidar.to(DEV))
    met = model.loss_backward(tgt.to(DEV))
    torch.cuda.synchronize()
    assert _rel(logits.detach(), o64["logits"]) < 1e-3
    # the committed fixture from the reference's own module
    g = np.load(os.path.join(golden_dir, f"g2_tiny_{variant}.npz"))
    assert _rel(logits.detach(), torch.from_numpy(g["logits_full"])) < 1e-3
    assert _rel(met["loss_per_class"], o64["loss_per_class"]) < 1e-5
    np.testing.assert_allclose(met["loss_per_class"].cpu().double().numpy(), g["step0/loss_per_class"], rtol=1e-4)
    torch.testing.assert_close(met["iou_per_instance_per_class"].cpu(), torch.from_numpy(g["step0/iou"]), rtol=1e-6, atol=1e-6, equal_nan=True)
    torch.testing.assert_close(met["acc_per_class"].cpu(), torch.from_numpy(g["step0/acc"]).float(), rtol=1e-6, atol=1e-6)
    for k, p in model.named_parameters():
        ref = g64[k]
        s = ref.abs().max().clamp_min(1e-30)
        err = ((p.grad.detach().cpu().double() - ref).abs().max() / s).item()
        noise = ((g32[k].double() - ref).abs().max() / s).item()
        assert err < max(3e-3, 4 * noise), (k, err, noise)
    sd = model.state_dict()
    for k, v in sd.items():
        if k.endswith(("running_mean", "running_var")):
            assert _rel(v, P64[k]) < 1e-4, k
        if k.endswith("num_batches_tracked"):
            assert int(v) == 1


G3_ARCH = dict(growth_rate=24, block_config=(2, 2, 2, 2), num_init_features=48)


@pytest.mark.parametrize("variant", ["early", "mid3"])
def test_odd_channel_counts_against_reference_vectors_fp32(variant, golden_dir):
    """K = 48 / 72 / 96 channels (chunk tails in every gather): logits and every parameter gradient against the fixture
    recorded from the reference module (tests/golden/g3_layers_*.npz)."""
    from oracle import restatement as R
    from tests.test_oracle_golden import _check_digest
    arch = _arch(R, G3_ARCH, variant)
    g = np.load(os.path.join(golden_dir, f"g3_layers_{variant}.npz"))
    model = _model(arch)
    model.load_state_dict(R.make_state(arch, seed=321))
    model = model.to(DEV).train()
    rgb, lidar, tgt = R.make_inputs(arch, 2, 64, 96, seed=7)
    logits = model(rgb.to(DEV), lidar.to(DEV))
    model.loss_backward(tgt.to(DEV))
    torch.cuda.synchronize()
    assert _rel(logits.detach(), torch.from_numpy(g["logits_full"])) < 1e-3
    # gradients: against the reference's vectors (fp32 CPU, its own summation noise) and, noise-aware, against the fp64 oracle
    o64, g64, _, _ = _oracle_step(R, arch, torch.float64, seed=7, wseed=321)
    o32, g32, _, _ = _oracle_step(R, arch, torch.float32, seed=7, wseed=321)
    for k, p in model.named_parameters():
        _check_digest(g, f"grad/{k}", p.grad.detach().cpu(), rtol=1e-2)
        ref = g64[k]
        s = ref.abs().max().clamp_min(1e-30)
        err = ((p.grad.detach().cpu().double() - ref).abs().max() / s).item()
        noise = ((g32[k].double() - ref).abs().max() / s).item()
        assert err < max(3e-3, 4 * noise), (k, err, noise)


@pytest.mark.parametrize("variant", ["no", "mid3"])
def test_eval_mode_and_autograd_path_fp32(variant, golden_dir):
    from oracle import restatement as R
    arch = _arch(R, TINY, variant)
    model = _model(arch)
    model.load_state_dict(R.make_state(arch, seed=123))
    model = model.to(DEV).train()
    rgb, lidar, tgt = R.make_inputs(arch, 2, 64, 96, seed=0)
    # reference-style step: unreduced loss, backward(ones) through torch autograd (A:247, A:264)
    pred = model(rgb.to(DEV), lidar.to(DEV))
    loss = torch.nn.BCEWithLogitsLoss(reduction="none")(pred, tgt.to(DEV))
    loss.backward(torch.ones_like(loss))
    g_auto = model.grad_arena.clone()
    model.loss_backward(tgt.to(DEV))
    torch.cuda.synchronize()
    assert _rel(g_auto, model.grad_arena) < 1e-5
    # eval mode on fresh data uses the running statistics set by that one training forward (golden 'eval1')
    model.eval()
    rgb2, lidar2, _ = R.make_inputs(arch, 2, 64, 96, seed=50)
    with torch.no_grad():
        out = model(rgb2.to(DEV), lidar2.to(DEV))
    g = np.load(os.path.join(golden_dir, f"g2_tiny_{variant}.npz"))
    # two training forwards happened here (autograd + fused), the golden had one: redo with a fresh model
    m2 = _model(arch)
    m2.load_state_dict(R.make_state(arch, seed=123))
    m2 = m2.to(DEV).train()
    with torch.no_grad():
        m2(rgb.to(DEV), lidar.to(DEV))
        m2.eval()
        out2 = m2(rgb2.to(DEV), lidar2.to(DEV))
    assert _rel(out2, torch.from_numpy(g["eval1/logits_full"])) < 1e-3
    assert out.shape == out2.shape
    # eval-mode samples are independent of their batch mates
    with torch.no_grad():
        solo = m2(rgb2[:1].to(DEV), lidar2[:1].to(DEV))
    assert _rel(solo, out2[:1]) < 1e-5


def test_adam_steps_follow_oracle():
    from oracle import restatement as R
    from dmmfods_amd.optim import FusedAdam
    arch = _arch(R, TINY, "early")
    P = R.make_state(arch, seed=123)
    tr = R.Trainer(arch, P)
    model = _model(arch)
    model.load_state_dict(R.make_state(arch, seed=123))
    model = model.to(DEV).train()
    opt = FusedAdam(model)
    for step in range(2):
        rgb, lidar, tgt = R.make_inputs(arch, 2, 64, 96, seed=step)
        tr.step(rgb, lidar, tgt)
        with torch.no_grad():
            model(rgb.to(DEV), lidar.to(DEV))
        model.loss_backward(tgt.to(DEV))
        opt.step()
    torch.cuda.synchronize()
    sd = model.state_dict()
    bad = tot = 0
    for k, ref in tr.leaves:
        d = (sd[k].cpu() - ref.detach()).abs()
        assert d.max() <= 2 * 2e-3 + 1e-4, k          # at most a sign flip of a noise-level gradient per step
        bad += int((d > 2e-4).sum())
        tot += d.numel()
    assert bad / tot < 0.05
    osd = opt.state_dict()
    ref_sd = tr.opt.state_dict()
    assert len(osd["state"]) == len(ref_sd["state"]) and set(osd["state"][0]) == {"step", "exp_avg", "exp_avg_sq"}


@pytest.mark.parametrize("variant", ["early", "mid3"])
def test_tiny_training_step_fp16(variant):
    from oracle import restatement as R
    arch = _arch(R, TINY, variant)
    oh, gh, _, _ = _oracle_step(R, arch, torch.float64, storage=torch.float16)
    model = _model(arch, "fp16")
    model.load_state_dict(R.make_state(arch, seed=123))
    model = model.to(DEV).train()
    rgb, lidar, tgt = R.make_inputs(arch, 2, 64, 96, seed=0)
    logits = model(rgb.to(DEV), lidar.to(DEV))
    met = model.loss_backward(tgt.to(DEV))
    torch.cuda.synchronize()
    assert _rel(logits.detach(), oh["logits"]) < 2e-2        # measured 0.6e-2 / 1.1e-2
    assert _rel(met["loss_per_class"], oh["loss_per_class"]) < 2e-3
    num = den = 0.0
    for k, p in model.named_parameters():
        num += (p.grad.detach().cpu().double() - gh[k]).pow(2).sum().item()
        den += gh[k].pow(2).sum().item()
    print(f"tiny fp16 {variant}: grad rel L2 vs the fp16-storage emulation {(num / den) ** 0.5:.3e}, logits {_rel(logits.detach(), oh['logits']):.3e}")
    assert (num / den) ** 0.5 < 0.09        # measured 6.2e-2 (early) / 2.9e-2 (mid3): fp16 storage rounding of cancelling BatchNorm sums


def test_c1_densenet121_golden_and_directional_derivative(golden_dir):
    """BASELINE configs[0]: d121 no-fusion 1x3x256x384 fp32 against the reference-generated fixture, plus an
    oracle-free check of the whole backward: <grad, d> equals the central difference of the summed loss."""
    from oracle import restatement as R
    arch = R.densenet_arch(121, concat_before_block_num=1, stream_2_in_channels=0)
    model = _model(arch)
    model.load_state_dict(R.make_state(arch, seed=123))
    model = model.to(DEV).train()
    rgb, lidar, tgt = R.make_inputs(arch, 1, 256, 384, seed=0)
    rgb, tgt = rgb.to(DEV), tgt.to(DEV)
    with torch.no_grad():
        logits = model(rgb, None)
    met = model.loss_backward(tgt)
    g = np.load(os.path.join(golden_dir, "g4_c1_d121_no.npz"))
    mom, sample = g["logits/mom"], torch.from_numpy(g["logits/sample"])
    got = logits.flatten()[:: int(mom[1])][: len(sample)].cpu()
    assert ((got - sample).abs().max() / mom[4]).item() < 1e-3
    np.testing.assert_allclose(met["loss_per_class"].cpu().double().numpy(), g["loss_per_class"], rtol=1e-4)
    # Gradients.  On this network the reference's own fp32 arithmetic is only good to ~1e-2 of a tensor's absmax
    # (median over tensors; up to 0.18) against an fp64 run -- measured with the CPU oracle -- so: (i) the sampled
    # gradients in the reference-generated fixture are matched at 3e-2, (ii) every tensor is compared with the fp64
    # oracle using the CPU-fp32 oracle's own error as yardstick, (iii) the global relative L2 error must not exceed
    # 3x that of CPU fp32.
    for k in ("features.conv0.weight", "features.denseblock3.denselayer24.conv2.weight", "decoder.Transposed_Convolution_2.weight",
              "dec_out_to_heat_maps.refine1.weight"):
        mom, sample = g[f"grad/{k}/mom"], torch.from_numpy(g[f"grad/{k}/sample"])
        p = dict(model.named_parameters())[k]
        got = p.grad.flatten()[:: int(mom[1])][: len(sample)].cpu()
        assert ((got - sample).abs().max() / mom[4]).item() < 3e-2, k
    _, g64, _, _ = _oracle_step(R, arch, torch.float64, B=1, H=256, W=384)
    _, g32, _, _ = _oracle_step(R, arch, torch.float32, B=1, H=256, W=384)
    # The noise is heavy-tailed (measured: per-tensor max-error ratios GPU/CPU-fp32 spread 0.1x .. 8x with median 1.1
    # and equal global L2), so the per-tensor bounds are norm-wise with a loose max-norm backstop.
    num = den = num32 = 0.0
    ratios = []
    for k, p in model.named_parameters():
        ref = g64[k]
        d = p.grad.detach().cpu().double() - ref
        d32 = g32[k].double() - ref
        s = ref.abs().max().clamp_min(1e-30)
        err, noise = (d.abs().max() / s).item(), (d32.abs().max() / s).item()
        assert err < max(3e-3, 12 * noise), (k, err, noise)
        rn = ref.norm().clamp_min(1e-30)
        assert (d.norm() / rn).item() < max(2e-3, 8 * (d32.norm() / rn).item()), (k, (d.norm() / rn).item(), (d32.norm() / rn).item())
        ratios.append(err / max(noise, 1e-12))
        num += d.pow(2).sum().item()
        num32 += d32.pow(2).sum().item()
        den += ref.pow(2).sum().item()
    assert float(np.median(ratios)) < 2.0, float(np.median(ratios))
    assert (num / den) ** 0.5 < max(1e-3, 2 * (num32 / den) ** 0.5), ((num / den) ** 0.5, (num32 / den) ** 0.5)
    # directional derivative along a random direction (fp32 forward noise limits eps from below)
    grad = model.grad_arena.clone()
    gen = torch.Generator(device=DEV).manual_seed(1)
    d = torch.randn(grad.shape, device=DEV, generator=gen)
    d = d / d.norm()
    w0 = model.param_arena.clone()
    eps = 2e-3
    losses = []
    for sgn in (+1, -1):
        model.param_arena.copy_(w0 + sgn * eps * d)
        with torch.no_grad():
            lg = model(rgb, None)
        losses.append(model.loss_metrics(lg, tgt)["loss_per_class"].double().sum().item())
    model.param_arena.copy_(w0)
    fd = (losses[0] - losses[1]) / (2 * eps)
    an = (grad.double() * d.double()).sum().item()
    assert abs(fd - an) <= 2e-2 * max(abs(an), 1.0) + 5e-2 * grad.double().norm().item() * eps, (fd, an)


@pytest.mark.gpu
def test_thin_logits_kernel_matches_generic_kernels():
    """The gather-once kernel of the 5x5 head conv (thin.hip) against the generic implicit-GEMM path on the same fp16 tensors:
    identical operands, only the fp32 summation order differs."""
    import torch
    from dmmfods_amd import _lib
    from dmmfods_amd.graphs.models.Dense_U_Net_lidar import densenet121_u_lidar
    from dmmfods_amd.utils.Dense_U_Net_lidar_helper import get_config
    cfg = get_config("/tmp/none")
    # The weights are seeded HERE: with the process-wide generator in whatever state the tests before left it, the difference
    # between the two summation orders ranged over 1.0e-5 ... 8.7e-5 of max|logit| across three seeds (tools/probes/thin_flaky.py,
    # round 4; the forward itself is bit-reproducible: thin/thin and generic/generic pairs differ by exactly 0) and one driver run
    # drew 1.06e-4 against the old bound of 1e-4.  1600 fp32 products per logit with heavy cancellation: the bound is 3e-4.
    torch.manual_seed(5)
    model = densenet121_u_lidar(config=cfg, compute_dtype="fp16").cuda().train()
    g = torch.Generator().manual_seed(5)
    for (H, W) in ((64, 96), (160, 288)):      # one x strip / three x strips with a ragged last one, several y strips
        rgb = (torch.rand(2, 3, H, W, generator=g) * 255).cuda()
        lidar = (torch.rand(2, 1, H, W, generator=g) * 80).cuda()
        out = {}
        for on in (1, 0):
            _lib.check(_lib.lib().dmm_set_option(b"thin_logits", on))
            model.close()                     # a plan fixes the kernel family of every launch when it is bound
            with torch.no_grad():
                out[on] = model(rgb, lidar).clone()
        _lib.check(_lib.lib().dmm_set_option(b"thin_logits", 1))
        model.close()
        scale = float(out[0].abs().max())
        assert scale > 0 and torch.isfinite(out[1]).all()
        assert float((out[1] - out[0]).abs().max()) <= 3e-4 * scale + 1e-5, (H, W, float((out[1] - out[0]).abs().max()), scale)
    with pytest.raises(ValueError):
        _lib.check(_lib.lib().dmm_set_option(b"no_such_option", 1))


@pytest.mark.gpu
def test_two_stream_backward_matches_single_stream():
    """Weight gradients and backward leaves on the side stream (Streams) against the same launches on one stream:
    the only difference allowed is the order of fp32 atomic adds."""
    from oracle import restatement as R
    from dmmfods_amd import _lib
    arch = _arch(R, dict(growth_rate=16, block_config=(2, 2, 2, 2), num_init_features=32), "mid3")
    model = _model(arch, dtype="fp16")
    model.load_state_dict(R.make_state(arch, seed=11))
    model = model.to(DEV).train()
    rgb, lidar, tgt = R.make_inputs(arch, 2, 128, 192, seed=3)
    rgb, lidar, tgt = rgb.to(DEV), lidar.to(DEV), tgt.to(DEV)
    L = _lib.lib()
    out = {}
    try:
        for rep in range(6):
            on = rep % 2
            _lib.check(L.dmm_set_option(b"overlap_wgrad", on))
            with torch.no_grad():
                model(rgb, lidar)
            model.loss_backward(tgt)
            torch.cuda.synchronize()
            g = model.grad_arena.double().clone()
            assert torch.isfinite(g).all()
            if 0 in out:
                assert ((g - out[0]).norm() / out[0].norm()).item() < 1e-5, (rep, on)
            else:
                out[0] = g
    finally:
        _lib.check(L.dmm_set_option(b"overlap_wgrad", 1))


@pytest.mark.gpu
@pytest.mark.parametrize("depth,variant", [(169, "mid3"), (201, "early"), (161, "mid2")])
def test_other_densenet_depths_forward_backward_fp32(depth, variant):
    """The C4 / C5 architectures (DenseNet-169 / -201; -161 has growth 48 and 96 stem channels) at a small image size: logits and
    all gradients against the fp64 oracle, noise-aware like the other gradient checks."""
    from oracle import restatement as R
    arch = _arch(R, R.DENSENETS[depth], variant)
    o64, g64, _, _ = _oracle_step(R, arch, torch.float64, B=1, H=64, W=96, seed=5, wseed=77)
    o32, g32, _, _ = _oracle_step(R, arch, torch.float32, B=1, H=64, W=96, seed=5, wseed=77)
    model = _model(arch)
    model.load_state_dict(R.make_state(arch, seed=77))
    model = model.to(DEV).train()
    rgb, lidar, tgt = R.make_inputs(arch, 1, 64, 96, seed=5)
    logits = model(rgb.to(DEV), lidar.to(DEV))
    met = model.loss_backward(tgt.to(DEV))
    torch.cuda.synchronize()
    assert _rel(logits.detach(), o64["logits"]) < 1e-3
    assert _rel(met["loss_per_class"], o64["loss_per_class"]) < 1e-4
    num = den = num32 = 0.0
    for k, p in model.named_parameters():
        ref = g64[k]
        num += float((p.grad.detach().cpu().double() - ref).pow(2).sum())
        num32 += float((g32[k].double() - ref).pow(2).sum())
        den += float(ref.pow(2).sum())
    err, noise = (num / den) ** 0.5, (num32 / den) ** 0.5
    assert err < max(5e-3, 4 * noise), (err, noise)   # global relative L2 of all gradients vs the CPU-fp32 noise


@pytest.mark.gpu
@pytest.mark.parametrize("dtype", ["fp16", "bf16"])
def test_conv3_halo_kernels_match_generic_kernels(dtype):
    """The LDS-halo kernels of the dense layers' 3x3 convolutions (conv3.hip: forward with BN+ReLU prologue and statistics
    epilogue; data gradient with the deferred-correction prologue and the fused BN/ReLU backward epilogue) against the generic
    implicit-GEMM path on the same 16-bit tensors, on a net whose dense layers have the DenseNet shapes (128 -> 32 channels)
    and whose maps are not multiples of the 8 x 16 tile (20 x 28 ... 5 x 7 pixels).  The only differences allowed: fp32
    summation order, and the 16-bit staging of the data gradient before its ReLU mask."""
    from oracle import restatement as R
    from dmmfods_amd import _lib
    arch = _arch(R, dict(growth_rate=32, block_config=(2, 3, 2, 2), num_init_features=64), "mid3")
    model = _model(arch, dtype=dtype)
    model.load_state_dict(R.make_state(arch, seed=17))
    model = model.to(DEV).train()
    rgb, lidar, tgt = R.make_inputs(arch, 2, 160, 224, seed=4)
    rgb, lidar, tgt = rgb.to(DEV), lidar.to(DEV), tgt.to(DEV)
    L = _lib.lib()
    out = {}
    try:
        for on in (1, 0):
            _lib.check(L.dmm_set_option(b"conv3", on))
            model.close()                     # a plan fixes the kernel family of every launch when it is bound
            with torch.no_grad():
                logits = model(rgb, lidar).clone()
            met = model.loss_backward(tgt)
            torch.cuda.synchronize()
            out[on] = (logits, met["loss_per_class"].clone(), model.grad_arena.double().clone())
            model._tracked_arena.zero_()
    finally:
        _lib.check(L.dmm_set_option(b"conv3", 1))
        model.close()
    assert torch.isfinite(out[1][0]).all() and torch.isfinite(out[1][2]).all()
    tol = 1e-2 if dtype == "fp16" else 1.6e-2   # measured 3.1e-3 / 0 on the logits, 4.9e-3 / 3e-4 on the gradients
    e_log = _rel(out[1][0], out[0][0])
    e_g = ((out[1][2] - out[0][2]).norm() / out[0][2].norm()).item()
    print(f"conv3 vs generic ({dtype}): logits {e_log:.3e}, loss {_rel(out[1][1], out[0][1]):.3e}, grads rel L2 {e_g:.3e}")
    assert e_log < tol and _rel(out[1][1], out[0][1]) < tol
    assert e_g < 3 * tol
    # The growth convolution's weight-gradient kernel alone (wg3.hip vs the generic transposed form): with every other kernel
    # unchanged both see bit-identical operands, so only the fp32 summation order differs.  (Across the conv3 switch above these
    # particular gradients move by ~15 %: they are cancelling sums of the BatchNorm-corrected gradient that 16-bit storage
    # resolves to ~20 % on either path - measured against the oracle's emulation, tools/gpu_lab.py.)
    grads = {}
    try:
        for on in (1, 0):
            _lib.check(L.dmm_set_option(b"wg3", on))
            model.close()                     # a plan fixes the kernel family of every launch when it is bound
            with torch.no_grad():
                model(rgb, lidar)
            model.loss_backward(tgt)
            torch.cuda.synchronize()
            grads[on] = {k: p.grad.detach().double().clone() for k, p in model.named_parameters() if k.endswith("conv2.weight")}
    finally:
        _lib.check(L.dmm_set_option(b"wg3", 1))
        model.close()
    worst = max(((grads[1][k] - grads[0][k]).norm() / grads[0][k].norm()).item() for k in grads[1])
    print(f"   wg3 vs generic on identical operands: worst conv2.weight gradient rel L2 {worst:.3e}")
    assert len(grads[1]) >= 9 and worst < 2e-4


@pytest.mark.gpu
@pytest.mark.parametrize("dtype", ["fp16", "bf16"])
def test_parity_phase_weight_gradient_kernel_matches_generic(dtype):
    """wgp.hip (weight gradients of the ConvTranspose stages and of the head's 3x3 over the upsampled map, all taps of a parity
    phase per LDS tile) against the generic kernel inside DenseNet-121 (decoder 1024 -> 512 -> 256 -> 128 channels), maps that
    are not multiples of the 8 x 16 tile.  Nothing upstream of a weight gradient changes with the switch, so both kernels see
    bit-identical operands and only the fp32 summation order differs."""
    import ctypes as C
    from oracle import restatement as R
    from dmmfods_amd import _lib
    arch = _arch(R, R.DENSENETS[121], "early")
    model = _model(arch, dtype=dtype)
    model.load_state_dict(R.make_state(arch, seed=23))
    model = model.to(DEV).train()
    rgb, lidar, tgt = R.make_inputs(arch, 2, 96, 160, seed=6)
    rgb, lidar, tgt = rgb.to(DEV), lidar.to(DEV), tgt.to(DEV)
    L = _lib.lib()
    grads = {}
    try:
        for on in (1, 0):
            _lib.check(L.dmm_set_option(b"wgp", on))
            model.close()                     # a plan fixes the kernel family of every launch when it is bound
            with torch.no_grad():
                model(rgb, lidar)
            model.loss_backward(tgt)
            torch.cuda.synchronize()
            grads[on] = {k: p.grad.detach().double().clone() for k, p in model.named_parameters()
                         if "Transposed_Convolution_" in k and k.endswith(".weight") and "Sequence" not in k or k.endswith("refine0.weight")}
            if on:   # the plan labels its launches by the kernel that runs them: the phases with 2 or 4 taps must be on wgp
                plan = model._last[0]
                labels = []
                for i in range(L.dmm_plan_profile_num_ops(plan.handle, 1)):
                    label, fl, by = C.c_char_p(), C.c_double(), C.c_double()
                    L.dmm_plan_profile_op(plan.handle, 1, i, C.byref(label), C.byref(fl), C.byref(by))
                    labels.append((label.value or b"").decode())
    finally:
        _lib.check(L.dmm_set_option(b"wgp", 1))
        model.close()
    on_wgp = [x for x in labels if x.startswith("wgp.")]
    # round 5: the four phases of a ConvTranspose stage (the one-tap phase included) are ONE wgp launch, as are the head's four
    assert len(on_wgp) == 5 and sum("/d.TC_" in x for x in on_wgp) == 4, on_wgp
    assert len(grads[1]) == 5
    worst = max(((grads[1][k] - grads[0][k]).norm() / grads[0][k].norm()).item() for k in grads[1])
    print(f"wgp vs generic on identical operands ({dtype}): worst rel L2 {worst:.3e}; launches on wgp: {len(on_wgp)}")
    assert all(torch.isfinite(v).all() for v in grads[1].values()) and worst < 2e-4
    # the same for the two convolutions with an 8-channel operand (wg5.hip vs the generic kernel): the head's 5x5 onto the classes
    # (transposed form) and the stem's 7x7 stride 2 over the raw input (normal form, deferred-correction prologue on the gradient)
    # ... and the raw-input segment of the head's 3x3 (normal form, BN+ReLU prologue on the 8-channel operand, one pass for all four parities)
    names = ["dec_out_to_heat_maps.refine1.weight", "features.conv0.weight", "dec_out_to_heat_maps.refine0.weight"]
    g5 = {}
    # (round 5: by default wg5.hip's launch for refine1 ALSO yields norm1's BatchNorm-backward sums - its factor form, covered by
    # test_head_norm1_batchnorm_sums_from_the_5x5_weight_gradient and test_conv5_weight_gradient_and_batchnorm_sums_in_one_pass -, so the
    # family switch would change operands upstream of every other gradient; this comparison pins the plain form on both sides)
    import os
    keep = os.environ.get("DMM_NO_R1_STATS")
    os.environ["DMM_NO_R1_STATS"] = "1"
    try:
        for on in (1, 0):
            _lib.check(L.dmm_set_option(b"wg5", on))
            model.close()                     # a plan fixes the kernel family of every launch when it is bound
            with torch.no_grad():
                model(rgb, lidar)
            model.loss_backward(tgt)
            torch.cuda.synchronize()
            g5[on] = {k: p.grad.detach().double().clone() for k, p in model.named_parameters() if k in names}
    finally:
        _lib.check(L.dmm_set_option(b"wg5", 1))
        model.close()
        if keep is None:
            os.environ.pop("DMM_NO_R1_STATS", None)
        else:
            os.environ["DMM_NO_R1_STATS"] = keep
    # (round 5: + the finish launch of the raw-input segment, which turns wg5.hip's factor correlations into the packed gradient)
    assert sorted(x.split("/")[1] for x in labels if x.startswith("wg5.n")) == ["f.conv0", "h.refine0.raw", "h.refine1"], [x for x in labels if x.startswith("wg5.")]
    assert sum(x.startswith("wg5.rawfin") for x in labels) == 1
    for k in names:
        e5 = ((g5[1][k] - g5[0][k]).norm() / g5[0][k].norm()).item()
        print(f"wg5 vs generic on identical operands ({dtype}) {k}: rel L2 {e5:.3e}, max |g| {g5[0][k].abs().max().item():.3e}")
        assert torch.isfinite(g5[1][k]).all() and g5[0][k].abs().max() > 0 and e5 < 2e-4


@pytest.mark.gpu
@pytest.mark.parametrize("dtype", ["fp16", "bf16"])
@pytest.mark.parametrize("opt", ["bw1"])
def test_fused_dense_layer_backward_matches_separate_kernels(opt, dtype):
    """bw1.hip (data gradient + weight gradient of every dense layer's 1x1 bottleneck convolution in one pass; the same fusion for
    the 3x3 growth convolution was built, verified with this test and dropped: see DESIGN.md) against the separate kernels inside DenseNet-121, on maps whose pixel counts are not multiples of the tiles and input
    widths that are not multiples of the 128-channel slice (64 ... 1024 channels).  The weight gradients see (nearly) identical
    operands; the data gradients differ by the 16-bit rounding of gradients that are stored between layers on either path."""
    import ctypes as C
    from oracle import restatement as R
    from dmmfods_amd import _lib
    arch = _arch(R, R.DENSENETS[121], "early")
    model = _model(arch, dtype=dtype)
    model.load_state_dict(R.make_state(arch, seed=29))
    model = model.to(DEV).train()
    rgb, lidar, tgt = R.make_inputs(arch, 2, 96, 160, seed=8)
    rgb, lidar, tgt = rgb.to(DEV), lidar.to(DEV), tgt.to(DEV)
    L = _lib.lib()
    out = {}
    try:
        for on in (1, 0):
            _lib.check(L.dmm_set_option(opt.encode(), on))
            model.close()                     # the fusion is decided when the plan is built
            with torch.no_grad():
                model(rgb, lidar)
            met = model.loss_backward(tgt)
            torch.cuda.synchronize()
            plan = model._last[0]
            labels = []
            for i in range(L.dmm_plan_profile_num_ops(plan.handle, 1)):
                label, fl, by = C.c_char_p(), C.c_double(), C.c_double()
                L.dmm_plan_profile_op(plan.handle, 1, i, C.byref(label), C.byref(fl), C.byref(by))
                labels.append((label.value or b"").decode())
            out[on] = (met["loss_per_class"].clone(), {k: p.grad.detach().double().clone() for k, p in model.named_parameters()},
                       sum(1 for x in labels if x.startswith(opt + ".")))
            model._tracked_arena.zero_()
    finally:
        _lib.check(L.dmm_set_option(opt.encode(), 1))
        model.close()
    assert out[1][2] >= 58 and out[0][2] == 0, (out[1][2], out[0][2])      # every dense layer of DenseNet-121 (+ a 128-wide decoder 1x1)
    assert _rel(out[1][0], out[0][0]) < 1e-6                                # the forward pass is untouched
    g1, g0 = out[1][1], out[0][1]
    last = "features.denseblock4.denselayer16.conv%d.weight" % (1 if opt == "bw1" else 2)   # little or nothing fused runs upstream
    e_last = ((g1[last] - g0[last]).norm() / g0[last].norm()).item()
    num = sum(float((g1[k] - g0[k]).pow(2).sum()) for k in g1)
    den = sum(float(g0[k].pow(2).sum()) for k in g1)
    e_all = (num / den) ** 0.5
    print(f"{opt} vs separate kernels ({dtype}): last layer's weight gradient rel L2 {e_last:.3e}; all gradients rel L2 {e_all:.3e}")
    assert all(torch.isfinite(v).all() for v in g1.values())
    assert e_last < (3e-3 if dtype == "fp16" else 3e-2)       # bw1 measured 6.4e-4 (fp16) / 4.6e-3 (bf16)
    assert e_all < (2e-2 if dtype == "fp16" else 1e-1)        # bw1 measured 6.6e-4 (fp16) / 5.1e-3 (bf16)


@pytest.mark.gpu
@pytest.mark.parametrize("dtype,variant", [("fp32", "mid3"), ("fp16", "early")])
def test_graph_replay_matches_eager(dtype, variant):
    """Launch-list capture (hipGraph): training steps replayed from captured graphs against the same steps launched eagerly - forward
    logits bit for bit (no atomics in fp32 accumulation order... the statistics are fp64 atomics: equal to rounding), gradients to
    the order of the fp32 atomic adds - and the graphs must really be in use (replay counters), survive alternating input buffers,
    and be dropped when an option or the loss changes."""
    from oracle import restatement as R
    from dmmfods_amd import _lib
    L = _lib.lib()
    arch = _arch(R, dict(growth_rate=32, block_config=(2, 2, 2, 2), num_init_features=64) if dtype == "fp16" else TINY, variant)
    model = _model(arch, dtype=dtype)
    model.load_state_dict(R.make_state(arch, seed=3))
    model = model.to(DEV).train()
    batches = [tuple(t.to(DEV) for t in R.make_inputs(arch, 2, 64, 96, seed=s)) for s in (0, 1)]
    out = {}
    try:
        for mode in (0, 1):
            _lib.check(L.dmm_set_option(b"graph", mode))
            model.close()
            res = []
            keep = []                       # (different logits / input tensors every step: the replayed segment touches no caller pointer)
            for step in range(8):
                rgb, lidar, tgt = batches[step % 2]
                with torch.no_grad():
                    logits = model(rgb, lidar)
                met = model.loss_backward(tgt)
                keep = (keep + [logits])[-2:]
                torch.cuda.synchronize()
                res.append((logits.clone(), met["loss_per_class"].clone(), model.grad_arena.double().clone()))
                model._tracked_arena.zero_()
            plan = model._last[0]
            out[mode] = (res, L.dmm_plan_num_graph_replays(plan.handle, 0), L.dmm_plan_num_graph_replays(plan.handle, 1))
    finally:
        _lib.check(L.dmm_set_option(b"graph", 0))      # (the default)
        model.close()
    assert out[0][1] == 0 and out[0][2] == 0
    assert out[1][1] >= 4 and out[1][2] >= 4, out[1][1:]        # the later steps were replays
    for (lg0, ls0, g0), (lg1, ls1, g1) in zip(out[0][0], out[1][0]):
        assert float((lg0 - lg1).abs().max()) <= 1e-6 * float(lg0.abs().max())
        assert _rel(ls1, ls0) < 1e-6
        assert ((g1 - g0).norm() / g0.norm()).item() < (1e-5 if dtype == "fp32" else 2e-3)


def test_plan_lifetime_is_explicit_and_teardown_is_traced(capfd):
    """Round 5: GPU objects are released at a known statement.  `model.close()` destroys every plan of the
    model - dmm_plan_destroy synchronises the library's helper streams, returns its events to the process pool and reports any failing
    HIP call as DmmError instead of a swallowed exception in a garbage-collected __del__; the model stays usable (the next forward
    builds a new plan with the same results), evicting a plan from the cache and moving the model close its plans too, a closed plan
    refuses a late backward, and DMM_TRACE_DESTROY names every teardown step on stderr."""
    from oracle import restatement as R
    arch = _arch(R, TINY, "mid3")
    model = _model(arch)
    model.load_state_dict(R.make_state(arch, seed=123))
    model = model.to(DEV).train()
    rgb, lidar, tgt = (t.to(DEV) for t in R.make_inputs(arch, 2, 64, 96, seed=0))
    with torch.no_grad():
        l0 = model(rgb, lidar).clone()
    model.loss_backward(tgt)
    g0 = model.grad_arena.clone()
    plan = model._last[0]
    assert not plan.closed
    model.close()                      # no synchronize in front of it: the teardown orders itself
    assert plan.closed and model._last is None and len(model._plans) == 0
    plan.close()                       # idempotent
    with pytest.raises(RuntimeError):
        model.loss_backward(tgt)       # needs a new training forward
    with torch.no_grad():
        l1 = model(rgb, lidar).clone()
    model.loss_backward(tgt)
    assert torch.equal(l0, l1)
    assert ((model.grad_arena - g0).norm() / g0.norm()).item() < 1e-5
    # the autograd route keeps a reference to its plan: closing in between is an error at backward, not a crash
    out = model(rgb, lidar)
    model.close()
    with pytest.raises(RuntimeError, match="closed"):
        out.sum().backward()
    # the plan cache holds two plans: a third size closes the oldest
    sizes = [(64, 96), (96, 96), (64, 128)]
    plans = []
    for H, W in sizes:
        r, l, t = (x.to(DEV) for x in R.make_inputs(arch, 1, H, W, seed=1))
        with torch.no_grad():
            model(r, l)
        plans.append(model._last[0])
    assert plans[0].closed and not plans[1].closed and not plans[2].closed
    model.float()                      # _apply re-points the arenas: the plans bound to the old ones are closed
    assert all(p.closed for p in plans)
    torch.cuda.synchronize()
    # breadcrumbs (read by the library when it was loaded or at the first teardown: run in a child so the variable is seen)
    import subprocess, sys
    code = ("import torch, sys; sys.path.insert(0, %r)\n"
            "from tests.test_model_gpu import _arch, _model, TINY\n"
            "from oracle import restatement as R\n"
            "a = _arch(R, TINY, 'no'); m = _model(a).to('cuda').train()\n"
            "r, l, t = (x.to('cuda') for x in R.make_inputs(a, 1, 64, 96, seed=0))\n"
            "m(r, l); m.loss_backward(t); m.close(); print('closed ok')\n") % os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    res = subprocess.run([sys.executable, "-c", code], env=dict(os.environ, DMM_TRACE_DESTROY="1"), capture_output=True, text=True, timeout=600)
    assert res.returncode == 0 and "closed ok" in res.stdout, res.stderr[-2000:]
    for step in ("begin", "synchronise the side stream", "events back to the pool", "delete", "done"):
        assert f"[dmm] destroy: {step}" in res.stderr, (step, res.stderr[-2000:])
