"""Parity of the kernels the benchmark actually times (fp16 / bf16 storage, DenseNet-121 shapes): bw1 (fused backward of the
1x1 bottleneck convolutions), conv3 (3x3 forward / data gradient with the deferred-correction prologue), wg3 (3x3 weight gradient,
transposed form), wgp / cvp (ConvTranspose phases), wg5, thin.  Round 2's tests reached them only through self-comparison with the
generic kernels or on nets whose channel counts made them decline.

(1) per kernel, through the C ABI, against autograd of torch.nn.functional on the CPU (fp32 reference of the same op on the same
    16-bit-rounded operands): 3e-3 (fp16) / 2.5e-2 (bf16) of the tensor's max - the tolerances the forward kernels meet;
(2) the C2 and C3 networks (DenseNet-121 early / mid-3 from the reference's factories, M:335-388) in fp16 at two sizes against the
    oracle's fp16-storage emulation, with the emulation's own distance from the fp64 oracle as yardstick, asserting from the
    plan's launch labels that every timed kernel family is in the launch list;
(3) bf16 at layer depth (a two-block net whose widths engage the same kernels) so that errors do not compound through 200 BatchNorms."""
import ctypes as C

import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda"


@pytest.fixture(scope="module")
def lab():
    assert torch.cuda.is_available()
    from tools import gpu_lab
    return gpu_lab


def plan_labels(plan, lists=(0, 1)):
    from dmmfods_amd import _lib
    L = _lib.lib()
    out = []
    for which in lists:
        for i in range(L.dmm_plan_profile_num_ops(plan.handle, which)):
            label, fl, by = C.c_char_p(), C.c_double(), C.c_double()
            _lib.check(L.dmm_plan_profile_op(plan.handle, which, i, C.byref(label), C.byref(fl), C.byref(by)))
            out.append((label.value or b"").decode())
    return out


# ------------------------------------------------------------------------------------------------ (1) per kernel
@pytest.mark.parametrize("dtype", [1, 2], ids=["fp16", "bf16"])
@pytest.mark.parametrize("with_q,acc", [(1, 1), (1, 0), (0, 1), (0, 0)], ids=["q-acc", "q-assign", "mat-acc", "mat-assign"])
def test_bw1_fused_backward_of_1x1_bottleneck(lab, dtype, with_q, acc):
    """bw1_kernel<T, PQ, ACC>: every template variant; C_in below / equal to / above one 128-channel slice, a ragged last slice
    (160, 224 = DenseNet-121 block 1), several slices (512), ragged 64-pixel row tiles and a map smaller than one tile."""
    for (B, H, W, Cin) in [(2, 12, 20, 64), (1, 16, 24, 160), (2, 9, 7, 224), (1, 24, 32, 512), (1, 3, 5, 128)]:
        assert lab.backward_case(f"1x1 {Cin}->128 {B}x{H}x{W}", dtype, B, H, W, Cin, 128, 1, 1, 0, with_q=with_q, acc=acc, what="fused", expect="bw1")


@pytest.mark.parametrize("dtype", [1, 2], ids=["fp16", "bf16"])
@pytest.mark.parametrize("with_q", [1, 0], ids=["effgrad", "materialised"])
def test_dense_3x3_weight_gradient_transposed_form(lab, dtype, with_q):
    """wg3_kernel<T, PQ> (128 -> 32 channels, the only shape it takes) on ragged 8x16 tiles, one tile, many tiles."""
    for (B, H, W) in [(2, 12, 20), (1, 8, 16), (3, 5, 3), (1, 40, 60)]:
        assert lab.backward_case(f"3x3 128->32 {B}x{H}x{W}", dtype, B, H, W, 128, 32, 3, 3, 1, with_q=with_q, what="wgradT", expect="wg3")


@pytest.mark.parametrize("dtype", [1, 2], ids=["fp16", "bf16"])
@pytest.mark.parametrize("with_q,acc", [(1, 0), (1, 1), (0, 0)], ids=["effgrad", "effgrad-acc", "plain"])
def test_dense_3x3_data_gradient_with_deferred_correction(lab, dtype, with_q, acc):
    """conv3_kernel<..., EPI_BNBWD, PRO=2> (32 -> 128 channels): the production prologue that round 2's per-kernel test never hit."""
    for (B, H, W) in [(2, 12, 20), (1, 8, 16), (2, 7, 9)]:
        assert lab.backward_case(f"3x3 128->32 {B}x{H}x{W}", dtype, B, H, W, 128, 32, 3, 3, 1, with_q=with_q, acc=acc, what="dgrad", expect="conv3")


@pytest.mark.parametrize("dtype", [1, 2], ids=["fp16", "bf16"])
def test_parity_phase_weight_gradients_with_deferred_correction(lab, dtype):
    """wgp_kernel (ConvTranspose phases) fed the effective gradient, as in the plan's decoder stages."""
    for (B, H, W, Ci, Co) in [(2, 12, 20, 128, 64), (1, 9, 17, 256, 128)]:
        assert lab.backward_case(f"convT {Ci}->{Co}", dtype, B, H, W, Ci, Co, 3, 3, 1, transposed=1, with_q=1, what="wgrad", expect="wgp")


@pytest.mark.parametrize("dtype", [1, 2], ids=["fp16", "bf16"])
def test_parity_phase_weight_gradients_wave_specialised(lab, dtype):
    """Round 5: wgpw.hip - wgp.hip's tiles run by eight waves (four matrix waves on the accumulators, four loader waves with two
    register sets of inline-assembly loads, two LDS image sets, one raw barrier per tile) for the MATERIALISED output gradient, which
    is what every launch of the benchmarked plans feeds it; now also the ConvTranspose's one-tap (0, 0) phase.  Against fp32 torch
    autograd of conv_transpose2d on the same 16-bit-rounded operands: 64 and 128-column result tiles (NJ = 2 / 4), one and several
    128-channel input tiles, ragged 8 x 16 pixel tiles, fewer tiles than workgroups, an odd number of tiles per workgroup."""
    for (B, H, W, Ci, Co) in [(2, 12, 20, 128, 64), (1, 9, 17, 256, 128), (1, 8, 16, 128, 128), (3, 5, 3, 128, 64), (1, 24, 40, 256, 64)]:
        assert lab.backward_case(f"convT {Ci}->{Co} {B}x{H}x{W}", dtype, B, H, W, Ci, Co, 3, 3, 1, transposed=1, with_q=0, what="wgrad", expect="wgpw")
    # with the deferred correction on the gradient the four-wave kernel keeps the launch
    assert lab.backward_case("convT 128->64 q", dtype, 2, 12, 20, 128, 64, 3, 3, 1, transposed=1, with_q=1, what="wgrad", expect="wgp")


PRODUCTION = [  # name, B, H, W, Cin, Cout, R, stride, pad, bn, transposed, family: the launches of BASELINE configs[1] (C2) that run on LDS pipelines
    ("stem 7x7s2 8->64 @1280x1920", 4, 1280, 1920, 8, 64, 7, 2, 3, 0, 0, "conv3"),
    ("dense 3x3 128->32 @320x480", 4, 320, 480, 128, 32, 3, 1, 1, 1, 0, "cf"),      # large maps: the wave-specialised kernel (cf.hip, round 4)
    ("dense 3x3 128->32 @160x240", 4, 160, 240, 128, 32, 3, 1, 1, 1, 0, "conv3"),   # 1200 tiles: below cf.hip's threshold of eight per CU
    ("dense 3x3 128->32 @80x120", 4, 80, 120, 128, 32, 3, 1, 1, 1, 0, "conv3"),
    ("dense 1x1 224->128 @320x480", 4, 320, 480, 224, 128, 1, 1, 0, 1, 0, "pig"),
    ("dense 1x1 992->128 @80x120", 4, 80, 120, 992, 128, 1, 1, 0, 1, 0, "pig"),      # K-deep: 16 stages behind counted waits (pig.hip)
    ("dense 1x1 1024->128 @40x60", 4, 40, 60, 1024, 128, 1, 1, 0, 1, 0, "pig"),     # ... on 75 tiles
    ("convT 128->128 @320x480", 4, 320, 480, 128, 128, 3, 2, 1, 1, 1, "cvw"),       # round 5: the wave-specialised persistent form (cvw.hip: stages
    ("convT 256->256 @160x240", 4, 160, 240, 256, 256, 3, 2, 1, 1, 1, "cvw"),       # of one or two channel groups), noted beside its family cvp
    ("convT 256->256 @44x52 ragged", 2, 44, 52, 256, 256, 3, 2, 1, 1, 1, "cvw"),    # ragged tiles in both directions, fewer items than workgroups x phases
    ("convT 512->512 @80x120", 4, 80, 120, 512, 512, 3, 2, 1, 1, 1, "cvp"),         # deeper stages stay on cvp.hip (measured: DESIGN 4)
    ("convT 1024->1024 @40x60", 4, 40, 60, 1024, 1024, 3, 2, 1, 1, 1, "cvp"),
]


@pytest.mark.parametrize("dtype", [1, 2], ids=["fp16", "bf16"])
def test_forward_kernels_at_production_size_are_right_and_reproducible(lab, dtype):
    """Round 3 found the stem's 7x7 convolution (conv3.hip) returning ~0.1 % wrong outputs at 4 x 1280 x 1920 - different ones every
    run, none at parity-test sizes: a refill of the weight ring could land before a queued fragment read had executed.  Every LDS
    pipeline on the forward path is therefore also run with the chip full, three times, and must match torch and itself bit for bit."""
    for c in PRODUCTION:
        assert lab.production_forward_case(c[0], dtype, *c[1:-1], expect=c[-1])


@pytest.mark.parametrize("dtype", [1, 2], ids=["fp16", "bf16"])
def test_backward_kernels_at_production_size(lab, dtype):
    """The backward LDS kernels with the chip full (block-1 / block-3 maps of C2, batch 4): the persistent tile walks, the XCD
    grouping of bw1's channel slices and the end-of-walk atomics only take their production paths at these sizes.  Reference:
    torch autograd on the GPU, fp32, of the same op on the same 16-bit-rounded operands."""
    B = 4
    for (H, W, Cin) in [(320, 480, 224), (80, 120, 992)]:
        assert lab.backward_case(f"1x1 {Cin}->128 @{H}x{W}", dtype, B, H, W, Cin, 128, 1, 1, 0, with_q=1, acc=1, what="fused", ref_dev="cuda", expect="bw1")
    for (H, W) in [(320, 480), (80, 120)]:
        assert lab.backward_case(f"3x3 128->32 @{H}x{W}", dtype, B, H, W, 128, 32, 3, 3, 1, with_q=1, what="wgradT", ref_dev="cuda", expect="wg3")
        assert lab.backward_case(f"3x3 128->32 @{H}x{W}", dtype, B, H, W, 128, 32, 3, 3, 1, with_q=1, acc=1, what="dgrad", ref_dev="cuda", expect="conv3")
    assert lab.backward_case("convT 256->256 @160x240", dtype, B, 160, 240, 256, 256, 3, 3, 1, transposed=1, with_q=1, what="wgrad", ref_dev="cuda", expect="wgp")
    # round 5: the wave-specialised form (materialised gradient) with the chip full: the decoder's second and last stages
    assert lab.backward_case("convT 256->256 @160x240 ws", dtype, B, 160, 240, 256, 256, 3, 3, 1, transposed=1, with_q=0, what="wgrad", ref_dev="cuda", expect="wgpw")
    assert lab.backward_case("convT 128->128 @320x480 ws", dtype, B, 320, 480, 128, 128, 3, 3, 1, transposed=1, with_q=0, what="wgrad", ref_dev="cuda", expect="wgpw")
    assert lab.backward_case("convT 512->512 @80x120 ws", dtype, B, 80, 120, 512, 512, 3, 3, 1, transposed=1, with_q=0, what="wgrad", ref_dev="cuda", expect="wgpw")
    assert lab.backward_case("convT 256->256 @160x240", dtype, B, 160, 240, 256, 256, 3, 3, 1, transposed=1, with_q=0, what="dgrad", ref_dev="cuda", expect="cvp")
    # round 4: the head's kernels with the chip full - the stride-2 data gradient towards the decoder (cvd<64, true>: the
    # 3x3 over the nearest-x2 upsampled map, four sub-grids of merged taps) and the 8-channel weight gradients (wg5: the 5x5 onto the
    # classes in the transposed form, the 3x3 over the raw input in the normal form)
    assert lab.backward_case("up2 3x3 128->64 @160x240", dtype, 2, 160, 240, 128, 64, 3, 3, 1, mode=1, with_q=0, what="dgrad", ref_dev="cuda", expect="cvp")
    assert lab.backward_case("5x5 64->3 @320x480", dtype, 2, 320, 480, 64, 3, 5, 5, 2, with_q=0, what="wgradT", ref_dev="cuda", expect="wg5")


@pytest.mark.parametrize("dtype", [1, 2], ids=["fp16", "bf16"])
def test_eight_channel_and_upsampled_kernels_against_torch(lab, dtype):
    """Round 4: wg5.hip's three variants and the head's stride-2 data gradient, each through the C ABI against fp32
    torch autograd on the same 16-bit-rounded operands at 3e-3 / 2.5e-2 of the tensor's maximum, asserting the family that ran:
    the 5x5 onto the classes (transposed form, 64 -> 3), the stem's 7x7 stride 2 over the 8-channel raw input (normal form, no
    BatchNorm in front), the 3x3 over the normalised 8-channel raw input (normal form, 8 -> 64), the 3x3 over the upsampled map."""
    assert lab.backward_case("5x5 64->3", dtype, 2, 24, 40, 64, 3, 5, 5, 2, with_q=0, what="wgradT", expect="wg5")
    assert lab.backward_case("5x5 64->3 ragged", dtype, 1, 13, 21, 64, 3, 5, 5, 2, with_q=0, what="wgradT", expect="wg5")
    assert lab.conv_case("7x7s2 8->64", dtype, 1, 2, 32, 48, 8, 64, 7, 7, 2, 3, bn=0, expect_wgrad="wg5")
    assert lab.backward_case("3x3 8->64 raw", dtype, 2, 24, 40, 8, 64, 3, 3, 1, with_q=1, what="wgrad", expect="wg5")
    assert lab.backward_case("up2 3x3 128->64", dtype, 2, 12, 20, 128, 64, 3, 3, 1, mode=1, with_q=0, what="dgrad", expect="cvp")
    assert lab.backward_case("up2 3x3 128->64 ragged", dtype, 1, 9, 7, 128, 64, 3, 3, 1, mode=1, with_q=0, what="dgrad", expect="cvp")
    # (with the deferred correction on the incoming gradient this launch falls back to the generic kernel - the family assertion
    # found that in its first run; the plan materialises that gradient, so the head never runs it that way - still right:)
    assert lab.backward_case("up2 3x3 128->64 q", dtype, 1, 9, 7, 128, 64, 3, 3, 1, mode=1, with_q=1, what="dgrad", expect="generic")


# ------------------------------------------------------------------------------------------------ (2) the timed networks, fp16
def _model(arch, dtype, factory=None):
    from dmmfods_amd.graphs.models import Dense_U_Net_lidar as M
    from dmmfods_amd.utils.Dense_U_Net_lidar_helper import get_config
    cfg = get_config("/tmp/dmm_test")
    cfg.model.growth_rate, cfg.model.block_config, cfg.model.num_init_features = arch.growth_rate, arch.block_config, arch.num_init_features
    cfg.model.concat_before_block_num, cfg.model.stream_2_in_channels = arch.concat_before_block_num, arch.stream_2_in_channels
    if factory:
        return getattr(M, factory)(pretrained=False, config=cfg, compute_dtype=dtype)
    return M.Dense_U_Net_lidar(cfg, compute_dtype=dtype)


def _oracle(R, arch, B, H, W, seed, wseed, storage=None):
    P = {k: (t.double() if t.is_floating_point() else t.clone()) for k, t in R.make_state(arch, seed=wseed).items()}
    tr = R.Trainer(arch, P, storage=storage)
    rgb, lidar, tgt = R.make_inputs(arch, B, H, W, seed=seed)
    out = tr.step(rgb.double(), lidar.double(), tgt.double(), do_update=False)
    return out, {k: t.grad.clone() for k, t in tr.leaves}


def _rel(a, b):
    a, b = a.double().cpu(), b.double().cpu()
    return ((a - b).abs().max() / b.abs().max().clamp_min(1e-30)).item()


def _errors(model, logits, met, emu, g_emu, o64, g64):
    num = den = ynum = 0.0
    worst = (0.0, None)
    per = {}
    for k, p in model.named_parameters():
        got = p.grad.detach().cpu().double()
        num += float((got - g_emu[k]).pow(2).sum())
        ynum += float((g_emu[k] - g64[k]).pow(2).sum())
        den += float(g_emu[k].pow(2).sum())
        if p.dim() == 4:  # convolution weights, per tensor: rel L2 against the emulation
            e = float((got - g_emu[k]).norm() / g_emu[k].norm().clamp_min(1e-30))
            worst = max(worst, (e, k))
            per[k] = e
    return dict(per_conv=per, logits=_rel(logits.detach(), emu["logits"]), y_logits=_rel(emu["logits"], o64["logits"]),
                loss=_rel(met["loss_per_class"], emu["loss_per_class"]),
                grads=(num / den) ** 0.5, y_grads=(ynum / den) ** 0.5, worst_conv=worst)


TIMED_FAMILIES = ("bw1.", "bw1.reduce", "pig.", "conv3.store", "conv3.bnbwd", "wg3.", "wgp.", "cvp.store", "cvp.bnbwd", "wg5.", "thin.logits", "hf.store")


@pytest.mark.parametrize("variant,H,W", [("early", 64, 96), ("early", 128, 192), ("mid3", 64, 96), ("mid3", 128, 192)])
def test_c2_c3_networks_fp16_against_oracle_emulation(variant, H, W):
    """BASELINE configs[1] / configs[2]'s networks in the timed arithmetic (fp16 storage, fp32 accumulate), every timed kernel
    family in the launch list.  Bounds = measured on MI355X + margin (the measured values are printed)."""
    from oracle import restatement as R
    cbb, s2 = {"early": (1, 3), "mid3": (3, 3)}[variant]
    arch = R.densenet_arch(121, concat_before_block_num=cbb, stream_2_in_channels=s2)
    B = 2
    emu, g_emu = _oracle(R, arch, B, H, W, seed=H + cbb, wseed=2024, storage=torch.float16)
    o64, g64 = _oracle(R, arch, B, H, W, seed=H + cbb, wseed=2024)
    model = _model(arch, "fp16", factory="densenet121_u_lidar")
    model.load_state_dict(R.make_state(arch, seed=2024))
    model = model.to(DEV).train()
    rgb, lidar, tgt = R.make_inputs(arch, B, H, W, seed=H + cbb)
    logits = model(rgb.to(DEV), lidar.to(DEV))
    met = model.loss_backward(tgt.to(DEV))
    torch.cuda.synchronize()
    labels = plan_labels(model._last[0])
    for fam in TIMED_FAMILIES:
        assert any(lab.startswith(fam) for lab in labels), f"{fam} is not in the launch list: the test would not cover it"
    assert sum(lab.startswith("bw1.") for lab in labels) >= 58 and sum(lab.startswith("wg3.") for lab in labels) >= 58
    e = _errors(model, logits, met, emu, g_emu, o64, g64)
    print(f"d121 {variant} {H}x{W} fp16: logits {e['logits']:.3e} (emulation vs fp64 {e['y_logits']:.3e}), loss {e['loss']:.3e}, "
          f"grads rel L2 {e['grads']:.3e} (emulation vs fp64 {e['y_grads']:.3e}), worst conv tensor {e['worst_conv'][0]:.3e} {e['worst_conv'][1]}")
    top = sorted(e["per_conv"].items(), key=lambda kv: -kv[1])[:4]
    print("   worst conv tensors: " + ", ".join(f"{k} {v:.3f}" for k, v in top))
    other = sorted(((k, v) for k, v in e["per_conv"].items() if "denselayer" not in k), key=lambda kv: -kv[1])
    print("   outside the dense layers: " + ", ".join(f"{k} {v:.3f}" for k, v in other))
    assert torch.isfinite(logits).all() and torch.isfinite(model.grad_arena).all()
    assert e["loss"] < 5e-3, e
    assert e["logits"] < max(2e-2, 1.0 * e["y_logits"]), e      # the HIP path is closer to the emulation than the emulation to fp64
    assert e["grads"] < max(5e-2, 1.0 * e["y_grads"]), e
    # Per tensor (round 5).  16-bit storage noise ACCUMULATES on the way back from the loss: measured (MI355X,
    # this test, four cases) refine1 0.000, refine0 0.016 ... 0.039, the last ConvTranspose 0.026 ... 0.064, then +0.03 ... 0.05 per
    # decoder stage up to 0.27 ... 0.33 at the first one and 0.31 ... 0.41 in the encoder's stem / transitions; the dense layers'
    # conv1 / conv2 weights - noise-dominated on either side: two correct summation orders differ by 20 % there - 0.35 ... 0.55.  So the
    # bound is per group, 1.5 x the worst measured of the group, and SHARP on the tensors the head's and the last decoder stage's kernels
    # produce directly (hf / wgp.n64 merged, cvp_multi / cvd / wgp.n128 merged, thin / wg5): a wrong tap, phase or channel mapping in any
    # of them is an O(1) error in exactly these.
    sharp = {"dec_out_to_heat_maps.refine1.weight": 1e-3, "dec_out_to_heat_maps.refine0.weight": 0.06, "decoder.Transposed_Convolution_4.weight": 0.10}
    dense_bound = 1.5 * {("early", 64): 0.378, ("early", 128): 0.346, ("mid3", 64): 0.553, ("mid3", 128): 0.395}[(variant, H)]
    for k, v in e["per_conv"].items():
        bound = sharp.get(k) or (dense_bound if "denselayer" in k else (0.5 if k.startswith("decoder.") else 0.62))
        assert v < bound, (k, v, bound)


# ------------------------------------------------------------------------------------------------ (3) bf16 at layer depth
@pytest.mark.parametrize("dtype,storage,tol_log,tol_g", [("bf16", torch.bfloat16, 2e-2, 6e-2), ("fp16", torch.float16, 3e-3, 1e-2)])
def test_two_block_net_layer_level_16bit(dtype, storage, tol_log, tol_g):
    """One dense layer per block (growth 32, bottleneck 128, 64 stem channels: the DenseNet-121 layer shapes, so bw1 / conv3 / wg3 /
    cvp / wgp all engage) + transition + two decoder stages + head: 7 BatchNorms in series instead of 200, so the 16-bit storage
    error of each kernel is visible instead of compounded.  HIP vs the oracle's storage emulation on identical weights."""
    from oracle import restatement as R
    arch = R.Arch(growth_rate=32, block_config=(1, 1), num_init_features=64, concat_before_block_num=1, stream_2_in_channels=3)
    B, H, W = 2, 64, 96
    emu, g_emu = _oracle(R, arch, B, H, W, seed=3, wseed=11, storage=storage)
    o64, g64 = _oracle(R, arch, B, H, W, seed=3, wseed=11)
    model = _model(arch, dtype)
    model.load_state_dict(R.make_state(arch, seed=11))
    model = model.to(DEV).train()
    rgb, lidar, tgt = R.make_inputs(arch, B, H, W, seed=3)
    logits = model(rgb.to(DEV), lidar.to(DEV))
    met = model.loss_backward(tgt.to(DEV))
    torch.cuda.synchronize()
    labels = plan_labels(model._last[0])
    for fam in ("bw1.", "conv3.store", "conv3.bnbwd", "wg3.", "thin.logits"):
        assert any(lab.startswith(fam) for lab in labels), fam
    e = _errors(model, logits, met, emu, g_emu, o64, g64)
    print(f"two-block net {dtype}: logits {e['logits']:.3e} (emulation vs fp64 {e['y_logits']:.3e}), loss {e['loss']:.3e}, "
          f"grads rel L2 {e['grads']:.3e} (emulation vs fp64 {e['y_grads']:.3e}), worst conv tensor {e['worst_conv'][0]:.3e} {e['worst_conv'][1]}")
    print("   per conv tensor: " + ", ".join(f"{k} {v:.4f}" for k, v in sorted(e["per_conv"].items(), key=lambda kv: -kv[1])))
    print("   families: " + " ".join(sorted({lab.split("/")[0] for lab in labels})))
    assert e["logits"] < tol_log and e["loss"] < tol_log and e["grads"] < tol_g, e
    assert e["worst_conv"][0] < {"fp16": 0.15, "bf16": 0.4}[dtype], e["worst_conv"]   # measured 0.070 / 0.199 (block 2's conv1.weight)
    # Round 5: this net's last decoder stage (128 -> 128 channels) and head have the DenseNet-121 widths, so
    # the merged / wave-specialised launches are all in its list - hf.store (head forward), cvp.store with the four phases in one
    # launch, cvp.bnbwd, wgp.n128 (wgpw.hip, the ConvTranspose's four phases in one launch), wgp.n64 (the head's) - seven BatchNorms
    # away from the loss at most.  Per tensor against the emulation, bounds = 2 x measured (fp16 / bf16): refine1 0.0000 / 0.0003,
    # refine0 0.0141 / 0.0384, the last ConvTranspose 0.0207 / 0.0596, its conv_reduce 0.046 / 0.138.
    for fam in ("hf.store", "cvp.store", "cvp.bnbwd", "wgp.n128", "wgp.n64", "pig.store"):
        assert any(lab.startswith(fam) for lab in labels), fam
    assert sum(lab.startswith("cvp.store") for lab in labels) == 1 and sum(lab.startswith("wgp.n128") for lab in labels) == 1, "the phases are not merged"
    named = {"dec_out_to_heat_maps.refine1.weight": (1e-3, 1e-3), "dec_out_to_heat_maps.refine0.weight": (0.03, 0.08),
             "decoder.Transposed_Convolution_2.weight": (0.042, 0.12), "decoder.Transposed_Convolution_Sequence_2.conv_reduce.weight": (0.093, 0.28)}
    for k, (b16, bbf) in named.items():
        assert e["per_conv"][k] < (b16 if dtype == "fp16" else bbf), (k, e["per_conv"][k])


# ------------------------------------------------------------------------------------------------ (4) the compact effective gradient
@pytest.mark.parametrize("dtype", ["fp16", "bf16"])
def test_wg3_reads_the_compact_effective_gradient_bit_for_bit(dtype, monkeypatch):
    """Round 4: the dense 3x3 convolution's data gradient (conv3.hip) writes the effective output gradient of its tiles - prologue
    applied, 16-bit - as a compact [pixel][32] tensor and the weight gradient (wg3.hip) reads that instead of gathering and
    correcting 64 bytes per pixel from two wide buffers.  Same operand values, same summation order (per-workgroup slots added in
    slot order): with the hand-over switched off (DMM_NO_EFF_COMPACT=1, read when a plan is built) every gradient must be EQUAL."""
    from oracle import restatement as R
    arch = R.Arch(growth_rate=32, block_config=(2, 2), num_init_features=64, concat_before_block_num=1, stream_2_in_channels=3)
    B, H, W = 2, 72, 104      # ragged tiles in both directions
    model = _model(arch, dtype)
    model.load_state_dict(R.make_state(arch, seed=11))
    model = model.to(DEV).train()
    rgb, lidar, tgt = (t.to(DEV) for t in R.make_inputs(arch, B, 96, 128, seed=3))
    rgb, lidar, tgt = rgb[..., :H, :W].contiguous(), lidar[..., :H, :W].contiguous(), tgt[..., :H, :W].contiguous()
    H32, W32 = 64, 96         # the network wants multiples of 32: crop again (the tile grid of block 1 is 16 x 24 -> ragged in x)
    rgb, lidar, tgt = rgb[..., :H32, :W32].contiguous(), lidar[..., :H32, :W32].contiguous(), tgt[..., :H32, :W32].contiguous()
    grads = {}
    for off in (0, 1):
        if off:
            monkeypatch.setenv("DMM_NO_EFF_COMPACT", "1")
        else:
            monkeypatch.delenv("DMM_NO_EFF_COMPACT", raising=False)
        model.close()
        model._tracked_arena.zero_() if hasattr(model, "_tracked_arena") else None
        model(rgb, lidar)
        model.loss_backward(tgt)
        torch.cuda.synchronize()
        labels = plan_labels(model._last[0])
        assert sum(lab.startswith("wg3.") for lab in labels) == 4
        grads[off] = {k: p.grad.detach().clone() for k, p in model.named_parameters()}
    monkeypatch.delenv("DMM_NO_EFF_COMPACT", raising=False)
    model.close()
    for k in grads[0]:
        if k.endswith("conv2.weight"):    # wg3's results: no float atomics anywhere on their path
            assert torch.equal(grads[0][k], grads[1][k]), (k, float((grads[0][k] - grads[1][k]).abs().max()))
        else:                             # (other weight gradients are added with fp32 atomics: equal to their order)
            assert _rel(grads[0][k], grads[1][k]) < 2e-3, k
    assert any(float(grads[0][k].abs().max()) > 0 for k in grads[0] if k.endswith("conv2.weight"))


@pytest.mark.parametrize("dtype,tol", [("fp16", 2e-3), ("bf16", 2e-2)])
def test_two_pass_batchnorm_backward_of_the_head(dtype, tol, monkeypatch):
    """Round 4: the data gradient of the head's 5x5 convolution runs twice - reductions only, then (the correction constants of the
    BatchNorm in front of it being final) storing s*dz + q + r*x - instead of once plus an apply_corr pass over the 64-channel
    full-resolution gradient.  Against the one-pass + apply_corr schedule (DMM_NO_TWO_PASS=1): the same gradients up to ONE 16-bit
    rounding of that tensor (the old schedule rounds s*dz, then the corrected sum).  (Round 5 took the FIRST of the two passes away
    again - norm1's sums come from the 5x5 weight gradient, test_head_norm1_batchnorm_sums_from_the_5x5_weight_gradient -; this test
    pins the two-pass form itself, DMM_NO_R1_STATS=1.)"""
    from oracle import restatement as R
    arch = R.Arch(growth_rate=32, block_config=(1, 1), num_init_features=64, concat_before_block_num=1, stream_2_in_channels=3)
    model = _model(arch, dtype)
    model.load_state_dict(R.make_state(arch, seed=11))
    model = model.to(DEV).train()
    rgb, lidar, tgt = (t.to(DEV) for t in R.make_inputs(arch, 2, 64, 96, seed=3))
    grads, labels = {}, {}
    monkeypatch.setenv("DMM_NO_R1_STATS", "1")
    for off in (0, 1):
        if off:
            monkeypatch.setenv("DMM_NO_TWO_PASS", "1")
        else:
            monkeypatch.delenv("DMM_NO_TWO_PASS", raising=False)
        model.close()
        model(rgb, lidar)
        model.loss_backward(tgt)
        torch.cuda.synchronize()
        labels[off] = plan_labels(model._last[0])
        grads[off] = {k: p.grad.detach().clone() for k, p in model.named_parameters()}
    monkeypatch.delenv("DMM_NO_TWO_PASS", raising=False)
    monkeypatch.delenv("DMM_NO_R1_STATS", raising=False)
    model.close()
    n_on = sum(lab.startswith("conv3.bnbwd") and lab.endswith("h.refine1") for lab in labels[0])
    n_off = sum(lab.startswith("conv3.bnbwd") and lab.endswith("h.refine1") for lab in labels[1])
    assert (n_on, n_off) == (2, 1), (n_on, n_off)
    assert sum(lab.startswith("applycorr") for lab in labels[0]) == sum(lab.startswith("applycorr") for lab in labels[1]) - 1
    num = sum(float((grads[0][k].double() - grads[1][k].double()).pow(2).sum()) for k in grads[0])
    den = sum(float(grads[1][k].double().pow(2).sum()) for k in grads[0])
    assert (num / den) ** 0.5 < tol, (num / den) ** 0.5
    for k in grads[0]:
        assert torch.isfinite(grads[0][k]).all(), k


def test_deferred_head_and_decoder_weight_gradients(monkeypatch):
    """Round 4 experiment, kept as a switch (DMM_DEFER_WGRAD=1; measured slower, plan.cpp): the head's / decoder's multi-tap weight
    gradients enter the backward list where the main chain reaches the encoder instead of at the front.  Same launches, same
    operands: against the default order the gradients agree to the order of their fp32 atomics, the launch multiset is the same
    and the first weight-gradient launch of the head then FOLLOWS the decoder's last data gradient."""
    from oracle import restatement as R
    arch = R.Arch(growth_rate=32, block_config=(1, 1), num_init_features=64, concat_before_block_num=1, stream_2_in_channels=3)
    model = _model(arch, "fp16")
    model.load_state_dict(R.make_state(arch, seed=11))
    model = model.to(DEV).train()
    rgb, lidar, tgt = (t.to(DEV) for t in R.make_inputs(arch, 2, 64, 96, seed=3))
    grads, labels = {}, {}
    monkeypatch.setenv("DMM_NO_WGP_MERGE", "1")     # (held-back phase launches are not merged into one: compare like with like)
    monkeypatch.setenv("DMM_NO_RAW_STATS", "1")     # (nor does a held-back raw-segment weight gradient feed the norm's sums: round 5)
    monkeypatch.setenv("DMM_NO_R1_STATS", "1")      # (nor the held-back 5x5 weight gradient norm1's: the data-gradient chain would wait for it)
    for off in (0, 1):
        if off:
            monkeypatch.delenv("DMM_DEFER_WGRAD", raising=False)
        else:
            monkeypatch.setenv("DMM_DEFER_WGRAD", "1")
        model.close()
        model(rgb, lidar)
        model.loss_backward(tgt)
        torch.cuda.synchronize()
        labels[off] = plan_labels(model._last[0])
        grads[off] = {k: p.grad.detach().clone() for k, p in model.named_parameters()}
    monkeypatch.delenv("DMM_DEFER_WGRAD", raising=False)
    monkeypatch.delenv("DMM_NO_WGP_MERGE", raising=False)
    monkeypatch.delenv("DMM_NO_RAW_STATS", raising=False)
    monkeypatch.delenv("DMM_NO_R1_STATS", raising=False)
    model.close()
    assert sorted(labels[0]) == sorted(labels[1])
    first_head_w = {o: next(i for i, lab in enumerate(labels[o]) if lab.startswith(("wgp.", "wg5.")) and "/h." in lab) for o in (0, 1)}
    last_dec_dgrad = {o: max(i for i, lab in enumerate(labels[o]) if ".bnbwd" in lab and "/d." in lab) for o in (0, 1)}
    assert first_head_w[0] > last_dec_dgrad[0] and first_head_w[1] < last_dec_dgrad[1], (first_head_w, last_dec_dgrad)
    for k in grads[0]:
        assert _rel(grads[0][k], grads[1][k]) < 2e-3, k


@pytest.mark.parametrize("dtype,tol", [("fp16", 1e-3), ("bf16", 1e-3)])
def test_head_weight_gradient_phases_in_one_launch(dtype, tol, monkeypatch):
    """Round 4: the four output-parity phases of the head's 3x3 weight gradient (wgp.hip) run as ONE launch - phase p of a tile range
    beside the other phases of that range on one XCD, so the half-resolution input is fetched from HBM once.  Same arithmetic per
    phase: against four launches (DMM_NO_WGP_MERGE=1) the weight gradient agrees to the order of the fp32 atomics."""
    from oracle import restatement as R
    arch = R.Arch(growth_rate=32, block_config=(1, 1), num_init_features=64, concat_before_block_num=1, stream_2_in_channels=3)
    model = _model(arch, dtype)
    model.load_state_dict(R.make_state(arch, seed=11))
    model = model.to(DEV).train()
    rgb, lidar, tgt = (t.to(DEV) for t in R.make_inputs(arch, 2, 96, 160, seed=3))    # 48 x 80 phase grid: 6 x 5 tiles, ragged in neither
    grads, nl = {}, {}
    for off in (0, 1):
        if off:
            monkeypatch.setenv("DMM_NO_WGP_MERGE", "1")
        else:
            monkeypatch.delenv("DMM_NO_WGP_MERGE", raising=False)
        model.close()
        model(rgb, lidar)
        model.loss_backward(tgt)
        torch.cuda.synchronize()
        labels = plan_labels(model._last[0])
        nl[off] = sum(lab.startswith("wgp.") and lab.endswith("h.refine0") for lab in labels)
        grads[off] = {k: p.grad.detach().clone() for k, p in model.named_parameters()}
    monkeypatch.delenv("DMM_NO_WGP_MERGE", raising=False)
    model.close()
    assert (nl[0], nl[1]) == (1, 4), nl
    k = "dec_out_to_heat_maps.refine0.weight"
    assert float(grads[1][k].abs().max()) > 0
    assert _rel(grads[0][k], grads[1][k]) < tol, _rel(grads[0][k], grads[1][k])
    for k2 in grads[0]:
        assert _rel(grads[0][k2], grads[1][k2]) < 2e-3, k2


@pytest.mark.parametrize("dtype,tol", [("fp16", 2e-3), ("bf16", 1.5e-2)])
def test_raw_input_batchnorm_sums_from_the_weight_gradient(dtype, tol, monkeypatch):
    """Round 5: the head's first convolution reads the raw input behind norm0 (reference M:123-127), so backward needs sum dz and
    sum dz * xhat over the full-resolution map for those (<= 8) channels - for nothing but their gamma / beta gradients.  Round 4 ran
    the 3x3 data gradient towards the raw input for that (one more pass over the 64-channel full-resolution gradient).  Now the
    weight-gradient launch of that segment (wg5.hip) correlates the gradient with the two factors of the activation,
    relu(bn(x)) = gamma (m xhat) + beta m, and a one-workgroup launch derives the packed weight gradient AND both sums from the result
    (sum dz = sum W S1, sum dz xhat = sum W S2).  Against the data-gradient path (DMM_NO_RAW_STATS=1): norm0's gamma / beta gradients
    on the raw-input channels, refine0's weight gradient on its raw-input segment, and every other gradient; the launch list holds
    the finish launch and no data gradient towards the raw input."""
    from oracle import restatement as R
    arch = R.densenet_arch(121, concat_before_block_num=1, stream_2_in_channels=3)
    model = _model(arch, dtype)
    model.load_state_dict(R.make_state(arch, seed=23))
    model = model.to(DEV).train()
    rgb, lidar, tgt = (t.to(DEV) for t in R.make_inputs(arch, 2, 96, 160, seed=8))
    grads, labs = {}, {}
    for off in (0, 1):
        if off:
            monkeypatch.setenv("DMM_NO_RAW_STATS", "1")
        else:
            monkeypatch.delenv("DMM_NO_RAW_STATS", raising=False)
        model.close()
        model(rgb, lidar)
        model.loss_backward(tgt)
        torch.cuda.synchronize()
        labs[off] = plan_labels(model._last[0], lists=(1,))
        grads[off] = {k: p.grad.detach().double().cpu().clone() for k, p in model.named_parameters()}
    monkeypatch.delenv("DMM_NO_RAW_STATS", raising=False)
    model.close()
    assert sum(lab.startswith("wg5.rawfin") for lab in labs[0]) == 1 and not any(lab.startswith("wg5.rawfin") for lab in labs[1])
    raw_dgrad = [lab for lab in labs[1] if ".bnbwd.n32/h.refine0" in lab]
    assert len(raw_dgrad) == 1 and not any(".bnbwd.n32/h.refine0" in lab for lab in labs[0]), (raw_dgrad, [l for l in labs[0] if "refine0" in l])
    nfl = 128
    for k in ("dec_out_to_heat_maps.norm0.weight", "dec_out_to_heat_maps.norm0.bias"):
        a, b = grads[0][k][nfl:], grads[1][k][nfl:]
        assert float(b.abs().max()) > 0 and a.numel() == 6
        assert float((a - b).abs().max() / b.abs().max()) < tol, (k, a, b)
    a, b = grads[0]["dec_out_to_heat_maps.refine0.weight"][:, nfl:], grads[1]["dec_out_to_heat_maps.refine0.weight"][:, nfl:]
    assert float((a - b).norm() / b.norm()) < tol, float((a - b).norm() / b.norm())
    for k in grads[0]:
        assert _rel(grads[0][k], grads[1][k]) < max(tol, 2e-3), k


@pytest.mark.parametrize("dtype", [1, 2])
def test_conv5_weight_gradient_and_batchnorm_sums_in_one_pass(dtype):
    """Round 5: dmm_conv5_wgrad_stats (wg5.hip PA = 3 + wg5_fin64_kernel) against fp64 torch on the same 16-bit operands - the weight
    gradient of the head's 5x5 convolution and norm1's two BatchNorm-backward sums, the latter to 1e-6 of the largest channel sum
    (fp32 accumulation per workgroup, fp64 across workgroups; the sums sit on the data-gradient chain).  Shapes: several tiles per
    workgroup, a ragged picture, four classes."""
    from tools import gpu_lab as lab
    assert lab.conv5_stats_case("5x5 64->3", dtype, 2, 24, 40)
    assert lab.conv5_stats_case("5x5 64->3 ragged", dtype, 1, 13, 21, seed=1)
    assert lab.conv5_stats_case("5x5 64->4 @96x160", dtype, 2, 96, 160, Cout=4, seed=2)
    assert lab.conv5_stats_case("5x5 64->3 @320x480", dtype, 2, 320, 480, seed=3, ref_dev="cuda")


@pytest.mark.parametrize("dtype,tol_n,tol_w,tol_all", [("fp16", 5e-5, 2e-4, 3e-3), ("bf16", 5e-4, 2e-3, 2e-2)])
def test_head_norm1_batchnorm_sums_from_the_5x5_weight_gradient(dtype, tol_n, tol_w, tol_all, monkeypatch):
    """Round 5: the head's last convolution (5x5, 64 -> classes; reference M:128-131) sits behind norm1 + ReLU, so its backward needs
    sum dz and sum dz * xhat of dz = m * (W^T dY) over the full-resolution map BEFORE the gradient towards refine0 can be stored - the
    two-pass data gradient ran a reductions-only first pass over the 64-channel activation for them.  The weight gradient of the same
    convolution reads the same activation and the same logits gradient: with the activation entered as its two factors,
    relu(bn(x)) = scale (m x) + shift m, the launch (wg5.hip, PA = 3) yields S2 = corr(m x, dY) and S1 = corr(m, dY), and a small
    launch derives dW = scale S2 + shift S1 AND both sums (sum dz = sum W S1, sum dz xhat = (sum W S2 - mean sum W S1) invstd).
    Against the two-pass path (DMM_NO_R1_STATS=1): the launch list holds the finish launch and ONE 5x5 data-gradient pass instead of
    two; norm1's gamma / beta gradients (= the two sums) and refine1's weight gradient agree to 3e-6 / 2e-5 in f16 (measured) - the
    distance is the OLD path's: its epilogue stages dz in 16 bits before summing (3e-6 of a sum in f16, 3e-5 in bf16), while the new
    sums are within 4e-8 of fp64 (test_conv5_weight_gradient_and_batchnorm_sums_in_one_pass).  Everything else in the network sits
    behind norm1's backward, and DenseNet-121 amplifies a per-channel offset of the head's gradient ~1e3-fold on the way to conv0:
    global relative L2 over all gradients 6e-4 (f16) / 5e-3 (bf16) measured; both paths are bit-reproducible run to run
    (tools/probes/r05_r1stats.py)."""
    from oracle import restatement as R
    arch = R.densenet_arch(121, concat_before_block_num=1, stream_2_in_channels=3)
    model = _model(arch, dtype)
    model.load_state_dict(R.make_state(arch, seed=29))
    model = model.to(DEV).train()
    rgb, lidar, tgt = (t.to(DEV) for t in R.make_inputs(arch, 2, 96, 160, seed=5))
    grads, labs = {}, {}
    for off in (0, 1):
        if off:
            monkeypatch.setenv("DMM_NO_R1_STATS", "1")
        else:
            monkeypatch.delenv("DMM_NO_R1_STATS", raising=False)
        model.close()
        model(rgb, lidar)
        model.loss_backward(tgt)
        torch.cuda.synchronize()
        labs[off] = plan_labels(model._last[0], lists=(1,))
        grads[off] = {k: p.grad.detach().double().cpu().clone() for k, p in model.named_parameters()}
    monkeypatch.delenv("DMM_NO_R1_STATS", raising=False)
    model.close()
    assert sum(lab.startswith("wg5.fin64") for lab in labs[0]) == 1 and not any(lab.startswith("wg5.fin64") for lab in labs[1])
    passes = [sum("conv3.bnbwd.n64/h.refine1" in lab for lab in labs[o]) for o in (0, 1)]
    assert passes == [1, 2], (passes, [l for l in labs[0] if "refine1" in l])
    for k, tol in (("dec_out_to_heat_maps.norm1.weight", tol_n), ("dec_out_to_heat_maps.norm1.bias", tol_n),
                   ("dec_out_to_heat_maps.refine1.weight", tol_w)):
        a, b = grads[0][k], grads[1][k]
        assert float(b.abs().max()) > 0
        assert float((a - b).norm() / b.norm()) < tol, (k, float((a - b).norm() / b.norm()))
    num = sum(float((grads[0][k] - grads[1][k]).pow(2).sum()) for k in grads[0])
    den = sum(float(grads[1][k].pow(2).sum()) for k in grads[0])
    assert (num / den) ** 0.5 < tol_all, (num / den) ** 0.5
    assert all(torch.isfinite(v).all() for v in grads[0].values())


@pytest.mark.parametrize("dtype", ["fp16", "bf16"])
def test_decoder_weight_gradient_phases_in_one_launch(dtype, monkeypatch):
    """Round 5: the four output-parity phases of a decoder ConvTranspose's weight gradient (1, 2, 2 and 4 taps) run as ONE launch of the
    wave-specialised kernel (wgpw.hip: the phase's tap count is taken per workgroup, 64 output channels per workgroup in every phase,
    the four phases of a tile range side by side on one XCD).  Round 4 ran them as three wgp launches and one generic launch, each
    sweeping the stage's input.  densenet121 widths (1024 / 512 / 256 / 128 decoder channels: several input and output channel tiles).
    Against four launches (DMM_NO_WGP_MERGE=1) every ConvTranspose weight gradient agrees to the order of the fp32 atomics, and so
    does every other gradient; the merged list holds ONE wgp launch per decoder stage (and the head's)."""
    from oracle import restatement as R
    arch = R.densenet_arch(121, concat_before_block_num=1, stream_2_in_channels=3)
    model = _model(arch, dtype)
    model.load_state_dict(R.make_state(arch, seed=14))
    model = model.to(DEV).train()
    rgb, lidar, tgt = (t.to(DEV) for t in R.make_inputs(arch, 2, 96, 160, seed=5))
    grads, nl = {}, {}
    for off in (0, 1):
        if off:
            monkeypatch.setenv("DMM_NO_WGP_MERGE", "1")
        else:
            monkeypatch.delenv("DMM_NO_WGP_MERGE", raising=False)
        model.close()
        model(rgb, lidar)
        model.loss_backward(tgt)
        torch.cuda.synchronize()
        labels = plan_labels(model._last[0], lists=(1,))
        nl[off] = sum(lab.startswith("wgp.") and "/d.TC_" in lab for lab in labels)
        grads[off] = {k: p.grad.detach().clone() for k, p in model.named_parameters()}
    monkeypatch.delenv("DMM_NO_WGP_MERGE", raising=False)
    model.close()
    assert (nl[0], nl[1]) == (4, 16), nl
    for j in range(1, 5):
        k = f"decoder.Transposed_Convolution_{j}.weight"
        assert float(grads[1][k].abs().max()) > 0
        assert _rel(grads[0][k], grads[1][k]) < 1e-3, (k, _rel(grads[0][k], grads[1][k]))
    for k2 in grads[0]:
        assert _rel(grads[0][k2], grads[1][k2]) < 2e-3, k2


@pytest.mark.parametrize("dtype", ["fp16", "bf16"])
def test_head_forward_phases_in_one_launch(dtype, monkeypatch):
    """Round 4: the four output-parity phases of the head's first convolution (conv3.hip: 2x2 merged taps over the half-resolution
    decoder output + 3x3 stride-2 taps over the raw input) run as ONE launch that walks (tile, phase) pairs.  Same arithmetic per
    phase, no float atomics on stored values: against four launches (DMM_NO_C3_MERGE=1) the logits must be EQUAL bit for bit.
    (DMM_NO_HF=1: where hf.hip takes the launch - 128 decoder channels - conv3.hip's path is what is compared here.)"""
    monkeypatch.setenv("DMM_NO_HF", "1")
    from oracle import restatement as R
    arch = R.Arch(growth_rate=32, block_config=(1, 1), num_init_features=64, concat_before_block_num=1, stream_2_in_channels=3)
    model = _model(arch, dtype)
    model.load_state_dict(R.make_state(arch, seed=11))
    model = model.to(DEV).train()
    rgb, lidar, tgt = (t.to(DEV) for t in R.make_inputs(arch, 2, 96, 160, seed=3))
    outs, nl = {}, {}
    for off in (0, 1):
        if off:
            monkeypatch.setenv("DMM_NO_C3_MERGE", "1")
        else:
            monkeypatch.delenv("DMM_NO_C3_MERGE", raising=False)
        model.close()
        with torch.no_grad():
            outs[off] = model(rgb, lidar).clone()
        torch.cuda.synchronize()
        labels = plan_labels(model._last[0])
        nl[off] = sum(lab.startswith("conv3.store") and lab.endswith("h.refine0") for lab in labels)
    monkeypatch.delenv("DMM_NO_C3_MERGE", raising=False)
    model.close()
    assert (nl[0], nl[1]) == (1, 4), nl
    assert torch.isfinite(outs[0]).all() and float(outs[0].abs().max()) > 0
    assert torch.equal(outs[0], outs[1]), float((outs[0] - outs[1]).abs().max())


def test_mid_fusion_forward_interleaves_the_encoders_and_packs_late_weights_aside(monkeypatch):
    """Round 4: with mid fusion the two encoders' forward records are emitted alternately (the second stream's as one chain on the side
    stream), the weights of the early layers are packed on the launch stream and the rest by a launch on a stream of its own, joined
    in front of the first layer that needs them (plan.cpp emit_forward_records / pack_cut_rec).  Against the round-3 order - one
    encoder after the other, everything but the stem's weights packed behind a join after the stem - the logits must be EQUAL and
    the gradients equal up to the order of the fp32 atomics: a missing dependency between the three streams would show here."""
    from oracle import restatement as R
    arch = R.Arch(growth_rate=32, block_config=(2, 2, 2), num_init_features=64, concat_before_block_num=2, stream_2_in_channels=3)
    model = _model(arch, "fp16")
    model.load_state_dict(R.make_state(arch, seed=5))
    model = model.to(DEV).train()
    rgb, lidar, tgt = (t.to(DEV) for t in R.make_inputs(arch, 2, 128, 192, seed=9))
    res, labels = {}, {}
    for old in (0, 1, 0):
        if old:
            monkeypatch.setenv("DMM_PACK_CUT", "1")
            monkeypatch.setenv("DMM_NO_S2_INTERLEAVE", "1")
        else:
            monkeypatch.setenv("DMM_PACK_CUT", "16")   # (the test net is far below the element count that places the cut by itself)
            monkeypatch.delenv("DMM_NO_S2_INTERLEAVE", raising=False)
        model.close()
        for _ in range(2):   # the second pass runs with the streams and events of the first already made
            with torch.no_grad():
                logits = model(rgb, lidar).clone()
            model.loss_backward(tgt)
        torch.cuda.synchronize()
        res.setdefault(old, []).append((logits, model.grad_arena.clone().double()))
        labels[old] = plan_labels(model._last[0], lists=(0,))   # the training forward list
    monkeypatch.delenv("DMM_PACK_CUT", raising=False)
    monkeypatch.delenv("DMM_NO_S2_INTERLEAVE", raising=False)
    model.close()
    new, oldl = labels[0], labels[1]
    assert sorted(l for l in new if not l.startswith(("other", "pack"))) == sorted(l for l in oldl if not l.startswith(("other", "pack")))
    s1 = [i for i, l in enumerate(new) if "/f.b1." in l]
    s2 = [i for i, l in enumerate(new) if "/stream_2_f.b1." in l]
    assert s1 and s2 and min(s1) < max(s2) and min(s2) < max(s1), "the encoders' records are not interleaved"
    o1 = [i for i, l in enumerate(oldl) if "/f.b1." in l]
    o2 = [i for i, l in enumerate(oldl) if "/stream_2_f.b1." in l]
    assert max(o2) < min(o1)
    jp = [i for i, l in enumerate(new) if l.endswith("join.pack")]
    assert len(jp) == 1 and jp[0] > max(s1 + s2), (jp, max(s1 + s2))
    (l0, g0), (l0b, g0b) = res[0]
    (l1, g1), = res[1]
    assert torch.isfinite(l0).all() and float(l0.abs().max()) > 0
    assert torch.equal(l0, l1) and torch.equal(l0, l0b), float((l0 - l1).abs().max())
    for a, b in ((g0, g1), (g0, g0b)):
        assert ((a - b).norm() / b.norm()).item() < 2e-3


@pytest.mark.parametrize("dtype", ["fp16", "bf16"])
def test_pack_and_unpack_tile_kernels_against_the_generic_kernels(dtype, monkeypatch):
    """Round 4: weights are packed (and packed gradients scattered back) by tile kernels that turn 32 x 32 x taps blocks of a master
    tensor in LDS (pointwise.hip pack_tiles_kernel / unpack_tiles_kernel: 1x1 and 3x3 tensors, both contiguity classes, the parity
    phases of a ConvTranspose as one group) instead of one gathered element per load.  Packing is a permutation + conversion, so the
    logits must be EQUAL to those of the generic kernels (DMM_NO_PACK_TILES=1); every gradient tensor must agree to the noise of the
    fp32 atomics of the weight-gradient kernels - a wrong tap or channel mapping is an O(1) difference in that tensor."""
    from oracle import restatement as R
    arch = R.Arch(growth_rate=32, block_config=(2, 2, 2), num_init_features=64, concat_before_block_num=2, stream_2_in_channels=3)
    model = _model(arch, dtype)
    model.load_state_dict(R.make_state(arch, seed=21))
    model = model.to(DEV).train()
    rgb, lidar, tgt = (t.to(DEV) for t in R.make_inputs(arch, 2, 96, 160, seed=4))
    res = {}
    for generic in (0, 1):
        if generic:
            monkeypatch.setenv("DMM_NO_PACK_TILES", "1")
        else:
            monkeypatch.delenv("DMM_NO_PACK_TILES", raising=False)
        model.close()
        with torch.no_grad():
            logits = model(rgb, lidar).clone()
        model.loss_backward(tgt)
        torch.cuda.synchronize()
        res[generic] = (logits, {n: p.grad.detach().double().clone() for n, p in model.named_parameters()})
    monkeypatch.delenv("DMM_NO_PACK_TILES", raising=False)
    model.close()
    (l0, g0), (l1, g1) = res[0], res[1]
    assert torch.isfinite(l0).all() and float(l0.abs().max()) > 0
    assert torch.equal(l0, l1), float((l0 - l1).abs().max())
    worst = max(((g0[n] - g1[n]).norm() / (g1[n].norm() + 1e-30)).item() for n in g0)
    assert all(float(g1[n].abs().max()) > 0 for n in g1)
    assert worst < 2e-3, worst


def test_decoder_transposed_convolution_phases_in_one_launch(monkeypatch):
    """Round 4: the four output-parity phases (1, 2, 2, 4 taps) of each decoder ConvTranspose run as ONE cvp.hip launch that deals the
    workgroups of all phases, most taps first (separate launches of 600 / 1200 workgroups left the last round on the chip's 512 slots
    mostly empty).  Same arithmetic per phase, no float atomics on stored values: against sixteen launches (DMM_NO_CVP_MERGE=1) the
    logits must be EQUAL bit for bit.  densenet121 widths (the decoder's 1024 / 512 / 256 / 128 channels are what cvp.hip takes)."""
    from oracle import restatement as R
    arch = R.densenet_arch(121, concat_before_block_num=2, stream_2_in_channels=3)
    model = _model(arch, "fp16")
    model.load_state_dict(R.make_state(arch, seed=2))
    model = model.to(DEV).train()
    rgb, lidar, tgt = (t.to(DEV) for t in R.make_inputs(arch, 2, 64, 96, seed=6))
    outs, nl = {}, {}
    for off in (0, 1):
        if off:
            monkeypatch.setenv("DMM_NO_CVP_MERGE", "1")
        else:
            monkeypatch.delenv("DMM_NO_CVP_MERGE", raising=False)
        model.close()
        with torch.no_grad():
            outs[off] = model(rgb, lidar).clone()
        torch.cuda.synchronize()
        labels = plan_labels(model._last[0], lists=(0,))
        nl[off] = sum(lab.startswith("cvp.store") for lab in labels)
    monkeypatch.delenv("DMM_NO_CVP_MERGE", raising=False)
    model.close()
    assert (nl[0], nl[1]) == (4, 16), nl
    assert torch.isfinite(outs[0]).all() and float(outs[0].abs().max()) > 0
    assert torch.equal(outs[0], outs[1]), float((outs[0] - outs[1]).abs().max())


@pytest.mark.parametrize("dtype", ["fp16", "bf16"])
def test_head_first_convolution_wave_specialised_kernel(dtype, monkeypatch):
    """Round 4: hf.hip - the head's first convolution (reference M:126-127) with one parity phase per 8-wave workgroup, the phase's
    weights in the matrix waves' registers, loader waves filling the next image set beside them.  densenet121 widths (128 decoder
    channels in front of the head: the shape hf.hip takes).  The K walk is conv3.hip's (tap, chunk; then the raw taps) and nothing
    stored goes through a float atomic: against conv3.hip's one-launch path (DMM_NO_HF=1) the logits must be EQUAL bit for bit, as
    must two runs of hf.hip; the launch list names the family."""
    from oracle import restatement as R
    arch = R.densenet_arch(121, concat_before_block_num=1, stream_2_in_channels=3)
    model = _model(arch, dtype)
    model.load_state_dict(R.make_state(arch, seed=8))
    model = model.to(DEV).train()
    rgb, lidar, tgt = (t.to(DEV) for t in R.make_inputs(arch, 2, 64, 96, seed=12))
    outs, nhf, nc3 = {}, {}, {}
    for k, nohf in (("hf", 0), ("conv3", 1), ("hf2", 0)):
        if nohf:
            monkeypatch.setenv("DMM_NO_HF", "1")
        else:
            monkeypatch.delenv("DMM_NO_HF", raising=False)
        model.close()
        with torch.no_grad():
            outs[k] = model(rgb, lidar).clone()
        torch.cuda.synchronize()
        labels = plan_labels(model._last[0], lists=(0,))
        nhf[k] = sum(lab.startswith("hf.store") and lab.endswith("h.refine0") for lab in labels)
        nc3[k] = sum(lab.startswith("conv3.store") and lab.endswith("h.refine0") for lab in labels)
    monkeypatch.delenv("DMM_NO_HF", raising=False)
    model.close()
    assert (nhf["hf"], nc3["hf"], nhf["conv3"], nc3["conv3"]) == (1, 0, 0, 1), (nhf, nc3)
    assert torch.isfinite(outs["hf"]).all() and float(outs["hf"].abs().max()) > 0
    assert torch.equal(outs["hf"], outs["hf2"]), "hf.hip is not reproducible"
    assert torch.equal(outs["hf"], outs["conv3"]), float((outs["hf"] - outs["conv3"]).abs().max())
