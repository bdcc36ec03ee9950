#!/usr/bin/env python3
"""Bring-up lab: runs every kernel family against a torch reference and the tiny model against the oracle,
printing errors for ALL cases (never stops at the first failure).  Usage on the GPU box:
    python tools/gpu_lab.py [kernels] [model] > gpurun_out/lab.log
"""
import ctypes as C
import os
import sys
import time
import traceback

import torch
import torch.nn.functional as F

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from dmmfods_amd import _lib  # noqa: E402

DEV = "cuda"
L = _lib.lib()


def nhwc(x, dt):
    return x.permute(0, 2, 3, 1).contiguous().to(dt)


def nchw(x):
    return x.permute(0, 3, 1, 2).contiguous().float()


def relerr(a, b):
    a, b = a.double(), b.double()
    return ((a - b).abs().max() / b.abs().max().clamp_min(1e-30)).item()


def _check_family(res_name, expect):
    """Did the family the case names really run?  (The single-kernel entry points fall back to the generic kernels silently.)"""
    if expect is None:
        return True
    ran = _lib.impls_since_reset()
    ok = expect in ran
    if not ok:
        print(f"FAIL {res_name}: expected kernel family {expect!r}, ran {sorted(ran)}", flush=True)
    return ok


def conv_case(name, dtype, mfma, B, H, W, Cin, Cout, R, S, stride, pad, transposed=0, mode=0, bn=1, seed=0, expect_wgrad=None):
    g = torch.Generator().manual_seed(seed)
    dt = {0: torch.float32, 1: torch.float16, 2: torch.bfloat16}[dtype]
    x = (torch.randn(B, Cin, H, W, generator=g) * 2 + 0.5)
    scale = torch.rand(Cin, generator=g) + 0.5
    shift = torch.randn(Cin, generator=g) * 0.5
    wshape = (Cin, Cout, 3, 3) if transposed else (Cout, Cin, R, S)
    w = torch.randn(wshape, generator=g) / (Cin * R * S) ** 0.5
    xq = x.to(dt).float()
    a = F.relu(xq * scale.view(1, -1, 1, 1) + shift.view(1, -1, 1, 1)) if bn else xq
    a = a.requires_grad_(True)
    wq = w.clone().requires_grad_(True)
    if transposed:
        y = F.conv_transpose2d(a, wq, stride=2, padding=1, output_padding=1)
    elif mode == 1:
        y = F.conv2d(F.interpolate(a, scale_factor=2, mode="nearest"), wq, padding=pad)
    elif mode == 2:
        y = F.avg_pool2d(F.conv2d(a, wq), 2, 2)
    else:
        y = F.conv2d(a, wq, stride=stride, padding=pad)
    dy = torch.randn(y.shape, generator=g)
    dyq = dy.to(dt).float()
    (y * dyq).sum().backward()
    d = _lib.ConvDesc(dtype=dtype, use_mfma=mfma, B=B, H=H, W=W, Cin=Cin, Cout=Cout, R=R, S=S, stride=stride, pad=pad,
                      transposed=transposed, mode=mode, bn_relu=bn)
    nsc = L.dmm_conv_scratch_bytes(C.byref(d))
    scratch = torch.zeros(nsc, dtype=torch.uint8, device=DEV)
    xd = nhwc(x, dt).to(DEV)
    # dgrad entry point wants [shift | mean | invstd]; any mean/invstd define xhat for the second reduction
    mean = torch.randn(Cin, generator=g)
    invstd = torch.rand(Cin, generator=g) + 0.5
    wd, sd, hd = w.to(DEV), scale.to(DEV), torch.cat([shift, mean, invstd]).to(DEV)
    yd = torch.full((B, y.shape[2], y.shape[3], Cout), float("nan"), dtype=dt, device=DEV)
    stats = torch.zeros(2 * Cout, dtype=torch.float64, device=DEV)
    st = _lib.stream_ptr()
    res = {}
    _lib.check(L.dmm_conv_forward(C.byref(d), xd.data_ptr(), wd.data_ptr(), sd.data_ptr(), hd.data_ptr(), yd.data_ptr(),
                                  stats.data_ptr(), scratch.data_ptr(), st))
    torch.cuda.synchronize()
    yo = nchw(yd).cpu()
    res["fwd"] = relerr(yo, y.detach())
    res["sum"] = relerr(stats[:Cout].cpu(), yo.double().sum(dim=(0, 2, 3)))
    res["sq"] = relerr(stats[Cout:].cpu(), (yo.double() ** 2).sum(dim=(0, 2, 3)))
    # wgrad
    dyd = nhwc(dy, dt).to(DEV)
    dwd = torch.full(wshape, float("nan"), device=DEV)
    _lib.impls_since_reset()
    _lib.check(L.dmm_conv_wgrad(C.byref(d), xd.data_ptr(), dyd.data_ptr(), sd.data_ptr(), hd.data_ptr(), dwd.data_ptr(),
                                scratch.data_ptr(), st))
    torch.cuda.synchronize()
    fam_ok = _check_family(name + " wgrad", expect_wgrad)
    res["wgrad"] = relerr(dwd.cpu(), wq.grad)
    # dgrad (BN fused)
    if bn and not (mode == 0 and not transposed and stride != 1):
        z = xq * scale.view(1, -1, 1, 1) + shift.view(1, -1, 1, 1)
        dz = a.grad * (z > 0)
        gx_ref = dz * scale.view(1, -1, 1, 1)
        gxd = torch.full((B, H, W, Cin), float("nan"), dtype=dt, device=DEV)
        red = torch.zeros(2 * Cin, dtype=torch.float64, device=DEV)
        _lib.check(L.dmm_conv_dgrad(C.byref(d), xd.data_ptr(), dyd.data_ptr(), wd.data_ptr(), sd.data_ptr(), hd.data_ptr(),
                                    gxd.data_ptr(), red.data_ptr(), scratch.data_ptr(), st))
        torch.cuda.synchronize()
        res["dgrad"] = relerr(nchw(gxd).cpu(), gx_ref)
        res["red1"] = relerr(red[:Cin].cpu(), dz.double().sum(dim=(0, 2, 3)))
        xhat = (xq.double() - mean.double().view(1, -1, 1, 1)) * invstd.double().view(1, -1, 1, 1)
        res["red2"] = relerr(red[Cin:].cpu(), (dz.double() * xhat).sum(dim=(0, 2, 3)))
    tol = {0: 2e-5, 1: 3e-3, 2: 2.5e-2}[dtype]   # relative to the tensor's max: fp32 / f16 (11 bits) / bf16 (8 bits) storage
    bad = [k for k, v in res.items() if not (v < tol)]
    print(f"{'FAIL' if bad else 'ok  '} {name:28s} dt={dtype} mfma={mfma} " + " ".join(f"{k}={v:.2e}" for k, v in res.items()), flush=True)
    return not bad and fam_ok


def _eff_setup(g, dt, B, Cout, Ho, Wo, with_q):
    """Incoming gradient of a convolution output as the plan's launches see it: raw gradient dy, and optionally the deferred
    BatchNorm-backward correction q + r * yfwd of the layer behind it (yfwd = that layer's input = this convolution's output)."""
    kw = dict(generator=g, device=g.device)
    dy = torch.randn(B, Cout, Ho, Wo, **kw)
    dyq = dy.to(dt).float()
    if not with_q:
        return dy, dyq, None, None, None, dyq
    yf = torch.randn(B, Cout, Ho, Wo, **kw) * 1.5
    yfq = yf.to(dt).float()
    q = torch.randn(Cout, **kw) * 0.3
    r = torch.randn(Cout, **kw) * 0.2
    eff = dyq + q.view(1, -1, 1, 1) + r.view(1, -1, 1, 1) * yfq
    return dy, dyq, yf, q, r, eff


def backward_case(name, dtype, B, H, W, Cin, Cout, R, S, pad, transposed=0, with_q=1, acc=0, what="dgrad", seed=0, ref_dev="cpu", mode=0,
                  expect=None):
    """The backward launches of the timed configuration, each through its C-ABI entry point against autograd of
    torch.nn.functional on the CPU (fp32 reference of the same op, 16-bit-rounded operands):
      what = "fused":  dmm_conv1x1_backward_fused (bw1.hip): data + weight gradient of a 1x1 bottleneck convolution
             "dgrad":  dmm_conv_dgrad_ex with the effective-gradient prologue (conv3.hip PRO=2 for the dense 3x3)
             "wgradT": dmm_conv_wgrad_ex in the transposed form (wg3.hip for the dense 3x3)
             "wgrad":  dmm_conv_wgrad_ex, normal form, with the effective-gradient prologue (wgp.hip for ConvTranspose phases)"""
    # ref_dev="cuda": the torch reference itself runs on the GPU (production sizes; fp32 autograd of the same op)
    g = torch.Generator(device=ref_dev).manual_seed(seed)
    dt = {0: torch.float32, 1: torch.float16, 2: torch.bfloat16}[dtype]
    kw = dict(generator=g, device=ref_dev)
    x = (torch.randn(B, Cin, H, W, **kw) * 2 + 0.5)
    scale = torch.rand(Cin, **kw) + 0.5
    shift = torch.randn(Cin, **kw) * 0.5
    mean = torch.randn(Cin, **kw)
    invstd = torch.rand(Cin, **kw) + 0.5
    wshape = (Cin, Cout, 3, 3) if transposed else (Cout, Cin, R, S)
    w = torch.randn(wshape, **kw) / (Cin * R * S) ** 0.5
    xq = x.to(dt).float()
    z = xq * scale.view(1, -1, 1, 1) + shift.view(1, -1, 1, 1)
    a = F.relu(z).requires_grad_(True)
    wq = w.clone().requires_grad_(True)
    if transposed:
        y = F.conv_transpose2d(a, wq, stride=2, padding=1, output_padding=1)
    elif mode == 1:   # the head's 3x3 over the nearest-x2 upsampled input (M:120, M:126)
        y = F.conv2d(F.interpolate(a, scale_factor=2, mode="nearest"), wq, padding=pad)
    else:
        y = F.conv2d(a, wq, padding=pad)
    Ho, Wo = y.shape[2], y.shape[3]
    dy, dyq, yf, q, r, eff = _eff_setup(g, dt, B, Cout, Ho, Wo, with_q)
    (y * eff).sum().backward()
    dz = a.grad * (z > 0)
    gold = (torch.randn(B, Cin, H, W, **kw)).to(dt) if acc else None
    gx_ref = dz * scale.view(1, -1, 1, 1) + (gold.float() if acc else 0.0)
    xhat = (xq.double() - mean.double().view(1, -1, 1, 1)) * invstd.double().view(1, -1, 1, 1)
    d = _lib.ConvDesc(dtype=dtype, use_mfma=1, B=B, H=H, W=W, Cin=Cin, Cout=Cout, R=R, S=S, stride=2 if transposed else 1, pad=pad,
                      transposed=transposed, mode=mode, bn_relu=1)
    scratch = torch.zeros(L.dmm_conv_scratch_bytes(C.byref(d)), dtype=torch.uint8, device=DEV)
    xd, dyd = nhwc(x, dt).to(DEV), nhwc(dy, dt).to(DEV)
    yfd = nhwc(yf, dt).to(DEV) if with_q else None
    qd, rd = (q.to(DEV), r.to(DEV)) if with_q else (None, None)
    wd, sd, hd = w.to(DEV), scale.to(DEV), torch.cat([shift, mean, invstd]).to(DEV)
    ptr = lambda t: t.data_ptr() if t is not None else None  # noqa: E731
    st = _lib.stream_ptr()
    res = {}
    gxd = nhwc(gold.float(), dt).to(DEV) if acc else torch.full((B, H, W, Cin), float("nan"), dtype=dt, device=DEV)
    red = torch.zeros(2 * Cin, dtype=torch.float64, device=DEV)
    dwd = torch.full(wshape, float("nan"), device=DEV)
    _lib.impls_since_reset()
    if what == "fused":
        _lib.check(L.dmm_conv1x1_backward_fused(C.byref(d), xd.data_ptr(), dyd.data_ptr(), wd.data_ptr(), sd.data_ptr(), hd.data_ptr(),
                                                ptr(yfd), ptr(qd), ptr(rd), gxd.data_ptr(), acc, dwd.data_ptr(), red.data_ptr(),
                                                scratch.data_ptr(), st))
    elif what == "dgrad":
        _lib.check(L.dmm_conv_dgrad_ex(C.byref(d), xd.data_ptr(), dyd.data_ptr(), wd.data_ptr(), sd.data_ptr(), hd.data_ptr(),
                                       ptr(yfd), ptr(qd), ptr(rd), gxd.data_ptr(), acc, red.data_ptr(), scratch.data_ptr(), st))
    else:
        _lib.check(L.dmm_conv_wgrad_ex(C.byref(d), xd.data_ptr(), dyd.data_ptr(), sd.data_ptr(), hd.data_ptr(), ptr(yfd), ptr(qd), ptr(rd),
                                       1 if what == "wgradT" else 0, dwd.data_ptr(), scratch.data_ptr(), st))
    torch.cuda.synchronize()
    fam_ok = _check_family(f"{what} {name}", expect)
    if what in ("fused", "dgrad"):
        res["dgrad"] = relerr(nchw(gxd).cpu(), gx_ref.cpu())
        res["red1"] = relerr(red[:Cin].cpu(), dz.double().sum(dim=(0, 2, 3)).cpu())
        res["red2"] = relerr(red[Cin:].cpu(), (dz.double() * xhat).sum(dim=(0, 2, 3)).cpu())
    if what != "dgrad":
        res["wgrad"] = relerr(dwd.cpu(), wq.grad.cpu())
    tol = {0: 2e-5, 1: 3e-3, 2: 2.5e-2}[dtype]
    bad = [k for k, v in res.items() if not (v < tol)]
    print(f"{'FAIL' if bad else 'ok  '} {what:6s} {name:28s} dt={dtype} q={with_q} acc={acc} " + " ".join(f"{k}={v:.2e}" for k, v in res.items()), flush=True)
    return not bad and fam_ok


def conv5_stats_case(name, dtype, B, H, W, Cout=3, seed=0, ref_dev="cpu"):
    """dmm_conv5_wgrad_stats (round 5; wg5.hip PA = 3 + wg5_fin64_kernel): the head's 5x5 convolution behind BN+ReLU - weight gradient
    AND the two BatchNorm-backward sums of the norm in front of it from one pass over x and dy - against fp64 torch on the same
    16-bit-rounded operands: dz = [bn(x) > 0] * conv_dgrad(dy, w rounded to the storage type).  The sums are held to 1e-6 (f16) of the
    largest channel sum: they feed a norm ON the data-gradient chain, where a per-channel offset is amplified ~1e4-fold by the
    encoder behind it (the data-gradient pass these sums used to come from staged dz in 16 bits: 3e-6 / 3e-5 off)."""
    g = torch.Generator(device=ref_dev).manual_seed(seed)
    dt = {1: torch.float16, 2: torch.bfloat16}[dtype]
    kw = dict(generator=g, device=ref_dev)
    Cin = 64
    x = torch.randn(B, Cin, H, W, **kw) * 2 + 0.5
    scale = torch.rand(Cin, **kw) + 0.5
    shift = torch.randn(Cin, **kw) * 0.5
    mean = torch.randn(Cin, **kw)
    invstd = torch.rand(Cin, **kw) + 0.5
    w = torch.randn(Cout, Cin, 5, 5, **kw) / (Cin * 25) ** 0.5
    dy = torch.rand(B, Cout, H, W, **kw) - 0.3            # like sigmoid(logit) - target: a non-zero mean
    xq, dyq, wq = x.to(dt).double(), dy.to(dt).double(), w.to(dt).double()
    z = xq.float() * scale.view(1, -1, 1, 1) + shift.view(1, -1, 1, 1)       # the forward's own fp32 fma ...
    m = (z.to(dt).float() > 0).double()                                       # ... rounded to the storage type, then ReLU
    a = m * (xq * scale.double().view(1, -1, 1, 1) + shift.double().view(1, -1, 1, 1))   # scale (m x) + shift m: the activation, unrounded
    dz = m * F.conv_transpose2d(dyq, wq, padding=2)                           # conv_dgrad of a unit-stride conv = conv_transpose
    xhat = (xq - mean.double().view(1, -1, 1, 1)) * invstd.double().view(1, -1, 1, 1)
    red1_ref, red2_ref = dz.sum(dim=(0, 2, 3)), (dz * xhat).sum(dim=(0, 2, 3))
    wr = wq.clone().requires_grad_(True)
    (F.conv2d(a, wr, padding=2) * dyq).sum().backward()
    dw_ref = wr.grad
    d = _lib.ConvDesc(dtype=dtype, use_mfma=1, B=B, H=H, W=W, Cin=Cin, Cout=Cout, R=5, S=5, stride=1, pad=2, transposed=0, mode=0, bn_relu=1)
    scratch = torch.zeros(L.dmm_conv_scratch_bytes(C.byref(d)), dtype=torch.uint8, device=DEV)
    xd, dyd = nhwc(x, dt).to(DEV), nhwc(dy, dt).to(DEV)
    wd, sd, hd = w.to(DEV), scale.to(DEV), torch.cat([shift, mean, invstd]).to(DEV)
    red = torch.full((2 * Cin,), float("nan"), dtype=torch.float64, device=DEV)
    dwd = torch.full((Cout, Cin, 5, 5), float("nan"), device=DEV)
    _lib.impls_since_reset()
    _lib.check(L.dmm_conv5_wgrad_stats(C.byref(d), xd.data_ptr(), dyd.data_ptr(), wd.data_ptr(), sd.data_ptr(), hd.data_ptr(), dwd.data_ptr(),
                                       red.data_ptr(), scratch.data_ptr(), _lib.stream_ptr()))
    torch.cuda.synchronize()
    fam_ok = _check_family(f"conv5 {name}", "wg5")
    res = {"red1": relerr(red[:Cin].cpu(), red1_ref.cpu()), "red2": relerr(red[Cin:].cpu(), red2_ref.cpu()),
           "wgrad": relerr(dwd.cpu(), dw_ref.cpu())}
    tol = {"red1": 1e-6, "red2": 1e-6, "wgrad": {1: 3e-3, 2: 2.5e-2}[dtype]}
    bad = [k for k, v in res.items() if not (v < tol[k])]
    print(f"{'FAIL' if bad else 'ok  '} conv5  {name:28s} dt={dtype} " + " ".join(f"{k}={v:.2e}" for k, v in res.items()), flush=True)
    return not bad and fam_ok


def production_forward_case(name, dtype, B, H, W, Cin, Cout, R, stride, pad, bn, transposed=0, reps=3, expect=None):
    """A forward convolution at a PRODUCTION size through the C ABI, several times on identical operands: against torch's GPU
    convolution (fp32 on the same 16-bit-rounded operands) and run to run (bitwise).  Timing-dependent hazards of the LDS pipelines
    (a refill landing before a queued fragment read has executed) only show with the chip full; parity-test shapes cannot see them."""
    g = torch.Generator(device=DEV).manual_seed(1)
    dt = {1: torch.float16, 2: torch.bfloat16}[dtype]
    x = torch.randn(B, Cin, H, W, device=DEV, generator=g) * 2 + 0.5
    scale = torch.rand(Cin, device=DEV, generator=g) + 0.5
    shift = torch.randn(Cin, device=DEV, generator=g) * 0.5
    wshape = (Cin, Cout, 3, 3) if transposed else (Cout, Cin, R, R)
    w = torch.randn(wshape, device=DEV, generator=g) / (Cin * R * R) ** 0.5
    xq = x.to(dt).float()
    a = F.relu(xq * scale.view(1, -1, 1, 1) + shift.view(1, -1, 1, 1)).to(dt).float() if bn else xq
    wq = w.to(dt).float()
    ref = F.conv_transpose2d(a, wq, stride=2, padding=1, output_padding=1) if transposed else F.conv2d(a, wq, stride=stride, padding=pad)
    d = _lib.ConvDesc(dtype=dtype, use_mfma=1, B=B, H=H, W=W, Cin=Cin, Cout=Cout, R=R, S=R, stride=stride, pad=pad, transposed=transposed,
                      mode=0, bn_relu=bn)
    scratch = torch.zeros(L.dmm_conv_scratch_bytes(C.byref(d)), dtype=torch.uint8, device=DEV)
    xd = x.permute(0, 2, 3, 1).contiguous().to(dt)
    outs = []
    _lib.impls_since_reset()
    for _ in range(reps):
        yd = torch.full((B, ref.shape[2], ref.shape[3], Cout), float("nan"), dtype=dt, device=DEV)
        stats = torch.zeros(2 * Cout, dtype=torch.float64, device=DEV)
        _lib.check(L.dmm_conv_forward(C.byref(d), xd.data_ptr(), w.data_ptr(), scale.data_ptr(), shift.data_ptr(), yd.data_ptr(),
                                      stats.data_ptr(), scratch.data_ptr(), _lib.stream_ptr()))
        torch.cuda.synchronize()
        outs.append(yd.permute(0, 3, 1, 2).float())
    top = float(ref.abs().max())
    tol = {1: 3e-3, 2: 2.5e-2}[dtype]
    errs = [float((o - ref).abs().max()) / top for o in outs]
    nbad = [int(((o - ref).abs() > tol * top).sum()) for o in outs]
    ndiff = [int((o != outs[0]).sum()) for o in outs[1:]]
    ok = max(errs) < tol and not any(ndiff) and _check_family("production " + name, expect)
    print(f"{'ok  ' if ok else 'FAIL'} production {name:36s} dt={dtype} err " + " ".join(f"{e:.2e}" for e in errs) + f" bad {nbad} run-to-run differing {ndiff}", flush=True)
    return ok


CASES = [
    # name, B,H,W,Cin,Cout,R,S,stride,pad, transposed, mode, bn
    ("1x1 72->32", 2, 12, 20, 72, 32, 1, 1, 1, 0, 0, 0, 1),
    ("1x1 256->128", 1, 16, 24, 256, 128, 1, 1, 1, 0, 0, 0, 1),
    ("1x1 64->256", 1, 8, 8, 64, 256, 1, 1, 1, 0, 0, 0, 1),
    ("3x3 128->32", 2, 12, 20, 128, 32, 3, 3, 1, 1, 0, 0, 1),
    ("3x3 32->8", 1, 9, 7, 32, 8, 3, 3, 1, 1, 0, 0, 1),
    ("5x5 64->8", 1, 10, 14, 64, 8, 5, 5, 1, 2, 0, 0, 1),
    ("7x7s2 8->64", 2, 32, 32, 8, 64, 7, 7, 2, 3, 0, 0, 0),
    ("convT 64->64", 2, 6, 10, 64, 64, 3, 3, 2, 1, 1, 0, 1),
    ("convT 16->16", 1, 5, 3, 16, 16, 3, 3, 2, 1, 1, 0, 1),
    ("pool2 1x1 128->64", 2, 12, 20, 128, 64, 1, 1, 1, 0, 0, 2, 1),
    ("up2 3x3 128->64", 1, 6, 10, 128, 64, 3, 3, 1, 1, 0, 1, 1),
    ("up2 3x3 16->8", 2, 4, 4, 16, 8, 3, 3, 1, 1, 0, 1, 1),
]


def run_kernels():
    ok = True
    for dtype in (0, 1):
        for mfma in (0, 1):
            for c in CASES:
                try:
                    ok &= conv_case(c[0], dtype, mfma, *c[1:])
                except Exception:
                    ok = False
                    print(f"EXC  {c[0]} dt={dtype} mfma={mfma}\n" + traceback.format_exc(), flush=True)
    return ok


def run_model(variants=("no", "early", "mid3", "mid2", "mid4"), dtypes=("fp32", "fp16"), mfma=True, loss_scale=None):
    from oracle import restatement as R
    from dmmfods_amd.graphs.models.Dense_U_Net_lidar import Dense_U_Net_lidar
    from dmmfods_amd.utils.Dense_U_Net_lidar_helper import get_config
    V = {"no": (1, 0), "early": (1, 3), "mid2": (2, 3), "mid3": (3, 3), "mid4": (4, 3)}
    ok = True
    for v in variants:
        cbb, s2 = V[v]
        arch = R.Arch(growth_rate=8, block_config=(2, 2, 2, 2), num_init_features=16, concat_before_block_num=cbb,
                      stream_2_in_channels=s2)
        # oracle in fp64 (truth) and fp32 (noise floor)
        outs = {}
        for odt in (torch.float64, torch.float32):
            P = {k: (t.to(odt) if t.is_floating_point() else t.clone()) for k, t in R.make_state(arch, seed=123).items()}
            tr = R.Trainer(arch, P)
            rgb, lidar, tgt = R.make_inputs(arch, 2, 64, 96, seed=0)
            o = tr.step(rgb.to(odt), lidar.to(odt), tgt.to(odt), do_update=False)
            outs[odt] = (o, {k: t.grad.clone() for k, t in tr.leaves}, P)
        o64, g64, P64 = outs[torch.float64]
        o32, g32, _ = outs[torch.float32]
        # fp16-storage emulation (fp64 arithmetic, fp16 rounding where the HIP path stores fp16)
        Ph = {k: (t.double() if t.is_floating_point() else t.clone()) for k, t in R.make_state(arch, seed=123).items()}
        trh = R.Trainer(arch, Ph, storage=torch.float16)
        rgb, lidar, tgt = R.make_inputs(arch, 2, 64, 96, seed=0)
        oh = trh.step(rgb.double(), lidar.double(), tgt.double(), do_update=False)
        gh = {k: t.grad.clone() for k, t in trh.leaves}
        for dts in dtypes:
            try:
                cfg = get_config("/tmp/dmm")
                cfg.model.growth_rate, cfg.model.block_config, cfg.model.num_init_features = 8, (2, 2, 2, 2), 16
                cfg.model.concat_before_block_num, cfg.model.stream_2_in_channels = cbb, s2
                model = Dense_U_Net_lidar(cfg, compute_dtype=dts, use_mfma=mfma, loss_scale=loss_scale)
                model.load_state_dict(R.make_state(arch, seed=123))
                model = model.to(DEV).train()
                rgb, lidar, tgt = R.make_inputs(arch, 2, 64, 96, seed=0)
                logits = model(rgb.to(DEV), lidar.to(DEV))
                met = model.loss_backward(tgt.to(DEV))
                torch.cuda.synchronize()
                if dts == "fp16":  # judge the fp16 build against fp16-storage emulation
                    o64, g64 = oh, gh
                else:
                    o64, g64 = outs[torch.float64][0], outs[torch.float64][1]
                e_log = relerr(logits.detach().cpu(), o64["logits"])
                n_log = relerr(o32["logits"], o64["logits"])
                e_loss = relerr(met["loss_per_class"].cpu(), o64["loss_per_class"])
                iou_ok = torch.allclose(met["iou_per_instance_per_class"].cpu(), o64["iou"].float(), atol=2e-2, equal_nan=True)
                worst = []
                num = den = 0.0
                for k, p in model.named_parameters():
                    ref = g64[k]
                    s = ref.abs().max().clamp_min(1e-30)
                    num += (p.grad.detach().cpu().double() - ref).pow(2).sum().item()
                    den += ref.pow(2).sum().item()
                    e = ((p.grad.detach().cpu().double() - ref).abs().max() / s).item()
                    n = ((g32[k].double() - ref).abs().max() / s).item()
                    worst.append((e, n, k))
                worst.sort(reverse=True)
                sd = model.state_dict()
                e_rm = max(relerr(sd[k].cpu(), P64[k]) for k in sd if k.endswith("running_mean"))
                e_rv = max(relerr(sd[k].cpu(), P64[k]) for k in sd if k.endswith("running_var"))
                tol = 1e-3 if dts == "fp32" else 5e-2
                bad = e_log > tol or worst[0][0] > (3e-3 if dts == "fp32" else 0.2) or not iou_ok
                ok &= not bad
                print(f"{'FAIL' if bad else 'ok  '} model {v:6s} {dts} mfma={int(mfma)} logits={e_log:.2e} (cpu32 {n_log:.1e}) loss={e_loss:.2e} "
                      f"iou_ok={iou_ok} rm={e_rm:.1e} rv={e_rv:.1e} gradL2={(num / den) ** 0.5:.2e} ls={loss_scale}", flush=True)
                for e, n, k in worst[:6]:
                    print(f"        grad err {e:.2e} (cpu32 {n:.1e}) {k}", flush=True)
            except Exception:
                ok = False
                print(f"EXC  model {v} {dts}\n" + traceback.format_exc(), flush=True)
    return ok


if __name__ == "__main__":
    what = sys.argv[1:] or ["kernels", "model"]
    print(torch.cuda.get_device_name(0), flush=True)
    t0 = time.time()
    ok = True
    if "kernels" in what:
        ok &= run_kernels()
    if "model" in what:
        ok &= run_model()
    if "scale" in what:
        for ls in (1.0, 64.0, 4096.0, 1.0 / 64):
            run_model(variants=("no",), dtypes=("fp16",), loss_scale=ls)
        run_model(variants=("no",), dtypes=("fp16",), mfma=False)
    if "model_scalar" in what:
        ok &= run_model(variants=("no",), dtypes=("fp32",), mfma=False)
    print(f"LAB {'PASS' if ok else 'FAIL'} in {time.time() - t0:.1f}s", flush=True)
