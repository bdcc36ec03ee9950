"""CPU oracle for the Dense_U_Net_lidar hot path  --  TEST INFRASTRUCTURE ONLY.

This file is a checker.  Only ``tests/``, ``__graft_entry__.smoke()`` and the
``cpu_baseline`` leg of ``bench.py`` may import it; nothing under
``dmmfods_amd/`` (the product) may.  It is a from-scratch functional restatement
(plain ``torch.nn.functional`` on a flat ``{state_dict key: tensor}`` mapping, CPU
fp32) of the reference algorithm:

  * network topology / forward ........ reference dmmfods/graphs/models/Dense_U_Net_lidar.py:29-267
  * dense layer / block / transition .. torchvision.models.densenet (``_DenseLayer``,
    ``_DenseBlock``, ``_Transition``); third-party, NOT vendored in /root/reference and not
    version-pinned there (requirements.txt:8).  Restated from the published DenseNet-BC
    algorithm (BN-ReLU-1x1(bn_size*k)-BN-ReLU-3x3(k), channel concat, transition =
    BN-ReLU-1x1(C/2)-AvgPool2).  The reference's only constraints on it are its call sites
    M:85-92, M:97-98, M:169-176, M:180-181, M:186 and the key-remap regex M:281-282.
  * training step ..................... reference dmmfods/agents/Dense_U_Net_lidar_Agent.py:244-265
  * metrics ........................... reference dmmfods/utils/Dense_U_Net_lidar_helper.py:311-401
  * focal losses ...................... reference dmmfods/graphs/losses/FocalLoss.py:41-50, 78-91
  * batched-file slicing .............. reference dmmfods/datasets/WaymoData.py:87-103
  * per-batch metric aggregation ...... reference dmmfods/agents/Dense_U_Net_lidar_Agent.py:252-260

Parity pin (G1-G5, G7 model; G6 focal losses / dataset slicing / agent metrics): ``oracle/make_golden.py`` imports the
reference's own ``Dense_U_Net_lidar`` class (and ``FocalLoss``, ``WaymoDataset``, the helper's metric functions) in
the build container (five absent third-party modules shimmed in memory), runs it on seeded
inputs and commits the results under ``tests/golden/``; ``tests/test_oracle_golden.py`` checks
this restatement against those vectors.  The reference ships no golden vectors of its own.
"""
from __future__ import annotations

import dataclasses
import math
from collections import OrderedDict
from typing import Dict, List, Optional, Sequence, Tuple

import numpy as np
import torch
import torch.nn.functional as F

BN_EPS = 1e-5
BN_MOMENTUM = 0.1


# --------------------------------------------------------------------------------------
# architecture description (mirrors config.model.* fields, M:42-55)
# --------------------------------------------------------------------------------------
@dataclasses.dataclass(frozen=True)
class Arch:
    growth_rate: int = 32
    block_config: Tuple[int, ...] = (6, 12, 24, 16)
    num_init_features: int = 64
    bn_size: int = 4
    num_classes: int = 3
    concat_before_block_num: int = 2
    num_layers_before_blocks: int = 4
    stream_1_in_channels: int = 3
    stream_2_in_channels: int = 1

    @property
    def fusion(self) -> str:  # M:57-65
        cbb, s2 = self.concat_before_block_num, self.stream_2_in_channels
        if cbb == 1 and s2 == 0:
            return "no"
        if cbb == 1 and s2 > 0:
            return "early"
        if 1 < cbb <= len(self.block_config):
            return "mid"
        raise AttributeError("invalid fusion configuration")

    @property
    def net_in_channels(self) -> int:  # M:56-61
        c = self.stream_1_in_channels
        if self.fusion == "early":
            c += self.stream_2_in_channels
        return c

    def block_channels(self) -> Tuple[List[int], List[int]]:
        """(input channels, output channels) of every dense block (Appendix A algebra)."""
        cin, cout = [], []
        c = self.num_init_features
        for i, n in enumerate(self.block_config):
            cin.append(c)
            c = c + n * self.growth_rate
            cout.append(c)
            if i != len(self.block_config) - 1:
                c //= 2
        return cin, cout

    def decoder_widths(self) -> List[Tuple[int, int]]:
        """[(num_in, nf)] of Transposed_Convolution_Sequence_k (M:81-119)."""
        _, bout = self.block_channels()
        stack = [self.num_init_features + 2 * self.growth_rate] + bout
        num_in = stack.pop()
        out = []
        for _ in self.block_config:
            nf = stack.pop()
            out.append((num_in, nf))
            num_in = 2 * nf
        return out


DENSENETS = {
    121: dict(growth_rate=32, block_config=(6, 12, 24, 16), num_init_features=64),
    161: dict(growth_rate=48, block_config=(6, 12, 36, 24), num_init_features=96),
    169: dict(growth_rate=32, block_config=(6, 12, 32, 32), num_init_features=64),
    201: dict(growth_rate=32, block_config=(6, 12, 48, 32), num_init_features=64),
}


def densenet_arch(depth: int, **kw) -> Arch:
    return Arch(**DENSENETS[depth], **kw)


# --------------------------------------------------------------------------------------
# parameter table in state_dict order
# --------------------------------------------------------------------------------------
def _bn_entries(prefix: str, c: int):
    return [
        (prefix + ".weight", (c,), "bn_w"),
        (prefix + ".bias", (c,), "bn_b"),
        (prefix + ".running_mean", (c,), "bn_rm"),
        (prefix + ".running_var", (c,), "bn_rv"),
        (prefix + ".num_batches_tracked", (), "bn_nbt"),
    ]


def _encoder_entries(prefix: str, arch: Arch, in_ch: int, upto_block: Optional[int]):
    """conv0/norm0 + dense blocks/transitions; ``upto_block`` = number of blocks (None = all)."""
    k, bs = arch.growth_rate, arch.bn_size
    ent = [(prefix + ".conv0.weight", (arch.num_init_features, in_ch, 7, 7), "conv")]
    ent += _bn_entries(prefix + ".norm0", arch.num_init_features)
    c = arch.num_init_features
    nblocks = len(arch.block_config)
    for bi, nl in enumerate(arch.block_config):
        if upto_block is not None and bi == upto_block:
            break
        for li in range(nl):
            p = f"{prefix}.denseblock{bi + 1}.denselayer{li + 1}"
            cin = c + li * k
            ent += _bn_entries(p + ".norm1", cin)
            ent += [(p + ".conv1.weight", (bs * k, cin, 1, 1), "conv")]
            ent += _bn_entries(p + ".norm2", bs * k)
            ent += [(p + ".conv2.weight", (k, bs * k, 3, 3), "conv")]
        c += nl * k
        if bi != nblocks - 1:
            p = f"{prefix}.transition{bi + 1}"
            ent += _bn_entries(p + ".norm", c)
            ent += [(p + ".conv.weight", (c // 2, c, 1, 1), "conv")]
            c //= 2
    return ent


def param_table(arch: Arch) -> List[Tuple[str, Tuple[int, ...], str]]:
    """(key, shape, kind) for every state_dict tensor, in the reference's registration order
    (features, decoder, dec_out_to_heat_maps, then stream_2_features, concat_module; M:71-192)."""
    ent = _encoder_entries("features", arch, arch.net_in_channels, None)
    for j, (nin, nf) in enumerate(arch.decoder_widths(), start=1):
        p = f"decoder.Transposed_Convolution_Sequence_{j}"
        ent += _bn_entries(p + ".norm0", nin)
        ent += [(p + ".conv_reduce.weight", (nf, nin, 1, 1), "conv")]
        ent += _bn_entries(p + ".norm1", nf)
        ent += [(f"decoder.Transposed_Convolution_{j}.weight", (nf, nf, 3, 3), "convT")]
    nf = arch.decoder_widths()[-1][1]
    hin = nf + arch.stream_1_in_channels + arch.stream_2_in_channels
    ent += _bn_entries("dec_out_to_heat_maps.norm0", hin)
    ent += [("dec_out_to_heat_maps.refine0.weight", (nf // 2, hin, 3, 3), "conv")]
    ent += _bn_entries("dec_out_to_heat_maps.norm1", nf // 2)
    ent += [("dec_out_to_heat_maps.refine1.weight", (arch.num_classes, nf // 2, 5, 5), "conv")]
    if arch.fusion == "mid":
        cbb = arch.concat_before_block_num
        ent += _encoder_entries("stream_2_features", arch, arch.stream_2_in_channels, cbb - 1)
        cin, _ = arch.block_channels()
        c = cin[cbb - 1]
        ent += _bn_entries("concat_module.norm", 2 * c)
        ent += [("concat_module.conv.weight", (c, 2 * c, 1, 1), "conv")]
    return ent


def num_params(arch: Arch) -> int:
    return sum(int(np.prod(s)) for _, s, kind in param_table(arch) if kind in ("conv", "convT", "bn_w", "bn_b"))


# --------------------------------------------------------------------------------------
# build-owned counter-based weight / input generators (no torch RNG; reproducible anywhere)
# --------------------------------------------------------------------------------------
def _philox_uniform(seed: int, stream: int, n: int) -> np.ndarray:
    gen = np.random.Generator(np.random.Philox(key=[int(seed) & 0xFFFFFFFFFFFFFFFF, int(stream)]))
    return gen.random(n, dtype=np.float64)


def make_state(arch: Arch, seed: int = 123, perturb_bn: bool = True) -> "OrderedDict[str, torch.Tensor]":
    """Deterministic weights: conv ~ uniform with kaiming fan-in variance (2/fan_in), ConvT with
    variance 1/(3*fan_in) (PyTorch default for ConvTranspose2d, SURVEY 7 'Init parity'); BN gamma
    near 1 / beta near 0 (perturbed so that gamma/beta paths are exercised), running stats 0/1."""
    sd: "OrderedDict[str, torch.Tensor]" = OrderedDict()
    for idx, (key, shape, kind) in enumerate(param_table(arch)):
        n = int(np.prod(shape)) if shape else 1
        if kind in ("conv", "convT"):
            u = _philox_uniform(seed, idx, n)
            if kind == "conv":
                fan_in = shape[1] * shape[2] * shape[3]
                std = math.sqrt(2.0 / fan_in)
            else:
                fan_in = shape[0] * shape[2] * shape[3]
                std = math.sqrt(1.0 / (3.0 * fan_in))
            w = (u - 0.5) * (2.0 * math.sqrt(3.0) * std)
            sd[key] = torch.from_numpy(w.astype(np.float32)).reshape(shape)
        elif kind == "bn_w":
            u = _philox_uniform(seed, idx, n) if perturb_bn else np.full(n, 0.5)
            sd[key] = torch.from_numpy((1.0 + 0.2 * (u - 0.5)).astype(np.float32))
        elif kind == "bn_b":
            u = _philox_uniform(seed, idx, n) if perturb_bn else np.full(n, 0.5)
            sd[key] = torch.from_numpy((0.2 * (u - 0.5)).astype(np.float32))
        elif kind == "bn_rm":
            sd[key] = torch.zeros(shape, dtype=torch.float32)
        elif kind == "bn_rv":
            sd[key] = torch.ones(shape, dtype=torch.float32)
        elif kind == "bn_nbt":
            sd[key] = torch.zeros((), dtype=torch.int64)
        else:  # pragma: no cover
            raise AssertionError(kind)
    return sd


def make_inputs(arch: Arch, batch: int, height: int, width: int, seed: int = 0):
    """Synthetic batch per SURVEY 8(d): RGB ~ U[0,255]; LiDAR ~90 % exact zeros, rest U[0,255];
    targets (U>0.9) with the reference's heat-map levels {.3,.5,.75,1} on the pedestrian plane."""
    s1, s2 = arch.stream_1_in_channels, max(arch.stream_2_in_channels, 1)
    n1 = batch * s1 * height * width
    rgb = (_philox_uniform(seed, 1000, n1) * 255.0).astype(np.float32).reshape(batch, s1, height, width)
    n2 = batch * s2 * height * width
    u = _philox_uniform(seed, 1001, n2)
    v = _philox_uniform(seed, 1002, n2)
    lidar = np.where(u > 0.9, v * 255.0, 0.0).astype(np.float32).reshape(batch, s2, height, width)
    nt = batch * arch.num_classes * height * width
    t = _philox_uniform(seed, 1003, nt)
    lv = _philox_uniform(seed, 1004, nt)
    levels = np.array([0.3, 0.5, 0.75, 1.0], dtype=np.float32)
    tgt = np.where(t > 0.9, 1.0, 0.0).astype(np.float32).reshape(batch, arch.num_classes, height, width)
    if arch.num_classes > 1:
        ped = levels[np.minimum((lv * 4).astype(np.int64), 3)].reshape(batch, arch.num_classes, height, width)
        tgt[:, 1] = np.where(tgt[:, 1] > 0, ped[:, 1], 0.0)
    return torch.from_numpy(rgb), torch.from_numpy(lidar), torch.from_numpy(tgt)


# --------------------------------------------------------------------------------------
# functional forward
# --------------------------------------------------------------------------------------
class _RoundSTE(torch.autograd.Function):
    """Round to a storage dtype in forward, identity in backward (emulates the HIP path's fp16 storage of
    activations and packed weights while keeping fp32/fp64 arithmetic)."""

    @staticmethod
    def forward(ctx, x, dt):
        return x.to(dt).to(x.dtype)

    @staticmethod
    def backward(ctx, g):
        return g, None


class _Ctx:
    def __init__(self, P: Dict[str, torch.Tensor], training: bool, storage=None):
        self.P, self.training, self.storage = P, training, storage

    def st(self, x):
        return x if self.storage is None else _RoundSTE.apply(x, self.storage)

    def bn_relu(self, x: torch.Tensor, prefix: str) -> torch.Tensor:
        P = self.P
        y = F.batch_norm(
            x, P[prefix + ".running_mean"], P[prefix + ".running_var"], P[prefix + ".weight"], P[prefix + ".bias"],
            self.training, BN_MOMENTUM, BN_EPS,
        )
        if self.training:
            P[prefix + ".num_batches_tracked"] += 1
        return F.relu(y)

    def conv(self, x, key, stride=1, padding=0, store=True):
        y = F.conv2d(x, self.st(self.P[key + ".weight"]), None, stride, padding)
        return self.st(y) if store else y


def _dense_block(ctx: _Ctx, x: torch.Tensor, prefix: str, nlayers: int) -> torch.Tensor:
    feats = [x]
    for li in range(nlayers):
        p = f"{prefix}.denselayer{li + 1}"
        cat = torch.cat(feats, 1)
        b = ctx.conv(ctx.bn_relu(cat, p + ".norm1"), p + ".conv1")
        feats.append(ctx.conv(ctx.bn_relu(b, p + ".norm2"), p + ".conv2", padding=1))
    return torch.cat(feats, 1)


def _transition(ctx: _Ctx, x: torch.Tensor, prefix: str) -> torch.Tensor:
    return ctx.st(F.avg_pool2d(ctx.conv(ctx.bn_relu(x, prefix + ".norm"), prefix + ".conv", store=False), 2, 2))


def _stem(ctx: _Ctx, x: torch.Tensor, prefix: str):
    y = ctx.bn_relu(ctx.conv(x, prefix + ".conv0", stride=2, padding=3), prefix + ".norm0")
    return y, ctx.st(F.max_pool2d(y, 3, 2, 1))


def forward(P: Dict[str, torch.Tensor], arch: Arch, stream_1: torch.Tensor, stream_2: Optional[torch.Tensor],
            training: bool = True, storage=None, capture: Optional[dict] = None) -> torch.Tensor:
    """Logits (B, num_classes, H, W).  BN running stats / num_batches_tracked in ``P`` are updated in
    place when ``training`` (as nn.BatchNorm2d does).  Mirrors M:210-267.

    ``storage`` (e.g. torch.float16) rounds inputs, packed weights and every tensor the HIP path stores
    (conv / pool / transposed-conv outputs) to that dtype with straight-through gradients; the arithmetic stays
    in the tensors' own dtype.  It is the yardstick for the fp16 build: what fp16 storage alone does to the result.

    ``capture`` (a dict) receives the outputs at the reference's module boundaries, keyed by the reference module name
    (``features.denseblock2``, ``features.transition1``, ``decoder.Transposed_Convolution_3``, ...): the layer-level golden
    vectors (tests/golden/g3_layers_*.npz) are forward-hook outputs of exactly those modules."""
    ctx = _Ctx(P, training, storage)

    def cap(name, t):
        if capture is not None:
            capture[name] = t
        return t
    stream_1 = ctx.st(stream_1)
    if stream_2 is not None:
        stream_2 = ctx.st(stream_2)
    fusion = arch.fusion
    nb = len(arch.block_config)
    if fusion == "no":
        raw, x = stream_1, stream_1
    else:
        raw = torch.cat((stream_1, stream_2), 1)
        x = raw if fusion == "early" else stream_1
    if x.shape[2] % 32 or x.shape[3] % 32:
        # the reference fails with ValueError inside ConvTranspose2d(output_size=...) (M:261)
        raise ValueError("spatial size must be a multiple of 32")

    s2_feat = None
    if fusion == "mid":  # whole second encoder prefix first (M:233)
        _, z = _stem(ctx, stream_2, "stream_2_features")
        for bi in range(arch.concat_before_block_num - 1):
            z = _dense_block(ctx, z, f"stream_2_features.denseblock{bi + 1}", arch.block_config[bi])
            z = _transition(ctx, z, f"stream_2_features.transition{bi + 1}")
        s2_feat = z

    skips, sizes = [raw], []
    y0, x = _stem(ctx, x, "features")
    cap("features.relu0", y0)
    cap("features.pool0", x)
    sizes.append(y0.shape[2:])
    for bi in range(nb):
        if fusion == "mid" and bi == arch.concat_before_block_num - 1:  # after transition_{cbb-1} (M:242-245)
            assert x.shape == s2_feat.shape, f"{tuple(x.shape)} {tuple(s2_feat.shape)}"
            x = cap("concat_module", ctx.conv(ctx.bn_relu(torch.cat((x, s2_feat), 1), "concat_module.norm"), "concat_module.conv"))
        x = cap(f"features.denseblock{bi + 1}", _dense_block(ctx, x, f"features.denseblock{bi + 1}", arch.block_config[bi]))
        if bi != nb - 1:
            skips.append(x)
            sizes.append(x.shape[2:])
            x = cap(f"features.transition{bi + 1}", _transition(ctx, x, f"features.transition{bi + 1}"))

    for j in range(1, nb + 1):  # decoder (M:255-261)
        if j > 1:
            x = torch.cat((x, skips.pop()), 1)
        p = f"decoder.Transposed_Convolution_Sequence_{j}"
        x = cap(p, ctx.bn_relu(ctx.conv(ctx.bn_relu(x, p + ".norm0"), p + ".conv_reduce"), p + ".norm1"))
        hw = sizes.pop()
        x = ctx.st(F.conv_transpose2d(x, ctx.st(P[f"decoder.Transposed_Convolution_{j}.weight"]), None, stride=2,
                                      padding=1, output_padding=1))
        assert tuple(x.shape[2:]) == tuple(hw)
        cap(f"decoder.Transposed_Convolution_{j}", x)
    x = F.interpolate(x, scale_factor=2, mode="nearest")
    x = torch.cat((x, skips.pop()), 1)  # raw input again (M:264)
    x = cap("dec_out_to_heat_maps.refine0", ctx.conv(ctx.bn_relu(x, "dec_out_to_heat_maps.norm0"), "dec_out_to_heat_maps.refine0", padding=1))
    x = ctx.conv(ctx.bn_relu(x, "dec_out_to_heat_maps.norm1"), "dec_out_to_heat_maps.refine1", padding=2, store=False)
    return x


# --------------------------------------------------------------------------------------
# loss, metrics, training step
# --------------------------------------------------------------------------------------
def bce_with_logits(x: torch.Tensor, t: torch.Tensor) -> torch.Tensor:
    """Unreduced BCE-with-logits, l = max(x,0) - x*t + log1p(exp(-|x|))  (A:54, A:247)."""
    return torch.clamp(x, min=0) - x * t + torch.log1p(torch.exp(-torch.abs(x)))


def iou_whole_img_batch(a: torch.Tensor, b: torch.Tensor, thr: float = 0.7) -> torch.Tensor:
    """(B, C) IoU of thresholded maps; 0/0 -> NaN kept (H:311-367).  Symmetric in (a, b)."""
    pa, pb = a >= thr, b >= thr
    inter = (pa & pb).sum(dim=(2, 3)).float()
    union = (pa | pb).sum(dim=(2, 3)).float()
    return inter / union


def accuracy_per_class(gt: torch.Tensor, pred: torch.Tensor, thr: float = 0.7) -> torch.Tensor:
    """(C,) fraction of pixels whose thresholded prediction equals thresholded target (H:369-401)."""
    eq = ((pred >= thr) == (gt >= thr)).sum(dim=(0, 2, 3))
    return eq / (gt.numel() / gt.shape[1])


def focal_loss(x: torch.Tensor, t: torch.Tensor, alpha, gamma, logits: bool = True) -> torch.Tensor:
    """Unreduced (class-wise) focal loss  alpha_c * (1 - exp(-bce))**gamma_c * bce  on (B, C, H, W) tensors (L:41-50; per class
    L:78-91).  Scalars broadcast over the classes.  ``logits=False``: inputs are probabilities (L:43-44)."""
    bce = bce_with_logits(x, t) if logits else F.binary_cross_entropy(x, t, reduction="none")
    a = torch.as_tensor(alpha, dtype=x.dtype).reshape(1, -1, 1, 1)
    g = torch.as_tensor(gamma, dtype=x.dtype).reshape(1, -1, 1, 1)
    return a * (1.0 - torch.exp(-bce)) ** g * bce


def split_batch(batch: torch.Tensor):
    """(N, 7, H, W) batched file -> image (N,3,H,W), lidar (N,1,H,W), heat maps (N,3,H,W)   (D:87-103)."""
    return batch[:, :3, :, :], batch[:, 3, :, :].unsqueeze(1), batch[:, 4:, :, :]


def batch_metrics(logits: torch.Tensor, target: torch.Tensor, thr: float = 0.7):
    """The agent's per-batch metric block (A:252-260): per-class IoU as the NaN-ignoring mean over samples with an all-NaN class
    mapped to 0, the per-class NaN count, and the thresholded accuracy."""
    iou = iou_whole_img_batch(logits, target, thr)
    nan = torch.isnan(iou)
    cnt = (~nan).sum(dim=0)
    mean = torch.where(nan, torch.zeros_like(iou), iou).sum(dim=0) / cnt.clamp_min(1)
    iou_pc = torch.where(cnt > 0, mean, torch.zeros_like(mean))
    return iou_pc, nan.sum(dim=0), accuracy_per_class(target, logits, thr)


def leaf_params(P: Dict[str, torch.Tensor], arch: Arch) -> List[Tuple[str, torch.Tensor]]:
    """Trainable tensors in ``nn.Module.parameters()`` order (the order Adam sees, A:57)."""
    return [(k, P[k]) for k, _, kind in param_table(arch) if kind in ("conv", "convT", "bn_w", "bn_b")]


class Trainer:
    """The agent's training step A:244-265 on CPU: forward, unreduced BCE, metrics on raw logits,
    zero_grad, backward(ones), Adam(lr 1e-3, betas .9/.999, eps 1e-8, wd 0, amsgrad False; H:146-159)."""

    def __init__(self, arch: Arch, P: Dict[str, torch.Tensor], lr=1e-3, betas=(0.9, 0.999), eps=1e-8,
                 iou_threshold: float = 0.7, storage=None, loss_fn=None):
        self.arch, self.P, self.thr, self.storage = arch, P, iou_threshold, storage
        self.loss_fn = loss_fn or bce_with_logits  # e.g. lambda x, t: focal_loss(x, t, alpha, gamma)
        self.leaves = leaf_params(P, arch)
        for _, t in self.leaves:
            t.requires_grad_(True)
        self.opt = torch.optim.Adam([t for _, t in self.leaves], lr=lr, betas=betas, eps=eps, weight_decay=0,
                                    amsgrad=False)

    def step(self, rgb, lidar, target, do_update: bool = True):
        logits = forward(self.P, self.arch, rgb, lidar, training=True, storage=self.storage)
        loss = self.loss_fn(logits, target)
        loss_per_class = loss.detach().sum(dim=(0, 2, 3))
        iou = iou_whole_img_batch(logits.detach(), target, self.thr)
        acc = accuracy_per_class(target, logits.detach(), self.thr)
        self.opt.zero_grad()
        loss.backward(torch.ones_like(loss))
        if do_update:
            self.opt.step()
        return dict(logits=logits.detach(), loss_per_class=loss_per_class, iou=iou, acc=acc)

    def evaluate(self, rgb, lidar, target):
        with torch.no_grad():
            logits = forward(self.P, self.arch, rgb, lidar, training=False, storage=self.storage)
            loss = bce_with_logits(logits, target)
            return dict(logits=logits, loss_per_class=loss.sum(dim=(0, 2, 3)),
                        iou=iou_whole_img_batch(logits, target, self.thr),
                        acc=accuracy_per_class(target, logits, self.thr))


def conv_flops_forward(arch: Arch, height: int, width: int) -> float:
    """2*MACs of every convolution for one image (SURVEY 8d definition)."""
    total = 0.0
    hw = {}

    def sp(level):  # level 0 = full res, 1 = /2, 2 = /4 ...
        return (height >> level) * (width >> level)

    k, bs = arch.growth_rate, arch.bn_size

    def encoder(in_ch, upto):
        t = 2.0 * arch.num_init_features * in_ch * 49 * sp(1)
        c = arch.num_init_features
        for bi, nl in enumerate(arch.block_config):
            if upto is not None and bi == upto:
                break
            px = sp(2 + bi)
            for li in range(nl):
                t += 2.0 * px * ((c + li * k) * bs * k + bs * k * k * 9)
            c += nl * k
            if bi != len(arch.block_config) - 1:
                t += 2.0 * px * c * (c // 2)
                c //= 2
        return t

    total += encoder(arch.net_in_channels, None)
    nb = len(arch.block_config)
    if arch.fusion == "mid":
        cbb = arch.concat_before_block_num
        total += encoder(arch.stream_2_in_channels, cbb - 1)
        c = arch.block_channels()[0][cbb - 1]
        total += 2.0 * sp(2 + cbb - 1) * 2 * c * c
    for j, (nin, nf) in enumerate(arch.decoder_widths()):
        px = sp(2 + nb - 1 - j)
        total += 2.0 * px * nin * nf + 2.0 * px * nf * nf * 9
    nf = arch.decoder_widths()[-1][1]
    hin = nf + arch.stream_1_in_channels + arch.stream_2_in_channels
    total += 2.0 * sp(0) * (hin * (nf // 2) * 9 + (nf // 2) * arch.num_classes * 25)
    return total
