#!/usr/bin/env python3
"""Generate the golden vectors under tests/golden/ from the REFERENCE's own module.

TEST INFRASTRUCTURE ONLY -- runs only in the build container, where /root/reference exists.
It never copies reference source: it imports ``dmmfods.graphs.models.Dense_U_Net_lidar`` and
``dmmfods.utils.Dense_U_Net_lidar_helper`` from /root/reference with the five absent third-party
modules (torchvision, easydict, tensorflow, waymo_open_dataset, tensorboard is not needed) replaced
by in-memory shims, runs them on inputs/weights from the build-owned generators in
``oracle/restatement.py`` and stores *data* (inputs are regenerated, outputs are stored).

What is reference code vs shim: topology, fusion, decoder, head, forward, factories,
config, metric functions are the reference's.  ``_DenseLayer/_DenseBlock/_Transition`` come from the
shim below (torchvision is not vendored in the reference), restated from the DenseNet-BC paper and
torchvision's public key layout; structure is pinned by exact parameter-count agreement with the
published torchvision DenseNet totals (checked in ``main``).

Usage:  python oracle/make_golden.py            (writes tests/golden/*.npz, *.json.gz)
"""
import gzip
import hashlib
import importlib.machinery
import json
import os
import sys
import types

sys.dont_write_bytecode = True  # /root/reference must stay untouched (no __pycache__)

import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
REF = "/root/reference"
GOLD = os.path.join(ROOT, "tests", "golden")
sys.path.insert(0, ROOT)

from oracle import restatement as R  # noqa: E402


# ------------------------------------------------------------------ shims
class _ShimDenseLayer(nn.Module):
    def __init__(self, num_input_features, growth_rate, bn_size, drop_rate, memory_efficient=False):
        super().__init__()
        self.norm1 = nn.BatchNorm2d(num_input_features)
        self.relu1 = nn.ReLU(inplace=True)
        self.conv1 = nn.Conv2d(num_input_features, bn_size * growth_rate, kernel_size=1, stride=1, bias=False)
        self.norm2 = nn.BatchNorm2d(bn_size * growth_rate)
        self.relu2 = nn.ReLU(inplace=True)
        self.conv2 = nn.Conv2d(bn_size * growth_rate, growth_rate, kernel_size=3, stride=1, padding=1, bias=False)
        self.drop_rate = float(drop_rate)

    def forward(self, feats):
        x = torch.cat(feats, 1) if isinstance(feats, (list, tuple)) else feats
        y = self.conv2(self.relu2(self.norm2(self.conv1(self.relu1(self.norm1(x))))))
        if self.drop_rate > 0:
            y = F.dropout(y, p=self.drop_rate, training=self.training)
        return y


class _ShimDenseBlock(nn.ModuleDict):
    def __init__(self, num_layers, num_input_features, bn_size, growth_rate, drop_rate, memory_efficient=False):
        super().__init__()
        for i in range(num_layers):
            self.add_module("denselayer%d" % (i + 1), _ShimDenseLayer(
                num_input_features + i * growth_rate, growth_rate, bn_size, drop_rate, memory_efficient))

    def forward(self, init_features):
        feats = [init_features]
        for _, layer in self.items():
            feats.append(layer(feats))
        return torch.cat(feats, 1)


class _ShimTransition(nn.Sequential):
    def __init__(self, num_input_features, num_output_features):
        super().__init__()
        self.add_module("norm", nn.BatchNorm2d(num_input_features))
        self.add_module("relu", nn.ReLU(inplace=True))
        self.add_module("conv", nn.Conv2d(num_input_features, num_output_features, kernel_size=1, stride=1, bias=False))
        self.add_module("pool", nn.AvgPool2d(kernel_size=2, stride=2))


class _EasyDict(dict):
    def __init__(self, d=None, **kw):
        super().__init__()
        for k, v in dict(d or {}, **kw).items():
            setattr(self, k, v)

    def __setattr__(self, k, v):
        if isinstance(v, dict) and not isinstance(v, _EasyDict):
            v = _EasyDict(v)
        super().__setitem__(k, v)
        super().__setattr__(k, v)

    __setitem__ = __setattr__


def _mod(name, **attrs):
    m = types.ModuleType(name)
    m.__spec__ = importlib.machinery.ModuleSpec(name, None)
    m.__dict__.update(attrs)
    sys.modules[name] = m
    return m


def install_shims():
    def _no_url(*a, **k):
        raise RuntimeError("no network: pretrained weights unavailable")

    tv = _mod("torchvision")
    tvm = _mod("torchvision.models")
    tvd = _mod("torchvision.models.densenet", model_urls={}, _DenseLayer=_ShimDenseLayer,
               _DenseBlock=_ShimDenseBlock, _Transition=_ShimTransition)
    tvu = _mod("torchvision.models.utils", load_state_dict_from_url=_no_url)
    tv.models, tvm.densenet, tvm.utils = tvm, tvd, tvu
    _mod("easydict", EasyDict=_EasyDict)
    _mod("tensorflow")
    wod = _mod("waymo_open_dataset")
    wu = _mod("waymo_open_dataset.utils")
    for sub in ("range_image_utils", "transform_utils", "frame_utils"):
        setattr(wu, sub, _mod("waymo_open_dataset.utils." + sub))
    wod.utils = wu
    wod.dataset_pb2 = _mod("waymo_open_dataset.dataset_pb2")


def import_reference():
    if not os.path.isdir(REF):
        print("reference not present; nothing to do", file=sys.stderr)
        sys.exit(0)
    install_shims()
    sys.path.insert(0, REF)
    import dmmfods.graphs.models.Dense_U_Net_lidar as M
    import dmmfods.utils.Dense_U_Net_lidar_helper as H
    return M, H


# ------------------------------------------------------------------ helpers
def digest(t, nsample=2048):
    """Compact fingerprint of a tensor: moments in float64 plus a strided sample."""
    a = t.detach().to(torch.float64).flatten()
    n = a.numel()
    stride = max(1, n // nsample)
    if stride > 1 and stride % 2 == 0:
        stride += 1
    return dict(
        # mom = [numel, stride, sum, l2, absmax]
        mom=np.array([n, stride, a.sum().item(), a.norm().item(), a.abs().max().item() if n else 0.0], dtype=np.float64),
        sample=t.detach().flatten()[::stride][:nsample].to(torch.float32).numpy().copy(),
    )


def put(store, name, d):
    for k, v in d.items():
        store[f"{name}/{k}"] = v


def ref_config(H, arch: R.Arch):
    cfg = H.create_config("/tmp/dmmfods_golden")
    cfg = sys.modules["easydict"].EasyDict(cfg)
    m = cfg.model
    m.growth_rate, m.block_config, m.num_init_features = arch.growth_rate, tuple(arch.block_config), arch.num_init_features
    m.bn_size, m.num_classes = arch.bn_size, arch.num_classes
    m.concat_before_block_num = arch.concat_before_block_num
    m.stream_1_in_channels, m.stream_2_in_channels = arch.stream_1_in_channels, arch.stream_2_in_channels
    return cfg


VARIANTS = {  # name -> (concat_before_block_num, stream_2_in_channels)
    "no": (1, 0), "early": (1, 3), "mid2": (2, 3), "mid3": (3, 3), "mid4": (4, 3),
}
TINY = dict(growth_rate=8, block_config=(2, 2, 2, 2), num_init_features=16)


def variant_arch(base: dict, name: str, **kw) -> R.Arch:
    cbb, s2 = VARIANTS[name]
    return R.Arch(**base, concat_before_block_num=cbb, stream_2_in_channels=s2, **kw)


# ------------------------------------------------------------------ G1 topology
def g1_topology(M, H):
    out = {}
    published = {121: 7978856, 169: 14149480, 201: 20013928, 161: 28681000}
    for depth in (121, 161, 169, 201):
        for vname in VARIANTS:
            arch = variant_arch(R.DENSENETS[depth], vname)
            cfg = ref_config(H, arch)
            factory = getattr(M, f"densenet{depth}_u_lidar")
            with torch.device("meta"):
                model = factory(pretrained=False, config=cfg)
            sd = model.state_dict()
            keys = [[k, list(v.shape)] for k, v in sd.items()]
            blob = ";".join(f"{k}:{','.join(map(str, s))}" for k, s in keys).encode()
            entry = dict(num_params=int(model.num_params), n_tensors=len(keys), fusion=model.fusion,
                         sha256=hashlib.sha256(blob).hexdigest())
            if depth == 121:
                entry["keys"] = keys
            out[f"d{depth}_{vname}"] = entry
        # structural pin against torchvision's published DenseNet totals (encoder + norm5 + classifier)
        arch = variant_arch(R.DENSENETS[depth], "no")
        enc = sum(int(np.prod(s)) for k, s, kind in R.param_table(arch)
                  if k.startswith("features.") and kind in ("conv", "bn_w", "bn_b"))
        c_last = arch.block_channels()[1][-1]
        total = enc + 2 * c_last + 1000 * c_last + 1000
        assert total == published[depth], (depth, total, published[depth])
        out[f"d{depth}_torchvision_total"] = total
    with gzip.open(os.path.join(GOLD, "g1_topology.json.gz"), "wt") as f:
        json.dump(out, f, separators=(",", ":"))
    print("G1 written:", len(out), "entries")


# ------------------------------------------------------------------ G2 tiny-net numerics
def run_reference_steps(M, H, arch, B, Hh, Ww, nsteps, weight_seed=123, data_seed=0):
    cfg = ref_config(H, arch)
    model = M.Dense_U_Net_lidar(cfg)
    sd0 = R.make_state(arch, seed=weight_seed)
    model.load_state_dict(sd0, strict=True)
    assert int(model.num_params) == R.num_params(arch)
    model.train()
    loss_fn = torch.nn.BCEWithLogitsLoss(reduction="none")  # A:54
    opt = torch.optim.Adam(model.parameters(), lr=cfg.optimizer.learning_rate,
                           betas=(cfg.optimizer.beta1, cfg.optimizer.beta2), eps=cfg.optimizer.eps,
                           weight_decay=cfg.optimizer.weight_decay, amsgrad=cfg.optimizer.amsgrad)  # A:57-61
    thr = cfg.agent.iou_threshold
    store = {}
    for step in range(nsteps):
        rgb, lidar, tgt = R.make_inputs(arch, B, Hh, Ww, seed=data_seed + step)
        pred = model(rgb, lidar)                                               # A:244
        cur = loss_fn(pred, tgt)                                               # A:247
        lpc = torch.sum(cur.detach(), dim=(0, 2, 3))                           # A:248
        iou = H.compute_IoU_whole_img_batch(pred.detach(), tgt.detach(), thr)  # A:252
        acc = H.compute_accuracy(tgt.detach(), pred.detach(), thr)             # A:259
        opt.zero_grad()                                                        # A:263
        cur.backward(torch.ones_like(cur.detach()))                            # A:264
        if step == 0:
            store["logits_full"] = pred.detach().numpy().copy()
            for k, p in model.named_parameters():
                put(store, f"grad0/{k}", digest(p.grad, nsample=192))
        opt.step()                                                             # A:265
        put(store, f"step{step}/logits", digest(pred))
        store[f"step{step}/loss_per_class"] = lpc.numpy().astype(np.float64)
        store[f"step{step}/iou"] = iou.numpy().copy()
        store[f"step{step}/acc"] = acc.numpy().copy()
        if step in (0, nsteps - 1):
            for k, v in model.state_dict().items():
                put(store, f"state{step}/{k}", digest(v.float(), nsample=192))
    # tight eval-mode pin: fresh weights, ONE train-mode forward (sets running stats), no optimiser step
    m2 = M.Dense_U_Net_lidar(ref_config(H, arch))
    m2.load_state_dict(R.make_state(arch, seed=weight_seed), strict=True)
    m2.train()
    with torch.no_grad():
        rgb, lidar, tgt = R.make_inputs(arch, B, Hh, Ww, seed=data_seed)
        m2(rgb, lidar)
        m2.eval()
        rgb, lidar, tgt = R.make_inputs(arch, B, Hh, Ww, seed=data_seed + 50)
        store["eval1/logits_full"] = m2(rgb, lidar).numpy().copy()
        for k, v in m2.state_dict().items():
            if "running" in k or "tracked" in k:
                put(store, f"eval1_state/{k}", digest(v.float(), nsample=192))
    model.eval()
    with torch.no_grad():
        rgb, lidar, tgt = R.make_inputs(arch, B, Hh, Ww, seed=data_seed + 100)
        pred = model(rgb, lidar)
        store["eval/logits_full"] = pred.numpy().copy()
        store["eval/iou"] = H.compute_IoU_whole_img_batch(pred, tgt, thr).numpy().copy()
        store["eval/acc"] = H.compute_accuracy(tgt, pred, thr).numpy().copy()
    return store


def g2_tiny(M, H):
    for vname in VARIANTS:
        arch = variant_arch(TINY, vname)
        store = run_reference_steps(M, H, arch, B=2, Hh=64, Ww=96, nsteps=3)
        store["meta/arch"] = np.frombuffer(json.dumps(dict(TINY, variant=vname, B=2, H=64, W=96, nsteps=3,
                                                          weight_seed=123, data_seed=0)).encode(), dtype=np.uint8)
        path = os.path.join(GOLD, f"g2_tiny_{vname}.npz")
        np.savez_compressed(path, **store)
        print("G2", vname, "->", os.path.getsize(path) // 1024, "KiB")


# ------------------------------------------------------------------ G4 C1 checksum
def g4_c1(M, H):
    arch = variant_arch(R.DENSENETS[121], "no")
    cfg = ref_config(H, arch)
    model = M.densenet121_u_lidar(pretrained=False, config=cfg)
    model.load_state_dict(R.make_state(arch, seed=123), strict=True)
    model.train()
    rgb, lidar, tgt = R.make_inputs(arch, 1, 256, 384, seed=0)
    pred = model(rgb, lidar)
    cur = torch.nn.BCEWithLogitsLoss(reduction="none")(pred, tgt)
    cur.backward(torch.ones_like(cur.detach()))
    store = {}
    put(store, "logits", digest(pred, nsample=8192))
    store["loss_per_class"] = torch.sum(cur.detach(), dim=(0, 2, 3)).numpy().astype(np.float64)
    for k in ("features.conv0.weight", "features.denseblock3.denselayer24.conv2.weight",
              "decoder.Transposed_Convolution_2.weight", "dec_out_to_heat_maps.refine1.weight",
              "features.denseblock1.denselayer1.norm1.weight", "features.norm0.bias"):
        put(store, f"grad/{k}", digest(dict(model.named_parameters())[k].grad))
    path = os.path.join(GOLD, "g4_c1_d121_no.npz")
    np.savez_compressed(path, **store)
    print("G4 ->", os.path.getsize(path) // 1024, "KiB")


# ------------------------------------------------------------------ G3 layer-level vectors
G3_ARCH = dict(growth_rate=24, block_config=(2, 2, 2, 2), num_init_features=48)  # K = 48, 72, 96: not powers of two, not x32
G3_MODULES = ["features.relu0", "features.pool0", "features.denseblock1", "features.denseblock1.denselayer2",
              "features.transition1", "features.denseblock3", "features.transition3", "features.denseblock4", "concat_module",
              "decoder.Transposed_Convolution_Sequence_1", "decoder.Transposed_Convolution_1",
              "decoder.Transposed_Convolution_Sequence_3", "decoder.Transposed_Convolution_3", "decoder.Transposed_Convolution_4",
              "dec_out_to_heat_maps.refine0", "dec_out_to_heat_maps"]


def g3_layers(M, H):
    """Forward-hook outputs of individual reference modules (one dense layer, dense blocks, transitions, fusion module, decoder
    stages incl. the ConvTranspose output_size path, head) and the gradients of their weights, at channel counts that are
    neither powers of two nor multiples of 32."""
    for vname in ("early", "mid3"):
        arch = variant_arch(G3_ARCH, vname)
        model = M.Dense_U_Net_lidar(ref_config(H, arch))
        model.load_state_dict(R.make_state(arch, seed=321), strict=True)
        model.train()
        outs = {}
        mods = dict(model.named_modules())
        for name in G3_MODULES:
            if name in mods:
                mods[name].register_forward_hook(lambda m, i, o, name=name: outs.__setitem__(name, o.detach()))
        rgb, lidar, tgt = R.make_inputs(arch, 2, 64, 96, seed=7)
        pred = model(rgb, lidar)
        cur = torch.nn.BCEWithLogitsLoss(reduction="none")(pred, tgt)
        cur.backward(torch.ones_like(cur.detach()))
        store = {"logits_full": pred.detach().numpy().copy()}
        for name, t in outs.items():
            put(store, f"out/{name}", digest(t, nsample=4096))
        for k, p_ in model.named_parameters():
            put(store, f"grad/{k}", digest(p_.grad, nsample=256))
        store["meta/arch"] = np.frombuffer(json.dumps(dict(G3_ARCH, variant=vname, B=2, H=64, W=96, weight_seed=321,
                                                          data_seed=7)).encode(), dtype=np.uint8)
        path = os.path.join(GOLD, f"g3_layers_{vname}.npz")
        np.savez_compressed(path, **store)
        print("G3", vname, "->", os.path.getsize(path) // 1024, "KiB", sorted(outs))


# ------------------------------------------------------------------ G5 shape / FLOP trace
G5_CONFIGS = {  # BASELINE.json configs: densenet, variant, batch, H, W
    "c1": (121, "no", 1, 256, 384), "c2": (121, "early", 4, 1280, 1920), "c3": (121, "mid3", 4, 1280, 1920),
    "c4": (169, "mid3", 2, 1280, 1920), "c5": (201, "mid3", 8, 640, 960),
}


def g5_trace(M, H):
    """Per-convolution shapes and forward FLOPs (2 MACs) of the reference module, traced on the meta device (no arithmetic)."""
    out = {}
    for cname, (dn, vname, B, Hh, Ww) in G5_CONFIGS.items():
        arch = variant_arch(R.DENSENETS[dn], vname)
        with torch.device("meta"):
            model = M.Dense_U_Net_lidar(ref_config(H, arch))
        rows = []

        def hook(m, i, o, name=None):
            x = i[0]
            if isinstance(m, nn.ConvTranspose2d):
                fl = 2.0 * m.in_channels * m.out_channels * m.kernel_size[0] * m.kernel_size[1] * x.shape[2] * x.shape[3]
            else:
                fl = 2.0 * m.in_channels * m.out_channels * m.kernel_size[0] * m.kernel_size[1] * o.shape[2] * o.shape[3]
            rows.append([name, type(m).__name__, m.in_channels, m.out_channels, m.kernel_size[0], m.stride[0],
                         int(o.shape[2]), int(o.shape[3]), fl])

        for name, m in model.named_modules():
            if isinstance(m, (nn.Conv2d, nn.ConvTranspose2d)):
                m.register_forward_hook(lambda m, i, o, name=name: hook(m, i, o, name))
        s1 = torch.empty(1, arch.stream_1_in_channels, Hh, Ww, device="meta")
        s2 = torch.empty(1, max(arch.stream_2_in_channels, 1), Hh, Ww, device="meta")
        model(s1, s2)
        out[cname] = dict(densenet=dn, variant=vname, batch=B, H=Hh, W=Ww, num_params=int(model.num_params),
                          fwd_gflop_per_img=sum(r[-1] for r in rows) / 1e9, convs=rows)
        print("G5", cname, round(out[cname]["fwd_gflop_per_img"], 2), "GFLOP/img,", len(rows), "convs")
    with gzip.open(os.path.join(GOLD, "g5_flop_trace.json.gz"), "wt") as f:
        json.dump(out, f)


# ------------------------------------------------------------------ G7 BASELINE architectures that the GPU tests did not cover
G7_CASES = {  # name -> (densenet, variant, B, H, W, weight seed, data seed)
    "d121_mid3_64": (121, "mid3", 2, 64, 96, 123, 11),      # C3 architecture (reference factory M:335-347)
    "d121_mid3_128": (121, "mid3", 1, 128, 192, 123, 12),
    "d201_mid3_64": (201, "mid3", 1, 64, 96, 77, 13),       # C5 architecture (reference factory M:377-388)
}


def g7_configs(M, H):
    """One training forward/backward of the reference's DenseNet-121 / -201 mid-fusion factories: logits, loss sums, metrics and
    a digest of EVERY parameter gradient."""
    for name, (dn, vname, B, Hh, Ww, wseed, dseed) in G7_CASES.items():
        arch = variant_arch(R.DENSENETS[dn], vname)
        cfg = ref_config(H, arch)
        model = getattr(M, f"densenet{dn}_u_lidar")(pretrained=False, config=cfg)
        model.load_state_dict(R.make_state(arch, seed=wseed), strict=True)
        model.train()
        rgb, lidar, tgt = R.make_inputs(arch, B, Hh, Ww, seed=dseed)
        pred = model(rgb, lidar)
        cur = torch.nn.BCEWithLogitsLoss(reduction="none")(pred, tgt)
        cur.backward(torch.ones_like(cur.detach()))
        thr = cfg.agent.iou_threshold
        store = {"logits_full": pred.detach().numpy().copy(),
                 "loss_per_class": torch.sum(cur.detach(), dim=(0, 2, 3)).numpy().astype(np.float64),
                 "iou": H.compute_IoU_whole_img_batch(pred.detach(), tgt, thr).numpy().copy(),
                 "acc": H.compute_accuracy(tgt, pred.detach(), thr).numpy().copy()}
        for k, p_ in model.named_parameters():
            put(store, f"grad/{k}", digest(p_.grad, nsample=64))
        store["meta/case"] = np.frombuffer(json.dumps(dict(densenet=dn, variant=vname, B=B, H=Hh, W=Ww, weight_seed=wseed,
                                                          data_seed=dseed)).encode(), dtype=np.uint8)
        path = os.path.join(GOLD, f"g7_{name}.npz")
        np.savez_compressed(path, **store)
        print("G7", name, "->", os.path.getsize(path) // 1024, "KiB")


# ------------------------------------------------------------------ G6 rows 8(f): focal losses, dataset reader, agent batch metrics
def g6_frows(M, H):
    import tempfile
    from dmmfods.graphs.losses.FocalLoss import ClassWiseFocalLoss, FocalLoss   # reference L:9-91
    from dmmfods.datasets.WaymoData import WaymoDataset, WaymoDataset_Loader    # reference D:9-213
    store = {}
    # ---- focal losses on fixed logits / targets (values and d(sum)/d(input)) ----
    B, C, Hh, Ww = 2, 3, 16, 24
    n = B * C * Hh * Ww
    x = torch.from_numpy(((R._philox_uniform(9, 1, n) - 0.5) * 12.0).astype(np.float32)).reshape(B, C, Hh, Ww)
    lv = np.array([0.0, 0.0, 0.0, 0.0, 0.3, 0.5, 0.75, 1.0], dtype=np.float32)
    t = torch.from_numpy(lv[np.minimum((R._philox_uniform(9, 2, n) * 8).astype(np.int64), 7)]).reshape(B, C, Hh, Ww)
    store["focal/x"], store["focal/t"] = x.numpy().copy(), t.numpy().copy()
    cases = {"focal_a1_g2": FocalLoss(alpha=1, gamma=2, logits=True, reduce=False),
             "focal_a025_g15": FocalLoss(alpha=0.25, gamma=1.5, logits=True, reduce=False),
             "classwise_default": ClassWiseFocalLoss(),
             "classwise_mixed": ClassWiseFocalLoss(alpha=[1.0, 2.0, 0.5], gamma=[2.0, 1.0, 3.0])}
    for name, fn in cases.items():
        xi = x.clone().requires_grad_(True)
        out = fn(xi, t)
        out.backward(torch.ones_like(out))
        store[f"focal/{name}/loss"] = out.detach().numpy().copy()
        store[f"focal/{name}/dx"] = xi.grad.numpy().copy()
    store["focal/focal_a1_g2/mean"] = np.array(FocalLoss(alpha=1, gamma=2, logits=True, reduce=True)(x, t).item())
    store["focal/prob_a1_g2/loss"] = FocalLoss(alpha=1, gamma=2, logits=False, reduce=False)(torch.sigmoid(x), t).numpy().copy()
    # ---- dataset reader: the reference's WaymoDataset on batched files written here ----
    with tempfile.TemporaryDirectory() as tmp:
        cfg = sys.modules["easydict"].EasyDict(H.create_config(tmp))
        cfg.dir.data.root = os.path.join(tmp, "data")
        cfg.dir.data.file_lists = os.path.join(tmp, "lists")
        nb, N, hh, ww = 2, 3, 16, 24
        for mode in ("train", "val"):
            d = os.path.join(cfg.dir.data.root, mode, "part0")
            os.makedirs(os.path.join(d, "labels"))
            for i in range(nb):
                u = R._philox_uniform(21, 10 * (mode == "val") + i, N * 7 * hh * ww).astype(np.float32).reshape(N, 7, hh, ww)
                batch = torch.from_numpy(u.copy())
                batch[:, :4] *= 255.0
                batch[:, 4:] = (batch[:, 4:] > 0.9).float()
                torch.save(batch, os.path.join(d, f"batch_{i}.pt"))
                store[f"data/{mode}/batch_{i}"] = batch.numpy().copy()
        ds = WaymoDataset("train", cfg)
        files = sorted(ds.files)
        store["data/train_files"] = np.frombuffer(json.dumps(files).encode(), dtype=np.uint8)
        store["data/len"] = np.array(len(ds))
        for i, f in enumerate(files):
            img, lid, hm = ds.get_batch(ds.files.index(f))                      # D:87-103
            store[f"data/get_batch/{i}/image"] = img.numpy().copy()
            store[f"data/get_batch/{i}/lidar"] = lid.numpy().copy()
            store[f"data/get_batch/{i}/ht_map"] = hm.numpy().copy()
        cfg.loader.num_workers = 0
        cfg.loader.pin_memory = False
        ld = WaymoDataset_Loader(cfg)                                            # D:160-213
        store["data/iterations"] = np.array([ld.train_iterations, ld.valid_iterations])
        first = next(iter(ld.valid_loader))
        store["data/loader_first_val_shapes"] = np.array([list(v.shape) for v in first])
    # ---- the agent's per-batch metric block A:247-260 replayed on fixed tensors ----
    B, C, Hh, Ww = 4, 3, 16, 24
    n = B * C * Hh * Ww
    pred = torch.from_numpy(((R._philox_uniform(31, 1, n) - 0.45) * 6.0).astype(np.float32)).reshape(B, C, Hh, Ww).clone()
    gt = torch.from_numpy((R._philox_uniform(31, 2, n) > 0.8).astype(np.float32)).reshape(B, C, Hh, Ww).clone()
    pred[1, 2] = -1.0; gt[1, 2] = 0.0          # empty union -> NaN for (sample 1, class 2)
    pred[3, 0] = -2.0; gt[3, 0] = 0.0          # ... and (sample 3, class 0)
    pred[:, 1] = -3.0; gt[:, 1] = 0.0          # class 1 empty in every sample: nan-mean of all-NaN -> NaN -> 0 (A:254)
    thr = 0.7
    loss = torch.nn.BCEWithLogitsLoss(reduction="none")(pred, gt)                # A:54, A:247
    store["agent/pred"], store["agent/gt"] = pred.numpy().copy(), gt.numpy().copy()
    store["agent/loss_per_class"] = torch.sum(loss, dim=(0, 2, 3)).numpy().astype(np.float64)       # A:248
    iou = H.compute_IoU_whole_img_batch(pred, gt, thr)                           # A:252
    import warnings
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        iou_pc = torch.tensor(np.nanmean(iou, axis=0))                           # A:253
    iou_pc[torch.isnan(iou_pc)] = 0                                              # A:254
    store["agent/iou_per_instance"] = iou.numpy().copy()
    store["agent/iou_per_class"] = iou_pc.numpy().copy()
    store["agent/iou_nans"] = torch.sum(torch.isnan(iou), axis=0).numpy().copy()  # A:256
    store["agent/acc_per_class"] = H.compute_accuracy(gt, pred, thr).numpy().copy()  # A:259
    path = os.path.join(GOLD, "g6_frows.npz")
    np.savez_compressed(path, **store)
    print("G6 ->", os.path.getsize(path) // 1024, "KiB")


def main():
    os.makedirs(GOLD, exist_ok=True)
    torch.manual_seed(0)
    torch.set_num_threads(8)
    M, H = import_reference()
    only = set(sys.argv[1:])   # e.g. `make_golden.py g6 g7` regenerates only those groups
    if only:
        for tag, fn in (("g1", g1_topology), ("g2", g2_tiny), ("g4", g4_c1), ("g3", g3_layers), ("g5", g5_trace), ("g6", g6_frows),
                        ("g7", g7_configs)):
            if tag in only:
                fn(M, H)
        return
    g1_topology(M, H)
    g2_tiny(M, H)
    g4_c1(M, H)
    g3_layers(M, H)
    g5_trace(M, H)
    g6_frows(M, H)
    g7_configs(M, H)
    leaked = [p for p, _, fs in os.walk(REF) if os.path.basename(p) == "__pycache__"]
    assert not leaked, f"bytecode leaked into the reference tree: {leaked}"


if __name__ == "__main__":
    main()
