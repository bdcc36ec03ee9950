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


def main():
    os.makedirs(GOLD, exist_ok=True)
    torch.manual_seed(0)
    torch.set_num_threads(8)
    M, H = import_reference()
    g1_topology(M, H)
    g2_tiny(M, H)
    g4_c1(M, H)
    g3_layers(M, H)
    g5_trace(M, H)
    leaked = [p for p, _, fs in os.walk(REF) if os.path.basename(p) == "__pycache__"]
    assert not leaked, f"bytecode leaked into the reference tree: {leaked}"


if __name__ == "__main__":
    main()
