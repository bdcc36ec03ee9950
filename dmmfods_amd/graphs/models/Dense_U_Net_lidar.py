"""Dense_U_Net_lidar on MI355X: same constructor, forward signature, attributes and state_dict layout as the
reference module (dmmfods/graphs/models/Dense_U_Net_lidar.py:18-388); all arithmetic runs in the HIP library
behind include/dmmfods_hip.h.  There is no CPU path: calling forward on CPU tensors raises.

Differences that are deliberate and documented:
  * parameters are views into one flat fp32 arena (and .grad into a flat gradient arena), so Adam and the
    data-parallel all-reduce each touch one contiguous buffer;
  * backward overwrites .grad instead of accumulating (the reference always zero_grad()s first, A:263);
  * ``pretrained=True`` needs torchvision's ImageNet checkpoint, which cannot be fetched offline.
"""
import ctypes as C
import math
import os
import re
import sys
from collections import OrderedDict

import torch
import torch.nn as nn

from ... import _lib
from ...utils.Dense_U_Net_lidar_helper import get_config

_DTYPES = {"fp32": _lib.DMM_F32, "float32": _lib.DMM_F32, "fp16": _lib.DMM_F16, "float16": _lib.DMM_F16,
           "bf16": _lib.DMM_BF16, "bfloat16": _lib.DMM_BF16}   # storage / MFMA operand type; accumulation is always fp32


class _Node(nn.Module):
    """Bare container; gives state_dict() the reference's dotted key layout."""


def _make_desc(model, batch, height, width):
    d = _lib.ModelDesc()
    d.growth_rate = int(model.growth_rate)
    bc = tuple(int(v) for v in model.block_config)
    if not 2 <= len(bc) <= 8:
        raise ValueError("block_config must have 2..8 entries")
    d.num_blocks = len(bc)
    for i, v in enumerate(bc):
        d.block_config[i] = v
    d.num_init_features = int(model.num_init_features)
    d.bn_size = int(model.bn_size)
    d.num_classes = int(model.num_classes)
    d.concat_before_block_num = int(model.concat_before_block_num)
    d.stream_1_in_channels = int(model.stream_1_in_channels)
    d.stream_2_in_channels = int(model.stream_2_in_channels)
    d.batch, d.height, d.width = int(batch), int(height), int(width)
    d.dtype = _DTYPES[model.compute_dtype]
    d.loss_scale = float(model.loss_scale)
    d.bn_momentum, d.bn_eps = 0.1, 1e-5
    d.iou_threshold = float(model.iou_threshold)
    d.use_mfma = 1 if model.use_mfma else 0
    return d


def _tensor_table(handle):
    L = _lib.lib()
    out = []
    for i in range(L.dmm_plan_num_tensors(handle)):
        name, kind, nd = C.c_char_p(), C.c_int32(), C.c_int32()
        shape, off = (C.c_int64 * 4)(), C.c_int64()
        _lib.check(L.dmm_plan_tensor_info(handle, i, C.byref(name), C.byref(kind), C.byref(nd), C.byref(shape), C.byref(off)))
        out.append((name.value.decode(), kind.value, tuple(shape[j] for j in range(nd.value)), off.value))
    return out


class _Plan:
    def __init__(self, model, batch, height, width):
        L = _lib.lib()
        self.key = (batch, height, width)
        self.handle = C.c_void_p()
        desc = _make_desc(model, batch, height, width)
        _lib.check(L.dmm_plan_create(C.byref(desc), C.byref(self.handle)))
        nbytes = L.dmm_plan_workspace_bytes(self.handle)
        dev = model._param_arena.device
        # DMM_GUARD_MB=n (tests): n MiB of a known byte pattern on either side of the workspace; check_guards() tells whether any
        # kernel wrote outside what the plan sized
        self._guard = (int(os.environ.get("DMM_GUARD_MB", "0")) << 20)
        self._raw = torch.empty(nbytes + 2 * self._guard, dtype=torch.uint8, device=dev)
        if self._guard:
            self._raw[:self._guard].fill_(0xA5)
            self._raw[self._guard + nbytes:].fill_(0xA5)
        self.workspace = self._raw[self._guard:self._guard + nbytes]
        self._alloc_stream = torch.cuda.current_stream(dev)
        self._streams = set()
        nc = int(model.num_classes)
        self.metrics = torch.zeros(2 * nc + batch * 2 * nc, dtype=torch.float64, device=dev)
        self.flops_forward = L.dmm_plan_forward_flops(self.handle)
        try:
            _lib.check(L.dmm_plan_bind(self.handle, self.workspace.data_ptr(), nbytes, model._param_arena.data_ptr(),
                                       model._grad_arena.data_ptr(), model._buffer_arena.data_ptr()))
        except Exception:
            self.close()
            raise
        self.loss_key = None
        # gradient buckets in the order backward finishes them: (offset, count) in elements of the gradient arena
        self.grad_buckets = []
        for i in range(L.dmm_plan_num_grad_buckets(self.handle)):
            off, cnt = C.c_int64(), C.c_int64()
            _lib.check(L.dmm_plan_grad_bucket(self.handle, i, C.byref(off), C.byref(cnt)))
            self.grad_buckets.append((off.value, cnt.value))

    def check_guards(self):
        """True when the guard bands around the workspace (DMM_GUARD_MB) still hold their pattern."""
        if not self._guard:
            raise RuntimeError("no guard bands: set DMM_GUARD_MB before the plan is created")
        g = self._guard
        return bool((self._raw[:g] == 0xA5).all()) and bool((self._raw[g + self.workspace.numel():] == 0xA5).all())

    def note_stream(self):
        """Remember the torch stream a launch list is about to be enqueued on (close() orders the workspace's release behind it)."""
        st = torch.cuda.current_stream(self._raw.device)
        if st != self._alloc_stream:
            self._streams.add(st)

    @property
    def closed(self):
        return not self.handle

    def close(self):
        """Destroy the plan NOW, at a known statement: dmm_plan_destroy synchronises the library's helper streams, hands the plan's
        events back to the process pool and frees the host structures; then the workspace goes back to torch's allocator (which
        orders its reuse behind the stream it was allocated on; any other stream the plan was run on is recorded).  Raises DmmError
        if a HIP call of the teardown failed - the plan is gone either way.  Idempotent."""
        if not self.handle:
            return
        h, self.handle = self.handle, None
        rc = _lib.lib().dmm_plan_destroy(h)
        for st in self._streams:
            self._raw.record_stream(st)
        self._streams.clear()
        self.workspace = self._raw = self.metrics = None
        _lib.check(rc)

    def __del__(self):
        # Fallback only: plans are closed explicitly (Dense_U_Net_lidar.close(), eviction from the plan cache, _apply()).  A plan
        # that is still open when the garbage collector finds it is closed here and the fact is reported, never swallowed.  While the
        # interpreter is finalising nothing is done: module globals (ctypes, torch, this module's _lib) may already be gone, and
        # the process's exit releases the plan - NOT because the HIP runtime would be unloaded (it is not: round 4's guess).
        if sys is None or sys.is_finalizing() or not getattr(self, "handle", None):   # (module-level import: nothing can be imported at shutdown)
            return
        try:
            self.close()
        except Exception as e:  # noqa: BLE001 - a destructor must not raise; say what happened instead
            sys.stderr.write(f"[dmmfods_amd] closing a plan from the garbage collector failed: {e!r}\n")


class _HipBackward(torch.autograd.Function):
    """Ties the HIP backward pass to torch autograd: d(loss)/d(logits) comes in, parameter gradients land in the
    model's gradient arena (p.grad are views of it)."""

    @staticmethod
    def forward(ctx, hook, logits, model, plan):
        ctx.model, ctx.plan = model, plan
        return logits.view_as(logits)

    @staticmethod
    def backward(ctx, grad_logits):
        model, plan = ctx.model, ctx.plan
        if plan.closed:
            raise RuntimeError("backward() through a forward whose plan has been closed (model.close() or a size change in between)")
        plan.note_stream()
        g = grad_logits.contiguous().float()
        _lib.check(_lib.lib().dmm_plan_backward(plan.handle, g.data_ptr(), _lib.stream_ptr()))
        model._attach_grads()
        return torch.zeros_like(model._hook), None, None, None


class Dense_U_Net_lidar(nn.Module):
    """U-Net-like detector head on a DenseNet encoder with an optional LiDAR stream (reference M:18-267)."""

    def __init__(self, config, compute_dtype=None, loss_scale=None, use_mfma=True):
        super().__init__()
        self.config = config
        m = config.model
        self.growth_rate = m.growth_rate
        self.block_config = tuple(m.block_config)
        self.num_init_features = m.num_init_features
        self.bn_size = m.bn_size
        self.drop_rate = m.drop_rate
        self.memory_efficient = m.memory_efficient
        self.num_classes = m.num_classes
        self.concat_before_block_num = m.concat_before_block_num
        self.num_layers_before_blocks = m.num_layers_before_blocks
        self.concat_after_module_idx = self.num_layers_before_blocks - 1 + 2 * (self.concat_before_block_num - 1)
        self.stream_1_in_channels = m.stream_1_in_channels
        self.stream_2_in_channels = m.stream_2_in_channels
        self.network_input_channels = self.stream_1_in_channels
        if self.concat_before_block_num == 1 and self.stream_2_in_channels == 0:
            self.fusion = "no"
        elif self.concat_before_block_num == 1 and self.stream_2_in_channels > 0:
            self.fusion = "early"
            self.network_input_channels += self.stream_2_in_channels
        elif 1 < self.concat_before_block_num <= len(self.block_config):
            self.fusion = "mid"
        else:
            raise AttributeError("invalid fusion configuration")
        if self.drop_rate:
            raise ValueError("drop_rate > 0 is not supported by the HIP path (reference default 0, H:120)")
        self.compute_dtype = compute_dtype or os.environ.get("DMMFODS_DTYPE", "fp32")
        if self.compute_dtype not in _DTYPES:
            raise ValueError(f"compute_dtype must be one of {sorted(_DTYPES)}")
        self.loss_scale = 1.0 if loss_scale is None else float(loss_scale)
        self.use_mfma = bool(use_mfma)
        try:
            self.iou_threshold = float(config.agent.iou_threshold)
        except (AttributeError, KeyError):
            self.iou_threshold = 0.7

        # ---- state_dict layout from the library's own layer table (shape-independent) ----
        L = _lib.lib()
        h = C.c_void_p()
        _lib.check(L.dmm_plan_create(C.byref(_make_desc(self, 1, 32, 32)), C.byref(h)))
        try:
            table = _tensor_table(h)
            nparams, nbuf = L.dmm_plan_num_params(h), L.dmm_plan_num_buffer_elems(h)
        finally:
            _lib.check(L.dmm_plan_destroy(h))
        self._table = table
        self._param_arena = torch.zeros(nparams, dtype=torch.float32)
        self._grad_arena = torch.zeros(nparams, dtype=torch.float32)
        self._buffer_arena = torch.zeros(max(nbuf, 1), dtype=torch.float32)
        n_bn = sum(1 for _, k, _, _ in table if k == _lib.T_BN_TRACKED)
        self._tracked_arena = torch.zeros(n_bn, dtype=torch.int64)
        self._hook = torch.zeros(1, requires_grad=True)
        self._slots = []  # (owner module, leaf name, kind, shape, offset)
        ti = 0
        for name, kind, shape, off in table:
            owner = self
            parts = name.split(".")
            for part in parts[:-1]:
                if part not in owner._modules:
                    owner.add_module(part, _Node())
                owner = owner._modules[part]
            leaf = parts[-1]
            n = int(math.prod(shape)) if shape else 1
            if kind <= _lib.T_BN_BIAS:
                owner.register_parameter(leaf, nn.Parameter(self._param_arena[off:off + n].view(shape)))
            elif kind <= _lib.T_BN_VAR:
                owner.register_buffer(leaf, self._buffer_arena[off:off + n].view(shape))
            else:
                owner.register_buffer(leaf, self._tracked_arena[ti])
                off = ti
                ti += 1
            self._slots.append((owner, leaf, kind, shape, off))
        self._init_weights()
        self.num_params = sum(p.numel() for p in self.parameters())
        self._plans = OrderedDict()
        self._last = None
        self._loss = (_lib.LOSS_BCE, None, None)

    # ------------------------------------------------------------------ loss epilogue
    def set_loss(self, kind="bce", alpha=None, gamma=None):
        """Loss computed by loss_backward() / loss_metrics(): "bce" = BCEWithLogitsLoss(reduction='none') (reference A:54);
        "focal" = alpha*(1-exp(-bce))**gamma*bce with scalar or per-class alpha / gamma (reference L:9-91)."""
        if kind == "bce":
            self._loss = (_lib.LOSS_BCE, None, None)
            return self
        if kind != "focal":
            raise ValueError("loss kind must be 'bce' or 'focal'")
        nc = int(self.num_classes)

        def per_class(v, default):
            v = default if v is None else v
            v = [float(v)] * nc if not hasattr(v, "__len__") else [float(e) for e in v]
            if len(v) != nc:
                raise ValueError(f"need {nc} per-class values")
            return tuple(v)
        self._loss = (_lib.LOSS_FOCAL, per_class(alpha, 1.0), per_class(gamma, 2.0))
        return self

    def _apply_loss(self, plan):
        if plan.loss_key == self._loss:
            return
        kind, alpha, gamma = self._loss
        nc = int(self.num_classes)
        a = (C.c_float * nc)(*(alpha or [1.0] * nc))
        g = (C.c_float * nc)(*(gamma or [2.0] * nc))
        _lib.check(_lib.lib().dmm_plan_set_loss(plan.handle, kind, a, g, nc))
        plan.loss_key = self._loss

    # ------------------------------------------------------------------ parameters
    def _init_weights(self):
        """kaiming_normal_ on Conv2d, PyTorch's default (kaiming_uniform a=sqrt(5)) on ConvTranspose2d, BN 1/0
        (reference M:198-205; ConvTranspose2d is not an nn.Conv2d subclass so it keeps its default)."""
        with torch.no_grad():
            for owner, leaf, kind, shape, off in self._slots:
                t = getattr(owner, leaf)
                if kind == _lib.T_CONV:
                    nn.init.kaiming_normal_(t)
                elif kind == _lib.T_CONVT:
                    nn.init.kaiming_uniform_(t, a=math.sqrt(5))
                elif kind == _lib.T_BN_WEIGHT or kind == _lib.T_BN_VAR:
                    t.fill_(1.0)
                elif kind == _lib.T_BN_BIAS or kind == _lib.T_BN_MEAN:
                    t.zero_()

    def _apply(self, fn, recurse=True):
        """Move the arenas, then re-point every parameter / buffer at its slice (keeps them views)."""
        def move(t, dtype):
            r = fn(t)
            return r.to(dtype) if r.dtype != dtype else r
        self._param_arena = move(self._param_arena, torch.float32)
        self._grad_arena = move(self._grad_arena, torch.float32)
        self._buffer_arena = move(self._buffer_arena, torch.float32)
        self._tracked_arena = move(self._tracked_arena, torch.int64)
        self._hook = torch.zeros(1, requires_grad=True, device=self._param_arena.device)
        for owner, leaf, kind, shape, off in self._slots:
            n = int(math.prod(shape)) if shape else 1
            if kind <= _lib.T_BN_BIAS:
                p = owner._parameters[leaf]
                p.data = self._param_arena[off:off + n].view(shape)
                p.grad = None
            elif kind <= _lib.T_BN_VAR:
                owner._buffers[leaf] = self._buffer_arena[off:off + n].view(shape)
            else:
                owner._buffers[leaf] = self._tracked_arena[off]
        self.close()   # the plans were bound to the old arenas
        return self

    def close(self):
        """Release every execution plan of this model (GPU workspaces of several GB each, the library's launch lists) at a known
        point.  The model stays usable: the next forward() builds a new plan.  The reference has no counterpart (it never frees
        anything explicitly, A:442-450); call it when a model is dropped or before the input size changes for good."""
        plans = list(getattr(self, "_plans", {}).values())
        last = getattr(self, "_last", None)
        if last is not None and last[0] not in plans:
            plans.append(last[0])
        self._plans = OrderedDict()
        self._last = None
        err = None
        for pl in plans:
            try:
                pl.close()
            except Exception as e:  # noqa: BLE001 - close them all, then report the first failure
                err = err or e
        if err is not None:
            raise err

    def _attach_grads(self):
        for owner, leaf, kind, shape, off in self._slots:
            if kind <= _lib.T_BN_BIAS:
                p = owner._parameters[leaf]
                if p.grad is None or p.grad.data_ptr() != self._grad_arena.data_ptr() + 4 * off:
                    p.grad = self._grad_arena[off:off + p.numel()].view(shape)

    @property
    def param_arena(self):
        return self._param_arena

    @property
    def grad_arena(self):
        return self._grad_arena

    # ------------------------------------------------------------------ plans
    def _get_plan(self, batch, height, width):
        key = (batch, height, width)
        plan = self._plans.get(key)
        if plan is None:
            while len(self._plans) >= 2:  # each plan owns a multi-GB workspace
                _, old = self._plans.popitem(last=False)
                if self._last is not None and self._last[0] is old:
                    self._last = None
                old.close()
            plan = _Plan(self, batch, height, width)
            self._plans[key] = plan
        else:
            self._plans.move_to_end(key)
        return plan

    # ------------------------------------------------------------------ forward / backward
    def forward(self, stream_1_data, stream_2_data):
        """stream_1_data (B, s1, H, W), stream_2_data (B, s2, H, W) float NCHW -> logits (B, num_classes, H, W)."""
        x1 = stream_1_data
        if not x1.is_cuda:
            raise RuntimeError("dmmfods_amd computes on the GPU only; move the model and inputs to 'cuda' (no CPU fallback)")
        if x1.device != self._param_arena.device:
            raise RuntimeError("model and inputs are on different devices")
        B, c1, H, W = x1.shape
        if c1 != self.stream_1_in_channels:
            raise RuntimeError(f"stream_1 has {c1} channels, expected {self.stream_1_in_channels}")
        x1 = x1.contiguous().float()
        x2 = None
        if self.fusion != "no":
            x2 = stream_2_data
            if self.fusion == "mid":
                assert tuple(x2.shape[2:]) == (H, W) and x2.shape[0] == B, f"{tuple(x1.shape)} {tuple(x2.shape)}"
            elif tuple(x2.shape[2:]) != (H, W) or x2.shape[0] != B:
                raise RuntimeError("Sizes of tensors must match except in dimension 1")
            if x2.shape[1] != self.stream_2_in_channels:
                raise RuntimeError(f"stream_2 has {x2.shape[1]} channels, expected {self.stream_2_in_channels}")
            x2 = x2.contiguous().float()
        plan = self._get_plan(B, H, W)
        plan.note_stream()
        logits = torch.empty(B, self.num_classes, H, W, dtype=torch.float32, device=x1.device)
        _lib.check(_lib.lib().dmm_plan_forward(plan.handle, x1.data_ptr(), x2.data_ptr() if x2 is not None else None,
                                               logits.data_ptr(), 1 if self.training else 0, _lib.stream_ptr()))
        if self.training:
            self._tracked_arena += 1
        # loss_backward() needs the saved activations / batch statistics of a TRAINING forward of this very plan
        self._last = (plan, logits) if self.training else None
        if self.training and torch.is_grad_enabled():
            return _HipBackward.apply(self._hook, logits, self, plan)
        return logits

    def loss_backward(self, target):
        """Fused training tail (reference A:247-264): per-pixel BCE-with-logits on the last forward's logits, metric
        counts, and backward of the SUM of all loss elements.  Returns a dict of device tensors."""
        if self._last is None:
            raise RuntimeError("loss_backward() needs a preceding training-mode forward()")
        plan, logits = self._last
        if plan.closed:
            raise RuntimeError("the plan of the last forward() has been closed")
        plan.note_stream()
        t = target.contiguous().float()
        self._apply_loss(plan)
        _lib.check(_lib.lib().dmm_plan_loss_backward(plan.handle, logits.data_ptr(), t.data_ptr(), plan.metrics.data_ptr(),
                                                     _lib.stream_ptr()))
        self._attach_grads()
        return self._metrics(plan, logits.shape)

    def loss_metrics(self, logits, target):
        """Loss sums and metric counts only (validation, A:345-358)."""
        plan = self._get_plan(logits.shape[0], logits.shape[2], logits.shape[3])
        t = target.contiguous().float()
        lg = logits.detach().contiguous().float()
        self._apply_loss(plan)
        _lib.check(_lib.lib().dmm_plan_loss_metrics(plan.handle, lg.data_ptr(), t.data_ptr(), plan.metrics.data_ptr(),
                                                    _lib.stream_ptr()))
        return self._metrics(plan, logits.shape)

    # ------------------------------------------------------------------ data-parallel hooks
    def grad_buckets(self):
        """[(offset, count)] ranges of grad_arena in the order the last training forward's plan finishes them in backward."""
        if self._last is None:
            raise RuntimeError("grad_buckets() needs a preceding training-mode forward()")
        return list(self._last[0].grad_buckets)

    def grad_bucket_wait(self, index, stream=None):
        """Make `stream` (default: the current stream) wait until bucket `index` of the enqueued backward is final."""
        if self._last is None:
            raise RuntimeError("grad_bucket_wait() needs a preceding training-mode forward()")
        sp = _lib.stream_ptr() if stream is None else C.c_void_p(stream.cuda_stream)
        _lib.check(_lib.lib().dmm_plan_grad_bucket_wait(self._last[0].handle, int(index), sp))

    def _metrics(self, plan, shape):
        B, nc, H, W = shape
        m = plan.metrics.clone()
        per = m[2 * nc:].view(B, 2, nc)
        inter, union = per[:, 0], per[:, 1]
        return {
            "loss_per_class": m[:nc].float(),
            "acc_per_class": (m[nc:2 * nc] / float(B * H * W)).float(),
            "iou_per_instance_per_class": (inter / union).float(),  # 0/0 -> NaN, as in the reference (H:337-341)
            "intersection": inter, "union": union,
        }


_LEGACY_KEY = re.compile(r"^(.*denselayer\d+\.(?:norm|relu|conv))\.([12]\.(?:weight|bias|running_mean|running_var))$")


def _find_checkpoint(source):
    """`source` is a file path or a torchvision arch name looked up as $DMM_PRETRAINED_DIR/<arch>.pth (no download: the
    reference fetches model_urls[arch] from the network, M:284; this build has none)."""
    if isinstance(source, (str, os.PathLike)) and os.path.isfile(source):
        return os.fspath(source)
    root = os.environ.get("DMM_PRETRAINED_DIR")
    if root:
        for ext in (".pth", ".pt"):
            cand = os.path.join(root, f"{source}{ext}")
            if os.path.isfile(cand):
                return cand
    raise RuntimeError(f"no local torchvision checkpoint for {source!r}: pass pretrained=<path to densenetNNN .pth> or set "
                       "DMM_PRETRAINED_DIR (pretrained weights cannot be downloaded in this environment)")


def _load_state_dict(model, config, model_url, progress=True):
    """Initialise the encoder(s) from a torchvision DenseNet checkpoint (reference M:269-309): old-style `norm.1` keys are
    renamed, `features.conv0.weight` is skipped when the stem does not take 3 channels (early fusion or a non-RGB stream 1),
    keys the model does not have (the classifier) are ignored, and for mid fusion the LiDAR stream starts as a copy of the
    RGB stream except for its stem convolution."""
    ckpt = torch.load(_find_checkpoint(model_url), map_location="cpu")
    if isinstance(ckpt, dict) and "state_dict" in ckpt and not any(k.startswith("features.") for k in ckpt):
        ckpt = ckpt["state_dict"]
    renamed = {}
    for key, value in ckpt.items():
        m = _LEGACY_KEY.match(key)
        renamed[m.group(1) + m.group(2) if m else key] = value
    if model.fusion == "early" or model.stream_1_in_channels != 3:
        renamed.pop("features.conv0.weight", None)
    own = model.state_dict()
    picked = {k: v for k, v in renamed.items() if k in own}
    for k, v in picked.items():
        if tuple(v.shape) != tuple(own[k].shape):
            raise RuntimeError(f"checkpoint tensor {k} has shape {tuple(v.shape)}, the model expects {tuple(own[k].shape)}")
    model.load_state_dict(picked, strict=False)
    if model.fusion == "mid":
        rgb = model.features.state_dict()
        lidar_keys = set(model.stream_2_features.state_dict())
        clone = {k: v for k, v in rgb.items() if k != "conv0.weight" and k in lidar_keys}
        model.stream_2_features.load_state_dict(clone, strict=False)
    return sorted(picked)


def _dense_u_net_lidar(arch, growth_rate, block_config, num_init_features, pretrained, progress, config, **kw):
    if config is None:
        config = get_config(os.path.join("content", "mnt", "My Drive", "Colab Notebooks", "DeepCV_Packages"))
    # for compatibility with the original densenet functions the factory overwrites these (reference M:323-325)
    config.model.growth_rate = growth_rate
    config.model.block_config = block_config
    config.model.num_init_features = num_init_features
    model = Dense_U_Net_lidar(config, **kw)
    if pretrained:  # True: $DMM_PRETRAINED_DIR/<arch>.pth; a string: that file
        _load_state_dict(model, config, arch if pretrained is True else pretrained, progress)
    return model


def densenet121_u_lidar(pretrained=False, progress=True, config=None, **kw):
    return _dense_u_net_lidar("densenet121", 32, (6, 12, 24, 16), 64, pretrained, progress, config, **kw)


def densenet161_u_lidar(pretrained=False, progress=True, config=None, **kw):
    return _dense_u_net_lidar("densenet161", 48, (6, 12, 36, 24), 96, pretrained, progress, config, **kw)


def densenet169_u_lidar(pretrained=False, progress=True, config=None, **kw):
    return _dense_u_net_lidar("densenet169", 32, (6, 12, 32, 32), 64, pretrained, progress, config, **kw)


def densenet201_u_lidar(pretrained=False, progress=True, config=None, **kw):
    return _dense_u_net_lidar("densenet201", 32, (6, 12, 48, 32), 64, pretrained, progress, config, **kw)
