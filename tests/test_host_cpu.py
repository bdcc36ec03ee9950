"""CPU-side checks (no GPU): the C-ABI library loads and exports every symbol the header declares, the plan's
state_dict layout equals the reference's (golden G1), host-side error behaviour, and the data-parallel
gradient exchange (world_size 2, gloo)."""
import ctypes as C
import gzip
import hashlib
import json
import os
import re
import subprocess
import sys

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
VARIANTS = {"no": (1, 0), "early": (1, 3), "mid2": (2, 3), "mid3": (3, 3), "mid4": (4, 3)}
DENSENETS = {121: (32, (6, 12, 24, 16), 64), 161: (48, (6, 12, 36, 24), 96), 169: (32, (6, 12, 32, 32), 64),
             201: (32, (6, 12, 48, 32), 64)}


@pytest.fixture(scope="module")
def lib():
    import __graft_entry__ as g
    g.build()
    from dmmfods_amd import _lib
    return _lib


def _cfg(depth, variant):
    from dmmfods_amd.utils.Dense_U_Net_lidar_helper import get_config
    cfg = get_config("/tmp/dmm_test")
    k, bc, nif = DENSENETS[depth]
    cfg.model.growth_rate, cfg.model.block_config, cfg.model.num_init_features = k, bc, nif
    cfg.model.concat_before_block_num, cfg.model.stream_2_in_channels = VARIANTS[variant]
    return cfg


def test_header_symbols_exported(lib):
    hdr = open(os.path.join(ROOT, "include", "dmmfods_hip.h")).read()
    declared = set(re.findall(r"\b(dmm_[a-z0-9_]+)\s*\(", hdr))
    declared -= {"dmm_status"}
    L = lib.lib()
    missing = [s for s in sorted(declared) if not hasattr(L, s)]
    assert not missing, missing
    assert set(lib.EXPORTS) <= declared


def test_state_dict_layout_matches_reference(lib, golden_dir):
    from dmmfods_amd.graphs.models import Dense_U_Net_lidar as M
    g1 = json.load(gzip.open(os.path.join(golden_dir, "g1_topology.json.gz"), "rt"))
    for depth in (121, 169, 201, 161):
        for v in VARIANTS:
            factory = getattr(M, f"densenet{depth}_u_lidar")
            model = factory(pretrained=False, config=_cfg(depth, v))
            sd = model.state_dict()
            blob = ";".join(f"{k}:{','.join(map(str, t.shape))}" for k, t in sd.items()).encode()
            ent = g1[f"d{depth}_{v}"]
            assert hashlib.sha256(blob).hexdigest() == ent["sha256"], (depth, v)
            assert model.num_params == ent["num_params"]
            assert model.fusion == ent["fusion"]
            assert [k for k, _ in model.named_parameters()] == [k for k, s in (ent.get("keys") or [[k, None] for k, _ in model.named_parameters()])
                                                                if not k.endswith(("running_mean", "running_var", "num_batches_tracked"))]


def test_parameters_are_arena_views_and_survive_load(lib):
    from dmmfods_amd.graphs.models.Dense_U_Net_lidar import densenet121_u_lidar
    m = densenet121_u_lidar(config=_cfg(121, "mid3"))
    base = m.param_arena.data_ptr()
    off = 0
    for p in m.parameters():
        assert p.data_ptr() == base + 4 * off
        off += p.numel()
    assert off == m.param_arena.numel() == 23567564
    sd = {k: torch.full_like(v, 0.5) if v.is_floating_point() else v for k, v in m.state_dict().items()}
    m.load_state_dict(sd)
    assert float(m.param_arena.min()) == 0.5 and float(m.param_arena.max()) == 0.5
    # reference init statistics: conv kaiming_normal (std sqrt(2/fan_in)), ConvTranspose default (std ~ sqrt(1/(3 fan_in)))
    m2 = densenet121_u_lidar(config=_cfg(121, "no"))
    w = m2.features.denseblock1.denselayer1.conv2.weight
    assert abs(w.std().item() / (2.0 / (128 * 9)) ** 0.5 - 1) < 0.05
    wt = m2.decoder.Transposed_Convolution_1.weight
    assert abs(wt.std().item() / (1.0 / (3 * 1024 * 9)) ** 0.5 - 1) < 0.05


def _fake_torchvision_checkpoint(depth, path):
    """A checkpoint with torchvision's densenet layout: `features.*` keys in the pre-0.3 spelling (`norm.1.weight` ...),
    a 3-channel stem and a classifier; values are random so that every tensor can be recognised after loading."""
    import re
    from dmmfods_amd.graphs.models.Dense_U_Net_lidar import Dense_U_Net_lidar
    donor = Dense_U_Net_lidar(_cfg(depth, "no"))
    g = torch.Generator().manual_seed(depth)
    ckpt = {}
    for k, v in donor.state_dict().items():
        if not k.startswith("features."):
            continue
        old = re.sub(r"(denselayer\d+\.)(norm|conv)([12])\.", r"\1\2.\3.", k)
        ckpt[old] = torch.randn(v.shape, generator=g) if v.is_floating_point() else v.clone()
    ckpt["classifier.weight"] = torch.randn(1000, 1024, generator=g)
    ckpt["classifier.bias"] = torch.randn(1000, generator=g)
    torch.save(ckpt, path)
    return ckpt


@pytest.mark.parametrize("variant", ["no", "early", "mid3"])
def test_pretrained_torchvision_checkpoint_loading(lib, tmp_path, monkeypatch, variant):
    """Reference M:269-309: legacy keys renamed, conv0 dropped unless the stem takes 3 channels, classifier ignored,
    stream 2 cloned from stream 1 (except conv0) for mid fusion."""
    import re
    from dmmfods_amd.graphs.models.Dense_U_Net_lidar import densenet121_u_lidar
    path = tmp_path / "densenet121.pth"
    ckpt = _fake_torchvision_checkpoint(121, path)
    fresh = densenet121_u_lidar(config=_cfg(121, variant))
    before = {k: v.clone() for k, v in fresh.state_dict().items()}
    monkeypatch.setenv("DMM_PRETRAINED_DIR", str(tmp_path))
    m = densenet121_u_lidar(pretrained=True, config=_cfg(121, variant))
    m2 = densenet121_u_lidar(pretrained=str(path), config=_cfg(121, variant))
    sd, sd2 = m.state_dict(), m2.state_dict()
    assert all(torch.equal(sd[k], sd2[k]) for k in sd if k.startswith("features.denseblock"))   # both spellings of `pretrained`
    new = lambda k: re.sub(r"(denselayer\d+\.)(norm|conv)\.([12])\.", r"\1\2\3.", k)
    own_features = [k for k in sd if k.startswith("features.")]
    loaded = 0
    for old_key, v in ckpt.items():
        k = new(old_key)
        if k.startswith("classifier"):
            assert k not in sd
            continue
        if k not in sd:            # mid fusion: stream 1 ends at the fusion point... it does not: features.* is the full encoder
            raise AssertionError(k)
        if k == "features.conv0.weight" and variant == "early":
            assert sd[k].shape[1] == 6 and not torch.equal(sd[k][:, :3], v)   # 6-channel stem keeps its own init
            continue
        assert torch.equal(sd[k], v), k
        loaded += 1
    assert loaded == len(own_features) - (1 if variant == "early" else 0)
    # arena views survive: parameters still alias the flat arena
    assert m.features.conv0.weight.data_ptr() == m.param_arena.data_ptr()
    if variant == "mid3":
        s2 = {k: v for k, v in sd.items() if k.startswith("stream_2_features.")}
        assert s2
        for k, v in s2.items():
            twin = "features." + k[len("stream_2_features."):]
            if k.endswith("conv0.weight"):
                assert v.shape == before[k].shape            # LiDAR stem: own shape, own initialisation
            elif not k.endswith("num_batches_tracked"):
                assert torch.equal(v, sd[twin]), k
    # untouched parts keep a fresh initialisation's statistics (decoder, head)
    assert abs(sd["decoder.Transposed_Convolution_1.weight"].std().item() / before["decoder.Transposed_Convolution_1.weight"].std().item() - 1) < 0.05
    monkeypatch.delenv("DMM_PRETRAINED_DIR")
    with pytest.raises(RuntimeError):
        densenet121_u_lidar(pretrained=True, config=_cfg(121, variant))


def test_error_behaviour_matches_reference(lib):
    from dmmfods_amd.graphs.models.Dense_U_Net_lidar import Dense_U_Net_lidar
    cfg = _cfg(121, "no")
    cfg.model.concat_before_block_num = 7
    with pytest.raises(AttributeError):     # reference M:65
        Dense_U_Net_lidar(cfg)
    m = Dense_U_Net_lidar(_cfg(121, "no"))
    with pytest.raises(RuntimeError, match="GPU only"):   # no CPU fallback
        m(torch.zeros(1, 3, 64, 64), None)
    d = lib.ModelDesc(growth_rate=32, num_blocks=4, num_init_features=64, bn_size=4, num_classes=3, concat_before_block_num=1,
                      stream_1_in_channels=3, stream_2_in_channels=0, batch=1, height=100, width=100, dtype=0, loss_scale=1.0,
                      bn_momentum=0.1, bn_eps=1e-5, iou_threshold=0.7, use_mfma=1)
    for i, v in enumerate((6, 12, 24, 16)):
        d.block_config[i] = v
    h = C.c_void_p()
    with pytest.raises(ValueError):         # reference: ValueError from ConvTranspose2d(output_size=...) (M:261)
        lib.check(lib.lib().dmm_plan_create(C.byref(d), C.byref(h)))


def test_conv5_entry_point_refuses_other_shapes_without_a_gpu(lib):
    """dmm_conv5_wgrad_stats (round 5) serves ONE shape family - the 5x5, pad 2, stride 1 convolution of 64 channels onto <= 4 behind
    BN+ReLU in 16-bit storage - and says so before it touches the HIP runtime: DMM_ERR_INVALID with a message, never a launch with a
    layout the kernel does not have (five classes would need a fifth column per tap)."""
    L = lib.lib()
    ok = dict(dtype=1, use_mfma=1, B=1, H=16, W=16, Cin=64, Cout=3, R=5, S=5, stride=1, pad=2, transposed=0, mode=0, bn_relu=1)
    for change in (dict(Cout=5), dict(Cin=128), dict(R=3, S=3, pad=1), dict(dtype=0), dict(bn_relu=0), dict(stride=2), dict(use_mfma=0)):
        d = lib.ConvDesc(**{**ok, **change})
        rc = L.dmm_conv5_wgrad_stats(C.byref(d), None, None, None, None, None, None, None, None, None)
        assert rc != 0 and b"5x5" in L.dmm_last_error(), (change, rc, L.dmm_last_error())


def test_plan_flops_match_survey(lib):
    L = lib.lib()
    for (cbb, s2, want) in ((1, 3, 938.7), (3, 3, 1133.1)):
        d = lib.ModelDesc(growth_rate=32, num_blocks=4, num_init_features=64, bn_size=4, num_classes=3, concat_before_block_num=cbb,
                          stream_1_in_channels=3, stream_2_in_channels=s2, batch=4, height=1280, width=1920, dtype=1, loss_scale=1.0,
                          bn_momentum=0.1, bn_eps=1e-5, iou_threshold=0.7, use_mfma=1)
        for i, v in enumerate((6, 12, 24, 16)):
            d.block_config[i] = v
        h = C.c_void_p()
        lib.check(L.dmm_plan_create(C.byref(d), C.byref(h)))
        assert abs(L.dmm_plan_forward_flops(h) / 4 / 1e9 - want) < 0.5
        assert L.dmm_plan_workspace_bytes(h) < 40 * 2**30
        lib.check(L.dmm_plan_destroy(h))


def test_plan_sizing_is_deterministic_and_independent_of_schedule_switches(lib, monkeypatch):
    """dmm_plan_create needs no GPU and sizes the workspace in a dry pass that the bound pass must repeat byte for byte.  Round 4 added
    tables whose size depends on what the plan decides (pack / unpack tile lists, the early / late pack cut, the interleaved encoders'
    launch order): two plans of one description must agree, and the A/B switches that only change the ORDER of launches must not change
    the size - a plan sized with one setting and bound with another would overrun its tables."""
    L = lib.lib()

    def size(cbb, s2, blocks, dtype):
        d = lib.ModelDesc(growth_rate=32, num_blocks=4, num_init_features=64, bn_size=4, num_classes=3, concat_before_block_num=cbb,
                          stream_1_in_channels=3, stream_2_in_channels=s2, batch=2, height=256, width=384, dtype=dtype, loss_scale=1.0,
                          bn_momentum=0.1, bn_eps=1e-5, iou_threshold=0.7, use_mfma=1)
        for i, v in enumerate(blocks):
            d.block_config[i] = v
        h = C.c_void_p()
        lib.check(L.dmm_plan_create(C.byref(d), C.byref(h)))
        n = L.dmm_plan_workspace_bytes(h)
        lib.check(L.dmm_plan_destroy(h))
        return n

    for cfg in ((1, 3, (6, 12, 24, 16), 1), (3, 3, (6, 12, 24, 16), 1), (3, 3, (6, 12, 48, 32), 2), (1, 0, (6, 12, 24, 16), 0)):
        base = size(*cfg)
        assert base > 0 and size(*cfg) == base
        for knob in ("DMM_NO_S2_INTERLEAVE", "DMM_NO_CVP_MERGE", "DMM_NO_C3_MERGE", "DMM_NO_WGP_MERGE"):
            monkeypatch.setenv(knob, "1")
            assert size(*cfg) == base, (cfg, knob)
            monkeypatch.delenv(knob)


@pytest.fixture(scope="module")
def host_drive(tmp_path_factory):
    """tools/hoststub: the library's HOST code (plan.cpp, capi.cpp and the host side of every kernel file) built for the CPU with
    AddressSanitizer + UBSan against a fake HIP runtime that counts teardown violations, plus its driver."""
    out = os.path.join(ROOT, "tools", "hoststub", "_build")
    subprocess.run([os.path.join(ROOT, "tools", "hoststub", "build.sh"), out], check=True, capture_output=True, timeout=900)
    return os.path.join(out, "drive")


HOST_DRIVE_CASES = [
    # (arch, dtype, batch, H, W, environment) - the networks, sizes and switch combinations the GPU tests build plans with
    ("tiny_mid", "bf16", 2, 96, 160, {}),                                    # the plan pair of the round-4 teardown crash ...
    ("tiny_mid", "bf16", 2, 96, 160, {"DMM_NO_PACK_TILES": "1"}),            # ... and its generic-kernel twin
    ("tiny_mid", "f16", 2, 128, 192, {"DMM_PACK_CUT": "16"}),
    ("tiny_mid", "f16", 2, 128, 192, {"DMM_PACK_CUT": "1", "DMM_NO_S2_INTERLEAVE": "1"}),
    ("tiny_early", "f16", 2, 64, 96, {"DMM_NO_TWO_PASS": "1", "DMM_NO_EFF_COMPACT": "1"}),
    ("tiny_no", "f32", 2, 64, 96, {}),
    ("g8_mid", "f32", 2, 64, 96, {}),                                        # smoke()'s network
    ("d121e", "f16", 2, 64, 96, {}),
    ("d121e", "f16", 2, 64, 96, {"DMM_NO_HF": "1"}),
    ("d121e", "bf16", 2, 64, 96, {"DMM_NO_RAW_STATS": "1"}),
    ("d121e", "f16", 2, 64, 96, {"DMM_NO_R1_STATS": "1"}),
    ("d121e", "f16", 2, 64, 96, {"DMM_NO_HF": "1", "DMM_NO_C3_MERGE": "1", "DMM_NO_CVP_MERGE": "1", "DMM_NO_WGP_MERGE": "1"}),
    ("d121e", "f16", 2, 64, 96, {"DMM_DEFER_WGRAD": "1", "DMM_NO_WGP_MERGE": "1"}),
    ("d121m", "f16", 2, 128, 192, {}),
    ("d121m2", "f16", 2, 64, 96, {}),
    ("d169m", "f16", 2, 64, 96, {}),
    ("d201m", "bf16", 2, 64, 96, {}),
    ("d161m", "f16", 1, 64, 96, {}),
    ("d121n", "f32", 1, 256, 384, {}),                                        # C1
    ("d121e", "f16", 4, 320, 480, {}),                                        # C2's launch geometry classes at a quarter of the size
]


def test_plan_life_under_sanitizers_and_the_teardown_contract(host_drive):
    """VERDICT round 4, item 1.  Every case: create -> bind -> training steps -> external-gradient backward -> eval -> metrics -> bucket
    waits -> profiled passes -> one more step left unsynchronised -> destroy, three plans in a row in one process.  Must hold:
    no AddressSanitizer / UBSan report in the sizing or the bound pass; every device pointer of every launch record inside the
    workspace or a caller arena; the bound pass takes the bytes the sizing pass reported (dmm_plan_bind refuses otherwise);
    dmm_plan_destroy returns DMM_OK having synchronised the helper streams it used - no stream or event is ever destroyed with work
    behind it (the fake runtime counts that), in fact no stream is destroyed at all and the SECOND and THIRD plan create no stream
    and no event (process pool)."""
    for arch, dtype, b, h, w, envx in HOST_DRIVE_CASES:
        env = {k: v for k, v in os.environ.items() if not k.startswith("DMM_")}
        env.update(envx, ASAN_OPTIONS="detect_leaks=0:abort_on_error=0", UBSAN_OPTIONS="print_stacktrace=1")
        r = subprocess.run([host_drive, arch, dtype, str(b), str(h), str(w), "3"], env=env, capture_output=True, text=True, timeout=600)
        tail = (r.stdout + r.stderr)[-3000:]
        assert r.returncode == 0 and "DRIVE OK" in r.stdout, (arch, dtype, envx, tail)
        lives = [ln for ln in r.stdout.splitlines() if ln.startswith("life ")]
        assert len(lives) == 3 and all(", 0 bad," in ln and ln.endswith("violations 0") for ln in lives), lives
        assert "streams alive 2 (created 2)" in lives[-1], lives[-1]          # the side and the pack stream, once per process


def test_bind_checks_the_bound_pass_against_the_sizing_pass(host_drive):
    """nothing compared what the bound pass took with what the sizing pass had reported.  Now (a) the sizing
    pass and the bound pass read the SAME switches (PlanSwitches, stored in the plan by dmm_plan_create - an environment variable set
    between create and bind cannot reach the bound pass) and run with the same null / non-null pattern of pointers; (b) every buffer is
    reserved by shape alone, so a kernel family switched off through dmm_set_option between the two calls changes launches, not bytes
    (checked here for every family); (c) dmm_plan_bind compares the three region sizes and returns DMM_ERR_STATE on any difference
    - provoked here by flipping a switch inside the plan after it was sized."""
    base = {k: v for k, v in os.environ.items() if not k.startswith("DMM_")}
    base.update(ASAN_OPTIONS="detect_leaks=0")
    for fam in ("conv3", "wg3", "bw1", "cvp", "pig", "wgp", "wg5", "thin_logits"):
        r = subprocess.run([host_drive, "d121e", "f16", "2", "64", "96", "1"], env=dict(base, DRIVE_TOGGLE_BETWEEN_CREATE_AND_BIND=fam),
                           capture_output=True, text=True, timeout=600)
        assert r.returncode == 0 and "DRIVE OK" in r.stdout, (fam, (r.stdout + r.stderr)[-2000:])
    r = subprocess.run([host_drive, "d121e", "f16", "2", "64", "96", "1"], env=dict(base, DRIVE_FLIP_SWITCH_BETWEEN_CREATE_AND_BIND="1"),
                       capture_output=True, text=True, timeout=600)
    assert r.returncode == 2 and "-> -4" in r.stderr and "sizing pass" in r.stderr, (r.stdout + r.stderr)[-2000:]


def test_config_fields_match_reference():
    from dmmfods_amd.utils.Dense_U_Net_lidar_helper import create_config, get_config
    cfg = get_config("/tmp/x")
    assert cfg.model.growth_rate == 32 and tuple(cfg.model.block_config) == (6, 12, 24, 16)
    assert cfg.model.concat_before_block_num == 2 and cfg.model.stream_2_in_channels == 1
    assert cfg.optimizer.learning_rate == 1e-3 and cfg.optimizer.beta2 == 0.999 and cfg.agent.iou_threshold == 0.7
    assert cfg.agent.checkpoint.state_dict == "state_dict" and cfg.agent.seed == 123
    assert set(create_config("/tmp/x")) == {"dir", "scripts", "model", "loss", "loader", "optimizer", "dataset", "agent"}


def _plan_buckets(lib, cbb, s2, bc=(6, 12, 24, 16)):
    L = lib.lib()
    d = lib.ModelDesc(growth_rate=32, num_blocks=4, num_init_features=64, bn_size=4, num_classes=3, concat_before_block_num=cbb,
                      stream_1_in_channels=3, stream_2_in_channels=s2, batch=1, height=64, width=96, dtype=1, loss_scale=1.0,
                      bn_momentum=0.1, bn_eps=1e-5, iou_threshold=0.7, use_mfma=1)
    for i, v in enumerate(bc):
        d.block_config[i] = v
    h = C.c_void_p()
    lib.check(L.dmm_plan_create(C.byref(d), C.byref(h)))
    out = []
    for i in range(L.dmm_plan_num_grad_buckets(h)):
        o, c = C.c_int64(), C.c_int64()
        lib.check(L.dmm_plan_grad_bucket(h, i, C.byref(o), C.byref(c)))
        out.append((o.value, c.value))
    n = L.dmm_plan_num_params(h)
    table = []
    for i in range(L.dmm_plan_num_tensors(h)):
        name, kind, nd = C.c_char_p(), C.c_int32(), C.c_int32()
        shape, off = (C.c_int64 * 4)(), C.c_int64()
        lib.check(L.dmm_plan_tensor_info(h, i, C.byref(name), C.byref(kind), C.byref(nd), C.byref(shape), C.byref(off)))
        table.append((name.value.decode(), kind.value, off.value))
    lib.check(L.dmm_plan_destroy(h))
    return out, n, table


def test_grad_buckets_cover_arena_in_backward_order(lib):
    """The plan's data-parallel buckets: whole tensors, an exact partition of the gradient arena, listed in the order backward
    finishes them (decoder/head before block 4 ... before the stems; the second stream of a mid-fusion net last)."""
    for cbb, s2 in ((1, 3), (3, 3)):
        buckets, n, table = _plan_buckets(lib, cbb, s2)
        assert len(buckets) >= 3
        spans = sorted(buckets)
        assert spans[0][0] == 0 and sum(c for _, c in spans) == n
        for (o0, c0), (o1, _) in zip(spans, spans[1:]):
            assert o0 + c0 == o1                                   # no gap, no overlap
        starts = {off for _, kind, off in table if kind <= lib.T_BN_BIAS}
        assert all(o in starts for o, _ in buckets)                # buckets begin at tensor boundaries
        where = {name: off for name, kind, off in table if kind <= lib.T_BN_BIAS}

        def pos(name):
            off = where[name]
            return next(i for i, (o, c) in enumerate(buckets) if o <= off < o + c)
        assert pos("decoder.Transposed_Convolution_4.weight") <= pos("features.denseblock4.denselayer16.conv2.weight")
        assert pos("features.denseblock4.denselayer16.conv2.weight") <= pos("features.denseblock1.denselayer1.conv1.weight")
        assert pos("features.conv0.weight") >= pos("features.denseblock3.denselayer1.conv1.weight")
        if cbb == 3:
            assert pos("stream_2_features.conv0.weight") == len(buckets) - 1
        # cuts inside the encoder (round 4): blocks 4 and 3 do not wait for conv0, and what is left behind the end of backward -
        # the bucket that becomes ready last - is the stem with blocks 1-2 (5.3 MB), not the whole encoder (25 MB in round 3)
        assert pos("features.denseblock4.denselayer1.conv1.weight") < pos("features.denseblock3.denselayer1.conv1.weight") < pos("features.conv0.weight")
        assert buckets[-1][1] * 4 <= 8 << 20, [round(c * 4 / 2 ** 20, 1) for _, c in buckets]
        assert max(c for _, c in buckets) * 4 <= 40 << 20                # the largest: the decoder's first ConvTranspose alone (37.7 MB)


_DP_SCRIPT = r"""
import json, os, sys, torch, torch.distributed as dist
sys.path.insert(0, %r)
from dmmfods_amd.parallel import GradAllReduce, broadcast_parameters
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
dist.init_process_group("gloo", rank=rank, world_size=world)
buckets, n = json.loads(os.environ["DMM_BUCKETS"]), int(os.environ["DMM_NPARAMS"])

class FakeModel:            # the surface GradAllReduce uses: grad_arena, grad_buckets(), param_arena / _buffer_arena
    def __init__(self):
        g = torch.Generator().manual_seed(1234 + rank)
        self.grad_arena = torch.randn(n, generator=g)
        self.param_arena = torch.full((n,), float(rank))
        self._buffer_arena = torch.full((17,), float(rank))
    def grad_buckets(self):
        return [tuple(b) for b in buckets]

m = FakeModel()
mine = m.grad_arena.clone()
r = GradAllReduce(m)
works = r.reduce_overlapped()                      # one collective per plan bucket, in readiness order
assert len(works) == len(buckets) and r.last_ranges == [tuple(b) for b in buckets]
GradAllReduce.wait(works)
other = torch.randn(n, generator=torch.Generator().manual_seed(1234 + (1 - rank)))
assert torch.equal(m.grad_arena, mine + other) or torch.equal(m.grad_arena, other + mine), rank     # SUM, every element once
broadcast_parameters(m, src=0)
assert float(m.param_arena.max()) == 0.0 and float(m._buffer_arena.max()) == 0.0
# the plain post-backward exchange with a bucket size that does not divide the arena
a2 = torch.arange(n, dtype=torch.float32) * (rank + 1)
r2 = GradAllReduce(a2, bucket_bytes=(7 << 20) + 12)
r2.all_reduce()
assert r2.last_ranges[-1][0] + r2.last_ranges[-1][1] == n and r2.last_ranges[-1][1] != r2.last_ranges[0][1]
assert torch.equal(a2, torch.arange(n, dtype=torch.float32) * 3)
dist.barrier(); dist.destroy_process_group()
print("rank", rank, "ok")
"""


def test_grad_allreduce_real_size_arena_gloo_world2(lib, tmp_path):
    """World-size-2 exchange over an arena of the C3 model's real size (23.6 M elements) cut into the plan's own buckets:
    coverage, order, SUM convention and the parameter broadcast."""
    import json
    buckets, n, _ = _plan_buckets(lib, 3, 3)
    assert n == 23567564
    script = tmp_path / "dp.py"
    script.write_text(_DP_SCRIPT % ROOT)
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29531", WORLD_SIZE="2", DMM_BUCKETS=json.dumps(buckets),
               DMM_NPARAMS=str(n))
    procs = [subprocess.Popen([sys.executable, str(script)], env=dict(env, RANK=str(r)), stdout=subprocess.PIPE,
                              stderr=subprocess.STDOUT) for r in range(2)]
    outs = [p.communicate(timeout=300)[0].decode() for p in procs]
    assert all(p.returncode == 0 for p in procs), outs


def test_inline_assembly_loads_are_waited_for_before_use():
    """wg3.hip's loader waves issue their global loads and the matching vmcnt waits from inline assembly (hipcc's own wait counting
    drained the second register set: see the comment in the kernel).  Nothing but the hand-written wait then orders a use behind the
    arrival of the data, so the ISA is checked: tools/check_asm_loads.py follows every path of the compiled kernel with the queue of
    loads in flight and fails if any instruction touches a register whose load has not been retired by a wait."""
    import shutil
    import subprocess
    import sys
    if shutil.which("hipcc") is None:
        pytest.skip("hipcc not on PATH")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, "tools", "check_asm_loads.py"),
                        os.path.join(root, "dmmfods_amd", "csrc", "wg3.hip"), "wg3_kernel"], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-2000:]
    assert r.stdout.count("0 violations") == 4, r.stdout     # f16 / bf16 x effective-gradient / materialised
    # hf.hip (round 4): the same loader pattern; the checker also reports scalar loads with a register offset (the form hipcc built for
    # a kernel-argument array indexed by the phase: misaligned base, wrong pointer - found at hf.hip's bring-up)
    r = subprocess.run([sys.executable, os.path.join(root, "tools", "check_asm_loads.py"),
                        os.path.join(root, "dmmfods_amd", "csrc", "hf.hip"), "hf_kernel"], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-2000:]
    assert r.stdout.count("0 violations") == 2, r.stdout     # f16 / bf16
    r = subprocess.run([sys.executable, os.path.join(root, "tools", "check_asm_loads.py"),
                        os.path.join(root, "dmmfods_amd", "csrc", "cf.hip"), "cf_kernel"], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-2000:]
    assert r.stdout.count("0 violations") == 2, r.stdout     # cf.hip (the dense 3x3 forward on the large maps): f16 / bf16


def test_fold_lane_algebra():
    """gather.h fold_to_lds (the per-channel reductions of every 16-bit epilogue) emulated lane by lane: v_permlane32_swap /
    v_permlane16_swap / DPP row_ror semantics as documented there, for every (slot columns, slot width) pair the kernels
    instantiate - the totals each LDS accumulator receives must be the column sums.  (Round 4: part of running down the round-3
    'fold anomaly' - the helper's algebra is exact; see the comment at the fold in igemm.hip.)"""
    import numpy as np

    def swap32(a, b):
        out = np.empty(64)
        out[:32] = a[:32] + a[32:]
        out[32:] = b[:32] + b[32:]
        return out

    def swap16(a, b):
        out = np.empty(64)
        for row in range(4):
            sl = slice(16 * row, 16 * row + 16)
            if row % 2 == 0:
                out[sl] = a[sl] + a[16 * (row + 1):16 * (row + 2)]
            else:
                out[sl] = b[16 * (row - 1):16 * row] + b[sl]
        return out

    def ror(v, n):
        return np.array([v[(l // 16) * 16 + (l % 16 + n) % 16] for l in range(64)])

    def pick(lane):
        g = lane >> 4
        return ((g & 1) << 1) | (g >> 1)

    rng = np.random.default_rng(0)
    for ncv, slot in [(32, 4), (16, 4), (8, 4), (16, 8), (8, 8), (4, 8)]:
        s1, s2 = rng.standard_normal((64, slot)), rng.standard_normal((64, slot))
        red = np.zeros(2 * slot * ncv)
        v = [s1[:, e].copy() for e in range(slot)] + [s2[:, e].copy() for e in range(slot)]
        cv = np.arange(64) % ncv
        if ncv == 32:
            for k in range(slot):
                t = swap32(v[2 * k], v[2 * k + 1])
                for l in range(64):
                    red[(2 * k + (l >> 5)) * ncv + cv[l]] += t[l]
        else:
            if ncv <= 8:
                v = [x + ror(x, 8) for x in v]
            if ncv <= 4:
                v = [x + ror(x, 4) for x in v]
            for m in range(slot // 2):
                w = swap16(swap32(v[4 * m], v[4 * m + 1]), swap32(v[4 * m + 2], v[4 * m + 3]))
                for l in range(64):
                    if (l & 15) < ncv:
                        red[(4 * m + pick(l)) * ncv + cv[l]] += w[l]
        want = np.zeros_like(red)
        for l in range(64):
            for e in range(slot):
                want[e * ncv + l % ncv] += s1[l, e]
                want[(slot + e) * ncv + l % ncv] += s2[l, e]
        assert np.abs(red - want).max() < 1e-12, (ncv, slot)
