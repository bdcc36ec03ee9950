"""Per-kernel parity on the GPU, through the C ABI: implicit-GEMM forward (+BN statistics epilogue), weight gradient
and BN/ReLU-fused data gradient of every convolution flavour on the hot path, against torch.nn.functional on the
CPU (fp32 reference of the same op).  Tolerances: fp32 2e-5, fp16 3e-3, bf16 2.5e-2 relative to the tensor's max."""
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def lab():
    assert torch.cuda.is_available()
    from tools import gpu_lab
    return gpu_lab


def _cases():
    from tools.gpu_lab import CASES
    return CASES


@pytest.mark.parametrize("dtype", [0, 1, 2], ids=["fp32", "fp16", "bf16"])
@pytest.mark.parametrize("mfma", [1, 0], ids=["mfma", "scalar"])
@pytest.mark.parametrize("case", range(12))
def test_conv_fwd_wgrad_dgrad(lab, case, mfma, dtype):
    c = lab.CASES[case]
    if dtype == 2 and not mfma:
        pytest.skip("bf16 has no scalar check kernels (the library rejects use_mfma=0 with DMM_BF16)")
    assert lab.conv_case(c[0], dtype, mfma, *c[1:])


@pytest.mark.parametrize("dtype", [0, 1, 2], ids=["fp32", "fp16", "bf16"])
def test_ragged_and_tail_shapes(lab, dtype):
    # M not a multiple of the 128-row tile, K tails, N tails, single-pixel maps
    for c in [("1x1 tail M=35", 1, 5, 7, 24, 40, 1, 1, 1, 0, 0, 0, 1),
              ("3x3 1-pixel", 1, 1, 1, 16, 8, 3, 3, 1, 1, 0, 0, 1),
              ("5x5 N=8 tail", 3, 3, 5, 8, 8, 5, 5, 1, 2, 0, 0, 1),
              ("convT 1x1 map", 2, 1, 1, 8, 8, 3, 3, 2, 1, 1, 0, 1),
              ("pool2 2x2 map", 1, 2, 2, 16, 8, 1, 1, 1, 0, 0, 2, 1)]:
        assert lab.conv_case(c[0], dtype, 1, *c[1:])


@pytest.mark.parametrize("dtype", [1, 2], ids=["fp16", "bf16"])
def test_parity_phase_weight_gradients_on_lds_tiles(lab, dtype):
    """wgp.hip (all taps of a ConvTranspose / upsampled-3x3 phase per LDS tile): shapes that take it - input channels a multiple of
    128, output a multiple of 64 - with ragged 8x16 tiles, several channel tiles and both accumulator shapes, against the torch
    reference AND against the generic kernel on identical operands."""
    import ctypes as C
    from dmmfods_amd import _lib
    L = _lib.lib()
    cases = [("convT 128->64 ragged", 2, 12, 20, 128, 64, 3, 3, 2, 1, 1, 0, 1),
             ("convT 256->128", 1, 9, 17, 256, 128, 3, 3, 2, 1, 1, 0, 1),
             ("convT 128->192", 1, 8, 16, 128, 192, 3, 3, 2, 1, 1, 0, 1),
             ("convT 384->256 ragged", 2, 11, 19, 384, 256, 3, 3, 2, 1, 1, 0, 1),   # forward on cvp.hip: 3 channel groups, 2 column tiles
             ("convT 128->128 1 tile", 1, 3, 5, 128, 128, 3, 3, 2, 1, 1, 0, 1),
             ("up2 3x3 128->64", 2, 10, 18, 128, 64, 3, 3, 1, 1, 0, 1, 1)]
    for c in cases:
        assert lab.conv_case(c[0], dtype, 1, *c[1:])
    try:
        _lib.check(L.dmm_set_option(b"wgp", 0))
        _lib.check(L.dmm_set_option(b"cvp", 0))
        for c in cases:
            assert lab.conv_case(c[0] + " (generic)", dtype, 1, *c[1:])
    finally:
        _lib.check(L.dmm_set_option(b"wgp", 1))
        _lib.check(L.dmm_set_option(b"cvp", 1))


def test_lane_swap_fold_helper_alone(tmp_path):
    """gather.h fold_to_lds (DPP row rotations + v_permlane32_swap / v_permlane16_swap) against a plain reduction for every
    (slot columns, slot width) pair the kernels instantiate, full and partial column validity: tools/probes/fold_probe.hip, built here
    with the toolchain of the box (the helper is header code; the probe instantiates combinations no shipped kernel uses any more)."""
    import os, shutil, subprocess
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        pytest.skip("no hipcc on this box")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = str(tmp_path / "fold_probe")
    subprocess.run([hipcc, "-O3", "-std=c++17", "--offload-arch=gfx950", "-munsafe-fp-atomics", "-Wno-unused-result",
                    "-I", os.path.join(root, "dmmfods_amd", "csrc"), os.path.join(root, "tools", "probes", "fold_probe.hip"), "-o", exe],
                   check=True, capture_output=True, timeout=600)
    out = subprocess.run([exe], capture_output=True, text=True, timeout=120)
    assert out.returncode == 0 and "FAIL" not in out.stdout, out.stdout + out.stderr
    assert out.stdout.count(": ok") >= 11, out.stdout
