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
