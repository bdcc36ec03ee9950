#!/usr/bin/env python3
"""Times single convolution launches through the C ABI (no reference check): tools/conv_time.py [reps]
Used with DMM_LIB_PATH pointing at experiment builds (IGEMM_DBG ablations)."""
import ctypes as C
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from dmmfods_amd import _lib  # noqa: E402

L = _lib.lib()
DEV = "cuda"
CASES = [  # name, B,H,W,Cin,Cout,R,S,stride,pad,transposed
    ("TC_1 convT 1024->512 40x60", 4, 40, 60, 1024, 512, 3, 3, 2, 1, 1),
    ("TC_3 convT 256->128 160x240", 4, 160, 240, 256, 128, 3, 3, 2, 1, 1),
    ("b3 1x1 640->128 80x120", 4, 80, 120, 640, 128, 1, 1, 1, 0, 0),
    ("b4 1x1 768->128 40x60", 4, 40, 60, 768, 128, 1, 1, 1, 0, 0),
    ("b2 1x1 320->128 160x240", 4, 160, 240, 320, 128, 1, 1, 1, 0, 0),
    ("b1 1x1 160->128 320x480", 4, 320, 480, 160, 128, 1, 1, 1, 0, 0),
]


def main():
    reps = int(sys.argv[1]) if len(sys.argv) > 1 else 20
    for name, B, H, W, Cin, Cout, R, S, stride, pad, tr in CASES:
        d = _lib.ConvDesc(dtype=1, use_mfma=1, B=B, H=H, W=W, Cin=Cin, Cout=Cout, R=R, S=S, stride=stride, pad=pad,
                          transposed=tr, mode=0, bn_relu=1)
        Ho, Wo = (2 * H, 2 * W) if tr else (H, W)
        x = torch.randn(B, H, W, Cin, device=DEV).half()
        w = torch.randn((Cin, Cout, 3, 3) if tr else (Cout, Cin, R, S), device=DEV) * 0.02
        sc = torch.rand(Cin, device=DEV) + 0.5
        sh = torch.randn(3 * Cin, device=DEV) * 0.5
        y = torch.empty(B, Ho, Wo, Cout, device=DEV, dtype=torch.half)
        dy = torch.randn(B, Ho, Wo, Cout, device=DEV).half()
        gx = torch.zeros(B, H, W, Cin, device=DEV, dtype=torch.half)
        dw = torch.zeros_like(w)
        stats = torch.zeros(2 * Cout, dtype=torch.float64, device=DEV)
        red = torch.zeros(2 * Cin, dtype=torch.float64, device=DEV)
        scratch = torch.zeros(L.dmm_conv_scratch_bytes(C.byref(d)), dtype=torch.uint8, device=DEV)
        st = _lib.stream_ptr()
        fns = {
            "fwd": lambda: L.dmm_conv_forward(C.byref(d), x.data_ptr(), w.data_ptr(), sc.data_ptr(), sh.data_ptr(), y.data_ptr(),
                                              stats.data_ptr(), scratch.data_ptr(), st),
            "dgrad": lambda: L.dmm_conv_dgrad(C.byref(d), x.data_ptr(), dy.data_ptr(), w.data_ptr(), sc.data_ptr(), sh.data_ptr(),
                                              gx.data_ptr(), red.data_ptr(), scratch.data_ptr(), st),
            "wgrad": lambda: L.dmm_conv_wgrad(C.byref(d), x.data_ptr(), dy.data_ptr(), sc.data_ptr(), sh.data_ptr(), dw.data_ptr(),
                                              scratch.data_ptr(), st),
        }
        gf = 2.0 * B * Ho * Wo * Cout * Cin * R * S / (stride * stride if tr else 1) / 1e9
        out = [f"{name:30s}"]
        for k, fn in fns.items():
            for _ in range(3):
                _lib.check(fn())
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(reps):
                fn()
            e1.record()
            torch.cuda.synchronize()
            ms = e0.elapsed_time(e1) / reps
            out.append(f"{k} {ms * 1000:7.1f} us {gf / ms:7.1f} TF/s")
        print("  ".join(out), flush=True)


if __name__ == "__main__":
    main()
