"""fp32 data gradients WITH the deferred-correction prologue (the in-model instantiations) at the tiny test model's shapes."""
import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", ".."))
from tools import gpu_lab as lab
for (B, H, W, Cin, Cout, R, pad) in [(2, 16, 24, 64, 32, 1, 0), (2, 32, 48, 64, 32, 1, 0), (2, 16, 24, 32, 32, 1, 0), (2, 8, 12, 64, 32, 1, 0),
                                     (2, 16, 24, 24, 32, 1, 0), (2, 16, 24, 32, 8, 3, 1), (2, 4, 6, 64, 32, 1, 0), (2, 16, 24, 128, 32, 1, 0)]:
    for q in (1, 0):
        for acc in (0, 1):
            lab.backward_case(f"{R}x{R} {Cin}->{Cout} @{B}x{H}x{W}", 0, B, H, W, Cin, Cout, R, R, pad, with_q=q, acc=acc, what="dgrad")
