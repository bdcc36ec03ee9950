"""fp32 dgrad channel sums of 1x1 / 3x3 convolutions at the tiny test model's shapes (DMM_LIB_PATH selects the build)."""
import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", ".."))
from tools import gpu_lab as lab
for (B, H, W, Cin, Cout, R, pad) in [(2, 32, 48, 64, 32, 1, 0), (2, 32, 48, 32, 32, 1, 0), (2, 16, 24, 64, 32, 1, 0), (2, 32, 48, 64, 8, 1, 0),
                                     (2, 16, 24, 24, 32, 1, 0), (2, 16, 24, 16, 32, 1, 0), (2, 32, 48, 32, 8, 3, 1), (2, 8, 12, 48, 32, 1, 0),
                                     (1, 8, 8, 64, 32, 1, 0), (2, 32, 48, 128, 32, 1, 0), (2, 64, 96, 40, 16, 3, 1)]:
    lab.conv_case(f"{R}x{R} {Cin}->{Cout} @{B}x{H}x{W}", 0, 1, B, H, W, Cin, Cout, R, R, 1, pad)
