#!/usr/bin/env python3
"""The forward LDS pipelines at production sizes, conv3 on / off (see tools/gpu_lab.py production_forward_case)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dmmfods_amd import _lib  # noqa: E402
from tools import gpu_lab  # noqa: E402
from tests.test_timed_kernels_gpu import PRODUCTION  # noqa: E402

for fam in (1, 0):
    _lib.check(_lib.lib().dmm_set_option(b"conv3", fam))
    print(f"== conv3 {'on' if fam else 'off'}")
    for c in PRODUCTION:
        gpu_lab.production_forward_case(c[0], 1, *c[1:])
_lib.check(_lib.lib().dmm_set_option(b"conv3", 1))
