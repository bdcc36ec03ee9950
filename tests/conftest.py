import ctypes
import os
import subprocess
import sys
import tempfile

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def _install_native_backtrace():
    """A native backtrace (with the thread id) in front of faulthandler's Python frames when the process dies of a signal: round 4's
    GPU suite ended in a segmentation fault on a runtime thread and the record held Python frames only (tests/crash_backtrace.c)."""
    try:
        out = os.path.join(tempfile.gettempdir(), f"dmm_crash_backtrace_{os.getuid()}.so")
        src = os.path.join(ROOT, "tests", "crash_backtrace.c")
        if not os.path.exists(out) or os.path.getmtime(out) < os.path.getmtime(src):
            subprocess.run(["gcc", "-O1", "-g", "-shared", "-fPIC", "-o", out, src], check=True, capture_output=True, timeout=120)
        lib = ctypes.CDLL(out)
        lib.crash_backtrace_install()
        return lib
    except Exception as e:  # noqa: BLE001 - an aid, never a reason to fail a run
        sys.stderr.write(f"[conftest] native backtrace handler not installed: {e!r}\n")
        return None


_CRASH_LIB = None


def pytest_sessionstart(session):
    # after pytest's faulthandler plugin has installed its own handlers (pytest_configure), so that ours runs first and chains to it
    global _CRASH_LIB
    _CRASH_LIB = _install_native_backtrace()


@pytest.fixture(scope="session")
def golden_dir():
    return os.path.join(ROOT, "tests", "golden")
