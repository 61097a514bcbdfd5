"""pytest configuration: registers the ``gpu`` marker and makes the repo importable."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
TESTS = os.path.dirname(os.path.abspath(__file__))
if TESTS not in sys.path:
    sys.path.insert(0, TESTS)


# PyTorch-ROCm's wheel bundles its own HIP runtime under the soname libmlsgpu_hip.so links against.  Loaded first, it is the
# one runtime of the process; loaded after /opt/rocm's copy it would be a SECOND runtime that finds no GPU.  The tests that
# use torch (device-side synthetic clouds) therefore need torch imported before the HIP library.
try:
    import torch  # noqa: F401
except Exception:   # noqa: BLE001 - the CPU suite runs without torch too
    pass


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: test needs a real MI355X (run with -m gpu)")
