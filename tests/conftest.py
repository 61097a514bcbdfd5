"""pytest configuration: registers the ``gpu`` marker, makes the repo importable and reports the sizes the full-size
tests actually ran (a test that silently shrinks to fit the box says so in the session's last lines)."""
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

SIZES_RUN = []       # (what, size actually run): filled by record_size(), printed by pytest_terminal_summary


def record_size(what, size):
    """A full-size test states what it ran (e.g. "cfg5", "1000000000 splats" or "125000000 splats (box lacks RAM)")."""
    SIZES_RUN.append((str(what), str(size)))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: test needs a real MI355X (run with -m gpu)")


def pytest_terminal_summary(terminalreporter, exitstatus, config):
    # `import conftest` from a test module and pytest's own load of this file may be two module objects: read both lists
    seen = list(SIZES_RUN)
    other = sys.modules.get("conftest")
    if other is not None and getattr(other, "SIZES_RUN", None) is not SIZES_RUN:
        seen += list(getattr(other, "SIZES_RUN", []))
    if seen:
        terminalreporter.write_line("sizes run: " + "; ".join("%s: %s" % kv for kv in seen))
