"""CPU-side checks of the drop-in boundary: the library loads and exports exactly what include/mlsgpu_hip.h declares."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    text = open(os.path.join(ROOT, "include", "mlsgpu_hip.h")).read()
    return sorted(set(re.findall(r"\b(mlsgpu_hip_[a-z0-9_]+)\s*\(", text)))


def test_library_exports_every_declared_symbol():
    import mlsgpu_amd
    path = mlsgpu_amd.library_path()
    assert os.path.exists(path), "build with __graft_entry__.build()"
    lib = ctypes.CDLL(path)
    names = declared_symbols()
    assert len(names) > 50
    missing = [n for n in names if not hasattr(lib, n)]
    assert missing == []


def test_binding_covers_the_header():
    import mlsgpu_amd
    L = mlsgpu_amd.lib()
    for n in declared_symbols():
        assert getattr(L, n).argtypes is not None or n == "mlsgpu_hip_last_error", n


def test_no_oracle_in_product_path():
    """The product must never route through the CPU oracle (or any CPU fallback)."""
    for base in ("mlsgpu_amd", "include"):
        for dirpath, _, files in os.walk(os.path.join(ROOT, base)):
            for f in files:
                if f.endswith((".py", ".hip", ".hpp", ".h", ".cpp")):
                    text = open(os.path.join(dirpath, f), errors="replace").read()
                    assert "liboracle" not in text and "oracle_binding" not in text, os.path.join(dirpath, f)
                    assert not re.search(r"#include\s+[\"<].*oracle", text), os.path.join(dirpath, f)


def test_error_reporting_without_gpu():
    """Argument checks run before any device work and report through the thread-local message."""
    import mlsgpu_amd
    L = mlsgpu_amd.lib()
    rc = L.mlsgpu_hip_ctx_create(0, None, None)
    assert rc == 1
    assert b"requirement failed" in L.mlsgpu_hip_last_error()
    assert L.mlsgpu_hip_compute_max_swathe(8192, 256, 8, 8) == 24      # src/workers.cpp:169-182


def test_missing_library_fails_loudly(monkeypatch):
    import mlsgpu_amd.binding as b
    monkeypatch.setattr(b, "_lib", None)
    monkeypatch.setattr(b, "library_path", lambda: "/nonexistent/libmlsgpu_hip.so")
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        b.lib()


def test_synth_is_counter_based():
    from mlsgpu_amd import synth
    a = synth.uniforms(synth.SEED_BASE + 2, 0, 1000, 4)
    b = synth.uniforms(synth.SEED_BASE + 2, 500, 500, 4)
    assert (a[:, 500:] == b).all()
    assert 0.0 <= a.min() and a.max() < 1.0
    s, g = synth.make_cloud("cfg2", scale=0.001)
    allb, buckets = synth.bucketize(s, g, 127)
    assert len(buckets) == 27 and sum(b.count for b in buckets) == len(allb)
    assert sum(b.cells for b in buckets) == 255 ** 3


def test_struct_layouts_match_the_header(tmp_path):
    """The header compiles as plain C, and every struct the ctypes binding mirrors has the C compiler's size and
    field offsets (a1: mlsgpu_splat is the reference's 32-byte Splat, src/splat.h:40-46)."""
    import ctypes as C
    import subprocess
    from mlsgpu_amd import binding as b
    pairs = [("mlsgpu_mesh", b.Mesh), ("mlsgpu_swathe", b.Swathe), ("mlsgpu_worker_config", b.WorkerConfig),
             ("mlsgpu_farm_config", b.FarmConfig), ("mlsgpu_grid", b.GridStruct), ("mlsgpu_bucket_params", b.BucketParams),
             ("mlsgpu_bucket", b.BucketStruct), ("mlsgpu_generator", b.Generator), ("mlsgpu_subitem", b.SubItem),
             ("mlsgpu_tree_build", b.TreeBuild)]
    lines = ['#include <stdio.h>', '#include <stddef.h>', '#include "mlsgpu_hip.h"', 'int main(void) {',
             'printf("mlsgpu_splat %zu %zu %zu %zu %zu\\n", sizeof(mlsgpu_splat), offsetof(mlsgpu_splat, position), '
             'offsetof(mlsgpu_splat, radius), offsetof(mlsgpu_splat, normal), offsetof(mlsgpu_splat, quality));']
    for cname, cls in pairs:
        fmt = " ".join(["%zu"] * (1 + len(cls._fields_)))
        args = ", ".join(["sizeof(%s)" % cname] + ["offsetof(%s, %s)" % (cname, f[0]) for f in cls._fields_])
        lines.append('printf("%s %s\\n", %s);' % (cname, fmt, args))
    lines += ["return 0; }"]
    src = tmp_path / "layout.c"
    src.write_text("\n".join(lines))
    exe = tmp_path / "layout"
    subprocess.check_call(["gcc", "-std=c99", "-Wall", "-Werror", "-I", os.path.join(ROOT, "include"), str(src), "-o", str(exe)])
    out = dict((l.split()[0], [int(v) for v in l.split()[1:]]) for l in subprocess.check_output([str(exe)]).decode().splitlines())
    assert out["mlsgpu_splat"] == [32, 0, 12, 16, 28]
    assert b.SPLAT_DTYPE.itemsize == 32 and [b.SPLAT_DTYPE.fields[n][1] for n in ("position", "radius", "normal", "quality")] == [0, 12, 16, 28]
    for cname, cls in pairs:
        want = [C.sizeof(cls)] + [getattr(cls, f[0]).offset for f in cls._fields_]
        assert out[cname] == want, cname
