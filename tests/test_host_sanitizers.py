"""Host code under AddressSanitizer and UBSan (g++ / gcc, no GPU): the tree builder, refit and four-wide conversion
over random scenes of every object kind; the C++ scene API and its scene.txt loader on the sample file, every prefix of
it and copies with a line dropped or a value garbled (with leak detection: a refused file must free what it had
built); and the CPU oracle itself, since a checker with undefined behaviour checks nothing.  Device code cannot be
sanitised on this hardware pool; its indexing is covered by the bit-exactness tests instead."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SAN = os.path.join(ROOT, "tests", "san")
# (the runtimes linked statically: a dynamic one insists on being first in the process's library list)
FLAGS = ["-O1", "-g", "-ffp-contract=off", "-fsanitize=address,undefined", "-fno-sanitize-recover=all", "-static-libasan", "-static-libubsan"]


def _build(tmp_path, name, compiler, sources, extra=()):
    exe = str(tmp_path / name)
    cmd = [compiler, *FLAGS, "-o", exe, *sources, *extra]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=600)
    if r.returncode != 0 and ("cannot find -lasan" in r.stderr or "cannot find -lubsan" in r.stderr or "libasan" in r.stderr and "No such file" in r.stderr):
        pytest.skip("no sanitizer runtime for %s here" % compiler)
    assert r.returncode == 0, r.stderr[-3000:]
    return exe


def _run(exe, *args):
    env = dict(os.environ)
    env["UBSAN_OPTIONS"] = "print_stacktrace=1"
    r = subprocess.run([exe, *args], capture_output=True, text=True, timeout=600, env=env)
    assert r.returncode == 0, (r.stdout[-2000:], r.stderr[-4000:])
    assert "ERROR: " not in r.stderr and "runtime error" not in r.stderr, r.stderr[-4000:]
    return r.stdout


def test_tree_builder_refit_and_wide_form(tmp_path):
    exe = _build(tmp_path, "bvh_san", "g++", [os.path.join(SAN, "bvh_san.cpp"), os.path.join(ROOT, "pathtracer_amd", "csrc", "pt_bvh.cpp")], ["-std=c++17"])
    out = _run(exe)
    assert "ok 72 cases" in out, out


def test_scene_api_and_scene_file_loader(tmp_path):
    lib_dir = os.path.join(ROOT, "pathtracer_amd")
    if not os.path.exists(os.path.join(lib_dir, "libpathtracer_amd.so")):
        pytest.skip("library not built")
    exe = _build(tmp_path, "host_san", "g++",
                 [os.path.join(SAN, "host_san.cpp"), os.path.join(lib_dir, "host", "host_capi.cpp")],
                 ["-std=c++17", "-L" + lib_dir, "-lpathtracer_amd", "-Wl,-rpath," + lib_dir])
    out = _run(exe, os.path.join(ROOT, "tests", "golden", "scene_sample.txt"), str(tmp_path))
    assert out.startswith("ok "), out
    loaded, refused = int(out.split()[1]), int(out.split()[4])
    assert loaded > 50 and refused > 300  # both outcomes are exercised


def test_oracle(tmp_path):
    exe = _build(tmp_path, "oracle_san", "gcc", [os.path.join(SAN, "oracle_san.c"), os.path.join(ROOT, "oracle", "pt_oracle.c")], ["-std=c11", "-fopenmp", "-lm"])
    out = _run(exe)
    assert out.startswith("ok "), out
