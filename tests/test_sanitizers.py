"""sanitizer run of the CPU checker (GPU AddressSanitizer is not available on the pool).
`make -C oracle asan` compiles the three oracle sources + oracle/asan_check.c under -fsanitize=address,undefined and runs the
driver over ragged / minimal / noisy / constant volumes, border and out-of-range query points and a tiny UV bake."""
import os
import shutil
import subprocess

import pytest

from conftest import ROOT


@pytest.mark.skipif(shutil.which("gcc") is None or shutil.which("make") is None, reason="needs gcc + make")
def test_oracle_is_clean_under_asan_and_ubsan():
    p = subprocess.run(["make", "-C", os.path.join(ROOT, "oracle"), "asan"], capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stdout[-2000:] + p.stderr[-4000:]
    assert "asan_check ok" in p.stdout
    assert "ERROR: AddressSanitizer" not in p.stderr and "runtime error" not in p.stderr


@pytest.mark.skipif(shutil.which("g++") is None, reason="needs g++")
def test_remesh_host_code_is_clean_under_asan_and_ubsan(tmp_path):
    """The host-pointer entry points sculpt_mesh_* run csrc/remesh_host.h (no GPU part): the driver decimates and remeshes a
    bumpy sphere (fine, coarse and default targets), an open strip, bad and empty input under -fsanitize=address,undefined."""
    exe = str(tmp_path / "asan_remesh")
    src = os.path.join(ROOT, "tests", "native", "asan_remesh.cpp")
    c = subprocess.run(["g++", "-std=c++17", "-O1", "-g", "-fsanitize=address,undefined", "-fno-sanitize-recover=undefined",
                        "-fno-omit-frame-pointer", "-Wall", "-Wextra", "-Werror", "-o", exe, src], capture_output=True, text=True, timeout=600)
    assert c.returncode == 0, c.stderr[-4000:]
    p = subprocess.run([exe], capture_output=True, text=True, timeout=600, env=dict(os.environ, ASAN_OPTIONS="detect_leaks=1"))
    assert p.returncode == 0, p.stdout[-2000:] + p.stderr[-4000:]
    assert "asan_remesh ok" in p.stdout
    assert "ERROR: AddressSanitizer" not in p.stderr and "runtime error" not in p.stderr
