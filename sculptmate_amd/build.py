"""Build libsculpt_hip.so (all HIP kernels + the C ABI) for gfx950, in-tree.

    python -m sculptmate_amd.build [--force]

hipcc cross-compiles without a GPU.  The .so is git-ignored but travels with gpurun snapshots.
"""
import glob
import os
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
SO = os.path.join(HERE, "libsculpt_hip.so")
ARCH = "gfx950"


def sources():
    return sorted(glob.glob(os.path.join(CSRC, "*.hip")))


def _deps():
    return sources() + glob.glob(os.path.join(CSRC, "*.h")) + [os.path.join(HERE, "..", "include", "sculpt_hip.h")]


def hipcc():
    for c in (shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if c and os.path.exists(c):
            return c
    raise RuntimeError("hipcc not found (need ROCm to build the gfx950 kernels)")


def is_fresh():
    return os.path.exists(SO) and all(os.path.getmtime(SO) >= os.path.getmtime(d) for d in _deps())


def build(force=False, verbose=False):
    if not force and is_fresh():
        return SO
    cmd = [hipcc(), "-O3", "--offload-arch=" + ARCH, "-std=c++17", "-fPIC", "-shared",
           "-Wno-unused-result", "-o", SO + ".tmp"] + os.environ.get("SCULPT_EXTRA_HIPCC_FLAGS", "").split() + sources()
    if verbose:
        print(" ".join(cmd))
    subprocess.check_call(cmd)
    os.replace(SO + ".tmp", SO)
    return SO


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
