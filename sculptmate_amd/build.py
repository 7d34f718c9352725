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
# triplane.hip: the SLP vectoriser packs the scalar fp32 adds / multiplies of the SiLU + split chunks into v_pk_add_f32 /
# v_pk_mul_f32, which cost ~10 issue cycles each and break the overlap with the bf16 MFMAs they are interleaved with
# (tools/micro/mfma_fill.hip); the kernels that want packed operations ask for them with vector types.
PER_FILE_FLAGS = {"triplane.hip": ["-fno-slp-vectorize"], "attention_pipe.hip": ["-fno-slp-vectorize"]}


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


def _headers():
    return glob.glob(os.path.join(CSRC, "*.h")) + [os.path.join(HERE, "..", "include", "sculpt_hip.h")]


def build(force=False, verbose=False):
    """One object file per .hip source (recompiled only when it or a header changed, up to 8 hipcc processes at a time),
    then one link.  Objects live in csrc/_obj/ (git-ignored, not needed on the GPU box: the linked .so travels)."""
    if not force and is_fresh():
        return SO
    from concurrent.futures import ThreadPoolExecutor

    objdir = os.path.join(CSRC, "_obj")
    os.makedirs(objdir, exist_ok=True)
    extra = os.environ.get("SCULPT_EXTRA_HIPCC_FLAGS", "").split()
    flags = ["-O3", "--offload-arch=" + ARCH, "-std=c++17", "-fPIC", "-Wno-unused-result"] + extra
    stamp = os.path.join(objdir, "flags.txt")
    stamp_text = " ".join(flags) + " | " + repr(sorted(PER_FILE_FLAGS.items()))
    if not os.path.exists(stamp) or open(stamp).read() != stamp_text:
        force = True
    hdr_time = max(os.path.getmtime(h) for h in _headers())
    jobs = []
    for src in sources():
        obj = os.path.join(objdir, os.path.basename(src)[:-4] + ".o")
        if force or not os.path.exists(obj) or os.path.getmtime(obj) < max(os.path.getmtime(src), hdr_time):
            jobs.append([hipcc()] + flags + PER_FILE_FLAGS.get(os.path.basename(src), []) + ["-c", src, "-o", obj])

    def run(cmd):
        if verbose:
            print(" ".join(cmd))
        subprocess.check_call(cmd)

    with ThreadPoolExecutor(max_workers=min(8, max(1, len(jobs)))) as ex:
        list(ex.map(run, jobs))
    with open(stamp, "w") as f:
        f.write(stamp_text)
    objs = [os.path.join(objdir, os.path.basename(src)[:-4] + ".o") for src in sources()]
    link = [hipcc(), "--offload-arch=" + ARCH, "-fPIC", "-shared", "-o", SO + ".tmp"] + objs
    run(link)
    os.replace(SO + ".tmp", SO)
    return SO


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
