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
PER_FILE_FLAGS = {"triplane.hip": ["-fno-slp-vectorize"], "density_filter.hip": ["-fno-slp-vectorize"], "attention_pipe.hip": ["-fno-slp-vectorize"],
                  "gemm_l3.hip": ["-fno-slp-vectorize"], "attention_l3.hip": ["-fno-slp-vectorize"], "attention_l2.hip": ["-fno-slp-vectorize"]}


def sources():
    return sorted(glob.glob(os.path.join(CSRC, "*.hip")))


def _deps():
    return sources() + glob.glob(os.path.join(CSRC, "*.h")) + [os.path.join(HERE, "..", "include", "sculpt_hip.h")]


def hipcc():
    for c in (shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if c and os.path.exists(c):
            return c
    raise RuntimeError("hipcc not found (need ROCm to build the gfx950 kernels)")


def source_digest(flags_text=None):
    """sha256 over every HIP source, header and the compile flags: what the library was built FROM.  It is compiled into the
    library (sculpt_source_digest(), csrc/api.hip) and written beside it, so that a stale .so is detected by content, not by
    file times (a snapshot copy does not keep them), and `_lib` refuses a library that does not match the sources it sits in."""
    import hashlib

    h = hashlib.sha256()
    for path in sorted(_deps(), key=lambda q: os.path.basename(q)):
        h.update(os.path.basename(path).encode() + b"\0")
        with open(path, "rb") as f:
            h.update(f.read())
        h.update(b"\0")
    h.update((flags_text if flags_text is not None else _flags_text()).encode())
    return h.hexdigest()[:32]


def _flags():
    extra = os.environ.get("SCULPT_EXTRA_HIPCC_FLAGS", "").split()
    return ["-O3", "--offload-arch=" + ARCH, "-std=c++17", "-fPIC", "-Wno-unused-result"] + extra


def _flags_text():
    return " ".join(_flags()) + " | " + repr(sorted(PER_FILE_FLAGS.items()))


DIGEST_FILE = SO + ".digest"


def built_digest():
    """The digest recorded when the library beside this file was linked ('' if there is none)."""
    try:
        with open(DIGEST_FILE) as f:
            return f.read().strip()
    except OSError:
        return ""


def is_fresh():
    return os.path.exists(SO) and built_digest() == source_digest()


def _content_hash(paths):
    import hashlib

    h = hashlib.sha256()
    for path in sorted(paths, key=lambda q: os.path.basename(q)):
        h.update(os.path.basename(path).encode() + b"\0")
        with open(path, "rb") as f:
            h.update(f.read())
        h.update(b"\0")
    return h.hexdigest()


def _headers():
    return glob.glob(os.path.join(CSRC, "*.h")) + [os.path.join(HERE, "..", "include", "sculpt_hip.h")]


def build(force=False, verbose=False):
    """One object file per .hip source (recompiled only when its content, a header or the flags changed, up to 8 hipcc processes at a time),
    then one link.  Objects live in csrc/_obj/ (git-ignored, not needed on the GPU box: the linked .so travels)."""
    if not force and is_fresh():
        return SO
    from concurrent.futures import ThreadPoolExecutor

    objdir = os.path.join(CSRC, "_obj")
    os.makedirs(objdir, exist_ok=True)
    flags = _flags()
    stamp = os.path.join(objdir, "flags.txt")
    stamp_text = _flags_text()
    if not os.path.exists(stamp) or open(stamp).read() != stamp_text:
        force = True
    digest = source_digest(stamp_text)
    hdr_hash = _content_hash(_headers())
    jobs = []
    for src in sources():
        base = os.path.basename(src)
        obj = os.path.join(objdir, base[:-4] + ".o")
        is_api = base == "api.hip"  # carries the digest: recompiled whenever anything changed
        # An object is reused only when the CONTENT it was compiled from is unchanged (source + every header + flags), recorded
        # beside it at compile time.  File times are not evidence: a snapshot copy gives old objects new mtimes.
        want = _content_hash([src]) + hdr_hash + stamp_text + repr(PER_FILE_FLAGS.get(base, []))
        tag = obj + ".src"
        have = open(tag).read() if os.path.exists(tag) and os.path.exists(obj) else None
        if force or is_api or have != want:
            jobs.append(([hipcc()] + flags + PER_FILE_FLAGS.get(base, [])
                         + (['-DSCULPT_SOURCE_DIGEST="%s"' % digest] if is_api else []) + ["-c", src, "-o", obj], tag, want))

    def run(cmd):
        if verbose:
            print(" ".join(cmd))
        subprocess.check_call(cmd)

    def compile_one(job):
        cmd, tag, want = job
        if os.path.exists(tag):
            os.remove(tag)
        run(cmd)
        with open(tag, "w") as f:
            f.write(want)

    with ThreadPoolExecutor(max_workers=min(8, max(1, len(jobs)))) as ex:
        list(ex.map(compile_one, jobs))
    with open(stamp, "w") as f:
        f.write(stamp_text)
    objs = [os.path.join(objdir, os.path.basename(src)[:-4] + ".o") for src in sources()]
    link = [hipcc(), "--offload-arch=" + ARCH, "-fPIC", "-shared", "-o", SO + ".tmp"] + objs
    run(link)
    os.replace(SO + ".tmp", SO)
    with open(DIGEST_FILE, "w") as f:
        f.write(digest + "\n")
    if verbose:
        print("linked %s from %d objects (%d compiled now), source digest %s" % (SO, len(objs), len(jobs), digest))
    return SO


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
