"""Build libcontextflow_hip.so for gfx950 with hipcc (cross-compiles without a GPU).

    python -m contextflow_amd.build            # rebuild if sources are newer than the library
    python -m contextflow_amd.build --force
"""
import concurrent.futures
import glob
import os
import subprocess
import sys

PKG = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(PKG, "csrc")
OUT = os.path.join(PKG, "libcontextflow_hip.so")
OBJ = os.path.join(PKG, "build")
ARCH = "gfx950"
FLAGS = ["--offload-arch=" + ARCH, "-O3", "-fPIC", "-std=c++17", "-Wno-comment"]
EXTRA_FLAGS = {}          # per-file extras: {"file.hip": [flags]} (A/B builds: tools/dev/make_abl.py)


def sources():
    return sorted(glob.glob(os.path.join(CSRC, "*.hip")))


def deps():
    return sources() + glob.glob(os.path.join(CSRC, "*.h")) + [os.path.join(os.path.dirname(PKG), "include", "contextflow_hip.h")]


def up_to_date():
    return os.path.exists(OUT) and all(os.path.getmtime(OUT) >= os.path.getmtime(d) for d in deps())


def _compile(src):
    obj = os.path.join(OBJ, os.path.basename(src)[:-4] + ".o")
    cmd = ["hipcc"] + FLAGS + EXTRA_FLAGS.get(os.path.basename(src), []) + ["-c", src, "-o", obj]
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError("hipcc failed for %s:\n%s" % (src, r.stderr[-4000:]))
    return obj


def build(force=False, verbose=True):
    if not force and up_to_date():
        return OUT
    os.makedirs(OBJ, exist_ok=True)
    with concurrent.futures.ThreadPoolExecutor(max_workers=min(8, os.cpu_count() or 1)) as ex:
        objs = list(ex.map(_compile, sources()))
    cmd = ["hipcc", "--offload-arch=" + ARCH, "-shared", "-fPIC", "-o", OUT] + objs
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError("link failed:\n%s" % r.stderr[-4000:])
    for junk in glob.glob(OUT + ".*"):          # hipcc leaves the unbundled device / host images of the link step behind
        os.remove(junk)
    if verbose:
        print("built %s (%d kernels files, %.1f KB)" % (OUT, len(objs), os.path.getsize(OUT) / 1024))
    return OUT


if __name__ == "__main__":
    build(force="--force" in sys.argv)
