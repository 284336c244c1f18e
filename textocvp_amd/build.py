"""
Builds libtocvp.so (all hand-written HIP kernels + the C-ABI) in-tree for gfx950.

    python -m textocvp_amd.build            # rebuild if any source is newer than the library

hipcc cross-compiles without a GPU, so this also runs in the CPU-only build container; the
resulting .so travels to the GPU box with the repo snapshot (it is git-ignored, not gpurun-ignored).
"""

import glob
import os
import subprocess
import sys

PKG = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(PKG)
CSRC = os.path.join(PKG, "csrc")
INCLUDE = os.path.join(ROOT, "include")
LIB_DIR = os.path.join(PKG, "_lib")
LIB_PATH = os.path.join(LIB_DIR, "libtocvp.so")


def sources():
    return sorted(glob.glob(os.path.join(CSRC, "*.hip")))


def extra_flags(src):
    """ per-file compiler flags: a source may carry ``// TOCVP_HIPCC_FLAGS: <flags>`` lines in its first 40 lines
    (e.g. ``-mllvm -amdgpu-mfma-vgpr-form`` for a kernel whose accumulators are rescaled by vector instructions) """
    flags = []
    with open(src) as f:
        for _, line in zip(range(40), f):
            if line.startswith("// TOCVP_HIPCC_FLAGS:"):
                flags += line.split(":", 1)[1].split()
    return flags


def _stale():
    if not os.path.exists(LIB_PATH):
        return True
    lib_m = os.path.getmtime(LIB_PATH)
    deps = sources() + glob.glob(os.path.join(CSRC, "*.h")) + glob.glob(os.path.join(INCLUDE, "*.h"))
    return any(os.path.getmtime(d) > lib_m for d in deps)


def build(force=False, verbose=False):
    """ Compile every .hip under csrc/ into one shared library.  Returns the library path. """
    if not force and not _stale():
        return LIB_PATH
    os.makedirs(LIB_DIR, exist_ok=True)
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    objs, procs = [], []
    headers = glob.glob(os.path.join(CSRC, "*.h")) + glob.glob(os.path.join(INCLUDE, "*.h"))
    hdr_m = max(os.path.getmtime(h) for h in headers)
    jobs = int(os.environ.get("TOCVP_BUILD_JOBS", "4"))
    for src in sources():
        obj = os.path.join(LIB_DIR, os.path.basename(src).replace(".hip", ".o"))
        objs.append(obj)
        # per-object incremental rebuild: a source newer than its object, or any header newer than it
        if not force and os.path.exists(obj) and os.path.getmtime(obj) > max(os.path.getmtime(src), hdr_m):
            continue
        cmd = [hipcc, "-O3", "--offload-arch=gfx950", "-fPIC", "-std=c++17", "-Wno-comment",
               f"-I{INCLUDE}", f"-I{CSRC}"] + extra_flags(src) + ["-c", src, "-o", obj]
        if verbose:
            print(" ".join(cmd))
        procs.append((cmd, subprocess.Popen(cmd)))
        while sum(p.poll() is None for _, p in procs) >= jobs:
            procs[0][1].wait() if procs[0][1].poll() is None else None
            for _, p in procs:
                if p.poll() is None:
                    p.wait()
                    break
    for cmd, p in procs:
        if p.wait() != 0:
            raise subprocess.CalledProcessError(p.returncode, cmd)
    cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB_PATH] + objs
    if verbose:
        print(" ".join(cmd))
    subprocess.run(cmd, check=True)
    return LIB_PATH


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
