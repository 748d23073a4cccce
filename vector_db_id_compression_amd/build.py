"""Builds libvidc.so (hand-written HIP for gfx950 + the C-ABI of include/vidc.h) in-tree with hipcc.

Staleness is decided by CONTENT, not by mtimes: the sha256 of every source the library is built from is kept next to
it (libvidc.so.srchash).  A checkout, a copy to the GPU box or a `touch` therefore never triggers a rebuild, an edit
always does.  The compiler writes to a temporary file that is renamed over the library under a file lock, so ranks of
one launch that find a stale library neither compile into each other's output nor load a half-written file.
"""
import concurrent.futures
import fcntl
import glob
import hashlib
import os
import shutil
import subprocess
import sys
import tempfile

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "libvidc.so")
HASH = LIB + ".srchash"
LOCK = LIB + ".lock"
ARCH = "gfx950"
FLAGS = [f"--offload-arch={ARCH}", "-O3", "-std=c++17", "-fPIC", "-shared", "-Wall", "-Wno-unused-function"]


def sources():
    return sorted(glob.glob(os.path.join(CSRC, "*.hip")))


def _deps():
    return sources() + sorted(glob.glob(os.path.join(CSRC, "*.h"))) + [os.path.join(HERE, "..", "include", "vidc.h")]


def source_hash():
    h = hashlib.sha256(" ".join(FLAGS).encode())
    for d in _deps():
        h.update(os.path.basename(d).encode() + b"\0")
        with open(d, "rb") as f:
            h.update(f.read())
        h.update(b"\0")
    return h.hexdigest()


def needs_build():
    if not os.path.exists(LIB):
        return True
    try:
        with open(HASH) as f:
            return f.read().strip() != source_hash()
    except OSError:
        return True


def find_hipcc():
    p = shutil.which("hipcc")
    if p:
        return p
    return "/opt/rocm/bin/hipcc" if os.path.exists("/opt/rocm/bin/hipcc") else None


def build(force=False, verbose=False):
    """hipcc --offload-arch=gfx950 ... -> vector_db_id_compression_amd/libvidc.so"""
    if not force and not needs_build():
        return LIB
    hipcc = find_hipcc()
    if hipcc is None:
        if os.path.exists(LIB) and not force:
            # a host that ships the prebuilt library without a compiler: use it, but say that it could not be checked
            print("vector_db_id_compression_amd: hipcc not found; using the prebuilt libvidc.so (source hash differs or is "
                  "missing)", file=sys.stderr)
            return LIB
        raise RuntimeError("libvidc.so has to be built and hipcc was not found (no CPU fallback exists)")
    want = source_hash()
    with open(LOCK, "w") as lk:
        fcntl.flock(lk, fcntl.LOCK_EX)
        try:
            if not force and not needs_build():  # another process built it while this one waited for the lock
                return LIB
            tmp = f"{LIB}.tmp.{os.getpid()}"
            objdir = tempfile.mkdtemp(prefix="vidc_obj_")
            try:
                # one compiler process per translation unit (they are independent), then one link
                def compile_one(src):
                    obj = os.path.join(objdir, os.path.basename(src) + ".o")
                    cmd = [hipcc] + [f for f in FLAGS if f != "-shared"] + ["-c", src, "-o", obj]
                    if verbose:
                        print(" ".join(cmd), file=sys.stderr)
                    subprocess.check_call(cmd)
                    return obj

                with concurrent.futures.ThreadPoolExecutor(max_workers=min(8, os.cpu_count() or 1)) as ex:
                    objs = list(ex.map(compile_one, sources()))
                cmd = [hipcc] + FLAGS + ["-o", tmp] + objs
                if verbose:
                    print(" ".join(cmd), file=sys.stderr)
                subprocess.check_call(cmd)
                os.replace(tmp, LIB)
            finally:
                shutil.rmtree(objdir, ignore_errors=True)
                if os.path.exists(tmp):
                    os.remove(tmp)
            with open(HASH + ".tmp", "w") as f:
                f.write(want + "\n")
            os.replace(HASH + ".tmp", HASH)
        finally:
            fcntl.flock(lk, fcntl.LOCK_UN)
    return LIB


if __name__ == "__main__":
    build(force="--force" in sys.argv, verbose=True)
    print(LIB)
