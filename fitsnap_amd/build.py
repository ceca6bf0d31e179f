"""Build libfsnap_hip.so (gfx950 only) in-tree with hipcc.

The library is the C-ABI boundary declared in include/fsnap_hip.h.  It is built
explicitly (`hipcc --offload-arch=gfx950 -shared -fPIC`) into fitsnap_amd/_lib/ so that
the artefact travels with the source tree to the GPU box; nothing is JIT-compiled into a
user cache.  hipcc cross-compiles without a GPU.
"""
from __future__ import annotations

import hashlib
import os
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIBDIR = os.path.join(HERE, "_lib")
LIBNAME = "libfsnap_hip.so"
SOURCES = ["fsnap_syrk_quad.hip:0", "fsnap_syrk_quad.hip:1", "fsnap_syrk_quad.hip:2", "fsnap_syrk.hip", "fsnap_syrk_short.hip", "fsnap_rows.hip", "fsnap_chol.hip", "fsnap_trsm.hip", "fsnap_fused.hip", "fsnap_capi.cpp", "fsnap_comm.cpp", "fsnap_p2p.cpp",
           "fsnap_rowspace.cpp", "fsnap_rowspace_host.cpp", "fsnap_solve.cpp"]
HEADERS = ["fsnap_kernels.h", "fsnap_device_common.h", "fsnap_ctx.h", "fsnap_rowspace_host.h", "fsnap_condest.h", "fsnap_p2p.h", os.path.join("..", "..", "include", "fsnap_hip.h")]
ARCH = "gfx950"


def lib_path() -> str:
    return os.path.join(LIBDIR, LIBNAME)


def _hipcc() -> str:
    exe = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(exe):
        raise RuntimeError("hipcc not found: libfsnap_hip.so cannot be built")
    return exe


def _source_digest() -> str:
    h = hashlib.sha256()
    for name in sorted({s.split(":")[0] for s in SOURCES}) + HEADERS:
        with open(os.path.join(CSRC, name), "rb") as f:
            h.update(f.read())
    return h.hexdigest()


def build_library(force: bool = False, verbose: bool = False) -> str:
    """Compile the translation units (in parallel) and link the shared library. Returns its path."""
    os.makedirs(LIBDIR, exist_ok=True)
    out = lib_path()
    stamp = out + ".sha256"
    digest = _source_digest()
    if not force and os.path.exists(out) and os.path.exists(stamp):
        with open(stamp) as f:
            if f.read().strip() == digest:
                return out
    hipcc = _hipcc()
    objs = []
    common = [f"--offload-arch={ARCH}", "-O3", "-std=c++17", "-fPIC", "-I", os.path.join(HERE, "..", "include")]
    procs = []
    # objects are kept between builds (fitsnap_amd/_lib/obj/, git-ignored and not shipped): a translation unit is
    # recompiled only when it, a header or its command line changed -- the SYRK kernels alone take three minutes
    objdir = os.path.join(LIBDIR, "obj")
    os.makedirs(objdir, exist_ok=True)
    hh = hashlib.sha256()
    for name in HEADERS:
        with open(os.path.join(CSRC, name), "rb") as f:
            hh.update(f.read())
    for entry in SOURCES:
        # "file:N" = part N of a translation unit that is compiled in pieces (-DFSNAP_QUAD_PART=N)
        src, _, part = entry.partition(":")
        obj = os.path.join(objdir, os.path.splitext(src)[0] + (f"_{part}" if part else "") + ".o")
        cmd = [hipcc, *common]
        if part:
            cmd += [f"-DFSNAP_QUAD_PART={part}"]
        if src in ("fsnap_capi.cpp", "fsnap_comm.cpp", "fsnap_p2p.cpp", "fsnap_rowspace.cpp"):
            cmd += ["-x", "hip"]
        if src in ("fsnap_solve.cpp", "fsnap_rowspace_host.cpp"):
            # host-only K x K solve: plain C++ (no device pass), AVX2+FMA baseline with AVX-512
            # function clones resolved at load time
            cmd = [hipcc, "-x", "c++", "-O3", "-std=c++17", "-fPIC", "-mavx2", "-mfma",
                   "-I", os.path.join(HERE, "..", "include")]
        cmd += ["-c", os.path.join(CSRC, src), "-o", obj]
        objs.append(obj)
        with open(os.path.join(CSRC, src), "rb") as f:
            odig = hashlib.sha256(hh.digest() + " ".join(cmd).encode() + f.read()).hexdigest()
        ostamp = obj + ".sha256"
        if not force and os.path.exists(obj) and os.path.exists(ostamp):
            with open(ostamp) as f:
                if f.read().strip() == odig:
                    continue
        if os.path.exists(ostamp):
            os.remove(ostamp)
        if verbose:
            print(" ".join(cmd), file=sys.stderr)
        procs.append((src, ostamp, odig, subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)))
    failed = None
    for src, ostamp, odig, p in procs:
        log, _ = p.communicate()
        if p.returncode != 0:
            failed = failed or f"hipcc failed on {src}:\n{log}"
            continue
        with open(ostamp, "w") as f:
            f.write(odig + "\n")
        if verbose and log.strip():
            print(log, file=sys.stderr)
    if failed:
        raise RuntimeError(failed)
    link = [hipcc, f"--offload-arch={ARCH}", "-shared", "-fPIC", "-o", out, *objs, "-Wl,-rpath,/opt/rocm/lib", "-lpthread", "-ldl"]
    r = subprocess.run(link, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    if r.returncode != 0:
        raise RuntimeError(f"link failed:\n{r.stdout}")
    with open(stamp, "w") as f:
        f.write(digest + "\n")
    return out


if __name__ == "__main__":
    print(build_library(force="--force" in sys.argv, verbose=True))
