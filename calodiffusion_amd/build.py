"""Build the gfx950 HIP library (and nothing else) in-tree.

    python -m calodiffusion_amd.build        # -> calodiffusion_amd/lib/libcalodiff_hip.so

hipcc cross-compiles for gfx950 without a GPU present.  The shared object is git-ignored but travels with the
working tree to the GPU box.
"""
from __future__ import annotations

import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIBDIR = os.path.join(HERE, "lib")
# CD_BUILD_TAG=<tag>: an experiment build (e.g. with CD_EXTRA_HIPCC_FLAGS=-DCD_ZS_EXPERIMENTS) next to the product library, in
# its own object directory; load it with CALODIFF_LIB=.../libcalodiff_hip_<tag>.so
TAG = os.environ.get("CD_BUILD_TAG", "")
LIB = os.path.join(LIBDIR, f"libcalodiff_hip{'_' + TAG if TAG else ''}.so")
SOURCES = ["kernels_conv.hip", "kernels_conv_zs.hip", "kernels_attn.hip", "kernels_conv_small.hip", "kernels_deep.hip", "kernels_wgrad16.hip", "kernels_norm_attn.hip", "kernels_misc.hip", "kernels_mlp.hip", "kernels_mlp_train.hip", "kernels_bwd.hip", "profiler.hip", "plan.hip"]
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-fno-gpu-rdc", "-Wall", "-Wno-unused-function", "-Wno-unused-value"]
FLAGS += os.environ.get("CD_EXTRA_HIPCC_FLAGS", "").split()  # e.g. -DCD_ZS_EXPERIMENTS (tools/zs_stamps.sh, tools/zs_power.sh)
# per-file flags.  kernels_conv_zs.hip: MFMA results in vector registers (the one-wave-per-SIMD kernel keeps its weights in the
# accumulation file and sums its partial tiles straight from the MFMA destinations; the default AGPR form costs a
# v_accvgpr_read per accumulator register and step)
FILE_FLAGS = {"kernels_conv_zs.hip": ["-mllvm", "-amdgpu-mfma-vgpr-form"]}


def _file_digest(paths, extra: str = "") -> str:
    import hashlib
    h = hashlib.sha256(extra.encode())
    for f in paths:
        h.update(os.path.basename(f).encode())
        h.update(open(f, "rb").read())
    return h.hexdigest()


def _read(path: str):
    try:
        return open(path).read().strip()
    except OSError:
        return None


def build_all(force: bool = False, verbose: bool = True) -> str:
    """Compile what changed and link.  Staleness is decided by CONTENT, not mtimes: every object carries `<obj>.hash` = sha256
    of its source, every header / include fragment and its flags, and the library's `.srchash` stamp is (re)written only when
    the link step ran -- after an rsync or a stash that preserves mtimes the stamp can therefore never vouch for a library
    that was not built from the hashed sources (engine.load_library compares it with source_hash())."""
    os.makedirs(LIBDIR, exist_ok=True)
    objdir = os.path.join(LIBDIR, "obj" + ("_" + TAG if TAG else ""))
    os.makedirs(objdir, exist_ok=True)
    # every header / include fragment is a dependency of every object (split16.h, gn_defer.h and train.inc are shared across
    # translation units: a stale object would mix producer and consumer layouts)
    inc = os.path.join(os.path.dirname(HERE), "include")
    headers = sorted(os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith((".h", ".inc")))
    headers += sorted(os.path.join(inc, f) for f in os.listdir(inc) if f.endswith(".h"))
    jobs = []
    objs = []
    for src in SOURCES:
        sp = os.path.join(CSRC, src)
        op = os.path.join(objdir, src.replace(".hip", ".o"))
        objs.append(op)
        flags = [*FLAGS, *FILE_FLAGS.get(os.path.basename(sp), [])]
        key = _file_digest([sp] + headers, " ".join(flags))
        if force or not os.path.exists(op) or _read(op + ".hash") != key:
            jobs.append(([HIPCC, *flags, "-c", sp, "-o", op], op, key))

    def run(cmd):
        if verbose:
            print(" ".join(cmd), flush=True)
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"hipcc failed:\n{r.stdout}\n{r.stderr}")
        if verbose and r.stderr.strip():
            print(r.stderr, file=sys.stderr)

    def compile_one(job):
        cmd, op, key = job
        if os.path.exists(op + ".hash"):
            os.remove(op + ".hash")
        run(cmd)
        with open(op + ".hash", "w") as fh:
            fh.write(key + "\n")

    with ThreadPoolExecutor(max_workers=4) as ex:
        list(ex.map(compile_one, jobs))
    want = source_hash()
    if force or jobs or not os.path.exists(LIB) or _read(LIB + ".srchash") != want:
        if os.path.exists(LIB + ".srchash"):
            os.remove(LIB + ".srchash")
        run([HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB, *objs])
        with open(LIB + ".srchash", "w") as fh:  # which sources the library next to it was built from
            fh.write(want + "\n")
    return LIB


def source_hash() -> str:
    """sha256 over every HIP source, header and include fragment (and the flags) that goes into the library."""
    import hashlib
    h = hashlib.sha256((" ".join(FLAGS) + repr(sorted(FILE_FLAGS.items()))).encode())
    inc = os.path.join(os.path.dirname(HERE), "include")
    files = [os.path.join(CSRC, f) for f in sorted(os.listdir(CSRC))] + [os.path.join(inc, f) for f in sorted(os.listdir(inc))]
    for f in files:
        if f.endswith((".hip", ".h", ".inc")):
            h.update(os.path.basename(f).encode())
            h.update(open(f, "rb").read())
    return h.hexdigest()


if __name__ == "__main__":
    print(build_all(force="--force" in sys.argv))
