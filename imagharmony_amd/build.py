"""In-tree build of libimh_hip.so for gfx950 (hipcc cross-compiles without a GPU).
Sources are compiled to objects in parallel, then linked; objects live under csrc/_obj (git-ignored)."""
import glob
import os
import shutil
import subprocess
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OBJ = os.path.join(CSRC, "_obj")
OUT = os.path.join(HERE, "libimh_hip.so")
CFLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wno-unused-value"]
# IMH_KERNEL (csrc/imh_common.h) carries __attribute__((target("no-packed-fp32-ops"))) for the gemm.hip kernels; the
# attribute only means something to the device pass, the host pass says "attribute ignored" -> -Wno-ignored-attributes.
CFLAGS.append("-Wno-ignored-attributes")
# IMH_EXPERIMENTAL=1: also compile the measured-but-not-selected kernel variants (csrc/imh_common.h IMH_EXP_ONLY; the library answers
# imh_debug_set(1, 0) with 1).  The default library holds what tuning.json and the default modes reach.
EXPERIMENTAL = os.environ.get("IMH_EXPERIMENTAL") == "1"
if EXPERIMENTAL:      # a second library next to the tools (use it with IMH_LIB_PATH=tools/tmp_libs/libimh_hip_experimental.so); the in-tree one stays the default build
    CFLAGS.append("-DIMH_EXPERIMENTAL")
    OBJ = os.path.join(HERE, "..", "tools", "tmp_libs", "_obj_experimental")
    OUT = os.path.join(HERE, "..", "tools", "tmp_libs", "libimh_hip_experimental.so")


def sources():
    return sorted(glob.glob(os.path.join(CSRC, "*.hip")))


def _headers():
    return glob.glob(os.path.join(CSRC, "*.h")) + [os.path.join(HERE, "..", "include", "imh.h")]


def needs_build():
    if not os.path.exists(OUT):
        return True
    t = os.path.getmtime(OUT)
    return any(os.path.getmtime(d) > t for d in sources() + _headers())


def build(force=False, verbose=True):
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not force and not needs_build():
        return OUT
    os.makedirs(OBJ, exist_ok=True)
    hdr_t = max(os.path.getmtime(h) for h in _headers())

    def compile_one(src):
        obj = os.path.join(OBJ, os.path.basename(src) + ".o")
        if not force and os.path.exists(obj) and os.path.getmtime(obj) > max(os.path.getmtime(src), hdr_t):
            return obj
        cmd = [hipcc] + CFLAGS + ["-c", src, "-o", obj]
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.run(cmd, check=True)
        return obj

    with ThreadPoolExecutor(max_workers=min(8, os.cpu_count() or 1)) as ex:
        objs = list(ex.map(compile_one, sources()))
    os.makedirs(os.path.dirname(OUT), exist_ok=True)
    cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", OUT] + objs
    if verbose:
        print(" ".join(cmd), flush=True)
    subprocess.run(cmd, check=True)
    return OUT


if __name__ == "__main__":
    build(force=True)
