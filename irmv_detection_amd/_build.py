"""Build libirmv_hip.so (gfx950) in-tree with hipcc.

hipcc cross-compiles without a GPU, so this runs in the CPU-only container; the
resulting .so is git-ignored but travels to the GPU box with the tree.
"""
from __future__ import annotations

import os
import shutil
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(_HERE, "csrc")
LIB_DIR = os.path.join(_HERE, "lib")
LIB_PATH = os.path.join(LIB_DIR, "libirmv_hip.so")
INCLUDE = os.path.join(os.path.dirname(_HERE), "include")

# (source, extra flags).  k_post.hip must not contract a*b+c: see its header.
SOURCES = [
    ("k_pre.hip", []),
    ("k_front.hip", []),
    ("k_c2f.hip", []),
    ("k_bneck.hip", []),
    ("k_kpt.hip", []),
    ("k_conv.hip", []),
    ("k_post.hip", ["-ffp-contract=off"]),
    ("k_light.hip", ["-ffp-contract=off"]),
    ("k_shuffle.hip", []),
    ("engine.cpp", ["-x", "hip"]),
]
# -amdgpu-mfma-vgpr-form: MFMA results land in VGPRs.  Left to itself the compiler gives kernels without a waves_per_eu
# bound (fused C2f blocks, direct convs) AGPR accumulators and copies every one of them out with a v_accvgpr_read before the
# epilogue: one vector instruction per output value in kernels whose vector-issue slots are the scarce resource.
COMMON = ["-O3", "-std=c++17", "-fPIC", "--offload-arch=gfx950", "-Wall", "-Wno-unused-function",
          "-mllvm", "-amdgpu-mfma-vgpr-form", "-I", INCLUDE]


def source_hash() -> str:
    """sha256 over everything that determines the kernels of libirmv_hip.so: the sources and headers of csrc/, the C ABI
    header and the compiler flags.  The counter files under profiles/ carry it (scripts/build_stamp.py) and bench.py drops
    their numbers when it differs from the tree it runs in -- a hash of the BINARY would not survive a rebuild elsewhere."""
    import hashlib
    h = hashlib.sha256()
    names = sorted(set([s for s, _ in SOURCES] + ["irmv_common.hpp", "pnp_device.hpp", "numa.hpp"]))
    for n in names:
        pth = os.path.join(CSRC, n)
        if os.path.exists(pth):
            h.update(n.encode() + b"\0" + open(pth, "rb").read())
    h.update(open(os.path.join(INCLUDE, "irmv_hip.h"), "rb").read())
    h.update(" ".join(COMMON[:-2]).encode())          # flags without the machine-specific include path
    for src, extra in SOURCES:
        h.update((src + " " + " ".join(extra)).encode())
    return h.hexdigest()


def hipcc() -> str:
    for cand in (shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found: the HIP extension cannot be built")


def _stale(target: str, deps) -> bool:
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


COMM_PATH = os.path.join(LIB_DIR, "libirmv_comm.so")


def build_comm(force: bool = False, verbose: bool = False) -> str:
    """libirmv_comm.so (include/irmv_comm.h): the RCCL weight broadcast of the multi-GPU path.  Host code only; linked
    against the ROCm installation's librccl (never torch's bundled copy)."""
    os.makedirs(LIB_DIR, exist_ok=True)
    src = os.path.join(CSRC, "comm.cpp")
    deps = [src, os.path.join(INCLUDE, "irmv_comm.h"), os.path.join(INCLUDE, "irmv_hip.h"), __file__]
    if force or _stale(COMM_PATH, deps):
        rocm_lib = os.path.join(os.path.dirname(os.path.dirname(os.path.realpath(hipcc()))), "lib")
        cmd = [hipcc(), "-O2", "-std=c++17", "-fPIC", "-shared", "-Wall", "-I", INCLUDE, src, "-o", COMM_PATH,
               "-L", rocm_lib, "-lrccl", "-Wl,-rpath," + rocm_lib]
        if verbose:
            print(" ".join(cmd))
        subprocess.check_call(cmd)
    return COMM_PATH


def build(force: bool = False, verbose: bool = False) -> str:
    os.makedirs(LIB_DIR, exist_ok=True)
    headers = [os.path.join(CSRC, "irmv_common.hpp"), os.path.join(CSRC, "pnp_device.hpp"), os.path.join(CSRC, "numa.hpp"), os.path.join(INCLUDE, "irmv_hip.h"), __file__]
    objs = []
    cc = hipcc()
    for src, extra in SOURCES:
        sp = os.path.join(CSRC, src)
        obj = os.path.join(LIB_DIR, os.path.splitext(src)[0] + ".o")
        if force or _stale(obj, [sp] + headers):
            cmd = [cc] + COMMON + extra + os.environ.get("IRMV_EXTRA_HIPCC_FLAGS", "").split() + ["-c", sp, "-o", obj]
            if verbose:
                print(" ".join(cmd))
            subprocess.check_call(cmd)
        objs.append(obj)
    if force or _stale(LIB_PATH, objs):
        cmd = [cc, "-shared", "-fPIC", "--offload-arch=gfx950", "-o", LIB_PATH] + objs
        if verbose:
            print(" ".join(cmd))
        subprocess.check_call(cmd)
    return LIB_PATH


if __name__ == "__main__":
    print(build(force=False, verbose=True))
    print(build_comm(force=False, verbose=True))
