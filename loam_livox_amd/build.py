"""Builds libloamlivox_hip.so (the C-ABI library of include/loam_livox_hip.h) in-tree with hipcc for gfx950.

hipcc cross-compiles without a GPU.  All translation units are compiled with -ffp-contract=off: label/index
sets and neighbour lists must reproduce the reference's un-contracted fp32/fp64 evaluation; the residual
accumulation (ll_reg_core.h block_accumulate) re-enables contraction locally.
"""
from __future__ import annotations

import os
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
# LL_LIB_OUT: build an instrumented / A-B variant next to the product library (e.g. with LL_EXTRA_HIPCC_FLAGS=-DLL_SOLVE_TIMING)
# and load it with LL_LIB_PATH (capi.py); the default is the product library.
LIB = os.environ.get("LL_LIB_OUT") or os.path.join(HERE, "libloamlivox_hip.so")
SOURCES = ["ll_api.hip", "ll_fe_kernels.hip", "ll_map_kernels.hip", "ll_reg_kernels.hip", "ll_reg_small_kernels.hip", "ll_knn_kernels.hip", "ll_voxel_kernels.hip", "ll_cellmap_kernels.hip"]
HEADERS = ["ll_device.h", "ll_fe_core.h", "ll_knn_core.h", "ll_knn_coop.h", "ll_knn_tile.h", "ll_reg_query.h", "ll_reg_big_path.h", "ll_reg_solve_common.h", "ll_reg_core.h", "ll_voxel.h", "ll_voxel_core.h", "ll_cellmap.h", "ll_cellmap_core.h", "../../include/loam_livox_hip.h"]
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-Wno-unused-result"]
FLAGS += os.environ.get("LL_EXTRA_HIPCC_FLAGS", "").split()  # e.g. -DLL_SOLVE_TIMING (instrumented solver, ll_reg_debug_cycles)


def hipcc() -> str:
    for cand in (shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found (ROCm toolchain required to build libloamlivox_hip.so)")



def needs_build(lib: str = LIB) -> bool:
    if not os.path.exists(lib):
        return True
    t = os.path.getmtime(lib)
    deps = [os.path.join(CSRC, f) for f in SOURCES + HEADERS] + [os.path.abspath(__file__)]
    return any(os.path.getmtime(d) > t for d in deps)


def build(force: bool = False, verbose: bool = False, lib: str = LIB, extra_flags=()) -> str:
    if not force and not needs_build(lib):
        return lib
    cc = hipcc()
    default = os.path.join(HERE, "libloamlivox_hip.so")
    objdir = os.path.join(HERE, "build" if lib == default else "build_" + os.path.basename(lib).replace(".so", ""))
    os.makedirs(objdir, exist_ok=True)
    objs = []
    procs = []
    for src in SOURCES:
        obj = os.path.join(objdir, src.replace(".hip", ".o"))
        cmd = [cc] + FLAGS + list(extra_flags) + ["-c", os.path.join(CSRC, src), "-o", obj]
        if verbose:
            print(" ".join(cmd), file=sys.stderr)
        procs.append((src, subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)))
        objs.append(obj)
    for src, p in procs:
        out, _ = p.communicate()
        if p.returncode != 0:
            raise RuntimeError(f"hipcc failed on {src}:\n{out.decode()}")
        if verbose and out:
            print(out.decode(), file=sys.stderr)
    cmd = [cc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", lib] + objs
    subprocess.check_call(cmd)
    return lib


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
