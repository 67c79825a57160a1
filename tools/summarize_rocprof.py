#!/usr/bin/env python3
"""Condense rocprofv3 CSV output into the per-(kernel, grid) summaries kept under profiles/.

  summarize_rocprof.py trace   <*_kernel_trace.csv> [lib] [--all]             -> kernel,grid_threads,workgroup,vgpr,lds_bytes,scratch_bytes,calls,total_ms,avg_us,min_us,max_us
                                                                                (--all: the library kernels of hipcub / rocprim too, names cut to 70 characters;
                                                                                 kernel names that contain a comma -- template arguments -- are quoted)
  summarize_rocprof.py pmc     <fetch *_counter_collection.csv> <write ...>  -> kernel,grid_threads,dispatches,fetch_kib_avg,fetch_mib_corrected_x2,write_kib_avg

  summarize_rocprof.py generic <*_counter_collection.csv> [...]              -> kernel,grid_threads,dispatches,<counter>_avg ... (any counters, per-dispatch averages)

  summarize_rocprof.py codeobj <libloamlivox_hip.so>                         -> kernel,vgpr,agpr,sgpr,vgpr_spill,sgpr_spill,scratch_bytes,lds_bytes,waves_per_simd
                                                                                (register / spill counts of the gfx950 code objects embedded in the library;
                                                                                 `trace <csv> <lib>` merges them into the trace summary -- rocprofv3's own
                                                                                 VGPR_Count column is the allocation granule count, half the real figure)

FETCH_SIZE / WRITE_SIZE come from two separate --pmc passes (MI355X_MICROARCH.md: never combined with trace domains);
the x2 on FETCH_SIZE is that guide's gfx950 correction (128-byte requests tallied at 64 B)."""
import csv
import os
import sys
from collections import defaultdict


LLVM_BIN = "/opt/rocm/lib/llvm/bin"
MAGIC = b"__CLANG_OFFLOAD_BUNDLE__"


def code_objects(lib):
    """{kernel name (demangled, as rocprofv3 prints it): dict of code-object metadata} for every gfx950 kernel in the library"""
    import os
    import re
    import subprocess
    import tempfile
    out = {}
    with tempfile.TemporaryDirectory() as td:
        fat = os.path.join(td, "fat.bin")
        subprocess.run([f"{LLVM_BIN}/llvm-objcopy", "--dump-section", f".hip_fatbin={fat}", lib, os.path.join(td, "dummy.so")],
                       check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
        blob = open(fat, "rb").read()
        starts = [m.start() for m in re.finditer(re.escape(MAGIC), blob)]  # one bundle per translation unit
        for i, a in enumerate(starts):
            piece = os.path.join(td, f"b{i}.bin")
            open(piece, "wb").write(blob[a:starts[i + 1] if i + 1 < len(starts) else len(blob)])
            co = os.path.join(td, f"b{i}.co")
            r = subprocess.run([f"{LLVM_BIN}/clang-offload-bundler", "--unbundle", "--type=o", f"--input={piece}",
                                "--targets=hipv4-amdgcn-amd-amdhsa--gfx950", f"--output={co}"], stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
            if r.returncode != 0 or not os.path.exists(co) or os.path.getsize(co) == 0:
                continue
            notes = subprocess.run([f"{LLVM_BIN}/llvm-readelf", "--notes", co], capture_output=True, text=True).stdout
            cur = {}
            for line in notes.splitlines() + ["  - .agpr_count: 0"]:  # (a kernel's entry starts at .agpr_count: its keys are sorted)
                m = re.match(r"\s*-?\s*\.(\w+):\s*(.*)$", line)
                if not m:
                    continue
                k, v = m.group(1), m.group(2).strip()
                if k == "agpr_count":
                    if "name" in cur:
                        out[cur["name"]] = dict(cur)
                    cur = {}
                if k in ("agpr_count", "vgpr_count", "sgpr_count", "vgpr_spill_count", "sgpr_spill_count", "private_segment_fixed_size",
                         "group_segment_fixed_size"):
                    cur[k] = int(v)
                elif k == "name" and v.startswith("_Z"):
                    cur["name"] = v
        if out:
            names = list(out)
            dem = subprocess.run(["c++filt"] + names, capture_output=True, text=True).stdout.splitlines()
            out = {short(d): out[n] for n, d in zip(names, dem)}
    return out


def waves_per_simd(vgpr, agpr):
    """gfx950: 512 registers per SIMD lane shared by the arch and accumulation files, allocated in granules of 8; at most 8 waves"""
    tot = max(8, (vgpr + agpr + 7) // 8 * 8)
    return max(1, min(8, 512 // tot))


def codeobj(lib):
    co = code_objects(lib)
    print("kernel,vgpr,agpr,sgpr,vgpr_spill,sgpr_spill,scratch_bytes,lds_bytes,waves_per_simd")
    for n in sorted(co):
        c = co[n]
        if not n.startswith("ll::"):
            continue
        print(f"{q(n)},{c.get('vgpr_count', 0)},{c.get('agpr_count', 0)},{c.get('sgpr_count', 0)},{c.get('vgpr_spill_count', 0)},"
              f"{c.get('sgpr_spill_count', 0)},{c.get('private_segment_fixed_size', 0)},{c.get('group_segment_fixed_size', 0)},"
              f"{waves_per_simd(c.get('vgpr_count', 0), c.get('agpr_count', 0))}")


def short(name):
    n = name.split("(")[0]
    return n[5:] if n.startswith("void ") else n


def q(name):
    """CSV field: quoted when it holds a comma (reg_solve_small_kernel<1, 16>)"""
    return '"' + name + '"' if "," in name else name


def trace(path, lib=None, include_all=False):
    agg = defaultdict(list)
    meta = {}
    co = code_objects(lib) if lib else {}
    for r in csv.DictReader(open(path)):
        n = short(r["Kernel_Name"])
        if not n.startswith("ll::"):
            if not include_all:
                continue
            n = n[:70]
        grid = int(r["Grid_Size_X"]) * int(r["Grid_Size_Y"]) * int(r["Grid_Size_Z"])
        key = (n, grid)
        agg[key].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
        meta[key] = (int(r["Workgroup_Size_X"]), int(r["VGPR_Count"]), int(r["LDS_Block_Size"]), int(r["Scratch_Size"]))
    # vgpr / agpr / vgpr_spill / waves_per_simd: from the code object when the library is given (the trace's VGPR_Count column is
    # the number of allocation granules, half the register count); otherwise the trace's column, labelled as such
    print("kernel,grid_threads,workgroup," + ("vgpr,agpr,vgpr_spill,sgpr_spill,waves_per_simd" if co else "vgpr_granules_rocprof") +
          ",lds_bytes,scratch_bytes,calls,total_ms,avg_us,min_us,max_us")
    for key, v in sorted(agg.items(), key=lambda kv: -sum(kv[1])):
        m = meta[key]
        if co:
            c = co.get(key[0], {})
            regs = (f"{c.get('vgpr_count', '')},{c.get('agpr_count', '')},{c.get('vgpr_spill_count', '')},{c.get('sgpr_spill_count', '')},"
                    f"{waves_per_simd(c['vgpr_count'], c.get('agpr_count', 0)) if 'vgpr_count' in c else ''}")
        else:
            regs = f"{m[1]}"
        print(f"{q(key[0])},{key[1]},{m[0]},{regs},{m[2]},{m[3]},{len(v)},{sum(v) / 1e3:.3f},{sum(v) / len(v):.1f},{min(v):.1f},{max(v):.1f}")


def counters(path, name):
    agg = defaultdict(list)
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] != name:
            continue
        n = short(r["Kernel_Name"])
        if n.startswith("ll::"):
            agg[(n, int(r["Grid_Size"]))].append(float(r["Counter_Value"]))
    return agg


def pmc(fetch_path, write_path):
    f, w = counters(fetch_path, "FETCH_SIZE"), counters(write_path, "WRITE_SIZE")
    print("kernel,grid_threads,dispatches,fetch_kib_avg,fetch_mib_corrected_x2,write_kib_avg")
    for key, v in sorted(f.items(), key=lambda kv: -sum(kv[1])):
        fa = sum(v) / len(v)
        wv = w.get(key, [0.0])
        print(f"{q(key[0])},{key[1]},{len(v)},{fa:.1f},{2 * fa / 1024:.2f},{sum(wv) / len(wv):.1f}")


def generic(paths):
    agg = defaultdict(lambda: defaultdict(list))
    names = []
    for path in paths:
        for r in csv.DictReader(open(path)):
            n = short(r["Kernel_Name"])
            if not n.startswith("ll::"):
                continue
            if r["Counter_Name"] not in names:
                names.append(r["Counter_Name"])
            agg[(n, int(r["Grid_Size"]))][r["Counter_Name"]].append(float(r["Counter_Value"]))
    print("kernel,grid_threads,dispatches," + ",".join(c + "_avg" for c in names))
    for key, d in sorted(agg.items(), key=lambda kv: -sum(sum(v) for v in kv[1].values())):
        nd = max(len(v) for v in d.values())
        print(f"{q(key[0])},{key[1]},{nd}," + ",".join(f"{sum(d[c]) / len(d[c]):.1f}" if d.get(c) else "" for c in names))


if __name__ == "__main__":
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    from build_id import stamp_line
    print(stamp_line())  # which kernel sources (and, when LL_GIT_COMMIT is set, which commit) the numbers below belong to
    if sys.argv[1] == "trace":
        rest = [a for a in sys.argv[2:] if a != "--all"]
        trace(rest[0], rest[1] if len(rest) > 1 else None, "--all" in sys.argv)
    elif sys.argv[1] == "codeobj":
        codeobj(sys.argv[2])
    elif sys.argv[1] == "generic":
        generic(sys.argv[2:])
    else:
        pmc(sys.argv[2], sys.argv[3])
