"""CPU tier: the pieces of bench.py and tools/ that turn committed rocprofv3 summaries into the numbers of the bench line --
a format drift in profiles/ must fail here, not silently null the `roofline` fields on the GPU box."""
import csv
import glob
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def test_bench_reads_the_committed_counter_summaries():
    import bench
    traffic, src = bench.pmc_traffic_bytes("reg_solve_kernel", 256)
    assert src and src.startswith("profiles/r") and os.path.exists(os.path.join(ROOT, src))
    assert 0.2e9 < traffic < 5e9  # bytes per B = 256 solver launch: above the 0.21 GB of block records, far below a TB
    assert bench.pmc_traffic_bytes("reg_solve_kernel", 8) == (None, None)  # measured at batch 256 only
    vf = bench.pmc_valu_fp64("reg_solve_kernel", 256)
    assert vf and vf["source"].startswith("profiles/r") and vf["fp64_wave_instructions"] < vf["valu_wave_instructions"]
    assert 1e9 < vf["flops"] < 1e11
    assert bench.pmc_valu_fp64("no_such_kernel", 256) is None
    kn = bench.pmc_knn_issue(256)
    assert kn and kn["bound"] == "valu-issue" and 0.2 < kn["lane_utilisation"] <= 1.0 and 0.05 < kn["valu_issue_frac_at_2p4GHz"] < 1.5
    assert all(os.path.exists(os.path.join(ROOT, p)) for p in kn["sources"]) and bench.pmc_knn_issue(8) is None
    # the newest summary wins (round tags sort lexicographically)
    newest = sorted(f for f in glob.glob(os.path.join(ROOT, "profiles", "r*_pmc_hbm_bytes.csv"))
                    if not any(t in os.path.basename(f) for t in ("_qpipe_", "_c3_", "_c4_", "_c5_")))[-1]   # (the default line's passes, not Q-pipe / C3 / C5)
    assert os.path.relpath(newest, ROOT) == src


def test_summarize_rocprof_generic_and_trace(tmp_path):
    tool = os.path.join(ROOT, "tools", "summarize_rocprof.py")
    pmc = tmp_path / "counter_collection.csv"
    with open(pmc, "w", newline="") as f:
        w = csv.writer(f)
        w.writerow(["Kernel_Name", "Grid_Size", "Counter_Name", "Counter_Value"])
        for v in (10.0, 30.0):
            w.writerow(["void ll::reg_solve_kernel<0>(ll::RegDev, ll::RegConst)", 131072, "SQ_INSTS_VALU", v])
        w.writerow(["void ll::reg_solve_kernel<0>(ll::RegDev, ll::RegConst)", 512, "SQ_INSTS_VALU", 7.0])
        w.writerow(["void other::kernel()", 64, "SQ_INSTS_VALU", 99.0])
    out = subprocess.run([sys.executable, tool, "generic", str(pmc)], capture_output=True, text=True, check=True).stdout.strip().split("\n")
    # first line: which kernel sources the numbers belong to (tools/build_id.py; bench.py compares it with the tree it runs on)
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    from build_id import build_id
    assert out[0] == "# build " + build_id()
    out = out[1:]
    assert out[0] == "kernel,grid_threads,dispatches,SQ_INSTS_VALU_avg"
    assert out[1] == "ll::reg_solve_kernel<0>,131072,2,20.0" and out[2] == "ll::reg_solve_kernel<0>,512,1,7.0" and len(out) == 3
    trace = tmp_path / "kernel_trace.csv"
    with open(trace, "w", newline="") as f:
        w = csv.writer(f)
        w.writerow(["Kernel_Name", "Grid_Size_X", "Grid_Size_Y", "Grid_Size_Z", "Workgroup_Size_X", "VGPR_Count", "LDS_Block_Size", "Scratch_Size",
                    "Start_Timestamp", "End_Timestamp"])
        w.writerow(["void ll::reg_knn_kernel(ll::RegDev)", 1024, 2, 1, 128, 56, 0, 0, 1000, 3000])
        w.writerow(["void ll::reg_knn_kernel(ll::RegDev)", 1024, 2, 1, 128, 56, 0, 0, 5000, 9000])
    out = subprocess.run([sys.executable, tool, "trace", str(trace)], capture_output=True, text=True, check=True,
                         env=dict(os.environ, LL_GIT_COMMIT="abc1234")).stdout.strip().split("\n")
    assert out[0] == "# build " + build_id() + " commit abc1234"
    out = out[1:]
    assert out[1] == "ll::reg_knn_kernel,2048,128,56,0,0,2,0.006,3.0,2.0,4.0"


def test_summarize_rocprof_reads_register_counts_from_the_code_object():
    """VGPR / spill figures come from the gfx950 code objects embedded in the built library, not from rocprofv3's VGPR_Count
    column (allocation granules: half the register count)"""
    lib = os.path.join(ROOT, "loam_livox_amd", "libloamlivox_hip.so")
    if not os.path.exists(lib):
        import pytest
        pytest.skip("library not built")
    tool = os.path.join(ROOT, "tools", "summarize_rocprof.py")
    out = subprocess.run([sys.executable, tool, "codeobj", lib], capture_output=True, text=True, check=True).stdout.strip().split("\n")
    assert out[0].startswith("# build ")
    out = out[1:]
    assert out[0] == "kernel,vgpr,agpr,sgpr,vgpr_spill,sgpr_spill,scratch_bytes,lds_bytes,waves_per_simd"
    rows = {r.split(",")[0]: r.split(",") for r in out[1:]}
    sol = rows["ll::reg_solve_kernel"]
    assert int(sol[1]) == 256 and int(sol[7]) > 150000 and int(sol[8]) == 2   # one 512-thread workgroup per CU, two waves per SIMD
    assert 64 <= int(rows["ll::reg_knn_kernel"][1]) <= 128 and int(rows["ll::reg_knn_kernel"][6]) == 0


def test_bench_gpus_flag_launches_its_own_ranks():
    """`python bench.py --gpus 2` without a launcher starts two ranks under torch.distributed.run itself (the flag was
    parsed and ignored); --dry-run keeps the hot path out so the plumbing -- rendezvous on 127.0.0.1, MAX over ranks, gathered
    per-rank rates, n_gpus = the world size the process group saw -- runs on the CPU tier with gloo."""
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    out = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--dry-run", "--batch", "4", "--steps", "2"],
                         capture_output=True, text=True, env=env, timeout=300)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1  # rank 0 prints ONE line
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["requested_gpus"] == 2 and len(d["per_rank_scans_per_s"]) == 2
    assert d["value"] == 2 * 4 * 2 / 0.020  # all ranks' scans over the slowest rank's time


def test_bench_gpus_8_dry_run_and_every_handle_binds_local_rank():
    """The first hardware run with 8 ranks is the driver's: it must not be the first time 8 ranks meet.  `bench.py --gpus 8 --dry-run`
    (gloo, no GPU): one line, world size 8, every rank reporting its own LOCAL_RANK as the device it would bind -- and, read off
    the source, every device handle bench.py / bench_c4.py create is given that device (`device=dev`, dev = LOCAL_RANK)."""
    import ast
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    out = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "8", "--dry-run", "--batch", "4", "--steps", "2"],
                         capture_output=True, text=True, env=env, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    d = json.loads(lines[0])
    assert d["n_gpus"] == 8 and d["devices"] == list(range(8)) and len(d["per_rank_scans_per_s"]) == 8
    assert d["value"] == 8 * 4 * 2 / 0.080
    handles = {"Livox_laser", "Map_buffer", "Point_cloud_registration", "VoxelGrid", "History_buffer", "Cell_map", "Laser_mapping"}
    for script in ("bench.py", "bench_c4.py"):
        tree = ast.parse(open(os.path.join(root, script)).read())
        calls = [n for n in ast.walk(tree) if isinstance(n, ast.Call) and isinstance(n.func, ast.Name) and n.func.id in handles]
        assert len(calls) >= 3, script
        for c in calls:
            kw = {k.arg: k.value for k in c.keywords}
            assert "device" in kw and isinstance(kw["device"], ast.Name) and kw["device"].id in ("dev", "local_rank"), (script, c.lineno)
        for a in (n for n in ast.walk(tree) if isinstance(n, ast.Assign) and any(isinstance(t, ast.Name) and t.id == "dev" for t in n.targets)):
            assert "local_rank" in ast.unparse(a.value), script   # dev = LOCAL_RANK
        for a in (n for n in ast.walk(tree) if isinstance(n, ast.Assign) and any(isinstance(t, ast.Name) and t.id == "local_rank" for t in n.targets)):
            assert "LOCAL_RANK" in ast.unparse(a.value), script


@pytest.mark.parametrize("n_slots", [1, 2, 3, 4])
@pytest.mark.parametrize("k_steps", [1, 2, 5, 20])
def test_pipeline_schedule_starts_and_collects_every_batch_once_and_never_reuses_a_busy_slot(n_slots, k_steps):
    """bench.py's batches-in-flight loop: all K batches are started and collected inside the call (nothing hoisted out of the timed
    region, nothing skipped), in order, at most n_slots in flight, and a slot is not started again before its batch was collected."""
    import bench
    events, busy, started, collected = [], {}, [], []

    def start(slot):
        assert slot not in busy, "slot started again while its batch is in flight"
        busy[slot] = len(started)
        started.append(len(started))
        events.append(("s", slot))
        assert len(busy) <= max(1, n_slots)

    def collect(slot):
        assert slot in busy, "collect of a slot with nothing in flight"
        collected.append(busy.pop(slot))
        events.append(("c", slot))
        return collected[-1]

    last = bench.pipeline_schedule(k_steps, n_slots, start, collect)
    assert started == list(range(k_steps)) and collected == list(range(k_steps)) and last == k_steps - 1 and not busy
    if n_slots > 1 and k_steps >= n_slots:
        assert max(sum(1 for e in events[:i + 1] if e[0] == "s") - sum(1 for e in events[:i + 1] if e[0] == "c") for i in range(len(events))) == n_slots
