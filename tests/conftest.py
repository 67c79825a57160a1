import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run on the GPU box with -m gpu)")


def pytest_collection_modifyitems(config, items):
    """GPU-tier tests are skipped (not failed) on a machine without a HIP device; the HIP library itself never falls back."""
    has_gpu = None
    for item in items:
        if item.get_closest_marker("gpu") is None:
            continue
        if has_gpu is None:
            try:
                import ctypes
                hip = ctypes.CDLL("libamdhip64.so")
                n = ctypes.c_int(0)
                has_gpu = hip.hipGetDeviceCount(ctypes.byref(n)) == 0 and n.value > 0
            except OSError:
                has_gpu = False
        if not has_gpu:
            item.add_marker(pytest.mark.skip(reason="no HIP device (GPU tier runs with -m gpu on an MI355X box)"))


@pytest.fixture(scope="session")
def small_world():
    """200k-point map (BASELINE config C1 size) + kd-trees of the oracle."""
    from loam_livox_amd import synth
    from oracle import orc
    world, corner, surf = synth.make_maps(200_000)
    return dict(world=world, corner=corner, surf=surf, tree_c=orc.KdTree(corner), tree_s=orc.KdTree(surf))


@pytest.fixture(scope="session")
def scans(small_world):
    from loam_livox_amd import synth
    return [synth.make_scan(small_world["world"], k) for k in range(4)]


@pytest.fixture(scope="session")
def gpu_lib():
    """The HIP library on a real device; GPU tests must not silently fall back to anything else."""
    from loam_livox_amd import capi
    L = capi.load()
    return L


def oracle_features(scan, current_time=1.0, min_blur=0.0, max_blur=1.0, params=None):
    from oracle import orc
    fe = orc.fe_extract(scan.xyzi, current_time, params)
    ci, si, fi = orc.fe_get_features(fe, min_blur, max_blur)
    return fe, ci, si, fi, orc.feature_cloud(fe, ci), orc.feature_cloud(fe, si)
