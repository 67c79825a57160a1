"""The C-ABI library loads without a GPU and exports every symbol include/loam_livox_hip.h declares.
No compute calls here (there is no CPU path to call)."""
import os
import re

import pytest

from loam_livox_amd import capi

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    text = open(os.path.join(ROOT, "include", "loam_livox_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(ll_[a-z0-9_]+)\s*\(", text)))


def test_library_is_built_in_tree():
    from loam_livox_amd import build
    assert os.path.exists(build.build())
    assert os.path.dirname(capi.LIB_PATH) == os.path.join(ROOT, "loam_livox_amd")


def test_every_declared_symbol_is_exported_and_bound():
    L = capi.load()
    names = declared_symbols()
    assert len(names) >= 30
    for n in names:
        assert hasattr(L, n), f"{n} declared in the header but not exported"
        assert n in capi.SYMBOLS, f"{n} has no ctypes prototype"
    assert sorted(capi.SYMBOLS) == names
    assert b"gfx950" in L.ll_version()


def test_default_params_match_reference_defaults():
    p = capi.fe_default_params()
    assert (p.thr_corner_curvature, p.thr_surface_curvature, p.minimum_view_angle) == (
        capi.C.c_float(0.05).value, capi.C.c_float(0.01).value, 10.0)  # LFX:152-154
    assert p.max_fov == 17.0 and abs(p.time_internal_pts - 1e-5) < 1e-12  # LFE:143,145
    r = capi.reg_default_params()
    assert (r.icp_max_iterations, r.ceres_max_iterations, r.ceres_prerun_times) == (20, 100, 2)  # PCR:89-91
    assert (r.maximum_dis_line_for_match, r.maximum_dis_plane_for_match) == (2.0, 50.0)  # PCR:64-65
    assert (r.huber_a, r.inliner_dis, r.inlier_ratio) == (0.1, 0.02, 0.8)  # PCR:220,97,98


def test_product_package_never_imports_the_oracle():
    """oracle/ is test infrastructure: no import / include / dlopen of it anywhere in the product package."""
    pkg = os.path.join(ROOT, "loam_livox_amd")
    pat = re.compile(r"^\s*(from\s+oracle|import\s+oracle|from\s+\.\.?oracle|#\s*include\s*[\"<].*oracle)|libll_oracle|orc\.", re.M)
    for dirpath, _, files in os.walk(pkg):
        if os.path.basename(dirpath) in ("build", "__pycache__"):
            continue
        for f in files:
            if f.endswith((".py", ".h", ".hip", ".cpp")):
                src = open(os.path.join(dirpath, f)).read()
                assert not pat.search(src), f"{f} uses oracle/"


def test_use_library_restores_the_product_library_when_the_variant_is_missing(tmp_path):
    """capi.use_library (how the tests reach the -DLL_AB_PATHS build): a variant that cannot be loaded raises and leaves the module
    bound to the library it had"""
    from loam_livox_amd import capi
    before = (capi._lib, capi.LIB_PATH)
    with pytest.raises(capi.LoamLivoxError):
        with capi.use_library(str(tmp_path / "no_such_variant.so")):
            pass
    assert (capi._lib, capi.LIB_PATH) == before
