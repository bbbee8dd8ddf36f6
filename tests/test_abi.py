"""The C-ABI library: it loads without a GPU and exports every symbol include/dt_hip.h declares; the
ctypes binding lists every one of them (no compute calls here)."""
import os
import re
import subprocess

import pytest

from distillation_trajectories_amd import _hip

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "dt_hip.h")


def declared_functions():
    text = open(HEADER).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(dt_[a-z0-9_]+)\s*\(", text)))


@pytest.fixture(scope="module")
def lib_path():
    from distillation_trajectories_amd.csrc.build import LIB, build
    return build() if not os.path.exists(LIB) else LIB


def test_header_declares_the_documented_surface():
    names = declared_functions()
    for must in ("dt_unet_create", "dt_unet_forward", "dt_cfg_update", "dt_sample_trajectory", "dt_traj_metrics",
                 "dt_traj_wasserstein", "dt_traj_resampled_distance", "dt_unet_time_bias", "dt_unet_workspace_bytes"):
        assert must in names


def test_library_exports_every_declared_symbol(lib_path):
    out = subprocess.run(["nm", "-D", "--defined-only", lib_path], capture_output=True, text=True, check=True).stdout
    exported = set(re.findall(r" T (dt_[a-z0-9_]+)", out))
    missing = [n for n in declared_functions() if n not in exported]
    assert not missing, f"declared in dt_hip.h but not exported: {missing}"


def test_binding_covers_every_declared_symbol(lib_path):
    assert sorted(_hip.SIGNATURES) == declared_functions()
    lib = _hip.load(lib_path)                      # sets argtypes for every symbol; no HIP call is made
    assert lib.dt_abi_version() == _hip.ABI_VERSION
    assert lib.dt_status_string(0) == b"ok" and lib.dt_status_string(-2).startswith(b"unsupported")
    assert lib.dt_profile_class_count() > 0


def test_library_is_not_pinned_to_a_rocm_runpath(lib_path):
    """It must bind to the HIP runtime torch already loaded, so it carries no RPATH/RUNPATH of its own."""
    dyn = subprocess.run(["readelf", "-d", lib_path], capture_output=True, text=True, check=True).stdout
    assert "RUNPATH" not in dyn and "RPATH" not in dyn
    assert "libamdhip64.so" in dyn


def test_missing_library_fails_loudly(tmp_path, monkeypatch):
    monkeypatch.setattr(_hip, "_LIB", None)
    with pytest.raises(_hip.HipLibraryError, match="no CPU fallback"):
        _hip.load(str(tmp_path / "libdt_hip.so"))
