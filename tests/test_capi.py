"""C-ABI library: builds for gfx950 without a GPU, loads, and exports every symbol include/grid_capi.h
declares.  No compute entry point is called here (no GPU in the build container)."""
import os
import re

import pytest

from gridcodegenerator_amd import host

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    with open(os.path.join(REPO, "include", "grid_capi.h")) as fh:
        text = re.sub(r"/\*.*?\*/", "", fh.read(), flags=re.S)
    return sorted(set(re.findall(r"\b(grid_[a-z_0-9]+)\s*\(", text)))


def test_header_and_binding_table_agree():
    assert declared_symbols() == sorted(name for (name, _, _) in host.CAPI_SIGNATURES)


@pytest.mark.parametrize("robot", ["mixed5", "iiwa7"])
def test_library_builds_loads_and_exports_all_symbols(robot):
    path = host.build_library(robot, "fp32")          # hipcc --offload-arch=gfx950 (cross-compiles here)
    assert os.path.exists(path)
    L = host.GridLibrary(robot)                        # resolves every symbol of CAPI_SIGNATURES or raises
    for sym in declared_symbols():
        assert hasattr(L.lib, sym), sym
    n = {"mixed5": 5, "iiwa7": 7}[robot]
    assert L.n == n and L.constants["NUM_JOINTS"] == n       # constant getters need no GPU
    assert L.constants["SUGGESTED_THREADS"] == 64
    assert L.lib.grid_robot_name().decode() == robot
    assert L.compute_dtype == "f32"


def test_library_contains_gfx950_code_object():
    path = host.build_library("mixed5", "fp32")
    with open(path, "rb") as fh:
        blob = fh.read()
    assert b"gfx950" in blob and b"forward_dynamics_gradient_kernel" in blob


def test_missing_library_fails_loudly(tmp_path):
    with pytest.raises(host.GridLibraryError):
        host.GridLibrary("iiwa7", path=str(tmp_path / "nope.so"))
