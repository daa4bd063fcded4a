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


def test_stream_argument_mapping():
    """GridHandle's `stream=` arguments: None -> `default_stream`, whose default is 0 = HIP's default stream = the C ABI's NULL
    (0 is what torch.cuda.current_stream().cuda_stream reports for PyTorch's default stream: a launch is then ordered with the
    torch work that produced its inputs); anything else unchanged.  The handle's own non-blocking streams are not what NULL
    means any more (that default raced with prefills); they come from grid_stream()."""
    from gridcodegenerator_amd import host

    class H:                     # (no GPU needed: the mapping is a pure function of the handle's default_stream)
        default_stream = 0
    f = host.GridHandle._stream
    assert f(H, None) is None and f(H, 0) is None and f(H, 0x7f00dead) == 0x7f00dead
    H.default_stream = 0x55aa
    assert f(H, None) == 0x55aa and f(H, 0) is None
    H.default_stream = None
    assert f(H, None) is None
    capi = open(os.path.join(REPO, "gridcodegenerator_amd", "csrc", "grid_capi.hip")).read()
    assert "return (hipStream_t)stream;" in capi and "h->streams[0]" not in capi.split("pick_stream")[1].split("}")[0]
    assert "grid_stream" in declared_symbols()
