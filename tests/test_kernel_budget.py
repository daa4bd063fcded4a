"""Compiler-reported resources of the shipped kernels (regression guard for DESIGN.md section 9: the round-1 GPU failures all came
from kernels that needed kilobytes of scratch and ~200 SGPR spills because every store address had become a spilled pointer
induction variable).  Reads the build logs of the libraries built by __graft_entry__.build(); no GPU needed."""
import os

import pytest

from gridcodegenerator_amd import host


def _resources(robot, precision):
    if not os.path.exists(host.library_paths(robot, precision)["log"]):
        pytest.skip("library not built (run __graft_entry__.build())")
    res = host.kernel_resources(robot, precision)
    assert len(res) >= 17, "build log incomplete"
    return res


ALL_LIBRARIES = [("iiwa7", "fp32"), ("mixed5", "fp32"), ("atlas30", "fp32"), ("iiwa7", "mixed"), ("mixed5", "mixed"), ("atlas30", "mixed")]


@pytest.mark.parametrize("robot,precision", ALL_LIBRARIES)
def test_every_kernel_is_within_the_build_guard(robot, precision):
    for k in _resources(robot, precision):
        assert k["scratch"] <= host.MAX_SCRATCH_BYTES_PER_LANE, k
        assert k["sgpr_spills"] <= host.MAX_SGPR_SPILLS, k
        assert k["vgprs"] + k.get("agprs", 0) <= 512, k


def test_small_robot_kernels_do_not_touch_scratch():
    for precision in ("fp32", "mixed"):
        for k in _resources("iiwa7", precision):
            assert k["sgpr_spills"] <= 32, k     # (mixed: double constants live in SGPR pairs; SGPR spills go to VGPR lanes, not scratch)
            capped = any(t in k["name"] for t in ("split2", "split3", "split4"))
            if not capped:
                assert k["scratch"] == 0, k
            else:      # the 2-, 3- and 4-way splits are compiled for <= 256 registers (two blocks per CU: a second stream's launch
                assert k["scratch"] <= 256, k      # can share the chip): a handful of spilled values, 0.8 % at K = 16384


def test_atlas_column_groups_are_spill_free_after_the_addressing_fix():
    """Round 1: inverse_dynamics_gradient_kernel_split4 676 B of scratch, forward_dynamics_gradient_kernel_split4 1604 B / 1022
    spills, 183-199 SGPR spills each.  The store addresses are no longer induction variables (helpers/_runtime_emit.py: flush_len)."""
    res = {k["name"]: k for k in _resources("atlas30", "fp32") if "split4" in k["name"] or "coop" in k["name"]}
    did = res["inverse_dynamics_gradient_kernel_split4"]
    assert did["scratch"] == 0 and did["vgpr_spills"] <= 8 and did["sgpr_spills"] <= 80, did
    dfd = res["forward_dynamics_gradient_kernel_split4"]
    assert dfd["scratch"] <= 700 and dfd["vgpr_spills"] <= 260, dfd
    coop = res["forward_dynamics_gradient_kernel_coop"]
    assert coop["scratch"] <= host.MAX_SCRATCH_BYTES_PER_LANE, coop


@pytest.mark.parametrize("robot,precision", ALL_LIBRARIES)
def test_no_kernel_writes_exec(robot, precision):
    """No lane-divergent control flow in any shipped kernel (gridcodegenerator_amd/isa_audit.py): hipcc placed
    spill code -- among it the copy of the lane id -- at the head of the join block of a divergent branch, before the `s_or_b64 exec`
    that restores the mask, and the masked lanes' spills never happened (wild stores, memory faults at small batches).  The ISA of
    every kernel of every library is disassembled here and must not contain a single instruction that writes EXEC."""
    from gridcodegenerator_amd import isa_audit
    lib = host.library_paths(robot, precision)["lib"]
    if not os.path.exists(lib):
        pytest.skip("library not built (run __graft_entry__.build())")
    res = isa_audit.audit(lib)
    assert len(res) >= 17
    assert all(n > 300 for (n, _) in res.values()), "disassembly incomplete"       # (the smallest kernel: mixed5 direct_minv_kernel_wave, 445)
    assert {isa_audit.short_name(k): e for k, (n, e) in res.items() if e} == {}


def test_the_audit_sees_exec_writes(tmp_path):
    """The audit's pattern against a kernel that does have a lane-divergent branch (compiled here, a few seconds)."""
    from gridcodegenerator_amd import isa_audit
    src = tmp_path / "k.hip"
    src.write_text("#include <hip/hip_runtime.h>\n__global__ void divergent(float *p){if (threadIdx.x & 1){p[threadIdx.x] = __sinf(p[threadIdx.x]);} p[threadIdx.x + 64] = 1.f;}\n"
                   "__global__ void uniform(float *p, int n){for (int i = 0; i < n; i++){p[threadIdx.x + 64*i] = 2.f;}}\n")
    obj = tmp_path / "k.o"
    import subprocess
    subprocess.check_call([host._hipcc(), "--offload-arch=gfx950", "-O2", "-c", str(src), "-o", str(obj)])
    res = {k: v for k, v in isa_audit.audit(str(obj)).items()}
    div = [v for k, v in res.items() if "divergent" in k][0]
    uni = [v for k, v in res.items() if "uniform" in k][0]
    assert div[1] >= 1 and uni[1] == 0, res


def test_resource_parser():
    text = """x.hip.h:1:1: remark: Function Name: _ZN10grid_iiwa723inverse_dynamics_kernelIffEEvPT_ [-Rpass-analysis=kernel-resource-usage]
x.hip.h:1:1: remark:     TotalSGPRs: 102 [-Rpass-analysis=kernel-resource-usage]
x.hip.h:1:1: remark:     VGPRs: 113 [-Rpass-analysis=kernel-resource-usage]
x.hip.h:1:1: remark:     AGPRs: 4 [-Rpass-analysis=kernel-resource-usage]
x.hip.h:1:1: remark:     ScratchSize [bytes/lane]: 16 [-Rpass-analysis=kernel-resource-usage]
x.hip.h:1:1: remark:     Occupancy [waves/SIMD]: 4 [-Rpass-analysis=kernel-resource-usage]
x.hip.h:1:1: remark:     SGPRs Spill: 3 [-Rpass-analysis=kernel-resource-usage]
x.hip.h:1:1: remark:     VGPRs Spill: 2 [-Rpass-analysis=kernel-resource-usage]"""
    (k,) = host.parse_kernel_resources(text)
    assert k["name"] == "inverse_dynamics_kernel" and k["sgprs"] == 102 and k["vgprs"] == 113 and k["agprs"] == 4
    assert k["scratch"] == 16 and k["occupancy"] == 4 and k["sgpr_spills"] == 3 and k["vgpr_spills"] == 2
