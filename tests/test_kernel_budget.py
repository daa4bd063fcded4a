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


@pytest.mark.parametrize("robot,precision", [("iiwa7", "fp32"), ("mixed5", "fp32"), ("atlas30", "fp32"), ("iiwa7", "mixed"), ("mixed5", "mixed")])
def test_every_kernel_is_within_the_build_guard(robot, precision):
    for k in _resources(robot, precision):
        assert k["scratch"] <= host.MAX_SCRATCH_BYTES_PER_LANE, k
        assert k["sgpr_spills"] <= host.MAX_SGPR_SPILLS, k
        assert k["vgprs"] + k.get("agprs", 0) <= 512, k


def test_small_robot_kernels_do_not_touch_scratch():
    for precision in ("fp32", "mixed"):
        for k in _resources("iiwa7", precision):
            assert k["sgpr_spills"] <= 16, k
            if "split2" not in k["name"]:      # (the 2-way split is capped at 256 registers for two waves per SIMD: 1 / 51 spilled values)
                assert k["scratch"] == 0, k
            else:
                assert k["scratch"] <= 256, k


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
