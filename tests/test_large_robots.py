"""How far the straight-line design scales, said by the generator instead of found out by the build (the
reference emits for any robot object -- GRiDCodeGenerator.py:37-46, helpers/_topology_helpers.py:193-258 -- and leaves the rest to
nvcc).  A synthetic 36-joint humanoid (torso 3, two arms of 9, head 1, two legs of 7: one tree of 22 joints and two of 7) is beyond
every fixed-size resource of the large-robot kernels: 2 x 22 gradient columns do not fit... the 64 lanes of a wave group do, but the
exchange region of the tile-cooperative kernels does not fit the LDS.  The generator must (a) still emit a complete, compilable
header, (b) name every kernel family it left out in a GridGenerationWarning and in the header's constants, (c) keep the kernels it
does emit inside the build guard -- checked here by cross-compiling the two cheapest ones for gfx950."""
import os
import re
import subprocess
import warnings

import numpy as np
import pytest

from gridcodegenerator_amd.robot_model import Joint, RobotModel


def synthetic_humanoid(arm=9, leg=7, name="synth36"):
    rng = np.random.default_rng(36)
    joints = []

    def chain(prefix, count, first_parent):
        parent = first_parent
        for i in range(count):
            nm = "%s%d" % (prefix, i)
            joints.append(Joint(nm, parent, axis=int(rng.integers(0, 3)), jtype="revolute", xyz=tuple(rng.uniform(-0.3, 0.3, 3)), rpy=(0.0, 0.0, 0.0),
                                damping=0.0, link_name=nm + "_link", mass=float(rng.uniform(0.5, 5.0)), com=tuple(rng.uniform(-0.05, 0.05, 3)),
                                inertia=(0.02, 0.0, 0.0, 0.03, 0.0, 0.01)))
            parent = nm
        return parent
    top = chain("torso", 3, None)
    chain("l_arm", arm, top)
    chain("head", 1, top)
    chain("r_arm", arm, top)
    chain("l_leg", leg, None)
    chain("r_leg", leg, None)
    return RobotModel(name, joints, base_link_name="pelvis")


@pytest.fixture(scope="module")
def generated(tmp_path_factory):
    from gridcodegenerator_amd import GRiDCodeGenerator
    from gridcodegenerator_amd.GRiDCodeGenerator import GridGenerationWarning
    robot = synthetic_humanoid()
    assert robot.get_num_pos() == 36
    d = tmp_path_factory.mktemp("synth36")
    cwd = os.getcwd()
    os.chdir(d)
    try:
        # (the gradient families that take minutes to trace for 36 joints are not what this test is about: no column splits, no two-pass)
        gen = GRiDCodeGenerator(robot, FILE_NAMESPACE="grid_synth36", grad_splits=[], pipeline=False)
        with warnings.catch_warnings(record=True) as caught:
            warnings.simplefilter("always")
            gen.gen_all_code()
    finally:
        os.chdir(cwd)
    return gen, [w for w in caught if issubclass(w.category, GridGenerationWarning)], str(d / "grid_synth36.hip.h")


def test_generator_names_what_it_leaves_out_beyond_32_joints(generated):
    gen, caught, header = generated
    code = gen.code_str
    assert len(caught) == 1
    text = str(caught[0].message)
    assert "synth36 (36 joints)" in text
    # the exchange regions (upper triangle of Minv of a 22-joint tree + two 7-joint trees = 309 slots + inputs + staging) exceed 160 KB
    assert "const int FD_DU_COOP_WAVES = 0;" in code and "FD_DU_COOP_WAVES = 0" in text
    assert "const int FD_DU_LEAN_WAVES = 0;" in code and "FD_DU_LEAN_WAVES = 0" in text
    # the wave-per-configuration kernels still fit (22 joints: 44 lanes) -- and say so by existing
    assert re.search(r"const int FD_DU_WAVE_WAVES = [1-9]", code) and "FD_DU_WAVE_WAVES = 0" not in text
    # the unsplit recomputing gradient kernel keeps Minv (22*23/2 + 2*28 = 309 values) in registers: predicted and reported
    assert gen.predicted_live["forward_dynamics_gradient_kernel"] > 300
    assert ("expect it to spill" in text) == (gen.predicted_live["forward_dynamics_gradient_kernel"] > 480)
    assert gen.generation_notes and all(note in text for note in gen.generation_notes)
    # every public kernel of the reference's interface is still there
    for name in ("inverse_dynamics_kernel", "direct_minv_kernel", "forward_dynamics_kernel", "inverse_dynamics_gradient_kernel",
                 "forward_dynamics_gradient_kernel"):
        assert "void %s(" % name in code


def test_a_wave_group_beyond_32_joints_drops_the_wave_kernels(tmp_path, monkeypatch):
    """... and a single tree of more than 32 joints (2m > 64 lanes) drops the wave-per-configuration kernels, by name."""
    from gridcodegenerator_amd.emit import wave
    from gridcodegenerator_amd.emit.model import RobotSpec
    robot = synthetic_humanoid(arm=14, leg=2, name="synth36b")          # torso 3 + 14 + 1 + 14 = one tree of 32... + 2 + 2
    spec = RobotSpec(robot)
    assert spec.n == 36 and wave.wave_groups(spec) is not None           # 32 joints: exactly 64 lanes
    robot = synthetic_humanoid(arm=15, leg=1, name="synth36c")          # one tree of 34 joints
    assert wave.wave_groups(RobotSpec(robot)) is None


def test_cheap_kernels_of_the_36_joint_robot_stay_inside_the_build_guard(generated, tmp_path):
    """RNEA and forward dynamics of the 36-joint robot cross-compiled for gfx950 (seconds each): within the guard host.build_library
    applies to every kernel (scratch per lane, SGPR spills), no instruction writes EXEC."""
    from gridcodegenerator_amd import host, isa_audit
    gen, caught, header = generated
    insts = {decl.split("::")[1].split("<")[0]: k for k, decl in enumerate(gen.kernel_instances)}
    for name in ("inverse_dynamics_kernel", "forward_dynamics_kernel"):
        obj = str(tmp_path / (name + ".o"))
        cmd = [host._hipcc(), "--offload-arch=" + host.ARCH, "-O3", "-ffp-contract=off", "-fPIC", "-std=c++17", "-I" + host.INCLUDE_DIR] + host.KERNARG_PRELOAD + [
            "-c", "-DGRID_HEADER=\"%s\"" % header, "-DGRID_NS=grid_synth36", "-Rpass-analysis=kernel-resource-usage", "-DGRID_INST=%d" % insts[name],
            host.KERNEL_INST_SRC, "-o", obj]
        proc = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=900)
        assert proc.returncode == 0, proc.stdout[-3000:]
        res = [k for k in host.parse_kernel_resources(proc.stdout) if k["name"] == name]
        assert res, proc.stdout[-2000:]
        for k in res:
            assert k.get("scratch", 0) <= host.MAX_SCRATCH_BYTES_PER_LANE and k.get("sgpr_spills", 0) <= host.MAX_SGPR_SPILLS, k
            assert k.get("vgprs", 0) <= 512
        assert not isa_audit.offenders(obj)
