"""The build variants that FAILED on the GPU in round 1, as buildable robots of their own: the same models
under another name with the generator options that reproduce the variant.  They are test infrastructure -- built by
tools/build_regression_variants.py (minutes to half an hour each), never by __graft_entry__.build(); the GPU tests that use
them skip when the library is absent.

  (a) atlas30_capped  register-capped column groups (waves_per_simd = 2): memory fault at K = 16384 in round 1
  (b) atlas30_fused   fused demand-ordered schedule for a 30-joint robot: wrong numbers from the USE_QDD_MINV kernel in round 1
  (c) iiwa7_fp64      all-double arithmetic: wrong results / hangs in round 1

All three are heavy spillers; what made them fail was spill code placed inside the reduced-EXEC region of a lane-divergent branch
(section 9.1).  The generated kernels no longer contain such branches, so they are expected to pass now -- which is the
regression test of that fix."""

VARIANTS = {
    "atlas30_capped": dict(base="atlas30", precision="fp32", gen=dict(waves_per_simd=2, allow_unverified=True)),
    "atlas30_fused": dict(base="atlas30", precision="fp32", gen=dict(grad_schedule="fused", grad_splits=[], allow_unverified=True)),
    "iiwa7_fp64": dict(base="iiwa7", precision="fp64", gen=dict(allow_unverified=True)),
}
# not a round-1 failure but the same regime taken further: all-double arithmetic for the 30-joint robot (never built before round 2;
# tests/test_round1_regressions.py::test_all_double_atlas runs it when the library is present)
EXTRA = {
    "atlas30_fp64": dict(base="atlas30", precision="fp64", gen=dict(allow_unverified=True, grad_splits=[4])),
}


def register():
    """Register every variant as a robot (idempotent); returns {name: precision}."""
    from gridcodegenerator_amd import host, robots
    for name, v in list(VARIANTS.items()) + list(EXTRA.items()):
        if name not in robots.REGISTERED_ROBOTS:
            robots.register_robot(name, (lambda b: (lambda: robots.get_robot(b)))(v["base"]))
        host.DEFAULT_GEN_KWARGS[name] = dict(v["gen"])
    return {name: v["precision"] for name, v in list(VARIANTS.items()) + list(EXTRA.items())}
