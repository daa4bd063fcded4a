"""Experiment variants of the Atlas-30 library around the register-lean 8-wave kernel (round 4): the same robot under other names with
other generator options, sharing the shipped library's object cache (host.OBJECT_CACHE_BASE) so that only the kernel under study is
compiled.  usage: python tools/lean_variants.py [name ...]     (builds them; the GPU scripts load them by name)"""
import os, sys, time
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)

VARIANTS = {
    # timing probes (NOT correct kernels): what a tile costs before its first gradient column / with only one wave per SIMD working
    "atlas30_probe_prefix": dict(experimental={"lean_probe": "prefix"}),
    "atlas30_probe_older": dict(experimental={"lean_probe": "older"}),
    "atlas30_probe_younger": dict(experimental={"lean_probe": "younger"}),
    # LDS reads issued ahead of their use (Tracer.emit read_ahead): a round trip is 64-128 cycles, ~400 of them per wave and tile
    "atlas30_ahead24": dict(experimental={"lean_read_ahead": 24}),
    "atlas30_ahead48": dict(experimental={"lean_read_ahead": 48}),
    "atlas30_ahead96": dict(experimental={"lean_read_ahead": 96}),
    # share of the gradient half-columns given to the waves dispatched second (cores.lean_plan: younger_speed; shipped default 0.6)
    "atlas30_xy80": dict(experimental={"lean_plan": {"younger_speed": 0.8}}),
    "atlas30_xy100": dict(experimental={"lean_plan": {"younger_speed": 1.0}}),
    "atlas30_xy125": dict(experimental={"lean_plan": {"younger_speed": 1.25}}),
    "atlas30_xy100p3": dict(experimental={"lean_plan": {"younger_speed": 1.0, "max_parked": 3}}),
    # wave-per-configuration kernels compiled for two waves per SIMD (<= 256 registers instead of 295: 1024 resident blocks instead of 512)
    "atlas30_wocc2": dict(experimental={"wave_occupancy": 2}),
    # X_j kept alive for the way back up the tree where the subtree below j has at most this many joints (instead of rebuilding it)
    "atlas30_kx2": dict(experimental={"lean_plan": {"keep_x_below": 2}}),
    "atlas30_kx4": dict(experimental={"lean_plan": {"keep_x_below": 4}}),
    "atlas30_kx8": dict(experimental={"lean_plan": {"keep_x_below": 8}}),
    # only the articulated-inertia chain is serial; the F recursions of the backward pass move to the per-column phase (all eight waves)
    "atlas30_cfc": dict(experimental={"lean_plan": {"columns_from_chain": True}}),
    # every wave takes a contiguous run of whole gradient columns (neighbours in the output row are flushed one after the other)
    "atlas30_runs": dict(experimental={"lean_plan": {"order": "runs"}}),
    # a column whose two halves are one wave's: two passes (d/dq recursion, d/dqd recursion) instead of one carrying both
    "atlas30_sep": dict(experimental={"lean_plan": {"separate_halves": True}}),
    # inputs straight from the configuration's row + u - c published instead of c and u (same planner, same sink)
    "atlas30_rl": dict(experimental={"lean_plan": {"umc": True, "order": "lpt", "aligned_flush": False, "chain_f": False}}),
    # ... + contiguous runs per wave + output cut at the 32-byte sectors of the row (AlignedPieces, grid_out_pieces)
    "atlas30_al": dict(experimental={"lean_plan": {"chain_f": False}}),
    # the accumulated force of the column finished last reused by its parent's column (runs walked towards the root): off
    "atlas30_nochain": dict(experimental={"lean_plan": {"chain_f": False}, "lean_id_plan": {"chain_f": False}}),
    "atlas30_runs2": dict(experimental={"lean_plan": {"order": "runs"}}),
    "atlas30_runs_pph": dict(experimental={"lean_plan": {"order": "runs", "products_per_half": True}}),
    # both halves of a column in one wave: one recursion, the two -Minv dc products one after the other (n accumulators instead of 2 n)
    "atlas30_pph": dict(experimental={"lean_plan": {"products_per_half": True}}),
    "atlas30_p4": dict(experimental={"lean_plan": {"max_parked": 4}}),
    "atlas30_p2": dict(experimental={"lean_plan": {"max_parked": 2}}),
    # scheduling fences every N statements inside the lean cores (smaller reordering windows for hipcc: fewer spills, less overlap)
    # (fence_every applies to every core of the header: such a variant recompiles all 25 kernels -- not registered)
}


def register():
    from gridcodegenerator_amd import host, robots
    for name, kw in VARIANTS.items():
        if name not in robots.REGISTERED_ROBOTS:
            robots.register_robot(name, lambda: robots.get_robot("atlas30"))
        host.DEFAULT_GEN_KWARGS[name] = dict(kw)
        host.OBJECT_CACHE_BASE[name] = "atlas30"
    return list(VARIANTS)


if __name__ == "__main__":
    from gridcodegenerator_amd import host
    names = register()
    for name in (sys.argv[1:] or names):
        t0 = time.time()
        path = host.build_library(name, "fp32", verbose=True)
        res = [k for k in host.kernel_resources(name, "fp32") if k["name"].endswith("coop8")]
        print("[lean variants] %s -> %s (%.0f s) %s" % (name, path, time.time() - t0, res), flush=True)
