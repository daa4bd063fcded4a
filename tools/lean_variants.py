"""Experiment variants of the Atlas-30 library around the register-lean 8-wave kernel (round 4): the same robot under other names with
other generator options, sharing the shipped library's object cache (host.OBJECT_CACHE_BASE) so that only the kernel under study is
compiled.  usage: python tools/lean_variants.py [name ...]     (builds them; the GPU scripts load them by name)"""
import os, sys, time
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)

# cores.lean_plan's options as they were when the kernel was first shipped this round (scattered half-column sets, one 30-value flush per
# half-column, c and u in LDS); the variants measured before the store-path work are expressed on top of it.  (That build also staged
# all 3n inputs through LDS in every wave; the lean kernels now always read their inputs from the configuration's row, so
# profiles/r04/lean_store_path.txt's first line -- 'atlas30' at the time -- cannot be rebuilt bit for bit; 'atlas30_rl' is the closest.)
FIRST = {"order": "lpt", "umc": False, "aligned_flush": False, "chain_f": False}


def first(**kw):
    return dict(experimental={"lean_plan": dict(FIRST, **kw)})


VARIANTS = {
    # timing probes (NOT correct kernels): what a tile costs before its first gradient column / with only one wave per SIMD working
    "atlas30_probe_prefix": dict(experimental={"lean_probe": "prefix", "lean_plan": dict(FIRST)}),
    "atlas30_probe_older": dict(experimental={"lean_probe": "older", "lean_plan": dict(FIRST)}),
    "atlas30_probe_younger": dict(experimental={"lean_probe": "younger", "lean_plan": dict(FIRST)}),
    # LDS reads issued ahead of their use (Tracer.emit read_ahead): a round trip is 64-128 cycles, ~400 of them per wave and tile
    "atlas30_ahead24": dict(experimental={"lean_read_ahead": 24, "lean_plan": dict(FIRST)}),
    "atlas30_ahead48": dict(experimental={"lean_read_ahead": 48, "lean_plan": dict(FIRST)}),
    # share of the gradient half-columns given to the waves dispatched second (younger_speed), parked columns per wave
    "atlas30_xy80": first(younger_speed=0.8),
    "atlas30_xy125": first(younger_speed=1.25),
    "atlas30_p4": first(max_parked=4),
    "atlas30_p2": first(max_parked=2),
    # wave-per-configuration kernels compiled for two waves per SIMD (<= 256 registers instead of 295: 1024 resident blocks instead of 512)
    "atlas30_wocc2": dict(experimental={"wave_occupancy": 2}),
    # X_j kept alive for the way back up the tree where the subtree below j has at most this many joints (instead of rebuilding it)
    "atlas30_kx4": first(keep_x_below=4),
    "atlas30_kx8": first(keep_x_below=8),
    # only the articulated-inertia chain is serial; the F recursions of the backward pass move to the per-column phase (all eight waves)
    "atlas30_cfc": first(columns_from_chain=True),
    # both halves of a column in one wave: one recursion, the two -Minv dc products one after the other (n accumulators instead of 2 n)
    "atlas30_pph": first(products_per_half=True),
    # ... two passes (d/dq recursion, d/dqd recursion) instead of one carrying both
    "atlas30_sep": first(separate_halves=True),
    # u - c published instead of c and u (same planner, same sink as the first form)
    "atlas30_rl": first(umc=True),
    # one contiguous run of d/dq columns and one of d/dqd columns per wave (neighbours in the output row flushed one after the other), same sink
    "atlas30_runs2": first(order="runs"),
    # ... + the output cut at the 32-byte sectors of the row (AlignedPieces, grid_out_pieces), u - c published: the shipped kernel without chain_f
    "atlas30_al": dict(experimental={"lean_plan": {"chain_f": False}}),
    # chain_f (the accumulated force of the column finished last reused by its parent's column) in BOTH lean kernels / in neither
    # two half-columns of one tree share ONE -Minv dc product: every Minv entry read once for both, packed multiply-adds (v_pk_fma_f32)
    "atlas30_pair": dict(experimental={"lean_plan": {"pair_products": True}}),
    "atlas30_chain_both": dict(experimental={"lean_id_plan": {"chain_f": True}}),
    "atlas30_nochain": dict(experimental={"lean_plan": {"chain_f": False}, "lean_id_plan": {"chain_f": False}}),
    # (scheduling fences every N statements apply to every core of the header: such a variant recompiles all 26 kernels -- not registered)
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
