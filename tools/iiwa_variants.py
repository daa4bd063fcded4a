"""Experiment variants of the iiwa-7 library around the headline kernel (round 4): column sets balanced on arithmetic + flush cost, and
the asymmetric 8-way split (4 heavy groups on the waves dispatched first, 4 light d/dqd groups on the waves behind them).
usage: python tools/iiwa_variants.py [name ...]   (builds them; tools/exp_iiwa_r04.py times them on the GPU box)"""
import os, sys, time
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)

VARIANTS = {
    "iiwa7_fl10": dict(experimental={"split_flush_slots": 10}),
    "iiwa7_fl20": dict(experimental={"split_flush_slots": 20}),
    "iiwa7_fl6": dict(experimental={"split_flush_slots": 6}),
    "iiwa7_fl14": dict(experimental={"split_flush_slots": 14}),
    "iiwa7_flx": dict(experimental={"split_flush_slots": "flush"}),      # the flush's own instruction count (iterations by run length) instead of a per-value rate
    "iiwa7_asym100": dict(experimental={"split_flush_slots": 10, "split_asym": 1.0}),
    "iiwa7_asym140": dict(experimental={"split_flush_slots": 10, "split_asym": 1.4}),
    "iiwa7_asym176": dict(experimental={"split_flush_slots": 10, "split_asym": 1.76}),
    "iiwa7_asym250": dict(experimental={"split_flush_slots": 10, "split_asym": 2.5}),
}


def register():
    from gridcodegenerator_amd import host
    for name, kw in VARIANTS.items():
        host.register_variant(name, "iiwa7", share_objects=True, **kw)
    return list(VARIANTS)


if __name__ == "__main__":
    from gridcodegenerator_amd import host
    names = register()
    for name in (sys.argv[1:] or names):
        t0 = time.time()
        path = host.build_library(name, "fp32")
        res = [(k["name"], k.get("vgprs"), k.get("scratch")) for k in host.kernel_resources(name, "fp32") if "forward_dynamics_gradient_kernel_split" in k["name"]]
        print("[iiwa variants] %s -> %s (%.0f s) %s" % (name, os.path.basename(path), time.time() - t0, res), flush=True)
