"""Build the round-1 failure variants (tests/regression_variants.py) so that tests/test_round1_regressions.py can run them on the
GPU.  usage: python tools/build_regression_variants.py [name ...]      (default: all; Atlas-30 variants take 15-40 minutes each)"""
import os, sys, time
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO); sys.path.insert(0, os.path.join(REPO, "tests"))
from gridcodegenerator_amd import host  # noqa: E402
import regression_variants  # noqa: E402

if __name__ == "__main__":
    todo = regression_variants.register()
    for name in (sys.argv[1:] or list(todo)):
        t0 = time.time()
        path = host.build_library(name, todo[name], verbose=True)
        print("[variants] %s -> %s (%.0f s)" % (name, path, time.time() - t0), flush=True)
        for k in host.kernel_resources(name, todo[name]):
            if k["scratch"] or k["sgpr_spills"] > 32:
                print("    %-50s scratch %5d B  SGPR spills %4d  VGPR spills %5d" % (k["name"], k["scratch"], k["sgpr_spills"], k["vgpr_spills"]))
