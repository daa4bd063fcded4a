"""Print, for every kernel of built libraries, how many instructions it has and how many of them write EXEC (lane-divergent control
flow; generated kernels are meant to have none: gridcodegenerator_amd/isa_audit.py).
usage: python tools/exec_audit.py [robot_precision ...]      (default: every libgrid_*.so under _build)"""
import glob, os, sys
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
from gridcodegenerator_amd import isa_audit  # noqa: E402

if __name__ == "__main__":
    build = os.path.join(REPO, "gridcodegenerator_amd", "_build")
    libs = [os.path.join(build, "libgrid_%s.so" % a) for a in sys.argv[1:]] or sorted(glob.glob(os.path.join(build, "libgrid_*.so")))
    bad = 0
    for lib in libs:
        for name, (n, e) in sorted(isa_audit.audit(lib).items()):
            print("%-28s %-44s [%s] %7d instructions  %4d write EXEC" % (os.path.basename(lib), isa_audit.short_name(name), name[-20:], n, e))
            bad += e > 0
    sys.exit(1 if bad else 0)
