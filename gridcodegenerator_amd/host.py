"""Python host side: build the per-robot shared object and drive it through the C ABI (ctypes).

This replaces the reference user's nvcc ``main()`` that includes ``grid.cuh`` (SURVEY.md section
8(b)): the generated ``grid_<robot>.hip.h`` is compiled together with ``csrc/grid_capi.hip`` into
``_build/libgrid_<robot>_<precision>.so`` and every call below is one C-ABI entry point declared in
``include/grid_capi.h``.  There is NO CPU fallback: a missing library raises ``GridLibraryError``.

``GridHandle`` mirrors the reference's host wrappers -- same names, argument meaning and buffer
layouts (``q_qd_u[K][3n]`` in; ``c``, ``Minv``, ``qdd``, ``dc_du``, ``df_du`` out) -- and adds the
``*_device`` variants that operate on device pointers (reference mode 2, ``_compute_only``).
"""
import ctypes
import hashlib
import os
import re
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

import numpy as np

from .GRiDCodeGenerator import GRiDCodeGenerator
from .robots import get_robot

PKG_DIR = os.path.dirname(os.path.abspath(__file__))
REPO_DIR = os.path.dirname(PKG_DIR)
BUILD_DIR = os.path.join(PKG_DIR, "_build")
CSRC = os.path.join(PKG_DIR, "csrc", "grid_capi.hip")
KERNEL_INST_SRC = os.path.join(PKG_DIR, "csrc", "grid_kernel_inst.hip")
INCLUDE_DIR = os.path.join(REPO_DIR, "include")
ARCH = "gfx950"
# arithmetic variants that ship (libgrid_<robot>_<precision>.so built by __graft_entry__.build()) and that bench.py offers; the
# all-double arithmetic (precision="fp64") is accepted by the generator and verified on the GPU as a regression variant only
VERIFIED_PRECISIONS = ("fp32", "mixed")
DEFAULT_PRECISION = "fp32"

ALG_ID, ALG_MINV, ALG_FD, ALG_ID_DU, ALG_FD_DU = range(5)
ALG_NAMES = {ALG_ID: "inverse_dynamics", ALG_MINV: "direct_minv", ALG_FD: "forward_dynamics",
             ALG_ID_DU: "inverse_dynamics_gradient", ALG_FD_DU: "forward_dynamics_gradient"}


class GridLibraryError(RuntimeError):
    pass


def _hipcc():
    for cand in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", "hipcc"):
        if cand and (os.path.isabs(cand) and os.path.exists(cand) or not os.path.isabs(cand)):
            return cand
    return "hipcc"


def _source_fingerprint(extra=""):
    h = hashlib.sha256()
    for root in (os.path.join(PKG_DIR, "emit"), os.path.join(PKG_DIR, "helpers"), os.path.join(PKG_DIR, "algorithms")):      # (every .py of the generator)
        for fn in sorted(os.listdir(root)):
            if fn.endswith(".py"):
                with open(os.path.join(root, fn), "rb") as fh:
                    h.update(fh.read())
    for fn in (os.path.join(PKG_DIR, "GRiDCodeGenerator.py"), os.path.join(PKG_DIR, "robots.py"),
               os.path.join(PKG_DIR, "robot_model.py"), CSRC, KERNEL_INST_SRC, os.path.join(INCLUDE_DIR, "grid_capi.h")):
        with open(fn, "rb") as fh:
            h.update(fh.read())
    h.update(extra.encode())
    return h.hexdigest()


def library_paths(robot_name, precision="fp32"):
    tag = "%s_%s" % (robot_name, precision)
    return dict(tag=tag, header=os.path.join(BUILD_DIR, "grid_%s.hip.h" % tag),
                lib=os.path.join(BUILD_DIR, "libgrid_%s.so" % tag), stamp=os.path.join(BUILD_DIR, "grid_%s.stamp" % tag),
                log=os.path.join(BUILD_DIR, "grid_%s.build.log" % tag))


_QUALIFIER_LINE = re.compile(r"^\s*(template\s*<.*>|__host__|__device__|__global__|inline|__forceinline__|__launch_bounds__\(.*\)|\s)+$")


def _header_blocks(text):
    """Cut a generated header into (name, text) blocks at the function doc comments (`/**` at namespace indentation).
    name = the function the block defines (None for anything else: constants, structs, helper templates)."""
    lines = text.splitlines(keepends=True)
    starts = [i for i, l in enumerate(lines) if l.startswith("    /**")]
    bounds = [0] + starts + [len(lines)]
    blocks = []
    for a, b in zip(bounds[:-1], bounds[1:]):
        if a == b:
            continue
        chunk = lines[a:b]
        name = None
        i = 0
        if chunk[0].startswith("    /**"):
            while i < len(chunk) and "*/" not in chunk[i]:
                i += 1
            i += 1
            while i < len(chunk) and _QUALIFIER_LINE.match(chunk[i]):
                i += 1
            if i < len(chunk):
                m = re.match(r"\s*struct\s+(\w+)", chunk[i]) or re.search(r"(\w+)\s*\(", chunk[i])
                name = m.group(1) if m else None
        blocks.append((name, "".join(chunk)))
    return blocks


def kernel_dependency_hashes(header_text, kernel_names):
    """For each kernel name: a hash of the part of the generated header its translation unit depends on -- everything
    outside the algorithm section (constants, structs, lane/staging helpers, init/close) plus the transitive closure of the
    algorithm functions (cores, _device, launchers ...) the kernel's text mentions.  Adding or changing ANOTHER kernel
    family then leaves this kernel's object cache entry valid (an Atlas-30 gradient kernel takes minutes to compile)."""
    blocks = _header_blocks(header_text)
    names = [b[0] for b in blocks]
    try:
        first = names.index("load_update_XImats_helpers")
        last = names.index("init_grid")
    except ValueError:
        whole = hashlib.sha256(header_text.encode()).hexdigest()
        return {k: whole for k in kernel_names}
    outside = "".join(t for (nm, t) in blocks[:first] + blocks[last:])
    # the explicit-instantiation macros at the end name every kernel: keep only the lines that are not per-kernel macros
    outside = "\n".join(l for l in outside.splitlines() if not l.startswith("#define GRID_KERNEL_INST_")
                        and not l.startswith("#define GRID_FOR_EACH_KERNEL_INST") and not l.startswith("#define GRID_NUM_KERNEL_INSTANCES"))
    # the accessor / sink structs of the helper section (`struct grid_xyz {` ... `};` with the template and comment lines in front)
    # are hashed only into the kernels that use them: a NEW sink or accessor leaves every other kernel's object valid
    always, helper = [], {}
    lines = outside.split("\n")
    i = 0
    while i < len(lines):
        m = re.match(r"^    struct (grid_\w+) \{$", lines[i])
        if not m:
            always.append(lines[i])
            i += 1
            continue
        head = []
        while always and re.match(r"^    (template\b|//|/\*\*| \*)", always[-1]):
            head.insert(0, always.pop())
        j = i
        while j < len(lines) and lines[j] != "    };":
            j += 1
        helper[m.group(1)] = "\n".join(head + lines[i:j + 1])
        i = j + 1
    always = "\n".join(always)
    region = blocks[first:last]
    by_name = {}
    for idx, (nm, t) in enumerate(region):
        if nm is not None:
            by_name.setdefault(nm, []).append(idx)
    ident = re.compile(r"\b[A-Za-z_]\w*\b")
    refs = []
    for (nm, t) in region:
        found = set(w for w in ident.findall(t) if w in by_name and w != nm)
        refs.append(found)
    helper_refs = {nm: set(w for w in ident.findall(t) if w in helper and w != nm) for nm, t in helper.items()}
    always_uses = set(w for w in ident.findall(always) if w in helper)          # (helpers that free functions of the section mention)
    whole = hashlib.sha256(header_text.encode()).hexdigest()
    out = {}
    for k in kernel_names:
        if k not in by_name:
            out[k] = whole            # a kernel whose block was not recognised depends on everything (never a stale object)
            continue
        seen = set()
        stack = [k]
        while stack:
            nm = stack.pop()
            if nm in seen or nm not in by_name:
                continue
            seen.add(nm)
            for idx in by_name[nm]:
                stack.extend(refs[idx])
        h = hashlib.sha256(always.encode())
        used = set(always_uses)
        for idx, (nm, t) in enumerate(region):
            if nm in seen or nm is None:
                h.update(t.encode())
                used.update(w for w in ident.findall(t) if w in helper)
        stack = list(used)
        used = set()
        while stack:
            nm = stack.pop()
            if nm not in used:
                used.add(nm)
                stack.extend(helper_refs[nm])
        for nm in sorted(used):
            h.update(helper[nm].encode())
        out[k] = h.hexdigest()
    return out


# Build-time guards.  (1) The ISA audit (isa_audit.py): no kernel of a library may write EXEC.  Round 1's GPU failures (a memory
# fault, silently wrong numbers, hangs) and one of round 2 were register-spill code that hipcc had placed inside the reduced-EXEC
# region of a lane-divergent branch; a kernel without such regions cannot be hit, however much it spills:
# the Atlas-30 mixed-precision kernels (2.4-3.1 KB of scratch per lane, 400-700 SGPR spills -- exactly the regime that failed in
# round 1) are bit-repeatable and within 6e-7 of the oracle on the GPU since the kernels are branch-free.  (2) A resource bound
# that only flags a runaway build (the fused Atlas-30 regression variant, 4.4 KB of scratch per lane, passes on the GPU).
MAX_SCRATCH_BYTES_PER_LANE = int(os.environ.get("GRID_MAX_SCRATCH", "8192"))
MAX_SGPR_SPILLS = int(os.environ.get("GRID_MAX_SGPR_SPILLS", "2048"))      # (1 061 in the mixed-arithmetic lean kernel: double constants live in SGPR pairs; they go to VGPR lanes, not to memory)


def parse_kernel_resources(log_text):
    """-Rpass-analysis=kernel-resource-usage remarks of a build log -> [{name, sgprs, vgprs, agprs, scratch, occupancy,
    sgpr_spills, vgpr_spills}] (one entry per compiled kernel, overloads included)."""
    out = []
    cur = None
    keys = {"TotalSGPRs": "sgprs", "VGPRs": "vgprs", "AGPRs": "agprs", "ScratchSize [bytes/lane]": "scratch",
            "Occupancy [waves/SIMD]": "occupancy", "SGPRs Spill": "sgpr_spills", "VGPRs Spill": "vgpr_spills"}
    for line in log_text.splitlines():
        m = re.search(r"remark: Function Name: (\S+)", line)
        if m:
            mangled = m.group(1)
            nm = re.match(r"_ZN\d+\w+?(\d+)(\w+)", mangled)
            name = mangled
            mm = re.match(r"_ZN(\d+)", mangled)
            if mm:      # _ZN<len><namespace><len><function>...
                i = 3 + len(mm.group(1)) + int(mm.group(1))
                m2 = re.match(r"(\d+)", mangled[i:])
                if m2:
                    j = i + len(m2.group(1))
                    name = mangled[j:j + int(m2.group(1))]
            cur = {"name": name, "mangled": mangled}
            out.append(cur)
            continue
        m = re.search(r"remark:\s+([A-Za-z \[\]/]+): (\d+)", line)
        if m and cur is not None and m.group(1).strip() in keys:
            cur[keys[m.group(1).strip()]] = int(m.group(2))
    return out


def kernel_resources(robot_name, precision="fp32"):
    """Compiler-reported resources of every kernel of a built library (from its build log)."""
    with open(library_paths(robot_name, precision)["log"]) as fh:
        return parse_kernel_resources(fh.read())


# Kernel arguments preloaded into SGPRs at wave launch (gfx950: 16 user SGPRs, 2 of them the kernarg pointer): the explicit arguments
# of every kernel here (<= 12 dwords) and the first hidden ones (block counts, x / y block extents) need no s_load -- a wave's first
# memory instruction is the load of its inputs (helpers/_runtime_emit.py: grid_block_threads).  The code object keeps a prologue that
# loads them the old way on firmware without the feature.
KERNARG_PRELOAD = ["-mllvm", "-amdgpu-kernarg-preload-count=16"]


def generate_header(robot, path, namespace, **gen_kwargs):
    """Run the generator for ``robot`` and move ``<namespace>.hip.h`` (written to the CWD, as the
    reference writes grid.cuh to the CWD, GRiDCodeGenerator.py:308) to ``path``."""
    import shutil
    import tempfile
    gen = GRiDCodeGenerator(robot, FILE_NAMESPACE=namespace, **gen_kwargs)
    cwd = os.getcwd()
    os.makedirs(os.path.dirname(path), exist_ok=True)
    # a directory of its own: variants that share a namespace (OBJECT_CACHE_BASE) write the same file name, possibly at the same time
    work = tempfile.mkdtemp(prefix=".gen_", dir=os.path.dirname(path))
    os.chdir(work)
    try:
        gen.gen_all_code()
        os.replace(os.path.join(work, gen.output_file_name()), path)
    finally:
        os.chdir(cwd)
        shutil.rmtree(work, ignore_errors=True)
    return gen


def build_library(robot_name, precision="fp32", force=False, verbose=False, extra_flags=(), **gen_kwargs):
    """Generate + compile the shared object for a built-in robot.  Returns the .so path.
    Safe to call from several processes at once (one rank per GPU): an exclusive file lock serialises the build and the
    late-comers find the finished library."""
    import fcntl
    os.makedirs(BUILD_DIR, exist_ok=True)
    with open(os.path.join(BUILD_DIR, ".lock_%s_%s" % (robot_name, precision)), "w") as lock:
        fcntl.flock(lock, fcntl.LOCK_EX)
        try:
            return _build_library_locked(robot_name, precision, force, verbose, extra_flags, gen_kwargs)
        finally:
            fcntl.flock(lock, fcntl.LOCK_UN)


# Experiment variants of a built-in robot (same model, other generator options, registered under another name: tools/lean_variants.py,
# tests/regression_variants.py) may share the BASE robot's object cache: the variant's header then uses the base's namespace and its
# kernel translation units the base's defines, so every kernel whose slice of the header is unchanged is a cache hit and only the
# kernels under study are compiled (an Atlas-30 library is 25 kernels and 15-35 minutes from scratch).  {variant name: base name}
OBJECT_CACHE_BASE = {}

# generator options used when a built-in robot is built without explicit options (tests rely on these)
DEFAULT_GEN_KWARGS = {"mixed5": {"pipeline": True, "grad_schedule": "recompute", "experimental": {"grad_table": True, "split_sets": True}}}   # the small test robot exercises the two-pass kernels and the recomputing (LDS table) schedule, prismatic joints included


def register_variant(name, base, share_objects=False, **gen_kwargs):
    """The robot `base` once more under `name`, built with other generator options (on top of the base's defaults): a library of its
    own, libgrid_<name>_<precision>.so.  share_objects: reuse the base library's kernel objects where the generated text agrees
    (OBJECT_CACHE_BASE).  Idempotent."""
    from . import robots
    if name not in robots.REGISTERED_ROBOTS:
        robots.register_robot(name, lambda: robots.get_robot(base))
    DEFAULT_GEN_KWARGS[name] = dict(DEFAULT_GEN_KWARGS.get(base, {}), **gen_kwargs)
    if share_objects:
        OBJECT_CACHE_BASE[name] = base
    return name


# the prismatic test robot with the REFERENCE's gradient seed (motion cross product, _test.py:311,437): its HIP gradients are held to the
# reference's own golden numbers (tests/test_gpu_parity.py::test_golden_fixtures); the shipped default stays "corrected"
REFERENCE_GRADIENT_VARIANT = ("mixed5_refgrad", "mixed5")


def _build_library_locked(robot_name, precision, force, verbose, extra_flags, gen_kwargs):
    if not gen_kwargs:
        gen_kwargs = dict(DEFAULT_GEN_KWARGS.get(robot_name, {}))
    p = library_paths(robot_name, precision)
    # Optimisation level per translation unit.  Small robots: -O3 everywhere.  Large robots (n > 12): their gradient kernels
    # are single basic blocks of 50-70 k instructions that the generator has already scheduled; hipcc needs 4-20 minutes
    # for one at -O2/-O3 and the result is not faster (Atlas-30 dID, K = 65536: 372 us at -O3, 306-359 us at -O1).
    # RNEA and FD are fastest at -O3 (seconds to compile), Minv at -O1 (65 vs 85 us).
    large = get_robot(robot_name).get_num_joints() > 12
    opt = "-O3"

    def kernel_opt(decl):
        if not large:
            return "-O3"
        name = decl.split("::")[1].split("<")[0]
        if name in ("inverse_dynamics_kernel", "forward_dynamics_kernel"):
            return "-O3"
        return "-O1"

    flags = ["--offload-arch=" + ARCH, opt, "-ffp-contract=off", "-fPIC", "-shared", "-std=c++17", "-I" + INCLUDE_DIR] + KERNARG_PRELOAD + list(extra_flags)
    # (no absolute paths in the fingerprint: the same tree is mounted at different locations on different boxes)
    robot_obj = get_robot(robot_name)
    model_hash = hashlib.sha256(repr([(robot_obj.get_parent_id(j), robot_obj.get_damping_by_id(j), np.asarray(robot_obj.get_Imat_by_id(j)).tolist(),
                                       np.asarray(robot_obj.get_Xmat_Func_by_id(j)(0.3)).tolist()) for j in range(robot_obj.get_num_pos())]).encode()).hexdigest()
    fp = _source_fingerprint(repr(sorted(gen_kwargs.items())) + precision + model_hash + " ".join(f for f in flags if not f.startswith("-I")))
    if not force and os.path.exists(p["lib"]) and os.path.exists(p["stamp"]):
        with open(p["stamp"]) as fh:
            if fh.read().strip() == fp:
                return p["lib"]
    os.makedirs(BUILD_DIR, exist_ok=True)
    cache_base = OBJECT_CACHE_BASE.get(robot_name)
    ns = "grid_" + (cache_base or robot_name)
    gen = generate_header(robot_obj, p["header"], ns, precision=precision, **gen_kwargs)
    # one translation unit per kernel + the C-ABI unit, compiled in parallel, then linked
    common = [_hipcc()] + [f for f in flags if f != "-shared"] + [
        "-c", "-DGRID_HEADER=\"%s\"" % p["header"], "-DGRID_NS=" + ns, "-DGRID_ROBOT_NAME=\"%s\"" % (cache_base or robot_name),
        "-Rpass-analysis=kernel-resource-usage"]
    objdir = os.path.join(BUILD_DIR, "obj_" + p["tag"])
    if cache_base and not os.path.isdir(objdir):
        import shutil
        src_dir = os.path.join(BUILD_DIR, "obj_" + library_paths(cache_base, precision)["tag"])
        if os.path.isdir(src_dir):
            shutil.copytree(src_dir, objdir)
    os.makedirs(objdir, exist_ok=True)
    jobs = [("capi", common + ["-DGRID_EXTERN_KERNELS", CSRC, "-o", os.path.join(objdir, "capi.o")])]
    for k in range(len(gen.kernel_instances)):
        kflags = [kernel_opt(gen.kernel_instances[k]) if f == opt else f for f in common]
        jobs.append(("kernel%d" % k, kflags + ["-DGRID_INST=%d" % k, KERNEL_INST_SRC, "-o", os.path.join(objdir, "kernel%d.o" % k)]))
    jobs.sort(key=lambda j: {"-O2": 0, "-O3": 1}.get(next((f for f in j[1] if f in ("-O1", "-O2", "-O3")), "-O3"), 2) if large else 0)   # longest first
    # per-object cache: an object is reused when the generated header, its own source and the flags are unchanged (a C-ABI
    # edit then recompiles capi.o only -- the largest Atlas-30 kernel alone takes ~20 minutes)
    with open(p["header"], "rb") as fh:
        header_bytes = fh.read()
    header_hash = hashlib.sha256(header_bytes).hexdigest()
    inst_names = [decl.split("::")[1].split("<")[0] for decl in gen.kernel_instances]
    dep_hash = kernel_dependency_hashes(header_bytes.decode(), set(inst_names))

    def object_key(cmd):
        inst = next((a for a in cmd if a.startswith("-DGRID_INST=")), None)
        # a kernel object depends on its own slice of the header (and its instantiation line); the C-ABI unit on all of it
        if inst is not None:
            k = int(inst.split("=")[1])
            h = hashlib.sha256((dep_hash[inst_names[k]] + gen.kernel_instances[k]).encode())
        else:
            h = hashlib.sha256(header_hash.encode())
        with open(cmd[-3], "rb") as fh:
            h.update(fh.read())
        if inst is None:        # only the C-ABI unit includes grid_capi.h
            with open(os.path.join(INCLUDE_DIR, "grid_capi.h"), "rb") as fh:
                h.update(fh.read())
        h.update(" ".join(a for a in cmd[1:-3] if not a.startswith("-I") and not a.startswith("-DGRID_HEADER=") and not a.startswith("-DGRID_INST=")).encode())
        return h.hexdigest()

    def run(job):
        name, cmd = job
        obj = cmd[-1]
        key = object_key(cmd)
        if not force and os.path.exists(obj) and os.path.exists(obj + ".key") and os.path.exists(obj + ".log"):
            with open(obj + ".key") as fh:
                if fh.read().strip() == key:
                    with open(obj + ".log") as lf:
                        return name, cmd, 0, "(cached object)\n" + lf.read()
        if verbose:
            print("[grid build] %s: %s" % (p["tag"], name), file=sys.stderr)
        proc = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
        if proc.returncode == 0:
            with open(obj + ".log", "w") as lf:
                lf.write(proc.stdout)
            with open(obj + ".key", "w") as fh:
                fh.write(key)
        return name, cmd, proc.returncode, proc.stdout

    workers = max(1, min(len(jobs), int(os.environ.get("GRID_BUILD_JOBS", os.cpu_count() or 4))))
    with ThreadPoolExecutor(max_workers=workers) as pool:
        results = list(pool.map(run, jobs))
    with open(p["log"], "w") as fh:
        for (name, cmd, rc, out) in results:
            fh.write("### %s (rc=%d)\n%s\n%s\n" % (name, rc, " ".join(cmd), out))
    for (name, cmd, rc, out) in results:
        if rc != 0:
            raise GridLibraryError("hipcc failed for %s/%s (see %s):\n%s" % (p["tag"], name, p["log"], out[-4000:]))
    if not gen_kwargs.get("allow_unverified"):
        for (name, cmd, rc, out) in results:
            for k in parse_kernel_resources(out):
                if k.get("scratch", 0) > MAX_SCRATCH_BYTES_PER_LANE or k.get("sgpr_spills", 0) > MAX_SGPR_SPILLS:
                    raise GridLibraryError(
                        "%s: kernel %s needs %d B of scratch per lane and %d SGPR spills (limits %d / %d): builds this heavy faulted or "
                        "miscomputed on the GPU (DESIGN.md section 9); pass allow_unverified=True to build it anyway"
                        % (p["tag"], k["name"], k.get("scratch", 0), k.get("sgpr_spills", 0), MAX_SCRATCH_BYTES_PER_LANE, MAX_SGPR_SPILLS))
    link = [_hipcc(), "--offload-arch=" + ARCH, "-shared", "-fPIC"] + [j[1][-1] for j in jobs] + ["-o", p["lib"] + ".tmp"]
    proc = subprocess.run(link, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    with open(p["log"], "a") as fh:
        fh.write("### link (rc=%d)\n%s\n%s\n" % (proc.returncode, " ".join(link), proc.stdout))
    if proc.returncode != 0:
        raise GridLibraryError("link failed for %s (see %s):\n%s" % (p["tag"], p["log"], proc.stdout[-4000:]))
    # no lane-divergent control flow in a shipped kernel (isa_audit.py: hipcc misplaced spill code inside reduced-EXEC regions).
    # The audit always runs and is logged; allow_unverified turns a finding into a logged warning (explicitly divergent options).
    from . import isa_audit
    bad = isa_audit.offenders(p["lib"] + ".tmp")
    with open(p["log"], "a") as fh:
        fh.write("### isa audit: %s\n" % (bad or "no kernel writes EXEC"))
    if bad and not gen_kwargs.get("allow_unverified"):
        os.remove(p["lib"] + ".tmp")
        raise GridLibraryError("%s: kernels with lane-divergent control flow (instructions writing EXEC): %s -- such builds are "
                               "unverified (DESIGN.md section 9); pass allow_unverified=True to build them anyway" % (p["tag"], bad))
    os.replace(p["lib"] + ".tmp", p["lib"])
    with open(p["stamp"], "w") as fh:
        fh.write(fp)
    return p["lib"]


API_HARNESS_SRC = os.path.join(REPO_DIR, "tests", "api_surface_harness.hip")


def api_harness_path(robot_name, precision="fp32"):
    return os.path.join(BUILD_DIR, "libapi_harness_%s_%s.so" % (robot_name, precision))


def build_api_harness(robot_name, precision="fp32", force=False):
    """TEST INFRASTRUCTURE: compile tests/api_surface_harness.hip (user-style code over the generated header: _compute_only /
    _launch / USE_COMPRESSED_MEM wrappers, _device and _inner tiers inside user kernels) against the robot's header, linked
    with its kernel library.  Built here so that the GPU box only has to load it.  Returns the .so path."""
    lib = build_library(robot_name, precision)
    p = library_paths(robot_name, precision)
    out = api_harness_path(robot_name, precision)
    h = hashlib.sha256()
    for fn in (p["header"], API_HARNESS_SRC):
        with open(fn, "rb") as fh:
            h.update(fh.read())
    key = h.hexdigest()
    if not force and os.path.exists(out) and os.path.exists(out + ".key"):
        with open(out + ".key") as fh:
            if fh.read().strip() == key:
                return out
    cmd = [_hipcc(), "--offload-arch=" + ARCH, "-O1", "-ffp-contract=off", "-fPIC", "-shared", "-std=c++17"] + KERNARG_PRELOAD + [
           "-DGRID_HEADER=\"%s\"" % p["header"], "-DGRID_NS=grid_" + robot_name, "-DGRID_EXTERN_KERNELS", API_HARNESS_SRC,
           "-L" + BUILD_DIR, "-l" + os.path.basename(lib)[3:-3], "-Wl,-rpath,$ORIGIN", "-o", out + ".tmp"]
    proc = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    if proc.returncode != 0:
        raise GridLibraryError("hipcc failed for the API-surface harness of %s:\n%s" % (robot_name, proc.stdout[-4000:]))
    os.replace(out + ".tmp", out)
    with open(out + ".key", "w") as fh:
        fh.write(key)
    return out


SINGLE_TIMING_SRC = os.path.join(REPO_DIR, "tests", "single_timing_harness.hip")


def single_timing_harness_path(robot_name, precision="fp32"):
    return os.path.join(BUILD_DIR, "single_timing_harness_%s_%s" % (robot_name, precision))


def build_single_timing_harness(robot_name, precision="fp32", force=False):
    """TEST INFRASTRUCTURE: the GRiD-style main() of tests/single_timing_harness.hip (reference mode 1: `Single Call <label>`
    latency twins) compiled against the robot's header; an executable, prebuilt so that the GPU box only runs it."""
    build_library(robot_name, precision)
    p = library_paths(robot_name, precision)
    out = single_timing_harness_path(robot_name, precision)
    h = hashlib.sha256()
    for fn in (p["header"], SINGLE_TIMING_SRC):
        with open(fn, "rb") as fh:
            h.update(fh.read())
    key = h.hexdigest()
    if not force and os.path.exists(out) and os.path.exists(out + ".key"):
        with open(out + ".key") as fh:
            if fh.read().strip() == key:
                return out
    cmd = [_hipcc(), "--offload-arch=" + ARCH, "-O1", "-ffp-contract=off", "-std=c++17"] + KERNARG_PRELOAD + ["-DGRID_HEADER=\"%s\"" % p["header"],
           "-DGRID_NS=grid_" + robot_name, SINGLE_TIMING_SRC, "-o", out + ".tmp"]
    if get_robot(robot_name).get_num_joints() > 12:
        # large robots: the library's kernels are linked in (minutes each to compile a second time); only the latency twins are built
        # here, and the forward-dynamics gradient is compared norm-wise (mode 0 dispatches the tile-cooperative kernel)
        lib = library_paths(robot_name, precision)["lib"]
        cmd = cmd[:-2] + ["-DGRID_EXTERN_KERNELS", "-DGRID_ST_REL_TOL=2e-5", "-L" + BUILD_DIR, "-l" + os.path.basename(lib)[3:-3], "-Wl,-rpath,$ORIGIN"] + cmd[-2:]
    proc = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    if proc.returncode != 0:
        raise GridLibraryError("hipcc failed for the single-timing harness of %s:\n%s" % (robot_name, proc.stdout[-4000:]))
    os.replace(out + ".tmp", out)
    with open(out + ".key", "w") as fh:
        fh.write(key)
    return out


_c_float_p = ctypes.POINTER(ctypes.c_float)
_c_int_p = ctypes.POINTER(ctypes.c_int)
_vp = ctypes.c_void_p

# (name, restype, argtypes) -- must list every symbol include/grid_capi.h declares (tests check this)
CAPI_SIGNATURES = [
    ("grid_robot_name", ctypes.c_char_p, []),
    ("grid_num_joints", ctypes.c_int, []),
    ("grid_constants", ctypes.c_int, [_c_int_p, ctypes.c_int]),
    ("grid_compute_dtype", ctypes.c_char_p, []),
    ("grid_last_error", ctypes.c_char_p, []),
    ("grid_init", ctypes.c_int, [ctypes.c_int, ctypes.POINTER(_vp)]),
    ("grid_alloc", ctypes.c_int, [_vp, ctypes.c_int]),
    ("grid_close", ctypes.c_int, [_vp]),
    ("grid_topology_helpers_count", ctypes.c_int, []),
    ("grid_read_model", ctypes.c_int, [_vp, _c_float_p, _c_int_p]),
    ("grid_inverse_dynamics", ctypes.c_int, [_vp, _c_float_p, _c_float_p, _c_float_p, ctypes.c_int, ctypes.c_float]),
    ("grid_direct_minv", ctypes.c_int, [_vp, _c_float_p, _c_float_p, ctypes.c_int]),
    ("grid_forward_dynamics", ctypes.c_int, [_vp, _c_float_p, _c_float_p, ctypes.c_int, ctypes.c_float]),
    ("grid_inverse_dynamics_gradient", ctypes.c_int, [_vp, _c_float_p, _c_float_p, _c_float_p, ctypes.c_int, ctypes.c_float]),
    ("grid_forward_dynamics_gradient", ctypes.c_int, [_vp, _c_float_p, _c_float_p, _c_float_p, _c_float_p, ctypes.c_int, ctypes.c_float]),
    ("grid_inverse_dynamics_device", ctypes.c_int, [_vp, _vp, _vp, ctypes.c_int, _vp, ctypes.c_int, ctypes.c_float, ctypes.c_int, ctypes.c_int, _vp]),
    ("grid_direct_minv_device", ctypes.c_int, [_vp, _vp, _vp, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, _vp]),
    ("grid_forward_dynamics_device", ctypes.c_int, [_vp, _vp, _vp, ctypes.c_int, ctypes.c_int, ctypes.c_float, ctypes.c_int, ctypes.c_int, _vp]),
    ("grid_inverse_dynamics_gradient_device", ctypes.c_int, [_vp, _vp, _vp, ctypes.c_int, _vp, ctypes.c_int, ctypes.c_float, ctypes.c_int, ctypes.c_int, _vp]),
    ("grid_forward_dynamics_gradient_device", ctypes.c_int, [_vp, _vp, _vp, ctypes.c_int, _vp, _vp, ctypes.c_int, ctypes.c_float, ctypes.c_int, ctypes.c_int, _vp]),
    ("grid_synchronize", ctypes.c_int, [_vp, _vp]),
    ("grid_stream", _vp, [_vp, ctypes.c_int]),
    ("grid_rollout_row_count", ctypes.c_int, []),
    ("grid_forward_dynamics_gradient_rollout_device", ctypes.c_int, [_vp, _vp, _vp, _vp, ctypes.c_int, ctypes.c_int, ctypes.c_float, ctypes.c_float,
                                                                     ctypes.c_int, ctypes.c_int, _vp]),
    ("grid_splits", ctypes.c_int, [ctypes.c_int, _c_int_p, ctypes.c_int]),
    ("grid_set_split", ctypes.c_int, [_vp, ctypes.c_int, ctypes.c_int]),
    ("grid_get_split", ctypes.c_int, [_vp, ctypes.c_int, ctypes.c_int]),
    ("grid_coop_available", ctypes.c_int, [ctypes.c_int]),
    ("grid_set_coop", ctypes.c_int, [_vp, ctypes.c_int, ctypes.c_int]),
    ("grid_get_coop", ctypes.c_int, [_vp, ctypes.c_int, ctypes.c_int]),
    ("grid_kernel_attributes_coop", ctypes.c_int, [ctypes.c_int, _c_int_p]),
    ("grid_lean_available", ctypes.c_int, [ctypes.c_int]),
    ("grid_kernel_attributes_lean", ctypes.c_int, [ctypes.c_int, _c_int_p]),
    ("grid_wave_available", ctypes.c_int, [ctypes.c_int]),
    ("grid_set_wave", ctypes.c_int, [_vp, ctypes.c_int, ctypes.c_int]),
    ("grid_get_wave", ctypes.c_int, [_vp, ctypes.c_int, ctypes.c_int]),
    ("grid_kernel_attributes_wave", ctypes.c_int, [ctypes.c_int, _c_int_p]),
    ("grid_workspace_count", ctypes.c_int, [ctypes.c_int]),
    ("grid_set_pipeline", ctypes.c_int, [_vp, ctypes.c_int, ctypes.c_int]),
    ("grid_time_device", ctypes.c_int, [_vp, ctypes.c_int, _vp, _vp, ctypes.c_int, _vp, _vp, ctypes.c_int, ctypes.c_float,
                                        ctypes.c_int, ctypes.c_int, _vp, ctypes.c_int, _c_float_p]),
    ("grid_kernel_attributes", ctypes.c_int, [ctypes.c_int, ctypes.c_int, _c_int_p]),
    ("grid_kernel_attributes_split", ctypes.c_int, [ctypes.c_int, ctypes.c_int, _c_int_p]),
]


class GridLibrary:
    """ctypes binding of one ``libgrid_<robot>_<precision>.so``."""

    def __init__(self, robot_name, precision="fp32", path=None):
        self.robot_name = robot_name
        self.path = path or library_paths(robot_name, precision)["lib"]
        if not os.path.exists(self.path):
            raise GridLibraryError(
                "HIP library %s is not built; run `python -c 'import __graft_entry__ as g; g.build()'` "
                "(there is deliberately no CPU fallback)" % self.path)
        self.lib = ctypes.CDLL(self.path, mode=ctypes.RTLD_LOCAL)
        for (name, restype, argtypes) in CAPI_SIGNATURES:
            fn = getattr(self.lib, name)  # AttributeError if the library does not export it
            fn.restype = restype
            fn.argtypes = argtypes
        self.n = int(self.lib.grid_num_joints())
        consts = (ctypes.c_int * 10)()
        self.lib.grid_constants(consts, 10)
        keys = ["NUM_JOINTS", "ID_DYNAMIC_SHARED_MEM_COUNT", "MINV_DYNAMIC_SHARED_MEM_COUNT", "FD_DYNAMIC_SHARED_MEM_COUNT",
                "ID_DU_DYNAMIC_SHARED_MEM_COUNT", "FD_DU_DYNAMIC_SHARED_MEM_COUNT", "ID_DU_MAX_SHARED_MEM_COUNT",
                "FD_DU_MAX_SHARED_MEM_COUNT", "SUGGESTED_THREADS", "SUGGESTED_MAX_BLOCKS"]
        self.constants = dict(zip(keys, list(consts)))
        self.compute_dtype = self.lib.grid_compute_dtype().decode()

    def last_error(self):
        return self.lib.grid_last_error().decode(errors="replace")

    def check(self, rc, what):
        if rc != 0:
            raise GridLibraryError("%s failed (code %d): %s" % (what, rc, self.last_error()))

    def splits(self, alg):
        out = (ctypes.c_int * 16)()
        k = self.lib.grid_splits(alg, out, 16)
        return [int(out[i]) for i in range(k)]

    def kernel_attributes(self, alg, variant=0, split=0, coop=False, wave=False):
        """Resources of the kernel for `alg`: variant 1 = the qdd / qdd+Minv input variant; split = S > 1: the S-way
        column-split kernel (what grid_get_split says a call dispatches); coop: the tile-cooperative kernel; wave: the
        wave-per-configuration kernel."""
        out = (ctypes.c_int * 4)()
        if wave:
            self.check(self.lib.grid_kernel_attributes_wave(alg, out), "grid_kernel_attributes_wave")
        elif coop == 2:       # the register-lean 8-wave variant (grid_get_coop's numbering)
            self.check(self.lib.grid_kernel_attributes_lean(alg, out), "grid_kernel_attributes_lean")
        elif coop:
            self.check(self.lib.grid_kernel_attributes_coop(alg, out), "grid_kernel_attributes_coop")
        elif split > 1:
            self.check(self.lib.grid_kernel_attributes_split(alg, split, out), "grid_kernel_attributes_split")
        else:
            self.check(self.lib.grid_kernel_attributes(alg, variant, out), "grid_kernel_attributes")
        return dict(numRegs=out[0], static_lds_bytes=out[1], scratch_bytes_per_lane=out[2], maxThreadsPerBlock=out[3])


def _f32(a, shape=None):
    a = np.ascontiguousarray(a, dtype=np.float32)
    if shape is not None and tuple(a.shape) != tuple(shape):
        raise ValueError("expected shape %s, got %s" % (shape, a.shape))
    return a


def _fp(a):
    return a.ctypes.data_as(_c_float_p)




class GridHandle:
    """One robot model on one GPU: ``init_robotModel`` + ``init_grid`` (+ ``init_gridData``)."""

    def __init__(self, robot_name, device=0, precision="fp32", max_timesteps=0, library=None):
        self.L = library or GridLibrary(robot_name, precision)
        self.n = self.L.n
        self._h = _vp()
        self.L.check(self.L.lib.grid_init(int(device), ctypes.byref(self._h)), "grid_init")
        self.max_timesteps = 0
        # stream used by the device-pointer methods when none is passed: 0 = HIP's default stream (what
        # `torch.cuda.current_stream().cuda_stream` reports for PyTorch's default stream), so a launch is ordered with the torch
        # work that produced its inputs.  The handle's own non-blocking streams (the reference's init_grid streams, NOT ordered
        # with the default stream) are available from own_stream().
        self.default_stream = 0
        if max_timesteps:
            self.alloc(max_timesteps)

    def _stream(self, stream):
        """The `stream` argument of the device-pointer methods -> what the C ABI gets.  None: `default_stream`.  0 (or None
        there): HIP's default stream, the C ABI's NULL.  Anything else: that stream handle."""
        if stream is None:
            stream = self.default_stream
        return None if (stream is None or int(stream) == 0) else int(stream)

    def own_stream(self, index=0):
        """One of the handle's own three non-blocking streams (init_grid<T>()) as an integer handle usable as `stream=`."""
        p = self.L.lib.grid_stream(self._h, int(index))
        if not p:
            raise GridLibraryError("grid_stream: " + self.L.last_error())
        return int(p)

    # lifecycle ------------------------------------------------------------------------------------
    def alloc(self, max_timesteps):
        self.L.check(self.L.lib.grid_alloc(self._h, int(max_timesteps)), "grid_alloc")
        self.max_timesteps = int(max_timesteps)

    def close(self):
        if self._h:
            h, self._h = self._h, _vp()
            self.L.check(self.L.lib.grid_close(h), "grid_close")

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _ensure(self, K):
        if K > self.max_timesteps:
            self.alloc(K)

    def read_model(self):
        XI = np.zeros(72 * self.n, dtype=np.float32)
        nt = int(self.L.lib.grid_topology_helpers_count())
        topo = np.zeros(max(nt, 1), dtype=np.int32)
        self.L.check(self.L.lib.grid_read_model(self._h, _fp(XI), topo.ctypes.data_as(_c_int_p)), "grid_read_model")
        return XI, topo[:nt]

    # host-buffer calls (reference host wrappers, mode 0) -----------------------------------------------
    def inverse_dynamics(self, q_qd_u, qdd=None, gravity=9.81):
        q_qd_u = _f32(q_qd_u); K = q_qd_u.shape[0]; n = self.n
        _f32(q_qd_u, (K, 3 * n)); self._ensure(K)
        qdd_p = _fp(_f32(qdd, (K, n))) if qdd is not None else None
        out = np.empty((K, n), dtype=np.float32)
        self.L.check(self.L.lib.grid_inverse_dynamics(self._h, _fp(q_qd_u), qdd_p, _fp(out), K, gravity), "grid_inverse_dynamics")
        return out

    def direct_minv(self, q_qd_u):
        q_qd_u = _f32(q_qd_u); K = q_qd_u.shape[0]; n = self.n
        _f32(q_qd_u, (K, 3 * n)); self._ensure(K)
        out = np.empty((K, n * n), dtype=np.float32)
        self.L.check(self.L.lib.grid_direct_minv(self._h, _fp(q_qd_u), _fp(out), K), "grid_direct_minv")
        return out

    def forward_dynamics(self, q_qd_u, gravity=9.81):
        q_qd_u = _f32(q_qd_u); K = q_qd_u.shape[0]; n = self.n
        _f32(q_qd_u, (K, 3 * n)); self._ensure(K)
        out = np.empty((K, n), dtype=np.float32)
        self.L.check(self.L.lib.grid_forward_dynamics(self._h, _fp(q_qd_u), _fp(out), K, gravity), "grid_forward_dynamics")
        return out

    def inverse_dynamics_gradient(self, q_qd_u, qdd=None, gravity=9.81):
        q_qd_u = _f32(q_qd_u); K = q_qd_u.shape[0]; n = self.n
        _f32(q_qd_u, (K, 3 * n)); self._ensure(K)
        qdd_a = _f32(qdd, (K, n)) if qdd is not None else None
        out = np.empty((K, 2 * n * n), dtype=np.float32)
        self.L.check(self.L.lib.grid_inverse_dynamics_gradient(self._h, _fp(q_qd_u), _fp(qdd_a) if qdd_a is not None else None,
                                                               _fp(out), K, gravity), "grid_inverse_dynamics_gradient")
        return out

    def forward_dynamics_gradient(self, q_qd_u, qdd=None, Minv=None, gravity=9.81):
        q_qd_u = _f32(q_qd_u); K = q_qd_u.shape[0]; n = self.n
        _f32(q_qd_u, (K, 3 * n)); self._ensure(K)
        qdd_a = _f32(qdd, (K, n)) if qdd is not None else None
        Minv_a = _f32(Minv, (K, n * n)) if Minv is not None else None
        out = np.empty((K, 2 * n * n), dtype=np.float32)
        self.L.check(self.L.lib.grid_forward_dynamics_gradient(
            self._h, _fp(q_qd_u), _fp(qdd_a) if qdd_a is not None else None, _fp(Minv_a) if Minv_a is not None else None,
            _fp(out), K, gravity), "grid_forward_dynamics_gradient")
        return out

    # device-pointer calls (reference mode 2); pointers are ints (e.g. torch.Tensor.data_ptr()) --------------
    def inverse_dynamics_device(self, d_c, d_q_qd, stride, K, d_qdd=None, gravity=9.81, blocks=0, threads=0, stream=None):
        self.L.check(self.L.lib.grid_inverse_dynamics_device(self._h, d_c, d_q_qd, stride, d_qdd, K, gravity, blocks, threads, self._stream(stream)),
                     "grid_inverse_dynamics_device")

    def direct_minv_device(self, d_Minv, d_q, stride, K, blocks=0, threads=0, stream=None):
        self.L.check(self.L.lib.grid_direct_minv_device(self._h, d_Minv, d_q, stride, K, blocks, threads, self._stream(stream)), "grid_direct_minv_device")

    def forward_dynamics_device(self, d_qdd, d_q_qd_u, stride, K, gravity=9.81, blocks=0, threads=0, stream=None):
        self.L.check(self.L.lib.grid_forward_dynamics_device(self._h, d_qdd, d_q_qd_u, stride, K, gravity, blocks, threads, self._stream(stream)),
                     "grid_forward_dynamics_device")

    def inverse_dynamics_gradient_device(self, d_dc_du, d_q_qd, stride, K, d_qdd=None, gravity=9.81, blocks=0, threads=0, stream=None):
        self.L.check(self.L.lib.grid_inverse_dynamics_gradient_device(self._h, d_dc_du, d_q_qd, stride, d_qdd, K, gravity, blocks, threads, self._stream(stream)),
                     "grid_inverse_dynamics_gradient_device")

    def forward_dynamics_gradient_device(self, d_df_du, d_q_qd_u, stride, K, d_qdd=None, d_Minv=None, gravity=9.81,
                                         blocks=0, threads=0, stream=None):
        self.L.check(self.L.lib.grid_forward_dynamics_gradient_device(self._h, d_df_du, d_q_qd_u, stride, d_qdd, d_Minv, K, gravity,
                                                                      blocks, threads, self._stream(stream)), "grid_forward_dynamics_gradient_device")

    def forward_dynamics_gradient_rollout_device(self, d_traj, d_x0, d_u_traj, K, num_steps, dt, gravity=9.81, blocks=0, threads=0, stream=None):
        """Semi-implicit Euler rollout with linearisation (include/grid_capi.h): d_traj is time-major
        [num_steps][K][rollout_row_count] = [x+ | A | B] per step."""
        self.L.check(self.L.lib.grid_forward_dynamics_gradient_rollout_device(self._h, d_traj, d_x0, d_u_traj, K, num_steps, dt, gravity,
                                                                               blocks, threads, self._stream(stream)), "grid_forward_dynamics_gradient_rollout_device")

    def rollout_row_count(self):
        return int(self.L.lib.grid_rollout_row_count())

    def set_split(self, alg, split):
        """0 = automatic (default), 1 = never split, S = force the S-way column-split kernel."""
        self.L.check(self.L.lib.grid_set_split(self._h, alg, int(split)), "grid_set_split")

    def set_coop(self, alg, mode):
        """Tile-cooperative kernel: 0 = automatic (default), 1 = never, 2 = always (4 waves per tile), 3 = always its register-lean
        8-wave variant."""
        self.L.check(self.L.lib.grid_set_coop(self._h, alg, int(mode)), "grid_set_coop")

    def get_coop(self, alg, K):
        """0: no tile-cooperative kernel for a call with K configurations, 1: the 4-wave kernel, 2: the register-lean 8-wave variant."""
        return max(0, int(self.L.lib.grid_get_coop(self._h, alg, int(K))))

    def lean_available(self, alg):
        return int(self.L.lib.grid_lean_available(alg)) == 1

    def coop_available(self, alg):
        return int(self.L.lib.grid_coop_available(alg)) == 1

    def set_wave(self, alg, mode):
        """Wave-per-configuration kernel (small batches): 0 = automatic (default), 1 = never, 2 = always."""
        self.L.check(self.L.lib.grid_set_wave(self._h, alg, int(mode)), "grid_set_wave")

    def get_wave(self, alg, K):
        return int(self.L.lib.grid_get_wave(self._h, alg, int(K))) == 1

    def wave_available(self, alg):
        return int(self.L.lib.grid_wave_available(alg)) == 1

    def set_pipeline(self, alg, mode):
        """0 = automatic (single kernel), 1 = single kernel, 2 = two-pass (workspace) variant."""
        self.L.check(self.L.lib.grid_set_pipeline(self._h, alg, int(mode)), "grid_set_pipeline")

    def get_split(self, alg, K):
        return int(self.L.lib.grid_get_split(self._h, alg, int(K)))

    def synchronize(self, stream=None):
        self.L.check(self.L.lib.grid_synchronize(self._h, self._stream(stream)), "grid_synchronize")

    def time_device(self, alg, d_out, d_in, stride, K, d_qdd=None, d_Minv=None, gravity=9.81, blocks=0, threads=0, stream=None, reps=20):
        ms = ctypes.c_float(0.0)
        self.L.check(self.L.lib.grid_time_device(self._h, alg, d_out, d_in, stride, d_qdd, d_Minv, K, gravity, blocks, threads, self._stream(stream),
                                                 reps, ctypes.byref(ms)), "grid_time_device")
        return float(ms.value)


def output_size(alg, n):
    return {ALG_ID: n, ALG_MINV: n * n, ALG_FD: n, ALG_ID_DU: 2 * n * n, ALG_FD_DU: 2 * n * n}[alg]


def algorithmic_bytes(alg, n):
    """compulsory input + output bytes per evaluation (fp32)."""
    return 4 * {ALG_ID: 2 * n + n, ALG_MINV: n + n * n, ALG_FD: 3 * n + n, ALG_ID_DU: 2 * n + 2 * n * n,
                ALG_FD_DU: 3 * n + 2 * n * n}[alg]
