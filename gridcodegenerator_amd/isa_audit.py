"""Static audit of built gfx950 code objects: which kernels contain instructions that WRITE the EXEC mask
(s_*_saveexec_*, s_or/and/andn2/... exec, v_cmpx_*), i.e. lane-divergent control flow.

Generated kernels are meant to contain none.  hipcc (ROCm 7.2) was seen to place register-spill code --
scratch stores, VGPR->AGPR copies -- at the head of the join block of a divergent branch, BEFORE the `s_or_b64 exec` that
restores the mask: for the lanes the branch had masked off the spill then never happened, and everything read back from those
slots later was garbage (a lane id in one kernel: wild stores, memory faults).  With no EXEC write in a kernel there is no
reduced-mask region to misplace anything into.  The build refuses a library that fails this audit."""
import os
import re
import struct
import subprocess
import tempfile

LLVM_BIN = "/opt/rocm/lib/llvm/bin"
EXEC_WRITE = re.compile(r"^\s*(s_\w+_saveexec_b64|s_(or|and|andn2|xor|mov|not|orn2|nand|nor|xnor|cselect|wqm)_b64\s+exec\b"
                        r"|s_(mov|or|and|andn2)_b32\s+exec_(lo|hi)\b|v_cmpx_)")


def code_objects(path, tmp, arch="gfx950"):
    """Yield ELF files written into the directory `tmp` (the caller owns and removes it), one per device code object of `arch`
    inside a host object / shared library (its .hip_fatbin section is a concatenation of clang offload bundles, one per
    translation unit)."""
    fat = os.path.join(tmp, "fat.bin")
    subprocess.check_call([os.path.join(LLVM_BIN, "llvm-objcopy"), "--dump-section", ".hip_fatbin=" + fat, path])
    with open(fat, "rb") as fh:
        d = fh.read()
    magic = b"__CLANG_OFFLOAD_BUNDLE__"
    p = d.find(magic)
    k = 0
    while p >= 0:
        cnt = struct.unpack_from("<Q", d, p + 24)[0]
        off = p + 32
        for _ in range(cnt):
            o, s, tl = struct.unpack_from("<QQQ", d, off)
            off += 24
            triple = d[off:off + tl].decode()
            off += tl
            if arch in triple and s > 0:
                elf = os.path.join(tmp, "co%d.elf" % k)
                k += 1
                with open(elf, "wb") as fh:
                    fh.write(d[p + o:p + o + s])
                yield elf
        p = d.find(magic, p + 1)


def short_name(mangled):
    m = re.match(r"_ZN(\d+)", mangled)                     # _ZN<len><namespace><len><kernel>I...
    if not m:
        return mangled
    rest = mangled[m.end() + int(m.group(1)):]
    m2 = re.match(r"(\d+)", rest)
    return mangled if not m2 else rest[m2.end():m2.end() + int(m2.group(1))]


def audit(path, arch="gfx950"):
    """-> {mangled kernel name: (instructions, instructions that write EXEC)} for every function in the code objects of `path`."""
    out = {}
    with tempfile.TemporaryDirectory(prefix="grid_isa_") as tmp:      # (removed on every exit path: the audit runs at every build)
        listings = [subprocess.run([os.path.join(LLVM_BIN, "llvm-objdump"), "-d", elf], stdout=subprocess.PIPE, text=True, check=True).stdout
                    for elf in code_objects(path, tmp, arch)]
    for txt in listings:
        cur = None
        for line in txt.splitlines():
            m = re.match(r"^[0-9a-f]+ <(\S+)>:", line)
            if m:
                cur = m.group(1)
                out[cur] = [0, 0]
                continue
            if cur is None or not line.startswith("\t"):
                continue
            ins = line.split("//")[0]
            out[cur][0] += 1
            if EXEC_WRITE.match(ins):
                out[cur][1] += 1
    return {k: tuple(v) for k, v in out.items()}


def offenders(path, arch="gfx950"):
    return {short_name(k) + " [" + k[-24:] + "]": v for k, v in audit(path, arch).items() if v[1] > 0}
