"""Disassemble the gfx950 code object of one kernel out of a built library (or object file).
usage: python tools/disasm_kernel.py <lib.so | kernelN.o> <substring of the mangled kernel name> <out.s>
The .hip_fatbin section of a multi-TU library is a concatenation of clang offload bundles; each is split here by hand."""
import os, struct, subprocess, sys, tempfile
LLVM = "/opt/rocm/lib/llvm/bin"
src, want, out = sys.argv[1], sys.argv[2], sys.argv[3]
tmp = tempfile.mkdtemp()
fat = os.path.join(tmp, "fat.bin")
subprocess.check_call([LLVM + "/llvm-objcopy", "--dump-section", ".hip_fatbin=" + fat, src])
d = open(fat, "rb").read()
magic = b"__CLANG_OFFLOAD_BUNDLE__"
p = d.find(magic)
while p >= 0:
    cnt = struct.unpack_from("<Q", d, p + 24)[0]
    off = p + 32
    for _ in range(cnt):
        o, s, tl = struct.unpack_from("<QQQ", d, off); off += 24
        t = d[off:off + tl].decode(); off += tl
        if "gfx950" in t and s > 0:
            elf = os.path.join(tmp, "co.elf")
            open(elf, "wb").write(d[p + o:p + o + s])
            syms = subprocess.run([LLVM + "/llvm-readelf", "-s", elf], stdout=subprocess.PIPE, text=True).stdout
            names = [l.split()[-1] for l in syms.splitlines() if " FUNC " in l]
            hit = [n for n in names if want in n]
            if hit:
                with open(out, "w") as fh:
                    subprocess.check_call([LLVM + "/llvm-objdump", "-d", elf], stdout=fh)
                print("\n".join(hit)); sys.exit(0)
    p = d.find(magic, p + 1)
sys.exit("no kernel matching %r" % want)
