"""Per-kernel resources of a HIP shared library, read from the code objects embedded in it (no GPU, no ROCm tools):
name, VGPRs, spilled VGPRs, private-segment (scratch) bytes, LDS bytes.  `python tools/kernel_resources.py [lib.so]`
prints the table; tests/test_host_cabi.py uses kernels() to assert that no shipped kernel uses scratch, and
risky_packed() (llvm-objdump of the same code objects) to assert that none holds the packed fp32 instruction form that
MI355X miscomputes in lanes 48-63 (tools/micro/pk_opsel.hip)."""
import os
import re
import struct
import subprocess
import sys
import tempfile

import msgpack

MAGIC = b"__CLANG_OFFLOAD_BUNDLE__"


def _code_objects(blob):
    """Every device ELF of every clang offload bundle inside `blob`."""
    pos = 0
    while True:
        pos = blob.find(MAGIC, pos)
        if pos < 0:
            return
        (n,) = struct.unpack_from("<Q", blob, pos + len(MAGIC))
        p = pos + len(MAGIC) + 8
        for _ in range(n):
            off, size, tl = struct.unpack_from("<QQQ", blob, p)
            triple = blob[p + 24:p + 24 + tl].decode()
            p += 24 + tl
            if "amdgcn" in triple and size > 0:
                yield triple, blob[pos + off:pos + off + size]
        pos += len(MAGIC)


def _notes(elf):
    """NT_AMDGPU_METADATA (type 32) note payloads of an ELF64 little-endian image."""
    if elf[:4] != b"\x7fELF":
        return
    shoff, = struct.unpack_from("<Q", elf, 0x28)
    shentsize, shnum = struct.unpack_from("<HH", elf, 0x3A)
    for i in range(shnum):
        sh = shoff + i * shentsize
        sh_type, = struct.unpack_from("<I", elf, sh + 4)
        if sh_type != 7:          # SHT_NOTE
            continue
        off, size = struct.unpack_from("<QQ", elf, sh + 0x18)
        p, end = off, off + size
        while p + 12 <= end:
            namesz, descsz, ntype = struct.unpack_from("<III", elf, p)
            p += 12
            p_desc = p + ((namesz + 3) & ~3)
            if ntype == 32:
                yield elf[p_desc:p_desc + descsz]
            p = p_desc + ((descsz + 3) & ~3)


def kernels(path):
    """[{name, vgprs, spills, scratch, lds, sgprs}] for every kernel of every gfx code object in the library."""
    blob = open(path, "rb").read()
    out = []
    for triple, elf in _code_objects(blob):
        for note in _notes(elf):
            md = msgpack.unpackb(note, raw=False, strict_map_key=False)
            for k in md.get("amdhsa.kernels", []):
                out.append({"name": k[".name"], "arch": triple.split("-")[-1], "vgprs": k.get(".vgpr_count", 0),
                            "spills": k.get(".vgpr_spill_count", 0), "scratch": k.get(".private_segment_fixed_size", 0),
                            "lds": k.get(".group_segment_fixed_size", 0), "sgprs": k.get(".sgpr_count", 0)})
    return out


OBJDUMP = "/opt/rocm/lib/llvm/bin/llvm-objdump"
_PK = re.compile(r"^\s*(v_pk_(?:mul|add|fma|min|max)\w*_f32)\s+(v\[[0-9:]+\]),\s*([^,]+),\s*([^, ]+)(?:,\s*[^, ]+)?\s*(.*?)\s*(?://.*)?$")


def _risky_line(line):
    """True for one disassembly line that holds the miscomputed form (see risky_packed)."""
    m = _PK.match(line)
    if not m:
        return False
    _op, _dst, s0, s1, mods = m.groups()
    sel = re.search(r"op_sel:\[([0-9,]+)\]", mods)
    if sel is None:
        return False
    bits = [int(x) for x in sel.group(1).split(",")]
    return len(bits) > 1 and bits[1] == 1 and s0.strip() != s1.strip()


def risky_packed(path):
    """[(kernel, instruction)] for every packed fp32 arithmetic instruction of the library whose LOW result lane reads
    the HIGH half of its second source (op_sel[1] = 1) from a register pair other than the first source's: the form
    that returns wrong low halves in lanes 48-63 next to MFMA / LDS traffic (v_pk_add_f32 a, a op_sel:[0,1]
    op_sel_hi:[1,0], the horizontal add of one pair, and every op_sel_hi form measured clean)."""
    blob = open(path, "rb").read()
    out = []
    for triple, elf in _code_objects(blob):
        with tempfile.NamedTemporaryFile(suffix=".co") as f:
            f.write(elf)
            f.flush()
            text = subprocess.run([OBJDUMP, "-d", "--no-show-raw-insn", f.name], capture_output=True, text=True, check=True).stdout
        kern = None
        for line in text.splitlines():
            m = re.match(r"^[0-9a-f]+ <(\S+)>:", line)
            if m:
                kern = m.group(1)
                continue
            if _risky_line(line):
                out.append((kern, line.strip()))
    return out


if __name__ == "__main__":
    path = sys.argv[1] if len(sys.argv) > 1 else os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "lanegcn-1_amd", "liblgcn.so")
    ks = kernels(path)
    names = subprocess.run(["c++filt"], input="\n".join(k["name"] for k in ks), capture_output=True, text=True).stdout.splitlines()
    print("%d kernels in %s; with scratch: %d" % (len(ks), path, sum(k["scratch"] > 0 for k in ks)))
    for k, n in sorted(zip(ks, names), key=lambda t: (-t[0]["scratch"], t[1])):
        if k["scratch"] > 0 or "-a" in sys.argv:
            print("%-90s vgprs %3d spills %3d scratch %4d B lds %6d" % (n[:90], k["vgprs"], k["spills"], k["scratch"], k["lds"]))
    rp = risky_packed(path)
    print("packed fp32 instructions with a selected high half of source 1 (op_sel[1] = 1, other register): %d" % len(rp))
    for kern, ins in rp[:20]:
        print("  %s: %s" % (kern[:80], ins))
