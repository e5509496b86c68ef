"""Reads the kernel records (AMDGPU metadata notes) of every gfx950 code object bundled into a HIP object file or shared library:
name, registers, LDS, private (scratch) bytes per lane, spilled registers.  Test infrastructure (tests/test_codeobj_cpu.py) and a
command-line tool:  python tests/codeobj_notes.py dsp_amd/libdsp_amd.so [regex]"""
import re
import struct
import subprocess
import sys

import msgpack

_MAGIC = b"__CLANG_OFFLOAD_BUNDLE__"


def _elf_notes(elf):
    shoff = struct.unpack_from("<Q", elf, 0x28)[0]
    shentsize, shnum, _ = struct.unpack_from("<HHH", elf, 0x3A)
    for s in range(shnum):
        _, typ, _, _, offset, size = struct.unpack_from("<IIQQQQ", elf, shoff + s * shentsize)
        if typ != 7:                      # SHT_NOTE
            continue
        p = offset
        while p < offset + size:
            namesz, descsz, ntype = struct.unpack_from("<III", elf, p)
            p += 12
            name = elf[p:p + namesz]
            p += (namesz + 3) & ~3
            desc = elf[p:p + descsz]
            p += (descsz + 3) & ~3
            yield name, ntype, desc


def kernels(path, arch="gfx950"):
    """[{name (demangled), mangled, scratch, vgpr, agpr, sgpr, spill_vgpr, spill_sgpr, lds, max_threads}, ...] of the file's code objects for `arch`"""
    data = open(path, "rb").read()
    out = []
    for m in re.finditer(_MAGIC, data):
        b = m.start()
        n = struct.unpack_from("<Q", data, b + 24)[0]
        o = b + 32
        for _ in range(n):
            off, size, tl = struct.unpack_from("<QQQ", data, o)
            o += 24
            triple = data[o:o + tl].decode()
            o += tl
            if arch not in triple or size == 0:
                continue
            elf = data[b + off:b + off + size]
            if elf[:4] != b"\x7fELF":
                continue
            for name, ntype, desc in _elf_notes(elf):
                if ntype != 32 or not name.startswith(b"AMDGPU"):
                    continue
                md = msgpack.unpackb(desc, raw=False, strict_map_key=False)
                for k in md.get("amdhsa.kernels", []):
                    out.append({"mangled": k[".name"], "scratch": k.get(".private_segment_fixed_size", 0), "vgpr": k.get(".vgpr_count", 0),
                                "agpr": k.get(".agpr_count", 0), "sgpr": k.get(".sgpr_count", 0), "spill_vgpr": k.get(".vgpr_spill_count", 0),
                                "spill_sgpr": k.get(".sgpr_spill_count", 0), "lds": k.get(".group_segment_fixed_size", 0),
                                "max_threads": k.get(".max_flat_workgroup_size", 0), "dynamic_stack": bool(k.get(".uses_dynamic_stack", False))})
    if out:
        r = subprocess.run(["c++filt"], input="\n".join(k["mangled"] for k in out), capture_output=True, text=True)
        names = r.stdout.split("\n") if r.returncode == 0 else [k["mangled"] for k in out]
        for k, nm in zip(out, names):
            k["name"] = nm
    return out


def short_name(name):
    """`void dspamd::p64::conv_row_duo<12, 2, false>(dspamd::ConvParams, int, int)` -> `conv_row_duo<12, 2, false>`"""
    s = name
    if s.startswith("void "):
        s = s[5:]
    depth = 0
    for i, c in enumerate(s):
        if c == "<":
            depth += 1
        elif c == ">":
            depth -= 1
        elif c == "(" and depth == 0:
            s = s[:i]
            break
    return s.split("::")[-1] if "<" not in s else re.sub(r"^(?:\w+::)+", "", s)


if __name__ == "__main__":
    pat = re.compile(sys.argv[2]) if len(sys.argv) > 2 else None
    ks = kernels(sys.argv[1])
    for k in sorted(ks, key=lambda k: (-k["scratch"], k["name"])):
        if pat and not pat.search(k["name"]):
            continue
        print(f'{k["scratch"]:5d} B scratch  {k["spill_vgpr"]:3d} spilled  v{k["vgpr"]:3d} a{k["agpr"]:3d} s{k["sgpr"]:3d}  lds {k["lds"]:6d}  {short_name(k["name"])}')
    print(f"{len(ks)} kernels, {sum(1 for k in ks if k['scratch'])} with scratch")
