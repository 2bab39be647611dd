#!/usr/bin/env python3
"""kernel_sha.py — the identity of a compiled kernel: sha256 over its machine code and its kernel descriptor.

    python3 kernel_sha.py build/rolling_simple.o rolling_simple_kernel [out.json]     (every instantiation of the kernel)

What ties a committed counter file (profiles/r*_pmc_hbm_traffic_bench_1e9.csv) to the code that runs: round 4 hashed three SOURCE
files, and an edit to common.h that touched nothing the benched kernel is compiled from (Interpolate structs) still made the
evidence "stale" - the driver's bench line lost its counter traffic.  The bytes of the kernel itself change exactly when the
kernel does.  Pure Python (ELF64 + the clang offload bundle of the .hip_fatbin section), no tool of the ROCm image needed: the
Makefile runs it after every build and writes ../libbowgpu.kernel_sha.json next to the library (it travels with it);
bench.py and tests/test_profiles_fresh.py read that file."""
import hashlib
import json
import re
import struct
import sys

BUNDLE_MAGIC = b"__CLANG_OFFLOAD_BUNDLE__"


def _sections(elf):
    assert elf[:4] == b"\x7fELF" and elf[4] == 2 and elf[5] == 1, "not a little-endian ELF64"
    shoff, = struct.unpack_from("<Q", elf, 0x28)
    shentsize, shnum, shstrndx = struct.unpack_from("<HHH", elf, 0x3A)
    if shnum == 0:   # more than 0xff00 sections (a host object full of kernel stubs): the real count sits in section 0's sh_size
        shnum, = struct.unpack_from("<Q", elf, shoff + 0x20)
    if shstrndx == 0xFFFF:
        shstrndx, = struct.unpack_from("<I", elf, shoff + 0x28)
    raw = []
    for i in range(shnum):
        name, typ, flags, addr, off, size, link, info, align, entsize = struct.unpack_from("<IIQQQQIIQQ", elf, shoff + i * shentsize)
        raw.append(dict(name_off=name, type=typ, addr=addr, off=off, size=size, link=link, entsize=entsize))
    strtab = raw[shstrndx]
    for s in raw:
        a = strtab["off"] + s["name_off"]
        s["name"] = elf[a:elf.index(b"\0", a)].decode()
    return raw


def _device_code_object(obj, arch):
    """the gfx950 code object inside a host object's .hip_fatbin (or the file itself when it already is one)"""
    secs = _sections(obj)
    fat = [s for s in secs if s["name"] == ".hip_fatbin"]
    if not fat:
        return obj
    blob = obj[fat[0]["off"]:fat[0]["off"] + fat[0]["size"]]
    assert blob[:len(BUNDLE_MAGIC)] == BUNDLE_MAGIC, "unknown fat binary format"
    n, = struct.unpack_from("<Q", blob, len(BUNDLE_MAGIC))
    p = len(BUNDLE_MAGIC) + 8
    for _ in range(n):
        off, size, tlen = struct.unpack_from("<QQQ", blob, p)
        triple = blob[p + 24:p + 24 + tlen].decode()
        p += 24 + tlen
        if triple.endswith(arch) and "amdgcn" in triple:
            return blob[off:off + size]
    raise SystemExit("no %s bundle in the fat binary" % arch)


def _pretty(symbol):
    """_ZN6bowgpu21rolling_simple_kernelILi0ELb0E...EEv... -> rolling_simple_kernel<0, false, ...> (integer and bool template arguments:
    all these kernels have), spelled as rocprofv3's Kernel_Name spells it"""
    m = re.match(r"_ZN6bowgpu(\d+)", symbol)
    if not m:
        return None
    n = int(m.group(1))
    name, rest = symbol[m.end():m.end() + n], symbol[m.end() + n:]
    if not rest.startswith("I"):
        return name
    args = []
    p = 1
    while rest[p] == "L":
        a = re.match(r"L([ib])(n?\d+)E", rest[p:])
        if not a:
            return None
        args.append(("true" if a.group(2) != "0" else "false") if a.group(1) == "b" else a.group(2).replace("n", "-"))
        p += a.end()
    return "%s<%s>" % (name, ", ".join(args))


def kernel_identities(path, kernel, arch="gfx950"):
    """{instantiation as rocprofv3 prints it: {sha, code_bytes, symbol}} for every instantiation of `kernel` in the object"""
    with open(path, "rb") as fh:
        co = _device_code_object(fh.read(), arch)
    secs = _sections(co)
    symtab = [s for s in secs if s["name"] == ".symtab"][0]
    strs = secs[symtab["link"]]
    code, kd = {}, {}
    for i in range(symtab["size"] // 24):
        name, info, other, shndx, value, size = struct.unpack_from("<IBBHQQ", co, symtab["off"] + 24 * i)
        if shndx in (0, 0xFFF1) or size == 0:
            continue
        a = strs["off"] + name
        nm = co[a:co.index(b"\0", a)].decode()
        if kernel not in nm:
            continue
        sec = secs[shndx]
        blob = co[sec["off"] + value - sec["addr"]: sec["off"] + value - sec["addr"] + size]
        if nm.endswith(".kd"):
            kd[nm[:-3]] = blob
        else:
            code[nm] = blob
    out = {}
    for sym, text in code.items():
        pretty = _pretty(sym)
        if pretty is None or sym not in kd or not pretty.startswith(kernel):
            continue
        d = bytearray(kd[sym])
        d[16:24] = bytes(8)   # kernel_code_entry_byte_offset: where the code lies relative to the descriptor - moves when ANOTHER kernel of the object grows
        h = hashlib.sha256()
        h.update(text)
        h.update(bytes(d))
        out[pretty] = {"sha": h.hexdigest()[:16], "code_bytes": len(text), "symbol": sym}
    if not out:
        raise SystemExit("no instantiation of %s in %s" % (kernel, path))
    return out


if __name__ == "__main__":
    ids = kernel_identities(sys.argv[1], sys.argv[2])
    doc = {"what": "per instantiation: sha256 (first 16 hex digits) over the kernel's machine code and its 64-byte kernel descriptor (entry "
                   "offset zeroed), taken from the object the library is linked from - bow_amd/csrc/kernel_sha.py",
           "arch": "gfx950", "kernels": dict(sorted(ids.items()))}
    text = json.dumps(doc, indent=1) + "\n"
    if len(sys.argv) > 3:
        with open(sys.argv[3], "w") as fh:
            fh.write(text)
    else:
        sys.stdout.write(text)
