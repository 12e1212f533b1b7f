"""Kernel metadata of the built liblbmpm_hip.so, read without any ROCm tool: the gfx950 code objects inside the library's
.hip_fatbin section (clang offload bundles) and their AMDGPU metadata notes (msgpack).  Used by tests and by tools/ to state
registers, LDS, spills and scratch of a kernel as the compiler decided them."""
import struct

BUNDLE_MAGIC = b"__CLANG_OFFLOAD_BUNDLE__"


def _elf_sections(b):
    assert b[:4] == b"\x7fELF" and b[4] == 2, "not a 64-bit ELF"
    shoff, = struct.unpack_from("<Q", b, 0x28)
    shentsize, shnum, shstrndx = struct.unpack_from("<HHH", b, 0x3A)
    secs = []
    for i in range(shnum):
        name, typ, _flags, _addr, off, size = struct.unpack_from("<IIQQQQ", b, shoff + i * shentsize)
        secs.append((name, typ, off, size))
    stroff = secs[shstrndx][2]
    out = {}
    for name, typ, off, size in secs:
        end = b.index(b"\0", stroff + name)
        out.setdefault(b[stroff + name:end].decode(), []).append((typ, off, size))
    return out


def _code_objects(lib_bytes):
    """device ELFs of every bundle in .hip_fatbin"""
    objs = []
    for _typ, off, size in _elf_sections(lib_bytes).get(".hip_fatbin", []):
        blob = lib_bytes[off:off + size]
        pos = blob.find(BUNDLE_MAGIC)
        while pos >= 0:
            n, = struct.unpack_from("<Q", blob, pos + len(BUNDLE_MAGIC))
            q = pos + len(BUNDLE_MAGIC) + 8
            for _ in range(n):
                eoff, esize, tsize = struct.unpack_from("<QQQ", blob, q)
                triple = blob[q + 24:q + 24 + tsize].decode()
                q += 24 + tsize
                if "amdgcn" in triple and esize:
                    objs.append(blob[pos + eoff:pos + eoff + esize])
            pos = blob.find(BUNDLE_MAGIC, pos + 1)
    return objs


def _notes(elf):
    for typ, off, size in [s for v in _elf_sections(elf).values() for s in v]:
        if typ != 7:                              # SHT_NOTE
            continue
        p, end = off, off + size
        while p + 12 <= end:
            namesz, descsz, ntype = struct.unpack_from("<III", elf, p)
            p += 12
            name = elf[p:p + namesz].rstrip(b"\0")
            p += (namesz + 3) & ~3
            desc = elf[p:p + descsz]
            p += (descsz + 3) & ~3
            if name == b"AMDGPU" and ntype == 32:  # NT_AMDGPU_METADATA
                yield desc


def kernels(lib_path):
    """{kernel symbol: metadata dict} -- '.vgpr_count', '.sgpr_count', '.group_segment_fixed_size' (LDS bytes),
    '.private_segment_fixed_size' (scratch bytes per lane), '.vgpr_spill_count', '.sgpr_spill_count', ..."""
    import msgpack
    with open(lib_path, "rb") as f:
        lib = f.read()
    out = {}
    for elf in _code_objects(lib):
        for desc in _notes(elf):
            md = msgpack.unpackb(desc, raw=False, strict_map_key=False)
            for k in md.get("amdhsa.kernels", []):
                out[k[".name"]] = k
    return out
