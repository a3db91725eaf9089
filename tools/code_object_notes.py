"""Per-kernel resource notes of a built HIP shared library, read straight from its embedded gfx950 code objects (no ROCm tool needed):
the clang offload bundles inside .hip_fatbin -> each amdgcn ELF -> NT_AMDGPU_METADATA (msgpack) -> amdhsa.kernels[*].
Used by tests/test_code_objects_cpu.py (zero scratch on every routed kernel).
  python tools/code_object_notes.py [path/to/lib.so]      prints name, vgpr, sgpr, spill, scratch, lds per kernel"""
import struct
import sys

MAGIC = b'__CLANG_OFFLOAD_BUNDLE__'


def _bundles(blob):
    pos = 0
    while True:
        i = blob.find(MAGIC, pos)
        if i < 0:
            return
        n, = struct.unpack_from('<Q', blob, i + len(MAGIC))
        p = i + len(MAGIC) + 8
        for _ in range(n):
            off, size, tl = struct.unpack_from('<QQQ', blob, p)
            triple = blob[p + 24:p + 24 + tl].decode()
            p += 24 + tl
            if 'amdgcn' in triple and size:
                yield triple, blob[i + off:i + off + size]
        pos = i + len(MAGIC)


def _notes(elf):
    assert elf[:4] == b'\x7fELF' and elf[4] == 2
    shoff, = struct.unpack_from('<Q', elf, 0x28)
    shentsize, shnum = struct.unpack_from('<HH', elf, 0x3A)
    for k in range(shnum):
        sh = elf[shoff + k * shentsize: shoff + (k + 1) * shentsize]
        sh_type, = struct.unpack_from('<I', sh, 4)
        off, size = struct.unpack_from('<QQ', sh, 0x18)
        if sh_type != 7:                                   # SHT_NOTE
            continue
        p = off
        while p < off + size:
            namesz, descsz, ntype = struct.unpack_from('<III', elf, p)
            p += 12
            name = elf[p:p + namesz].rstrip(b'\0')
            p += (namesz + 3) & ~3
            desc = elf[p:p + descsz]
            p += (descsz + 3) & ~3
            if name == b'AMDGPU' and ntype == 32:          # NT_AMDGPU_METADATA
                yield desc


def kernels(path):
    """[{name, vgpr_count, sgpr_count, vgpr_spill_count, sgpr_spill_count, private_segment_fixed_size, group_segment_fixed_size, arch}]"""
    import msgpack
    blob = open(path, 'rb').read()
    out = []
    for triple, elf in _bundles(blob):
        for desc in _notes(elf):
            md = msgpack.unpackb(desc, raw=False, strict_map_key=False)
            for k in md.get('amdhsa.kernels', []):
                out.append(dict(name=k['.name'], arch=triple.split('--')[-1], vgpr_count=k.get('.vgpr_count'), sgpr_count=k.get('.sgpr_count'),
                                vgpr_spill_count=k.get('.vgpr_spill_count', 0), sgpr_spill_count=k.get('.sgpr_spill_count', 0),
                                private_segment_fixed_size=k.get('.private_segment_fixed_size', 0),
                                group_segment_fixed_size=k.get('.group_segment_fixed_size', 0)))
    return out


if __name__ == '__main__':
    import os
    path = sys.argv[1] if len(sys.argv) > 1 else os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'pointdreamer_amd', 'libpdhip.so')
    ks = kernels(path)
    for k in sorted(ks, key=lambda k: (-k['private_segment_fixed_size'], k['name'])):
        print(f"{k['name'][:110]:110s} vgpr {k['vgpr_count']:3d} sgpr {k['sgpr_count']:3d} spill {k['vgpr_spill_count']:3d} scratch {k['private_segment_fixed_size']:5d} lds {k['group_segment_fixed_size']:6d} {k['arch']}")
    print(len(ks), 'kernels;', sum(1 for k in ks if k['private_segment_fixed_size'] or k['vgpr_spill_count']), 'with scratch')
