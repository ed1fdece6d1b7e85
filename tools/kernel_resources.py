"""Register / LDS budget of every kernel in the built library, read from the code objects' own metadata.

    python tools/kernel_resources.py [lib.so] [name-substring]

The shared library carries one clang offload bundle per translation unit; each holds a gfx950 ELF whose
NT_AMDGPU_METADATA note (msgpack) lists .vgpr_count / .sgpr_count / .group_segment_fixed_size per kernel.
tests/test_abi_cpu.py uses this to pin the budgets that co-residency depends on (NOTES.md, section 3).
"""
import re
import struct
import sys

import msgpack

MAGIC = b'__CLANG_OFFLOAD_BUNDLE__'


def code_objects(blob):
    for m in re.finditer(MAGIC, blob):
        base = m.start()
        p = base + len(MAGIC)
        (n,) = struct.unpack_from('<Q', blob, p)
        p += 8
        for _ in range(n):
            off, size, tlen = struct.unpack_from('<QQQ', blob, p)
            p += 24
            triple = blob[p:p + tlen].decode()
            p += tlen
            if 'gfx950' in triple and size:
                yield blob[base + off: base + off + size]


def notes(elf):
    assert elf[:4] == b'\x7fELF'
    shoff, = struct.unpack_from('<Q', elf, 0x28)
    shentsize, shnum = struct.unpack_from('<HH', elf, 0x3A)
    for i in range(shnum):
        sh = shoff + i * shentsize
        sh_type, = struct.unpack_from('<I', elf, sh + 4)
        off, size = struct.unpack_from('<QQ', elf, sh + 0x18)
        if sh_type != 7:      # SHT_NOTE
            continue
        p = off
        while p < off + size:
            namesz, descsz, ntype = struct.unpack_from('<III', elf, p)
            p += 12
            name = elf[p:p + namesz]
            p += (namesz + 3) & ~3
            desc = elf[p:p + descsz]
            p += (descsz + 3) & ~3
            if ntype == 32 and name.startswith(b'AMDGPU'):      # NT_AMDGPU_METADATA
                yield msgpack.unpackb(desc, raw=False, strict_map_key=False)


def kernel_resources(lib_path):
    """{mangled kernel name: {'vgpr', 'agpr', 'sgpr', 'lds', 'scratch', 'max_flat_workgroup_size'}}"""
    blob = open(lib_path, 'rb').read()
    out = {}
    for co in code_objects(blob):
        for md in notes(co):
            for k in md.get('amdhsa.kernels', []):
                out[k['.name']] = {'vgpr': k.get('.vgpr_count', 0), 'agpr': k.get('.agpr_count', 0),
                                   'sgpr': k.get('.sgpr_count', 0), 'lds': k.get('.group_segment_fixed_size', 0),
                                   'scratch': k.get('.private_segment_fixed_size', 0),
                                   'max_flat_workgroup_size': k.get('.max_flat_workgroup_size', 0)}
    return out


if __name__ == '__main__':
    import os
    lib = sys.argv[1] if len(sys.argv) > 1 else os.path.join(os.path.dirname(__file__), '..', 'aspire_amd', 'lib', 'libaspire_hip.so')
    sub = sys.argv[2] if len(sys.argv) > 2 else ''
    for name, r in sorted(kernel_resources(lib).items()):
        if sub in name:
            print(f"{r['vgpr']:4d} vgpr {r['agpr']:3d} agpr {r['sgpr']:4d} sgpr {r['lds']:6d} lds {r['scratch']:5d} scratch  {name}")
