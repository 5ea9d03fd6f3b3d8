#!/usr/bin/env python3
"""Per-kernel resources of a built object (.o / .so): VGPRs, AGPRs, SGPRs, scratch bytes, LDS
bytes and code size, read from the gfx950 code object's metadata notes.

    python scripts/kernel_resources.py soft_contrastive_learning_amd/csrc/convh.o [name-filter]
"""
import os
import re
import subprocess
import sys
import tempfile

LLVM = '/opt/rocm/lib/llvm/bin'


def code_objects(path, tmp):
    """Unbundle the gfx950 code object(s) of a host object / shared library."""
    out = os.path.join(tmp, 'dev.co')
    for kind in ('o', 'so'):
        r = subprocess.run([os.path.join(LLVM, 'clang-offload-bundler'), '--unbundle', '--type=' + kind,
                            '--targets=hipv4-amdgcn-amd-amdhsa--gfx950', '--input=' + path,
                            '--output=' + out], capture_output=True, text=True)
        if r.returncode == 0 and os.path.getsize(out) > 0:
            return [out]
    # a .so / .o holds the bundle in section .hip_fatbin
    fat = os.path.join(tmp, 'fat.bin')
    subprocess.run([os.path.join(LLVM, 'llvm-objcopy'), '-O', 'binary', '--only-section=.hip_fatbin',
                    path, fat], check=True)
    data = open(fat, 'rb').read()
    outs = []
    magic = b'__CLANG_OFFLOAD_BUNDLE__'
    pos = 0
    while True:
        pos = data.find(magic, pos)
        if pos < 0:
            break
        n = int.from_bytes(data[pos + 24:pos + 32], 'little')
        p = pos + 32
        for _ in range(n):
            off = int.from_bytes(data[p:p + 8], 'little')
            size = int.from_bytes(data[p + 8:p + 16], 'little')
            tl = int.from_bytes(data[p + 16:p + 24], 'little')
            triple = data[p + 24:p + 24 + tl].decode()
            p += 24 + tl
            if 'gfx950' in triple and size:
                o = os.path.join(tmp, 'co%d.elf' % len(outs))
                open(o, 'wb').write(data[pos + off:pos + off + size])
                outs.append(o)
        pos += 1
    return outs


def kernels(co):
    txt = subprocess.run([os.path.join(LLVM, 'llvm-readelf'), '--notes', co], capture_output=True,
                         text=True).stdout
    rows = []
    for blk in re.split(r'\n\s+- ', txt):
        name = re.search(r'\.name:\s+(\S+)', blk)
        if not name or '.vgpr_count' not in blk:
            continue
        def f(key):
            m = re.search(r'\.%s:\s+(\d+)' % key, blk)
            return int(m.group(1)) if m else 0
        rows.append((name.group(1), f('vgpr_count'), f('agpr_count'), f('sgpr_count'),
                     f('private_segment_fixed_size'), f('group_segment_fixed_size')))
    syms = subprocess.run([os.path.join(LLVM, 'llvm-readelf'), '-sW', co], capture_output=True, text=True).stdout
    size = {}
    for line in syms.splitlines():
        p = line.split()
        if len(p) >= 8 and p[3] == 'FUNC':
            size[p[7]] = int(p[2])
    return [r + (size.get(r[0], 0),) for r in rows]


def demangle(names):
    r = subprocess.run(['c++filt'], input='\n'.join(names), capture_output=True, text=True)
    return r.stdout.splitlines()


def main():
    path = sys.argv[1]
    flt = sys.argv[2] if len(sys.argv) > 2 else ''
    with tempfile.TemporaryDirectory() as tmp:
        rows = []
        for co in code_objects(path, tmp):
            rows += kernels(co)
    names = demangle([r[0] for r in rows])
    print('%6s %5s %5s %8s %7s %9s  %s' % ('vgpr', 'agpr', 'sgpr', 'scratch', 'lds', 'code B', 'kernel'))
    for r, n in sorted(zip(rows, names), key=lambda t: t[1]):
        n = n.replace('void ', '').replace('(anonymous namespace)::', '')
        depth = 0
        for i, ch in enumerate(n):            # cut the parameter list: the first '(' outside <...>
            depth += (ch == '<') - (ch == '>')
            if ch == '(' and depth == 0:
                n = n[:i]
                break
        if flt and flt not in n:
            continue
        print('%6d %5d %5d %8d %7d %9d  %s' % (r[1], r[2], r[3], r[4], r[5], r[6], n))


if __name__ == '__main__':
    main()
