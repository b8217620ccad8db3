#!/usr/bin/env python3
"""Compare two device assembly files (hipcc -S --cuda-device-only) kernel by kernel: opcode histogram, registers, LDS, scratch.

Used when a kernel template gains a parameter: the existing instantiations must come out of the compiler unchanged (same
opcode mix and resources) -- a check that needs no GPU.   tools/isa_diff.py base.s new.s [--ignore-suffix Lb0E]"""
import collections
import re
import sys


def kernels(path):
    txt = open(path).read()
    out = {}
    for m in re.finditer(r'^(_Z\w+):[^\n]*\n(.*?)\.end_amdhsa_kernel', txt, re.S | re.M):
        name, body = m.group(1), m.group(2)
        ops = collections.Counter()
        for line in body.split('\n'):
            line = line.split(';')[0].strip()
            if not line or line.startswith('.') or line.endswith(':'):
                continue
            ops[line.split()[0]] += 1
        res = {}
        for key in ('next_free_vgpr', 'next_free_sgpr', 'group_segment_fixed_size', 'private_segment_fixed_size', 'accum_offset'):
            mm = re.search(r'\.amdhsa_' + key + r'\s+(\d+)', body)
            if mm:
                res[key] = int(mm.group(1))
        out[name] = (ops, res)
    return out


def canon(name):
    """name of an instantiation in the NEW file with the trailing defaulted bool template argument dropped"""
    return re.sub(r'ELb0(EEEvNS_\d+\w+?E)$', r'\1', name)


def main():
    a, b = kernels(sys.argv[1]), kernels(sys.argv[2])
    bmap = {canon(k): k for k in b}
    bmap.update({k: k for k in b})
    rc = 0
    for k, (ops, res) in a.items():
        kb = bmap.get(k)
        if kb is None:
            print('MISSING', k)
            rc = 1
            continue
        ops2, res2 = b[kb]
        diff = {o: (ops[o], ops2[o]) for o in set(ops) | set(ops2) if ops[o] != ops2[o]}
        same = not diff and res == res2
        print(('same   ' if same else 'DIFFERS'), k[:90], sum(ops.values()), '->', sum(ops2.values()))
        if not same:
            rc = 1
            print('   resources', res, '->', res2)
            for o, (x, y) in sorted(diff.items()):
                print(f'   {o}: {x} -> {y}')
    for k in b:
        if k not in a and canon(k) not in a:
            print('new    ', k[:100], sum(b[k][0].values()), b[k][1])
    return rc


if __name__ == '__main__':
    sys.exit(main())
