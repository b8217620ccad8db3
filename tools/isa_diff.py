#!/usr/bin/env python3
"""Compare two device assembly files (hipcc -S --cuda-device-only) kernel by kernel: opcode histogram, registers, LDS, scratch
and (r05) the instruction stream itself, operands included (`same` = all of them equal).

Used when a kernel template gains a parameter: the existing instantiations must come out of the compiler unchanged (same
opcode mix and resources) -- a check that needs no GPU.   tools/isa_diff.py base.s new.s [--ignore-suffix Lb0E]"""
import collections
import hashlib
import re
import sys


def kernels(path):
    txt = open(path).read()
    out = {}
    for m in re.finditer(r'^(_Z\w+):[^\n]*\n(.*?)\.end_amdhsa_kernel', txt, re.S | re.M):
        name, body = m.group(1), m.group(2)
        ops = collections.Counter()
        seq = hashlib.sha256()          # the instruction STREAM (operands included; labels / symbol names normalised)
        for line in body.split('\n'):
            line = line.split(';')[0].strip()
            if not line or line.startswith('.') or line.endswith(':'):
                continue
            ops[line.split()[0]] += 1
            seq.update(re.sub(r'_Z\w+', 'SYM', re.sub(r'\.L\w+', 'LBL', line)).encode() + b'\n')
        res = {}
        for key in ('next_free_vgpr', 'next_free_sgpr', 'group_segment_fixed_size', 'private_segment_fixed_size', 'accum_offset'):
            mm = re.search(r'\.amdhsa_' + key + r'\s+(\d+)', body)
            if mm:
                res[key] = int(mm.group(1))
        res['stream'] = seq.hexdigest()[:16]
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
    brief = '--brief' in sys.argv

    def base(name):                 # mangled name up to the template argument list
        m = re.match(r'(_ZN?[\w]*?_k)(?:I|E)', name)
        return m.group(1) if m else name
    renamed = {}                    # old name -> new name: the template gained (defaulted) parameters, the mangled name changed
    for k, (ops, res) in a.items():
        if k in bmap:
            continue
        cands = [kb for kb in b if kb not in a and base(kb) == base(k)]
        exact = [kb for kb in cands if b[kb][0] == ops and b[kb][1] == res]
        pick = exact[0] if exact else None
        if pick is None:            # same leading template arguments, closest instruction count
            lead = re.sub(r'E+v.*$', '', k)
            pref = [kb for kb in cands if kb.startswith(lead)]
            if not pref:            # a parameter changed its type (bool -> int): same first argument
                first = re.match(r'.*?_kI(L\w\d+E)', k)
                pref = [kb for kb in cands if first and re.match(r'.*?_kI' + re.escape(first.group(1)), kb)] or \
                    ([] if first else cands)
            if pref:
                pick = min(pref, key=lambda kb: abs(sum(b[kb][0].values()) - sum(ops.values())))
        if pick is not None:
            renamed[k] = pick
            bmap[k] = pick
    for k, (ops, res) in a.items():
        kb = bmap.get(k)
        if kb is None:
            print('MISSING', k)
            rc = 1
            continue
        ops2, res2 = b[kb]
        diff = {o: (ops[o], ops2[o]) for o in set(ops) | set(ops2) if ops[o] != ops2[o]}
        same = not diff and res == res2
        print(('same   ' if same else 'DIFFERS'), k[:90], sum(ops.values()), '->', sum(ops2.values()),
              ('(now ' + kb[:90] + ')') if k in renamed else '')
        if not same:
            rc = 1
            print('   resources', res, '->', res2)
            mem = {o: v for o, v in diff.items() if o.startswith(('v_mfma', 'ds_', 'global_', 'buffer_', 'scratch_', 's_barrier'))}
            for o, (x, y) in sorted((mem if brief else diff).items()):
                print(f'   {o}: {x} -> {y}')
            if brief:
                print(f'   ({len(diff) - len(mem)} other opcodes differ in count: scalar / vector ALU, waits)')
    for k in b:
        if k not in a and canon(k) not in a and k not in renamed.values():
            print('new    ', k[:100], sum(b[k][0].values()), b[k][1])
    return rc


if __name__ == '__main__':
    sys.exit(main())
