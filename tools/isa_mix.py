"""Instruction mix of the kernels in a gfx950 assembly listing (hipcc -S --cuda-device-only): per kernel, and for its hottest
loop (the basic blocks between a label and the last backward branch to it).  usage: isa_mix.py file.s [name filter]"""
import collections
import re
import sys


def klass(op):
    if op.startswith('v_mfma'):
        return 'mfma'
    if op.startswith('ds_'):
        return op
    if op.startswith(('global_', 'buffer_', 'scratch_', 'flat_')):
        return '_'.join(op.split('_')[:2])
    if op.startswith('v_') and 'f64' in op:
        return 'valu64'
    if op.startswith('v_'):
        return 'valu'
    if op.startswith('s_waitcnt'):
        return 'waitcnt'
    if op.startswith('s_barrier'):
        return 'barrier'
    if op.startswith('s_'):
        return 'salu'
    return op


def main():
    txt = open(sys.argv[1]).read().split('\n')
    flt = sys.argv[2] if len(sys.argv) > 2 else ''
    starts = [i for i, l in enumerate(txt) if re.match(r'^_Z\w+:', l)]
    for a, b in zip(starts, starts[1:] + [len(txt)]):
        name = txt[a].split(':')[0]
        if flt not in name:
            continue
        body = txt[a:b]
        labels = {}
        ins = []
        for l in body:
            m = re.match(r'^(\.LBB\w+):', l)
            if m:
                labels[m.group(1)] = len(ins)
            elif l.startswith('\t') and not l.strip().startswith(('.', ';')):
                ins.append(l.strip())
            if l.startswith('\ts_endpgm'):
                pass
        tot = collections.Counter(klass(i.split()[0]) for i in ins)
        print(name[:70], len(ins), dict(tot.most_common(12)))
        # backward branches
        loops = []
        for k, i in enumerate(ins):
            m = re.match(r's_c?branch\w*\s+(\.LBB\w+)', i)
            if m and m.group(1) in labels and labels[m.group(1)] <= k:
                loops.append((k - labels[m.group(1)], labels[m.group(1)], k))
        for n, lo, hi in sorted(loops, reverse=True)[:3]:
            c = collections.Counter(klass(i.split()[0]) for i in ins[lo:hi + 1])
            print('   loop', n, dict(c.most_common(14)))


main()
