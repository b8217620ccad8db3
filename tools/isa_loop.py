"""Static look at the hottest loop of a kernel in a gfx950 assembly listing (hipcc -S --cuda-device-only): vector-ALU opcode
histogram, spill traffic (AGPR copies, v_readlane / v_writelane of spilled scalars, scratch), and how the MFMAs are spread --
the longest run of MFMAs with no other vector instruction between them and the vector-ALU instructions issued while an MFMA
can be in flight.  No GPU needed.   usage: isa_loop.py file.s <kernel name fragment> [steps per loop iteration]"""
import collections
import re
import sys


def kernel_body(txt, frag):
    starts = [i for i, l in enumerate(txt) if re.match(r'^_Z\w+:', l)]
    for a, b in zip(starts, starts[1:] + [len(txt)]):
        if frag in txt[a]:
            return txt[a].split(':')[0], txt[a:b]
    raise SystemExit(f'no kernel matching {frag!r}')


def hottest_loop(body):
    labels, ins = {}, []
    for l in body:
        m = re.match(r'^(\.LBB\w+):', l)
        if m:
            labels[m.group(1)] = len(ins)
        elif l.startswith('\t') and not l.strip().startswith(('.', ';')):
            ins.append(l.strip())
    best = None
    for k, i in enumerate(ins):
        m = re.match(r's_c?branch\w*\s+(\.LBB\w+)', i)
        if m and m.group(1) in labels and labels[m.group(1)] <= k:
            n = sum(1 for x in ins[labels[m.group(1)]:k + 1] if x.startswith('v_mfma'))
            if best is None or n > best[0]:
                best = (n, labels[m.group(1)], k)
    return ins, (ins[best[1]:best[2] + 1] if best else ins)


def report(path, frag, steps=1):
    txt = open(path).read().split('\n')
    name, body = kernel_body(txt, frag)
    meta = {}
    for l in body:
        m = re.match(r';\s*(NumVgprs|NumAgprs|TotalNumVgprs|ScratchSize|Occupancy|LDSByteSize|SGPRBlocks|NumSgprs):\s*(\d+)', l)
        if m:
            meta[m.group(1)] = int(m.group(2))
    ins, loop = hottest_loop(body)
    ops = [i.split()[0] for i in loop]
    valu = [o for o in ops if o.startswith('v_') and not o.startswith('v_mfma')]
    mfma = sum(o.startswith('v_mfma') for o in ops)
    print(f'{name[:90]}')
    print(f'  registers: {meta}')
    print(f'  hottest loop: {len(loop)} instructions = {steps} step(s); per step: {len(valu) / steps:.0f} vector-ALU, {mfma / steps:.0f} MFMA, '
          f'{sum(o.startswith("ds_read") for o in ops) / steps:.0f} LDS reads, {sum(o.startswith("ds_write") for o in ops) / steps:.0f} LDS writes, '
          f'{sum(o.startswith(("global_load", "buffer_load")) for o in ops) / steps:.0f} global loads, '
          f'{sum(o.startswith("s_waitcnt") for o in ops) / steps:.0f} waits, {sum(o.startswith("s_barrier") for o in ops) / steps:.0f} barrier')
    spill = collections.Counter(o for o in ops if o.startswith(('v_accvgpr', 'v_readlane', 'v_writelane', 'scratch_')))
    print(f'  spill traffic per step: ' + (', '.join(f'{v / steps:.0f} {k}' for k, v in spill.items()) or 'none'))
    h = collections.Counter(valu)
    print('  vector-ALU per step: ' + ', '.join(f'{v / steps:.0f} {k}' for k, v in h.most_common(12)))
    # MFMA spread: runs of consecutive MFMAs (ignoring scalar instructions and waits), and VALU between first and last MFMA of a step
    runs, cur = [], 0
    for o in ops:
        if o.startswith('v_mfma'):
            cur += 1
        elif o.startswith(('v_', 'ds_', 'global_', 'buffer_', 'scratch_')):
            if cur:
                runs.append(cur)
            cur = 0
    if cur:
        runs.append(cur)
    # vector work issued "under" MFMAs: between two MFMAs less than 40 instructions apart
    idx = [k for k, o in enumerate(ops) if o.startswith('v_mfma')]
    covered = 0
    for a, b in zip(idx, idx[1:]):
        if b - a < 40:
            covered += sum(1 for o in ops[a + 1:b] if o.startswith('v_') and not o.startswith('v_mfma'))
    # In-order issue model of ONE wave alone on its SIMD (the situation of a ~500-register kernel), no memory stalls: a vector /
    # LDS / memory instruction occupies the issue port for 4 cycles, a scalar one for 1, an MFMA for 4 and the matrix pipe for
    # `mfma_cycles` (16: v_mfma_f32_16x16x32_bf16 / 16x16x4_f32; 32: the 32x32 forms) -- the next MFMA waits for the pipe, a
    # vector instruction behind an MFMA does not.  A LOWER bound of a step (waits for LDS / HBM come on top), good for comparing
    # two schedules of the same work, not a prediction of time.
    t = pipe = 0
    for o in ops:
        if o.startswith('v_mfma'):
            start = max(t, pipe)
            t, pipe = start + 4, start + (32 if '32x32' in o else 16)
        elif o.startswith(('v_', 'ds_', 'global_', 'buffer_', 'scratch_')):
            t += 4
        else:
            t += 1
    t = max(t, pipe)
    busy = sum(32 if '32x32' in o else 16 for o in ops if o.startswith('v_mfma'))
    print(f'  issue model (one wave per SIMD, no memory stalls): {t / steps:.0f} cycles per step, matrix pipe busy {busy / steps:.0f} '
          f'({100.0 * busy / max(1, t):.0f} %)')
    print(f'  MFMA runs without a vector / memory instruction between them: {len(runs)} runs, longest {max(runs) if runs else 0}, '
          f'mean {sum(runs) / max(1, len(runs)):.1f}; vector-ALU instructions issued between MFMAs (< 40 apart): '
          f'{covered / steps:.0f} of {len(valu) / steps:.0f} per step')


if __name__ == '__main__':
    report(sys.argv[1], sys.argv[2], int(sys.argv[3]) if len(sys.argv) > 3 else 1)
