"""TEST INFRASTRUCTURE: random geometries for the opt-in kernel variants against their default kernels (the parity tests of
tests/test_gpu_zy_variants.py at fixed shapes; here clip counts, clip lengths, block counts, dropout, lockstep passes, channel
counts and job mixes are drawn at random).  Runs on an MI355X or, with S2AG_EMU=1, on the CPU device model.

    S2AG_EMU=1 python tools/fuzz_variants.py [seed] [n]      ->  profiles/r06_variants_fuzz.txt"""
import os
import random
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))
if os.environ.get('S2AG_EMU', '0') == '1':
    sys.path.insert(0, os.path.join(ROOT, 'tests', 'emu'))
    import harness
    harness.install()
import torch  # noqa: E402

import test_gpu_zy_variants as V  # noqa: E402


def main():
    seed = int(sys.argv[1]) if len(sys.argv) > 1 else 1
    n = int(sys.argv[2]) if len(sys.argv) > 2 else 24
    from speech2affective_gestures_amd import _lib as L
    from speech2affective_gestures_amd import config, ops
    lib = L.load()
    S = dict(L=L, lib=lib, config=config, ops=ops)
    rnd = random.Random(seed)
    bad = 0
    for it in range(n):
        # ---- TCN32_PAIR
        nP = rnd.choice([1, 1, 3])
        B = rnd.choice([2, 4, 6]) if nP > 1 else rnd.randint(1, 7)
        T, nb, drop = rnd.randint(3, 40), rnd.randint(1, 4), rnd.choice([0.0, 0.3])
        Cch = rnd.choice([260, 300, 320])
        g = torch.Generator().manual_seed(seed * 1000 + it)
        x = torch.randn(nP * B, T, Cch, generator=g).cuda()
        ws = [(torch.randn(Cch, 2, Cch, generator=g) * 0.04).cuda() for _ in range(2 * nb)]
        bs = [(torch.randn(Cch, generator=g) * 0.1).cuda() for _ in range(2 * nb)]
        gy = (torch.randn(B * T, Cch, generator=g) * 0.1).cuda()
        dils = [2 ** b for b in range(nb)]
        base = V._tcn32_run(S, x, ws, bs, dils, drop, nP, B, gy)
        form = rnd.choice([1, 2])                                     # TCN32_PAIR=1: two clips per workgroup, =2: one
        with config.override('TCN32_PAIR', form):
            var = V._tcn32_run(S, x, ws, bs, dils, drop, nP, B, gy)
        ok = all(torch.equal(base[k], var[k]) for k in base)
        bad += not ok
        print(f'tcn32 planes TCN32_PAIR={form} B={B} passes={nP} T={T} C={Cch} blocks={nb} drop={drop}: {"bit-identical" if ok else "DIFFERS"}', flush=True)
        # ---- WGRAD32_PIPE
        pieces, ring = rnd.choice([1, 2]), rnd.choice([1, 2])
        if rnd.random() < 0.5:
            Bg, In = rnd.randint(1, 6), rnd.choice([32, 88, 300, 600])
            fresh, jobs, _, keep = V._gru_jobs(S, Bg, rnd.randint(32, 40), 300, In, seed * 77 + it)
            nj, blocks, what = 3, rnd.choice([0, 7, 96]), f'gru B={Bg} In={In}'
        else:
            Bt, Tt = rnd.randint(1, 8), rnd.randint(32, 40)
            fresh, jobs, _, keep = V._tcn_jobs(S, Bt, Tt, rnd.choice([260, 300, 320]), 2, seed * 99 + it)
            nj, blocks, what = 4, rnd.choice([0, 13, 256]), f'tcn B={Bt} T={Tt}'
        with config.override('GRU_SPLIT', pieces):
            o0 = fresh()
            V._run_wgrad(S, jobs(o0), nj, blocks)
            with config.override('WGRAD32_PIPE', ring):
                o1 = fresh()
                V._run_wgrad(S, jobs(o1), nj, blocks)
        ok = all(torch.equal(a, b) for (na, a), (_, b) in zip(V._flat(o0), V._flat(o1)) if na.startswith('dw'))
        okb = all(V._rel(b, a) < 1e-6 for (na, a), (_, b) in zip(V._flat(o0), V._flat(o1)) if not na.startswith('dw'))
        bad += not (ok and okb)
        print(f'wgrad pipe   {what} pieces={pieces} ring={ring} blocks={blocks}: dw {"bit-identical" if ok else "DIFFERS"}, db {"ok" if okb else "OFF"}', flush=True)
    print(f'TOTAL: {2 * n} cases, {bad} bad')
    return 1 if bad else 0


if __name__ == '__main__':
    sys.exit(main())
