"""A/B timing of the opt-in kernel variants (config.py group 'opt-in kernel variants', tests/test_gpu_zy_variants.py) against the
default kernels -- ONE process, variants interleaved round by round (cdna_hip_programming.md rule 24: a perf delta is a
within-process interleaved measurement, N variants x M rounds, medians), HIP events on the launch stream.

    python tools/ab_variants.py [--rounds 15] [--only wgrad,cfg3,step]      (needs an MI355X; ~2 min)

Three levels per switch value:
  micro   the kernel alone at the shapes the step / BASELINE configs[3] launch it with
  cfg3    BASELINE configs[3] (WavEncoder + TextEncoderTCN forward + backward, B = 256), one captured iteration, fp32
  step    the GAN step at B = 128 (captured), clips/s
Prints a table and one JSON line (`AB {...}`); tools/gpu_first_call.sh keeps it under gpurun_out/first/ab_variants.txt.
A variant whose `micro` AND `cfg3`/`step` medians beat the default by more than the round-to-round spread is a candidate for the
default; until a GPU has said so every variant stays off."""
import argparse
import ctypes as C
import json
import os
import statistics
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))

if os.environ.get('S2AG_EMU', '0') == '1':      # DRY RUN of this script's logic on the CPU device model (tests/emu): times are meaningless
    sys.path.insert(0, os.path.join(ROOT, 'tests', 'emu'))
    import harness
    harness.install()
SMALL = os.environ.get('AB_SMALL', '0') == '1'      # dry-run sizes

# switch -> values to compare (first = default)
VARIANTS = {
    'WGRAD32_PIPE': [0, 1, 2],
    'TCN32_PAIR': [0, 1, 2],
    'BN_FOLD_APPLY': [0, 1],
    'EMB_BWD_ROWS': [0, 1],
}


# all variants together (the value each is expected to win with), timed at the cfg3 and step levels beside the single switches
ALL_ON = {'WGRAD32_PIPE': 2, 'TCN32_PAIR': 1, 'BN_FOLD_APPLY': 1, 'EMB_BWD_ROWS': 1}


def _all_on(config):
    import contextlib
    st = contextlib.ExitStack()
    for k, v in ALL_ON.items():
        st.enter_context(config.override(k, v))
    return st


def _p(t):
    return C.c_void_p(t.data_ptr())


def event_us(fn, stream):
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record(stream)
    fn()
    b.record(stream)
    b.synchronize()
    return a.elapsed_time(b) * 1e3


def interleaved(cases, rounds, stream, warm=3):
    """cases: {label: fn}.  Returns {label: (median_us, min_us, max_us)} from `rounds` rounds in which every case runs once, in
    rotating order."""
    labels = list(cases)
    for _ in range(warm):
        for k in labels:
            cases[k]()
    torch.cuda.synchronize()
    ts = {k: [] for k in labels}
    for r in range(rounds):
        for k in labels[r % len(labels):] + labels[:r % len(labels)]:
            ts[k].append(event_us(cases[k], stream))
    return {k: (statistics.median(v), min(v), max(v)) for k, v in ts.items()}


def wgrad_micro(rounds):
    """WGRAD32_PIPE: (a) the text TCN's eight weight gradients at B = 256 (BASELINE configs[3]); (b) one H = 300 GRU layer's three
    weight gradients at B = 128 (the step; 96 workgroups, as beside a cooperative recurrence)."""
    from speech2affective_gestures_amd import _lib as L
    from speech2affective_gestures_amd import config
    from test_gpu_zy_variants import _gru_jobs, _tcn_jobs
    lib = L.load()
    S = dict(L=L, lib=lib, config=config)
    st = torch.cuda.current_stream()
    sp = C.c_void_p(st.cuda_stream)
    out = {}
    for name, (fresh, jobs, _, keep), nj, blocks in (
            ('tcn_wgrad_B256', _tcn_jobs(S, 256 if not SMALL else 3, 34, 300, 4 if not SMALL else 1, 1), 8 if not SMALL else 2, 256),
            ('gru_wgrad_B128_in600', _gru_jobs(S, 128 if not SMALL else 2, 34, 300, 600, 2), 3, 0)):
        o = fresh()
        j = jobs(o)
        need = int(lib.s2ag_f32_wgrad_tr_scratch_floats_n(j, nj, blocks))
        sc = torch.empty(need, device='cuda')

        def launch(v):
            def fn():
                lib.s2ag_set_option(b'WGRAD32_PIPE', v)
                L.check(lib.s2ag_f32_wgrad_tr_n(j, nj, _p(sc), need, blocks, sp), 'f32_wgrad_tr')
            return fn
        res = interleaved({f'WGRAD32_PIPE={v}': launch(v) for v in VARIANTS['WGRAD32_PIPE']}, rounds, st)
        lib.s2ag_set_option(b'WGRAD32_PIPE', 0)
        out[name] = {k: dict(median_us=m, min_us=lo, max_us=hi) for k, (m, lo, hi) in res.items()}
    return out


def tcn_micro(rounds):
    """TCN32_PAIR: the clip-resident text TCN (4 blocks, C = 300, T = 34, dropout 0.3) forward and data-gradient chain at
    B = 256 (BASELINE configs[3]) and as the step's lockstep batch (3 passes x 128 clips, only the first saving)."""
    from speech2affective_gestures_amd import _lib as L
    from speech2affective_gestures_amd import config
    lib = L.load()
    st = torch.cuda.current_stream()
    sp = C.c_void_p(st.cuda_stream)
    out = {}
    Cch, T, nb = 300, 34, 4
    g = torch.Generator().manual_seed(3)
    ws = [(torch.randn(Cch, 2, Cch, generator=g) * 0.04).cuda() for _ in range(2 * nb)]
    bs = [(torch.randn(Cch, generator=g) * 0.1).cuda() for _ in range(2 * nb)]
    frag = torch.empty(int(lib.s2ag_tcn32_pack_elems(2 * nb)), dtype=torch.bfloat16, device='cuda')
    L.check(lib.s2ag_tcn32_pack((C.c_void_p * (2 * nb))(*[w.data_ptr() for w in ws]), 2 * nb, Cch, _p(frag), sp), 'pack')
    for name, B, nP in (('cfg3_B256', 256 if not SMALL else 4, 1), ('step_lockstep_3x128', 128 if not SMALL else 2, 3)):
        n_clips, rows = nP * B, B * T
        x = torch.randn(n_clips, T, Cch, generator=g).cuda()
        gy = (torch.randn(rows, Cch, generator=g) * 0.1).cuda()
        saved = torch.empty(3 * nb - 1, rows, Cch, device='cuda')
        y_last = torch.empty(n_clips * T, Cch, device='cuda')
        gx, gp = torch.empty(rows, Cch, device='cuda'), torch.empty(2 * nb, rows, Cch, device='cuda')
        keep = torch.zeros(int(lib.s2ag_tcn32_keep_bytes(n_clips, nb)), dtype=torch.uint8, device='cuda')
        noises = [torch.tensor([5, 10 + k], dtype=torch.int64, device='cuda') for k in range(nP)]
        a, b_ = L.Tcn32(), L.Tcn32()
        a.x, a.wfrag, b_.wfrag = x.data_ptr(), frag.data_ptr(), frag.data_ptr()
        for b in range(nb):
            for t_ in (a, b_):
                t_.h1[b], t_.h2[b] = saved[3 * b].data_ptr(), saved[3 * b + 1].data_ptr()
                t_.y[b] = saved[3 * b + 2].data_ptr() if b < nb - 1 else y_last.data_ptr()
                t_.dil[b] = 2 ** b
            b_.gp1[b], b_.gp2[b] = gp[2 * b].data_ptr(), gp[2 * b + 1].data_ptr()
            for j in range(2):
                a.bias[2 * b + j], a.site[2 * b + j] = bs[2 * b + j].data_ptr(), 40 + 2 * b + j
        a.n_blocks, a.n_clips, a.T, a.C, a.drop_p = nb, n_clips, T, Cch, 0.3
        a.rng, a.keep = noises[0].data_ptr(), keep.data_ptr()
        b_.n_blocks, b_.n_clips, b_.T, b_.C, b_.drop_p = nb, B, T, Cch, 0.3
        b_.gy, b_.gx = gy.data_ptr(), gx.data_ptr()
        rngs = (C.c_void_p * nP)(*[nz.data_ptr() for nz in noises])

        def fwd(v):
            def fn():
                lib.s2ag_set_option(b'TCN32_PAIR', v)
                L.check(lib.s2ag_tcn32_fwd_passes(C.byref(a), nP, rngs, B, sp), 'tcn32_fwd')
            return fn

        def bwd(v):
            def fn():
                lib.s2ag_set_option(b'TCN32_PAIR', v)
                L.check(lib.s2ag_tcn32_bwd(C.byref(b_), sp), 'tcn32_bwd')
            return fn
        for tag, mk in (('fwd', fwd), ('bwd', bwd)):
            res = interleaved({f'TCN32_PAIR={v}': mk(v) for v in VARIANTS['TCN32_PAIR']}, rounds, st)
            out[f'tcn_{tag}_{name}'] = {k: dict(median_us=m, min_us=lo, max_us=hi) for k, (m, lo, hi) in res.items()}
        lib.s2ag_set_option(b'TCN32_PAIR', 0)
    return out


def emb_micro(rounds):
    """EMB_BWD_ROWS: the word-embedding gradient at configs[3]'s shape (B = 256 x 34 ids, 85 % PAD, 300 columns, dropout 0.1)."""
    from speech2affective_gestures_amd import _lib as L
    lib = L.load()
    st = torch.cuda.current_stream()
    sp = C.c_void_p(st.cuda_stream)
    B, T, n_words, dim = (256 if not SMALL else 4), 34, 20000, 300
    g = torch.Generator().manual_seed(4)
    ids = torch.randint(1, n_words, (B * T,), generator=g)
    ids[torch.rand(B * T, generator=g) < 0.85] = 0
    ids, dy = ids.cuda(), torch.randn(B * T, dim, generator=g).cuda()
    dt = torch.zeros(n_words, dim, device='cuda')
    rng = torch.tensor([1, 2], dtype=torch.int64, device='cuda')
    e = L.Epilogue(0, 1.0, 0.1, C.c_void_p(rng.data_ptr()), 31)

    def mk(v):
        def fn():
            lib.s2ag_set_option(b'EMB_BWD_ROWS', v)
            L.check(lib.s2ag_embedding_bwd(_p(ids), _p(dy), dim, B * T, dim, n_words, _p(dt), 1, C.byref(e), sp), 'embedding_bwd')
        return fn
    res = interleaved({f'EMB_BWD_ROWS={v}': mk(v) for v in VARIANTS['EMB_BWD_ROWS']}, rounds, st)
    lib.s2ag_set_option(b'EMB_BWD_ROWS', 0)
    return {'embedding_bwd_B256': {k: dict(median_us=m, min_us=lo, max_us=hi) for k, (m, lo, hi) in res.items()}}


def bn_micro(rounds):
    """BN_FOLD_APPLY: fold + apply of a BatchNorm behind a stats-epilogue conv as two launches (default) or one, at MFCCEncoder's
    shape in the step (B = 128: 4 736 rows x 64 channels, 74 partial rows) and at the wave encoder's BatchNorm 3 in configs[3]
    (B = 256: 9 216 rows x 64 channels)."""
    from speech2affective_gestures_amd import _lib as L
    lib = L.load()
    st = torch.cuda.current_stream()
    sp = C.c_void_p(st.cuda_stream)
    out = {}
    for name, rows, cols, prow in (('mfcc_B128_64ch', 4736 if not SMALL else 150, 64, 74 if not SMALL else 3),
                                   ('wave_bn3_B256_64ch', 9216 if not SMALL else 144, 64, 144 if not SMALL else 3)):
        g = torch.Generator().manual_seed(6)
        x = torch.randn(rows, cols, generator=g).cuda()
        xb = x.double().view(prow, rows // prow, cols)
        part = torch.stack([xb.sum(1), (xb * xb).sum(1)]).contiguous()
        gamma, beta = torch.ones(cols, device='cuda'), torch.zeros(cols, device='cuda')
        rm, rv = torch.zeros(cols, device='cuda'), torch.ones(cols, device='cuda')
        nbt = torch.zeros(1, dtype=torch.int64, device='cuda')
        coef = torch.empty(4, cols, device='cuda')
        y = torch.empty(rows, cols, device='cuda')
        assert lib.s2ag_bn_fold_apply_supported(prow, cols, cols)

        def two():
            pp = part.clone()                                  # (s2ag_bn_fold may pre-fold its partials in place)
            L.check(lib.s2ag_bn_fold(_p(pp), prow, rows, cols, None, cols, _p(gamma), _p(beta), _p(rm), _p(rv), _p(nbt), 1e-5, 0.1, 1,
                                     _p(coef[0]), _p(coef[1]), _p(coef[2]), _p(coef[3]), sp), 'bn_fold')
            L.check(lib.s2ag_bn_apply(_p(x), rows, cols, cols, _p(coef[0]), _p(coef[1]), 0.3, _p(y), cols, sp), 'bn_apply')

        def one():
            pp = part.clone()                                  # (the same extra copy, so that the two sides differ in the launches only)
            L.check(lib.s2ag_bn_fold_apply(_p(pp), prow, rows, cols, None, cols, _p(gamma), _p(beta), _p(rm), _p(rv), _p(nbt), 1e-5,
                                           0.1, 1, _p(coef[0]), _p(coef[1]), _p(coef[2]), _p(coef[3]), _p(x), cols, 0.3, _p(y), cols,
                                           sp), 'bn_fold_apply')
        res = interleaved({'BN_FOLD_APPLY=0': two, 'BN_FOLD_APPLY=1': one}, rounds, st)
        out['bn_' + name] = {k: dict(median_us=m, min_us=lo, max_us=hi) for k, (m, lo, hi) in res.items()}
    return out


def cfg3_level(rounds):
    """One captured iteration of BASELINE configs[3] per switch value (bench.conv1d_roofline_run builds and times the graph;
    values are visited `rounds // 5 + 1` times in rotating order, the best median per value is kept)."""
    import bench
    from speech2affective_gestures_amd import config
    dev = torch.device('cuda')
    out = {}
    kw = dict(B=256, iters=30)
    if SMALL:                                   # dry run on the device model: no graph objects there, four clips, one iteration
        os.environ['S2AG_CFG3_EAGER'] = '1'
        kw = dict(B=4, iters=1)
    for sw, vals in VARIANTS.items():
        ms = {v: [] for v in vals}
        for r in range(rounds // 5 + 1):
            for v in vals[r % len(vals):] + vals[:r % len(vals)]:
                with config.override(sw, v):
                    ms[v].append(bench.conv1d_roofline_run(dev, cpu=False, mode='fp32', **kw)['ms_per_iter'])
        out[sw] = {f'{sw}={v}': dict(ms_per_iter=min(t), all=t, frac_hbm=kw['B'] * 5.75e6 / (min(t) * 1e-3) / 8e12) for v, t in ms.items()}
    ts = []
    for _ in range(rounds // 5 + 1):
        with _all_on(config):
            ts.append(bench.conv1d_roofline_run(dev, cpu=False, mode='fp32', **kw)['ms_per_iter'])
    out['ALL'] = {'ALL_ON ' + ','.join(f'{k}={v}' for k, v in ALL_ON.items()): dict(ms_per_iter=min(ts), all=ts, frac_hbm=kw['B'] * 5.75e6 / (min(ts) * 1e-3) / 8e12)}
    return out


def step_level(rounds):
    """The captured GAN step at B = 128 per switch value (a processor per value: the graph is captured under the switch)."""
    import bench
    from speech2affective_gestures_amd import config
    out = {}
    B, steps, warm, graph = (128, 30, 5, True) if not SMALL else (2, 1, 1, False)
    for sw, vals in VARIANTS.items():
        res = {}
        for v in vals:
            with config.override(sw, v):
                pr = bench.build_processor(B, graph, 34, bench.CONFIGS['step']['audio_len'])
                batch = bench.synthetic_batch(B, 0, pr.device, 34, bench.CONFIGS['step']['audio_len'])
                rates = []
                for _ in range(max(2, rounds // 5) if not SMALL else 1):
                    el = bench.timed_steps(pr, pr.dp, batch, steps, warm, sync=False)
                    rates.append(B * steps / el)
                res[f'{sw}={v}'] = dict(clips_per_s=statistics.median(rates), all=rates)
                del pr
                torch.cuda.empty_cache()
        out[sw] = res
    with _all_on(config):
        pr = bench.build_processor(B, graph, 34, bench.CONFIGS['step']['audio_len'])
        batch = bench.synthetic_batch(B, 0, pr.device, 34, bench.CONFIGS['step']['audio_len'])
        rates = [B * steps / bench.timed_steps(pr, pr.dp, batch, steps, warm, sync=False) for _ in range(max(2, rounds // 5) if not SMALL else 1)]
        out['ALL'] = {'ALL_ON': dict(clips_per_s=statistics.median(rates), all=rates)}
        del pr
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--rounds', type=int, default=15)
    ap.add_argument('--only', default='micro,cfg3,step')
    a = ap.parse_args()
    if not torch.cuda.is_available() and os.environ.get('S2AG_EMU', '0') != '1':
        raise SystemExit('tools/ab_variants.py needs an MI355X')
    only = set(a.only.split(','))
    res = {}
    if 'micro' in only:
        res['micro'] = dict(wgrad=wgrad_micro(a.rounds), tcn=tcn_micro(a.rounds), emb=emb_micro(a.rounds), bn=bn_micro(a.rounds))
        for grp in res['micro'].values():
            for name, r in grp.items():
                base = next(iter(r.values()))['median_us']          # (the first value of a switch is its default)
                for k, v in r.items():
                    print(f'micro {name:28s} {k:16s} {v["median_us"]:8.1f} us  [{v["min_us"]:.1f} .. {v["max_us"]:.1f}]  x{base / v["median_us"]:.2f}')
    if 'cfg3' in only:
        res['cfg3'] = cfg3_level(a.rounds)
        for sw, r in res['cfg3'].items():
            for k, v in r.items():
                print(f'cfg3  {k:24s} {v["ms_per_iter"]:.4f} ms/iter  {100 * v["frac_hbm"]:.1f} % of HBM')
    if 'step' in only:
        res['step'] = step_level(a.rounds)
        for sw, r in res['step'].items():
            for k, v in r.items():
                print(f'step  {k:24s} {v["clips_per_s"]:.0f} clips/s')
    # the decision the script is for, spelled out: a variant WINS a level when its gain over the default exceeds twice the
    # default's own spread at that level; a variant becomes the default only if it wins micro AND (configs[3] OR step)
    def gain_time(entries, key):          # {label: {key: t, ...}} -> {label: default_t / t}
        items = list(entries.items())
        base = items[0][1][key]
        return {k: base / v[key] for k, v in items[1:]}
    verdicts = {}
    for sw in VARIANTS:
        v = {}
        for grp in res.get('micro', {}).values():
            for name, r in grp.items():
                if next(iter(r)).startswith(sw + '='):
                    spread = (next(iter(r.values()))['max_us'] - next(iter(r.values()))['min_us']) / next(iter(r.values()))['median_us']
                    v.setdefault('micro', {})[name] = dict(gain=gain_time(r, 'median_us'), default_spread=spread)
        if sw in res.get('cfg3', {}):
            r = res['cfg3'][sw]
            d = next(iter(r.values()))['all']
            v['cfg3'] = dict(gain=gain_time(r, 'ms_per_iter'), default_spread=(max(d) - min(d)) / min(d))
        if sw in res.get('step', {}):
            r = res['step'][sw]
            base = next(iter(r.values()))
            v['step'] = dict(gain={k: x['clips_per_s'] / max(1e-30, base['clips_per_s']) for k, x in list(r.items())[1:]},
                             default_spread=(max(base['all']) - min(base['all'])) / max(1e-30, base['clips_per_s']))
        verdicts[sw] = v
        for lvl, e in v.items():
            for name, ee in (e.items() if lvl == 'micro' else [(lvl, e)]):
                for k, gval in ee['gain'].items():
                    mark = 'WINS' if gval > 1.0 + 2.0 * ee['default_spread'] else ('loses' if gval < 1.0 - 2.0 * ee['default_spread'] else 'within spread')
                    print(f'verdict {lvl:5s} {name:28s} {k:18s} x{gval:.3f} (default spread {100 * ee["default_spread"]:.1f} %): {mark}')
    res['verdicts'] = verdicts
    print('AB ' + json.dumps(res), flush=True)


if __name__ == '__main__':
    main()
