"""bench.py -- S2AG GAN train-step throughput on MI355X (BASELINE.json metric: train-step clips/s, 34-frame clips).

    python bench.py --gpus 1 --steps 50 --warmup 10
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

One step = Processor.train_step: 3 generator + 3 discriminator + 1 tri-modal forward, 2 backward, 2 Adam
(processor_v2.py:776-957 of the reference), on configs[1] of BASELINE.json: batch 128 per GPU, T = 34,
n_words 20000, 1371 speakers, synthetic TED-shaped inputs resident in HBM.  Storage, accumulation and
all element-wise arithmetic are fp32; the large matrix products (GRU recurrence and its input projections / input
gradients, the big convs' forward) multiply fp32 operands as bf16 pieces on the bf16 matrix pipe with fp32 accumulation:
2 pieces (16 mantissa bits per product, error vs fp64 ~1.5e-6) by default, 3 pieces (fp32-equivalent) with
S2AG_GRU_SPLIT=3, the f32 MFMA with S2AG_GRU_SPLIT=0 -- in every mode well above the bf16 BASELINE names, and inside
the 1e-3 parity bar by > 2 orders of magnitude (the line's config.matrix_products says which mode ran).  N > 1: one process per GPU, weak
scaling, RCCL all-reduce of the flat gradient arenas.  Rank 0 prints ONE JSON line.

Extra objects on that line:
  roofline     -- the dominant kernel of the step (the persistent GRU recurrence at H = 300), timed live with HIP
                  events on the launch stream around direct launches at the workload's exact shapes; `achieved`
                  = algorithmic FLOPs per launch / mean duration; peak = 157.3 TFLOP/s (fp32 MFMA == fp32 vector).
  cpu_baseline -- the CPU oracle (the pinned restatement of the reference, kind "port") timed on this host's
                  cores on a bounded sample of the same workload (rank 0, N = 1 only).
"""
import argparse
import json
import os
import sys
import time
import types

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

N_WORDS, N_SPK, T, POSE_DIM, AUDIO_LEN, MFCC_LEN, NUM_MFCC = 20000, 1371, 34, 27, 36267, 71, 37
MAX_WORDS_PER_CLIP = 8            # synthetic_batch draws k ~ U{2..8} words per clip

# BASELINE.json configs the bench can run as the timed workload (--config)
CONFIGS = {
    'step': dict(name='BASELINE configs[1]: full G+D GAN step (3 G fwd, 3 D fwd, 1 tri-modal fwd, 2 bwd, 2 Adam), '
                      '34-frame TED-shaped clips', batch=128, frames=34, audio_len=36267),
    # 146 000 samples -> exactly 136 wave-encoder frames; mfcc_length = ceil(146000 / 512) = 286 (processor_v2.py:124)
    'long': dict(name='BASELINE configs[4]: long-context full G+D GAN step, 136-frame clips (audio 146 000 samples, '
                      'mfcc_length 286), fp32', batch=64, frames=136, audio_len=146000),
}


class Vocab:                       # duck-typed speaker model (utils/vocab.py): class name must be 'Vocab'
    def __init__(self, n):
        self.n_words = n
        self.word2index = {'v%d' % i: i for i in range(n)}


def synthetic_batch(B, seed, device, frames=T, audio_len=AUDIO_LEN):
    """CPU generator seeded per rank, then copied."""
    g = torch.Generator().manual_seed(seed)
    text = torch.zeros(B, frames, dtype=torch.int64)
    for b in range(B):
        k = int(torch.randint(2, MAX_WORDS_PER_CLIP + 1, (1,), generator=g))
        pos = torch.randperm(frames, generator=g)[:k]
        text[b, pos] = torch.randint(4, N_WORDS, (k,), generator=g)
    audio = (torch.randn(B, audio_len, generator=g) * 0.05).clamp_(-1, 1)
    mfcc = torch.randn(B, NUM_MFCC, -(-audio_len // 512), generator=g) * 0.1
    target = torch.randn(B, frames, POSE_DIM, generator=g) * 0.2
    vid = torch.randint(0, N_SPK, (B,), generator=g)
    return [t.to(device) for t in (text, audio, mfcc, target, vid)]


HIDDEN = 300                      # (--dry-width: a launch-path dry run at reduced width, never the benchmark)


def make_cfg(frames=T):
    return types.SimpleNamespace(n_pre_poses=4, n_poses=frames, input_context='both', hidden_size=HIDDEN,
                                 hidden_size_s2eg=HIDDEN, n_layers=4, dropout_prob=0.3, freeze_wordembed=False,
                                 loss_warmup=0, loss_gan_weight=5.0, z_type='speaker', loss_reg_weight=0.05,
                                 loss_regression_weight=500, loss_kld_weight=0.1, wordembed_dim=300,
                                 learning_rate=5e-4, discriminator_lr_weight=0.2)


def build_processor(B, hip_graph, frames=T, audio_len=AUDIO_LEN):
    from speech2affective_gestures_amd import processor_v2 as P
    lang = types.SimpleNamespace(n_words=N_WORDS, word_embedding_weights=None)
    meta = types.SimpleNamespace(n_poses=frames, expected_audio_length=audio_len, num_mfcc_combined=NUM_MFCC,
                                 lang_model=lang, speaker_model=Vocab(N_SPK), n_samples=0)
    args = types.SimpleNamespace(batch_size=B, train_s2ag=True, work_dir_s2ag=None, save_log=False, print_log=False,
                                 hip_graph=hip_graph, max_words_per_clip=MAX_WORDS_PER_CLIP,
                                 overlap_passes=True)
    pr = P.Processor(ROOT, args, make_cfg(frames), {'train_data_s2ag': meta, 'val_data_s2ag': meta, 'test_data_s2ag': meta},
                     POSE_DIM, 3, 16000)
    pr.meta_info['epoch'] = 1            # discriminator branch active (epoch > loss_warmup)
    for m in (pr.s2ag_generator, pr.s2ag_discriminator, pr.trimodal_generator):
        m.train()
    return pr


def non_default_switches():
    from speech2affective_gestures_amd import config
    return {n: config.get(n) for n, sw in config.REGISTRY.items()
            if n not in ('HIP_LIB', 'CRASH_TRACE', 'BUILD_JOBS') and config.get(n) != sw.default}


# the value each opt-in variant is expected to win with (tools/ab_variants.py times every value separately)
VARIANTS_ALL_ON = {'WGRAD32_PIPE': 2, 'TCN32_PAIR': 1, 'BN_FOLD_APPLY': 1, 'EMB_BWD_ROWS': 1}


def variants_probe_child(a):
    """`bench.py --variants-probe` (started by variants_probe with S2AG_<switch> in the environment): 20 timed steps of the
    headline configuration and one captured configs[3] fp32 iteration; prints ONE line `VARIANTS_PROBE {json}`."""
    wl = CONFIGS['step']
    B = a.batch or wl['batch']
    pr = build_processor(B, not a.no_graph, wl['frames'], wl['audio_len'])
    from speech2affective_gestures_amd import noise
    noise.manual_seed(1234)
    batch = synthetic_batch(B, 0, pr.device, wl['frames'], wl['audio_len'])
    steps, warm = min(20, a.steps), min(5, a.warmup)
    el = timed_steps(pr, pr.dp, batch, steps, warm, sync=False)
    out = dict(switches=non_default_switches(), step_clips_per_s=B * steps / el, step_ms=el / steps * 1e3, steps=steps)
    del pr
    torch.cuda.empty_cache()
    out['conv1d_ms_per_iter'] = conv1d_roofline_run(torch.device('cuda'), cpu=False, mode='fp32').get('ms_per_iter')
    print('VARIANTS_PROBE ' + json.dumps(out), flush=True)


def variants_probe(a, default_value, default_conv1d_ms, timeout_s=180):
    """Child process of the default run: the same measurements with every opt-in kernel variant on.  Isolated on purpose -- the
    variants have never run on hardware; a fault or a time-out there costs this sub-object, not the line."""
    import shlex
    import subprocess
    env = dict(os.environ, **{'S2AG_' + k: str(v) for k, v in VARIANTS_ALL_ON.items()})
    # BENCH_PROBE_CMD: test hook -- the CPU suite starts the child through the device-model wrapper (tests/s2ag_emu_bench.py)
    self_cmd = shlex.split(os.environ['BENCH_PROBE_CMD']) if os.environ.get('BENCH_PROBE_CMD') else [sys.executable, os.path.abspath(__file__)]
    cmd = self_cmd + ['--variants-probe', '--steps', str(a.steps), '--warmup', str(a.warmup)] + \
          (['--batch', str(a.batch)] if a.batch else []) + (['--no-graph'] if a.no_graph else []) + \
          (['--dry-width', a.dry_width] if a.dry_width else [])
    try:
        p = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=timeout_s)
    except subprocess.TimeoutExpired:
        return dict(ok=False, error=f'no answer within {timeout_s} s', switches=VARIANTS_ALL_ON)
    ln = [x for x in p.stdout.splitlines() if x.startswith('VARIANTS_PROBE ')]
    if p.returncode != 0 or not ln:
        return dict(ok=False, error=f'rc {p.returncode}: ' + (p.stderr or p.stdout)[-400:], switches=VARIANTS_ALL_ON)
    d = json.loads(ln[-1][len('VARIANTS_PROBE '):])
    d.update(ok=True, step_vs_default=d['step_clips_per_s'] / default_value,
             conv1d_vs_default=(default_conv1d_ms / d['conv1d_ms_per_iter']) if d.get('conv1d_ms_per_iter') and default_conv1d_ms else None,
             note='20 timed steps / one captured configs[3] fp32 iteration in a child process with S2AG_<switch> set; parity of every '
                  'variant against its default kernel: tests/test_gpu_zy_variants.py; per-switch A/B: tools/ab_variants.py')
    return d


def matrix_products_mode():
    """How the large matrix products are formed in this run (see the module docstring)."""
    from speech2affective_gestures_amd import _lib as L
    np_ = int(L.load().s2ag_gru_coop_split_pieces())
    return {0: 'f32 MFMA (v_mfma_f32_16x16x4_f32) everywhere',
            1: 'bf16 step mode (S2AG_PRECISION=bf16_step): fp32 operands as ONE bf16 piece (one product, 8 mantissa bits per '
               'operand, fp32 accumulation) for the GRU recurrence, its input projections / input gradients and all large '
               'weight gradients; bf16 activations in the wave encoder and the text TCN; f32 MFMA elsewhere',
            2: 'fp32 operands as 2 bf16 pieces (3 products, 16 mantissa bits, fp32 accumulation) on the bf16 matrix pipe for '
               'the GRU recurrence, its input projections / input gradients / weight gradients and the text TCN (forward, '
               'data and weight gradients); f32 MFMA elsewhere',
            3: 'fp32 operands as 3 bf16 pieces (6 products, fp32-equivalent, fp32 accumulation) on the bf16 matrix pipe for '
               'the GRU recurrence, its input projections / input gradients and the big convs forward; f32 MFMA elsewhere'
            }[np_]


def dtype_label():
    """What the step computes in (storage / products), for the line's `dtype`."""
    from speech2affective_gestures_amd import _lib as L
    return {0: 'f32 (f32 MFMA products)', 1: 'bf16 products (one piece) / f32 storage of the GRU, bf16 Conv1d path',
            2: 'f32-storage/bf16x2-products', 3: 'f32-storage/bf16x3-products'}[
        int(L.load().s2ag_gru_coop_split_pieces())]


def pmc_traffic(kernel_prefix):
    """HBM bytes per launch of the kernel whose name starts with ``kernel_prefix``, from the tracked summary of the two
    separate rocprofv3 --pmc passes (profiles/r02_pmc_traffic.json, written by tools/pmc_traffic.py from
    FETCH_SIZE / WRITE_SIZE with the gfx950 corrections of the micro-architecture guide).  None when the file or the
    kernel is absent: the line then says traffic = null instead of quoting a stale constant."""
    import glob
    # newest round first (profiles/r<NN>_pmc_traffic.json, copied there from tools/profile_round.sh's output)
    for path in sorted(glob.glob(os.path.join(ROOT, 'profiles', 'r[0-9][0-9]_pmc_traffic.json')), reverse=True):
        fname = os.path.basename(path)
        try:
            with open(path) as f:
                table = json.load(f)
        except (OSError, ValueError):
            continue
        for name, ent in table.get('kernels', {}).items():
            if name.startswith(kernel_prefix):
                return float(ent['bytes_per_launch']), 'profiles/' + fname + ' (' + table.get('source', '?') + ')'
    return None, None


def pmc_traffic_per_iteration(mode):
    """HBM bytes of ONE iteration of the Conv1d roofline run (all its kernels), from the tracked summary of the two separate
    rocprofv3 --pmc passes over tools/run_cfg4.py (profiles/r03_pmc_traffic_cfg3_<mode>.json, tools/pmc_traffic.py).  None
    when the file is absent."""
    import glob
    cands = sorted(glob.glob(os.path.join(ROOT, 'profiles', f'r[0-9][0-9]_pmc_traffic_cfg3_{mode}.json')), reverse=True)
    path = cands[0] if cands else os.path.join(ROOT, 'profiles', f'r03_pmc_traffic_cfg3_{mode}.json')
    try:
        with open(path) as f:
            table = json.load(f)
        return float(table['bytes_per_iteration']), 'profiles/' + os.path.basename(path) + ' (' + table.get('source', '?') + ')'
    except (OSError, ValueError, KeyError):
        return None, None


def gru_roofline(B, iters=20, T=T):
    """Time the dominant kernel (the H=300 GRU forward recurrence, both directions, T frames) with HIP events on the
    stream it is launched on."""
    import ctypes as C
    from speech2affective_gestures_amd import _lib as L
    lib = L.load()
    H = 300
    dev = 'cuda'
    gi = torch.randn(B * T, 6 * H, device=dev) * 0.5
    whh = torch.randn(2, 3 * H, H, device=dev) * 0.05          # reference (state_dict) layout, both directions
    bhh = torch.randn(2, 3 * H, device=dev) * 0.05
    y = torch.empty(B * T, 2 * H, device=dev)
    yd = torch.empty_like(y)
    gates = torch.empty(2, B * T, 4 * H, device=dev)
    rng = torch.tensor([1, 0], dtype=torch.int64, device=dev)
    e = L.Epilogue(0, 1.0, 0.3, C.c_void_p(rng.data_ptr()), 1)
    s = torch.cuda.current_stream()
    sp = C.c_void_p(s.cuda_stream)

    coop = bool(lib.s2ag_gru_coop_supported(H))
    whhT = whh.transpose(1, 2).contiguous()
    ws = torch.empty(max(1, lib.s2ag_gru_coop_workspace_bytes(B, T, H, 0)), dtype=torch.uint8, device=dev)

    def launch():
        tail = (C.c_void_p(bhh.data_ptr()), C.c_void_p(y.data_ptr()), C.c_void_p(yd.data_ptr()),
                C.c_void_p(gates.data_ptr()), B, T, H, C.byref(e))
        if coop:
            L.check(lib.s2ag_gru_coop_fwd(C.c_void_p(gi.data_ptr()), C.c_void_p(whh.data_ptr()), *tail,
                                          C.c_void_p(ws.data_ptr()), sp), 'gru_coop_fwd')
        else:
            L.check(lib.s2ag_gru_seq_fwd(C.c_void_p(gi.data_ptr()), C.c_void_p(whh.data_ptr()),
                                         C.c_void_p(whhT.data_ptr()), *tail, sp), 'gru_seq_fwd')
    # The generator's three passes of a step run each decoder layer as ONE cooperative launch (s2ag_gru_coop_fwd_multi:
    # same kernel, three passes' slices side by side); the frozen baseline's four layers are single-pass launches.  A step
    # holds four launches of each kind, so the kernel's average launch -- what a rocprof summary of the step shows -- is
    # the mean of the two, with (3 + 1) / 2 passes of work.
    nP = 3
    multi = coop and bool(lib.s2ag_gru_coop_fwd_multi_supported(nP, B, H))
    if multi:
        gis = [gi] + [torch.randn_like(gi) * 0.5 for _ in range(nP - 1)]
        ys = [y] + [torch.empty_like(y) for _ in range(nP - 1)]
        yds = [yd] + [torch.empty_like(yd) for _ in range(nP - 1)]
        rngs = [rng] + [torch.tensor([1, k + 1], dtype=torch.int64, device=dev) for k in range(nP - 1)]
        arr = lambda ts: (C.c_void_p * nP)(*[None if t is None else t.data_ptr() for t in ts])
        wsm = torch.empty(lib.s2ag_gru_coop_fwd_multi_workspace_bytes(nP, B, T, H), dtype=torch.uint8, device=dev)
        a_gi, a_y, a_yd, a_g, a_r = arr(gis), arr(ys), arr(yds), arr([gates] + [None] * (nP - 1)), arr(rngs)

    def launch_multi():
        L.check(lib.s2ag_gru_coop_fwd_multi(nP, a_gi, C.c_void_p(whh.data_ptr()), C.c_void_p(bhh.data_ptr()), a_y, a_yd,
                                            a_g, B, T, H, 0.3, a_r, 1, C.c_void_p(wsm.data_ptr()), sp), 'gru_coop_fwd_multi')

    def timed(fn):
        for _ in range(3):
            fn()
        evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(iters)]
        for a, b in evs:
            a.record(s)
            fn()
            b.record(s)
        torch.cuda.synchronize()
        return sum(a.elapsed_time(b) for a, b in evs) / iters
    ms1 = timed(launch)
    flops1 = 2.0 * B * T * 2 * H * 3 * H          # recurrent mat-vec MACs x2, both directions, one pass
    if multi:
        ms3 = timed(launch_multi)
        ms, flops = 0.5 * (ms1 + ms3), 0.5 * (1 + nP) * flops1
    else:
        ms3 = None
        ms, flops = ms1, flops1
    achieved = flops / (ms * 1e-3) / 1e12
    np_ = int(lib.s2ag_gru_coop_split_pieces()) if coop else 0
    ns_ = int(lib.s2ag_gru_coop_fwd_slices(B)) if coop else 1
    name = ((f'gru_coop_fwd_sp2_k<300,32,{np_},2>' if ns_ == 2 else f'gru_coop_fwd_sp_k<300,32,{np_}>') if np_
            else 'gru_coop_fwd_k<300,32>') if coop else 'gru_seq_fwd_k<8>'
    # HBM / fabric bytes per launch: measured, from the tracked PMC summary -- only at the profiled shape
    traffic, source = pmc_traffic(name.split('<')[0]) if (coop and (B, T) == (128, 34)) else (None, None)
    pipe = {0: '12 waves x 38 f32 MFMAs (16x16x4) per CU and step',
            1: '12 waves x 5 bf16 MFMAs (16x16x32; ONE piece per fp32 operand, 8 mantissa bits per product: the bf16 step mode) '
               'per CU and slice step',
            2: '12 waves x 15 bf16 MFMAs (16x16x32; the 3 leading piece products of 2-piece bf16 splits of the fp32 '
               'operands: 16 mantissa bits per product) per CU and slice step',
            3: '12 waves x 30 bf16 MFMAs (16x16x32; the 6 leading piece products of exact 3-piece splits: '
               'fp32-equivalent) per CU and step'}[np_]
    algo_bytes = 4.0 * B * T * (6 * H + 2 * 2 * H + 2 * 4 * H) + 4.0 * 2 * 3 * H * H
    if multi:      # the two mate passes save no gates; W_hh is read once per launch
        algo_bytes = 0.5 * (algo_bytes + algo_bytes + (nP - 1) * 4.0 * B * T * (6 * H + 2 * 2 * H))
    # the products execute as `issued` bf16 MFMAs per fp32 product (2 pieces: 3, 3 pieces: 6; f32 MFMA: the f32 pipe itself)
    issued = {0: None, 1: 1, 2: 3, 3: 6}[np_]
    return dict(bound='mfma', kernel=name + f' (H=300, T={T}, 2 directions' + ('; average launch of the step: 4 three-pass lockstep launches + 4 one-pass launches)' if multi else ')'),
                achieved=achieved, peak=157.3, unit='TFLOP/s', frac=achieved / 157.3,
                # both denominators: `frac` prices the algorithmic fp32 FLOPs against the dense f32-MFMA
                # peak; the products EXECUTE as `issued` bf16 MFMAs each on the 2.5 PFLOP/s bf16 pipe, and that is the pipe
                # whose occupancy says how busy the matrix cores are
                frac_of_executing_pipe=(achieved * issued / 2500.0) if issued else achieved / 157.3,
                executing_pipe=(f'bf16 MFMA 16x16x32, dense peak 2500 TFLOP/s, {issued} issued per fp32 product' if issued
                                else 'f32 MFMA 16x16x4, dense peak 157.3 TFLOP/s'),
                frac_of_bf16_pipe=(achieved * issued / 2500.0) if issued else None,
                bf16_pipe_tflops_issued=(achieved * issued) if issued else None, traffic=traffic,
                traffic_source=source, algorithmic_bytes_per_launch=algo_bytes, ms_per_launch=ms,
                algorithmic_flops_per_launch=flops,
                launch_mix=(None if not multi else dict(
                    per_step='4 x three-pass lockstep launches (generator) + 4 x one-pass launches (frozen baseline)',
                    three_pass=dict(ms=ms3, tflops=nP * flops1 / (ms3 * 1e-3) / 1e12, frac=nP * flops1 / (ms3 * 1e-3) / 1e12 / 157.3),
                    one_pass=dict(ms=ms1, tflops=flops1 / (ms1 * 1e-3) / 1e12, frac=flops1 / (ms1 * 1e-3) / 1e12 / 157.3))),
                note='peak = dense f32 MFMA peak (the arithmetic is fp32); algorithmic FLOPs = 2 x 3H x H per clip, frame '
                     'and direction; algorithmic bytes = gi in, y / dropped y / saved gates out, W_hh once.  Sequential '
                     'recurrence, bound by the per-step exchange latency, not by a pipe: per time step ~1.3 us for the new '
                     'state to reach the peers (write-through tagged cells, polling loads), ' + pipe + ', 0.4 us gate '
                     'math, 0.5 us stores' + ('; a workgroup alternates between two 16-clip slices, 80 of 256 CUs per '
                     'launch' if ns_ == 2 else '; 160 of 256 CUs hold W_hh in registers') + '; ms_per_launch includes '
                     'the ~5 us exchange-buffer clear')


def _graph_timer(fn, iters, warm=3):
    """Capture ``fn`` into a HIP graph and return the median replay time in ms (HIP events on the launch stream)."""
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        for _ in range(warm):
            fn()
    torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        fn()
    g.replay()
    torch.cuda.synchronize()
    ts = []
    for _ in range(iters):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        g.replay()
        b.record()
        torch.cuda.synchronize()
        ts.append(a.elapsed_time(b))
    ts.sort()
    return ts[len(ts) // 2]


def _cpu_threads():
    try:
        import psutil
        phys = psutil.cpu_count(logical=False) or os.cpu_count()
    except Exception:
        phys = os.cpu_count()
    return phys, sorted({c for c in (8, 16, 32, 64, phys) if c <= phys})


def _cpu_time(fn, warm, timed, cand):
    """Median wall time of ``fn`` on the host at the fastest of a few thread counts (these are chains of small ops:
    oversubscribing a many-core host makes them slower)."""
    best, cores = None, cand[0]
    torch.set_num_threads(cand[0])
    for _ in range(warm):
        fn()
    for c in cand:
        torch.set_num_threads(c)
        t0 = time.perf_counter()
        fn()
        d = time.perf_counter() - t0
        if best is None or d < best:
            best, cores = d, c
    torch.set_num_threads(cores)
    ts = []
    for _ in range(timed):
        t0 = time.perf_counter()
        fn()
        ts.append(time.perf_counter() - t0)
    ts.sort()
    return ts[len(ts) // 2], cores


def gen_forward_ms(pr, device, cpu=True):
    """BASELINE metric tail 'gen fwd ms': PoseGenerator forward, eval mode, B = 4 and B = 128, median of 100 replays;
    beside it the CPU oracle's forward on the host cores (3 warm-up + 10 timed, median)."""
    out = {}
    G = pr.s2ag_generator
    was = G.training
    G.eval()
    try:
        for B in (4, 128):
            text, audio, mfcc, target, vid = synthetic_batch(B, 99, device)
            pre = pr._make_pre_seq(target)

            def fn():
                with torch.no_grad():
                    G(pre, text, mfcc, vid)
            out[f'b{B}'] = _graph_timer(fn, 100)
    finally:
        G.train(was)
    if cpu:
        from oracle import s2ag_oracle as O
        phys, cand = _cpu_threads()
        oc = O.ModelCfg()
        sd = O.recipe_state_dict(O.generator_shapes(oc, N_WORDS, N_SPK), 1)
        base = {}
        for B in (4, 128):
            inp = O.recipe_inputs(B, T, 7, N_WORDS, N_SPK)
            pre = O.make_pre_seq(inp['target'], 4)

            def one():
                with torch.no_grad():
                    O.pose_generator(sd, oc, pre, inp['in_text'], inp['in_mfcc'], inp['vid'], False, O.Noise(None), True)
            dt, cores = _cpu_time(one, 3, 10, cand)
            base[f'b{B}'] = dt * 1e3
            base[f'b{B}_cores'] = cores
        out['cpu_baseline'] = dict(value=base, unit='ms', kind='port', physical_cores=phys,
                                   sample='CPU oracle PoseGenerator forward (eval), 3 warm-up + 10 timed, median, fastest '
                                          f'of {cand} threads')
    return out


def conv1d_roofline_run(device, B=256, iters=30, cpu=True, mode='fp32'):
    """BASELINE configs[3] (the 'Conv1d roofline run'): WavEncoder + TextEncoderTCN forward + backward, train mode,
    dropout on, isolated, B = 256, in fp32 or in bf16 mode (bf16 activations in HBM, fp32 accumulation / statistics /
    master weights; csrc/conv_bf16.hip).  HBM-bound by design.  Algorithmic traffic per clip (every layer
    reads its input and writes its output once forward; backward re-reads x, reads dy, writes dx): 5.75 MB in fp32;
    2.95 MB in bf16 mode (the 145 KB waveform stays fp32, every activation and activation gradient halves).  Every
    iteration starts a new step generation, so the per-optimizer-step weight preparation (weight norm, tap-major / bf16
    operand copies) is inside the timed region, as in training."""
    from speech2affective_gestures_amd import bf16, ops
    from speech2affective_gestures_amd.net.multimodal_context_net_v2 import TextEncoderTCN, WavEncoder
    from speech2affective_gestures_amd.optim import ParamArena
    cfg = make_cfg()
    wav, txt = WavEncoder().to(device).train(), TextEncoderTCN(cfg, N_WORDS, 300, dropout=cfg.dropout_prob).to(device).train()
    arena = ParamArena(list(wav.parameters()) + list(txt.parameters()))
    text, audio, _, _, _ = synthetic_batch(B, 5, device)

    side = torch.cuda.Stream()
    two_streams = os.environ.get('S2AG_CFG3_STREAMS', '2') != '1'
    g_ones = torch.ones(B, T, 32, device=device)

    def fn():
        # the two encoders share nothing: as in the training step (where they are branches of one generator pass) they
        # run on two streams, forward and backward each
        ops.begin_step()
        arena.zero_grad()
        # the output gradient is a resident tensor of ones (d/dy of y.sum(), without the harness's reduce / fill launches)
        if not two_streams:
            wav(audio).backward(g_ones)
            txt(text)[0].backward(g_ones)
            return
        cur = torch.cuda.current_stream()
        side.wait_stream(cur)
        with torch.cuda.stream(side):
            txt(text)[0].backward(g_ones)
        y = wav(audio)
        ops.set_main_stream()                # as the trainer does: big weight gradients (leaves of the backward graph) run
        y.backward(g_ones)                   # beside the data-gradient chain on the weight-gradient stream
        ops.join_side_streams()
        cur.wait_stream(side)
    with bf16.precision(mode):
        if os.environ.get('S2AG_CFG3_EAGER', '0') == '1':       # counter passes (tools/profile_r03.sh): exactly `iters` eager runs
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(iters):
                fn()
            torch.cuda.synchronize()
            ms = (time.perf_counter() - t0) / iters * 1e3
            print(f'iterations_run={iters}', flush=True)
        else:
            ms = _graph_timer(fn, iters)
    clips = B / (ms * 1e-3)
    bytes_per_clip, flops_per_clip = (5.75e6 if mode == 'fp32' else 2.95e6), 413.8e6
    traffic, tsrc = pmc_traffic_per_iteration(mode) if B == 256 else (None, None)
    out = dict(workload=f'BASELINE configs[3]: WavEncoder + TextEncoderTCN fwd+bwd, B=256, T=34, {mode}, dropout on',
               config=dict(workload="BASELINE configs[3] (ablation-audio 'Conv1d roofline run'): the north-star's HBM figure "
                                    '(target >= 40 % of 8 TB/s) is roofline.frac of THIS sub-line', batch=B, frames=T,
                           mode=mode, streams=2 if two_streams else 1),
               ms_per_iter=ms, clips_per_s=clips,
               dtype='f32 storage, bf16x2 products (TCN, weight gradients) / f32 MFMA' if mode == 'fp32'
               else 'bf16 storage, fp32 accumulation / statistics / master weights',
               roofline=dict(bound='hbm', achieved=clips * bytes_per_clip / 1e9, peak=8000.0, unit='GB/s',
                             frac=clips * bytes_per_clip / 8e12, traffic=traffic, traffic_source=tsrc,
                             traffic_over_algorithmic=(traffic / (B * bytes_per_clip)) if traffic else None,
                             algorithmic_bytes_per_clip=bytes_per_clip,
                             algorithmic_bytes_per_iteration=B * bytes_per_clip,
                             frac_at_fp32_bytes=clips * 5.75e6 / 8e12,
                             note=f'{clips * flops_per_clip / 1e12:.1f} TFLOP/s of matrix work at this rate '
                                  '(413.8 MFLOP/clip); one "launch" of this line = one iteration (all kernels of the two '
                                  'encoders, forward + backward); kernel-level trace: profiles/r03_cfg3_fp32_kernel_stats.txt, '
                                  'profiles/r03_cfg3_bf16_kernel_stats.txt'))
    if cpu and mode == 'fp32':
        from oracle import s2ag_oracle as O
        phys, cand = _cpu_threads()
        oc = O.ModelCfg()
        sd = O.recipe_state_dict({**O._wav_encoder_shapes('wav.'),
                                  **O._text_encoder_shapes('txt.', N_WORDS, 300, oc.hidden_size, oc.n_layers)}, 11)
        inp = O.recipe_inputs(B, T, 5, N_WORDS, N_SPK)

        def one():
            leaf = {k: (v.detach().requires_grad_(True) if O.is_param(k) and '.net.' not in k else v)
                    for k, v in sd.items()}
            for k in list(leaf):
                if '.net.0.' in k or '.net.4.' in k:
                    leaf[k] = leaf[k.replace('.net.0.', '.conv1.').replace('.net.4.', '.conv2.')]
            y = O.wav_encoder(leaf, 'wav.', inp['in_audio'], True).sum() + \
                O.text_encoder_tcn(leaf, 'txt.', inp['in_text'], True, oc.dropout_prob, O.Noise(None)).sum()
            y.backward()
        dt, cores = _cpu_time(one, 3, 10, cand)
        out['cpu_baseline'] = dict(value=B / dt, unit='clips/s', cores=cores, kind='port', physical_cores=phys,
                                   sample=f'CPU oracle WavEncoder + TextEncoderTCN fwd+bwd at B={B}, 3 warm-up + 10 timed, '
                                          f'median, fastest of {cand} threads; {dt * 1e3:.0f} ms/iter')
    return out


def cpu_baseline(B, steps=5, frames=T, audio_len=AUDIO_LEN):      # 2 warm-up + 5 timed steps
    """The oracle's gan_step (ATen fused GRU, drawn dropout) on the host cores -- bounded sample."""
    from oracle import s2ag_oracle as O
    phys, cand = _cpu_threads()
    oc = O.ModelCfg(n_poses=frames)
    mfcc_len = -(-audio_len // 512)
    G = O.recipe_state_dict(O.generator_shapes(oc, N_WORDS, N_SPK, mfcc_length=mfcc_len), 1)
    D = O.recipe_state_dict(O.aff_discriminator_shapes(frames), 2)
    T3 = O.recipe_state_dict(O.trimodal_shapes(oc, N_WORDS, N_SPK), 3)
    gopt, dopt, scfg = O.AdamState(), O.AdamState(), O.StepCfg()
    inp = O.recipe_inputs(B, frames, 7, N_WORDS, N_SPK, audio_len=audio_len, mfcc_len=mfcc_len)

    def one():
        O.gan_step(G, D, T3, gopt, dopt, oc, scfg, inp['in_text'], inp['in_audio'], inp['in_mfcc'], inp['target'],
                   inp['vid'], epoch=1, noise=O.StepNoise.fresh(), fast=True)
    torch.set_num_threads(cand[0])
    one()                                             # warm-up (allocator, lazy inits)
    best, cores = None, cand[0]
    for c in cand:
        torch.set_num_threads(c)
        t0 = time.perf_counter()
        one()
        d = time.perf_counter() - t0
        if best is None or d < best:
            best, cores = d, c
    torch.set_num_threads(cores)
    t0 = time.perf_counter()
    for _ in range(steps):
        one()
    dt = (time.perf_counter() - t0) / steps
    return dict(value=B / dt, unit='clips/s', cores=cores, kind='port', physical_cores=phys,
                sample=f'{steps} timed GAN steps (+1 warm-up, +1 probe per thread count) of the CPU oracle at batch {B}, '
                       f'T={frames}, fp32, torch {torch.__version__}, {cores} threads (fastest of {cand}: the step is a '
                       f'chain of small ops, more threads are slower); {dt * 1e3:.0f} ms/step')


def dp_structure_run(steps):
    """The data-parallel STRUCTURE of the step on one GPU: a child process with S2AG_FORCE_DIST=1 opens a world-size-1 RCCL
    group, so the step runs as four graph segments with the eager collectives (id-count MAX, D arena, bucket A beside the
    encoders' backward, bucket B, touched-row all-gather) between them -- what every rank of an N-GPU run executes, minus the
    wire time.  Its rate against `value` is the cost of the segment / host round-trip structure itself."""
    import subprocess
    env = dict(os.environ, S2AG_FORCE_DIST='1', S2AG_BENCH_CHILD='1', RANK='0', WORLD_SIZE='1', LOCAL_RANK='0',
               MASTER_ADDR='127.0.0.1', MASTER_PORT=str(29400 + os.getpid() % 500))
    try:
        r = subprocess.run([sys.executable, os.path.abspath(__file__), '--no-extras', '--no-cpu-baseline', '--steps', str(steps),
                            '--warmup', '5'], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=600)
        line = json.loads(r.stdout.decode(errors='replace').strip().split('\n')[-1])
        return dict(clips_per_s=line['value'], ms_per_step=line['ms_per_step'], steps=steps,
                    gradient_exchange_bytes_per_rank=line['config'].get('gradient_exchange_bytes_per_rank'),
                    note='S2AG_FORCE_DIST=1: world-size-1 RCCL group, 4 graph segments + eager collectives between them')
    except Exception as e:       # noqa: BLE001  (an extra: never takes the headline down)
        return dict(error=repr(e)[:200])


def epoch_loop_rate(pr, B, n_clips=4096):
    """PCIe-inclusive rate (never `value`): Processor.per_train_epoch's loop -- yield_batch -> train_step -- over a synthetic
    TED-shaped numpy dataset resident in HOST memory (raw int16 audio, float64 poses, float16 MFCCs as the reference's cache
    holds them), with the prefetching feeder (data.BatchFeeder: pinned staging, own copy stream, device-side decode)."""
    import numpy as np
    rs = np.random.RandomState(0)
    text = np.zeros((n_clips, T), dtype=np.int64)
    for i in range(n_clips):
        k = rs.randint(2, MAX_WORDS_PER_CLIP + 1)
        text[i, rs.permutation(T)[:k]] = rs.randint(4, N_WORDS, k)
    samples = dict(extended_word_seq=text, vec_seq=rs.randn(n_clips, T, POSE_DIM) * 0.2,
                   audio=np.clip(rs.randn(n_clips, AUDIO_LEN) * 0.05 * 32767, -32767, 32767).astype(np.int16),
                   audio_max=np.ones(n_clips), mfcc_features=(rs.randn(n_clips, NUM_MFCC, MFCC_LEN) * 0.1).astype(np.float16),
                   vid_indices=rs.randint(0, N_SPK, n_clips))
    pr.train_samples, pr.num_train_samples = samples, n_clips
    pr.args.prefetch_batches = True
    rate = None
    for ep in range(2):                                  # epoch 0 warms up (pinned allocations, feeder thread)
        np.random.seed(ep)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        n = 0
        for text_b, vec, audio, mfcc, vids in pr.yield_batch(train=True):
            pr.train_step(text_b, audio, mfcc, vec, vids, sync=False)
            n += B
        torch.cuda.synchronize()
        rate = n / (time.perf_counter() - t0)
    return dict(clips_per_s=rate, clips_per_epoch=n_clips,
                note='host-resident dataset: gather + PCIe + device-side decode inside the timed loop (data.BatchFeeder)')


def timed_steps(pr, dp, batch, steps, warmup, sync):
    """W untimed + exactly K timed steps between barrier + synchronize on both sides; max over ranks."""
    text, audio, mfcc, target, vid = batch
    for _ in range(max(1, warmup)):       # also triggers graph capture (3 internal warm-up steps) on the 1st call
        pr.train_step(text, audio, mfcc, target, vid, sync=sync)
    torch.cuda.synchronize()
    dp.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        pr.train_step(text, audio, mfcc, target, vid, sync=sync)
    torch.cuda.synchronize()
    dp.barrier()
    torch.cuda.synchronize()
    return dp.max_over_ranks(time.perf_counter() - t0, pr.device)


def alt_modes(pr, dp, batch, B, steps=20):
    """The same step with the large matrix products formed differently (same process, graphs re-captured): 3 bf16
    pieces = fp32-equivalent products, 0 = the f32 MFMA everywhere.  The headline `value` is the default (2 pieces)."""
    from speech2affective_gestures_amd import _lib as L
    lib = L.load()
    out = {}
    prev = int(lib.s2ag_gru_coop_split_override())
    try:
        for pieces, tag in ((3, 'fp32_equivalent_3_bf16_pieces'), (0, 'f32_mfma_everywhere')):
            lib.s2ag_gru_coop_set_split_pieces(pieces)
            pr._graphed = None
            el = timed_steps(pr, dp, batch, steps, 2, False)
            out[tag] = dict(clips_per_s=B * dp.world_size * steps / el, ms_per_step=el / steps * 1e3, steps=steps)
    finally:
        lib.s2ag_gru_coop_set_split_pieces(prev)
        pr._graphed = None
    # BASELINE configs[1] names bf16: the same step with the Conv1d path (wave encoder, text TCN) in bf16 mode -- bf16
    # activations in HBM, fp32 accumulation / statistics / master weights (bf16.py; its own, looser parity tests)
    from speech2affective_gestures_amd import bf16
    try:
        with bf16.precision('bf16'):
            pr._graphed = None
            el = timed_steps(pr, dp, batch, steps, 2, False)
            out['bf16_conv_path'] = dict(clips_per_s=B * dp.world_size * steps / el, ms_per_step=el / steps * 1e3, steps=steps,
                                         dtype='bf16 activations in the wave encoder and the text TCN, fp32 elsewhere')
        # ... and with every large matrix product of the step on ONE bf16 piece per operand (one product instead of three):
        # the cooperative GRU's recurrence, its input projections / input gradients, the GRU / TCN / wave-encoder weight
        # gradients (bf16.precision('bf16_step'); storage of gi / y / gates, accumulation, statistics, Adam stay fp32)
        with bf16.precision('bf16_step'):
            pr._graphed = None
            el = timed_steps(pr, dp, batch, steps, 2, False)
            out['bf16_step'] = dict(clips_per_s=B * dp.world_size * steps / el, ms_per_step=el / steps * 1e3, steps=steps,
                                    dtype='bf16 products (one piece per fp32 operand) in the GRU, its projections and all large '
                                          'weight gradients + bf16 activations in the wave encoder and the text TCN; fp32 '
                                          'storage elsewhere, fp32 accumulation / statistics / master weights')
    finally:
        pr._graphed = None
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=50)
    ap.add_argument('--warmup', type=int, default=10)
    ap.add_argument('--config', choices=sorted(CONFIGS), default='step',
                    help="'step' = BASELINE configs[1] (the metric's configuration); 'long' = configs[4] (T=136, B=64)")
    ap.add_argument('--batch', type=int, default=None, help='clips per GPU (default: the config\'s)')
    ap.add_argument('--no-graph', action='store_true')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--no-extras', action='store_true',
                    help='skip gen-forward latency, the Conv1d roofline run, alternative product modes, the long config')
    ap.add_argument('--variants-probe', action='store_true',
                    help='(child process of the default run) the step rate and the configs[3] fp32 iteration with whatever '
                         'S2AG_* switches the environment sets, as one small JSON line')
    ap.add_argument('--dry-width', default=None, metavar='HIDDEN,N_WORDS,N_SPEAKERS',
                    help='DRY RUN of the launch path (rank env, process group, timing protocol, the one JSON line) at a reduced '
                         'model width -- what the CPU test of the N > 1 launch uses (tests/test_emu_suite.py).  The line it '
                         'prints is labelled as such and is not a measurement of the benchmark configuration')
    a = ap.parse_args()
    if a.dry_width:
        global HIDDEN, N_WORDS, N_SPK
        HIDDEN, N_WORDS, N_SPK = (int(v) for v in a.dry_width.split(','))

    if not torch.cuda.is_available():
        raise SystemExit('bench.py needs an MI355X; the product has no CPU path')
    if a.variants_probe:
        return variants_probe_child(a)
    wl = CONFIGS[a.config]
    B, frames, audio_len = a.batch or wl['batch'], wl['frames'], wl['audio_len']
    pr = build_processor(B, not a.no_graph, frames, audio_len)
    dp = pr.dp
    assert dp.world_size == a.gpus, f'--gpus {a.gpus} but WORLD_SIZE={dp.world_size}'
    from speech2affective_gestures_amd import noise
    noise.manual_seed(1234 + dp.rank)
    batch = synthetic_batch(B, dp.rank, pr.device, frames, audio_len)

    elapsed = timed_steps(pr, dp, batch, a.steps, a.warmup, sync=False)
    metric = pr._finish(pr._graphed['out']['comps'], pr._graphed['out']['dis']) if pr._graphed else None
    ms = elapsed / a.steps * 1e3
    value = B * dp.world_size * a.steps / elapsed
    # the epoch loop's call (per_train_epoch: one 40-byte loss read-back per step) -- the API path beside the async one
    el_sync = timed_steps(pr, dp, batch, max(5, a.steps // 2), 1, sync=True)
    sync_value = B * dp.world_size * max(5, a.steps // 2) / el_sync

    if dp.rank == 0:
        ex = pr._exchange()
        line = {
            'metric': 'gan_train_step_clips_per_sec' if not a.dry_width else
                      'DRY_RUN_reduced_width_not_the_benchmark__gan_train_step_clips_per_sec', 'value': value, 'unit': 'clips/s', 'n_gpus': dp.world_size,
            'steps': a.steps, 'warmup': a.warmup, 'ms_per_step': ms, 'higher_is_better': True, 'scaling': 'weak',
            'vs_baseline': None, 'dtype': dtype_label(), 'data': 'synthetic',
            'config': {'workload': wl['name'], 'batch_per_gpu': B, 'global_batch': B * dp.world_size, 'frames': frames,
                       'audio_samples': audio_len, 'n_words': N_WORDS, 'n_speakers': N_SPK, 'hidden_size': HIDDEN,
                       'parallelism': f'dp{dp.world_size}', 'hip_graph': not a.no_graph,
                       'matrix_products': matrix_products_mode(),
                       # every registry switch that is not at its default in this run (opt-in kernel variants included): a line
                       # measured with S2AG_TCN32_PAIR=1 etc. says so itself
                       'non_default_switches': non_default_switches(),
                       'gradient_exchange_bytes_per_rank': ex.bytes_per_step() if ex is not None else None,
                       'last_step_losses': pr.last_losses if metric is not None else None},
            'value_with_per_step_loss_readback': sync_value,
        }
        # the headline's large products carry 16 mantissa bits per operand (two bf16 pieces): the rate of the SAME step with
        # fp32-equivalent products (three pieces) -- the figure to hold against an fp32 reference -- at the top level too
        line['value_fp32_equivalent'] = None
        line['roofline'] = gru_roofline(B, T=frames) if not a.dry_width else None
        if dp.world_size == 1 and not a.no_extras and not dp.active:
            line['value_epoch_loop'] = epoch_loop_rate(pr, B) if a.config == 'step' else None
            line['value_dp_structure'] = dp_structure_run(a.steps)
        if dp.world_size == 1 and not a.no_extras:
            line['alt_modes'] = alt_modes(pr, dp, batch, B)
            if matrix_products_mode().startswith('fp32 operands as 2'):
                line['value_fp32_equivalent'] = line['alt_modes']['fp32_equivalent_3_bf16_pieces']['clips_per_s']
            elif int(__import__('speech2affective_gestures_amd._lib', fromlist=['load']).load().s2ag_gru_coop_split_pieces()) in (0, 3):
                line['value_fp32_equivalent'] = value
            line['gen_fwd_ms'] = gen_forward_ms(pr, pr.device, cpu=not a.no_cpu_baseline)
            line['conv1d_roofline_run'] = conv1d_roofline_run(pr.device, cpu=not a.no_cpu_baseline)
            line['conv1d_roofline_run_bf16'] = conv1d_roofline_run(pr.device, cpu=False, mode='bf16')
            if (not a.dry_width or os.environ.get('BENCH_PROBE_CMD')) and not (set(non_default_switches()) & set(VARIANTS_ALL_ON)):
                # the opt-in kernel variants (written with the GPU closed, never timed): the same step and the same configs[3]
                # iteration with all of them ON, in a child process with a time limit -- whatever happens there, this line stands
                line['opt_in_variants_all_on'] = variants_probe(a, value, line['conv1d_roofline_run'].get('ms_per_iter'))
        if dp.world_size == 1 and not a.no_cpu_baseline:
            line['cpu_baseline'] = cpu_baseline(B, frames=frames, audio_len=audio_len)
            line['gpu_over_cpu'] = value / line['cpu_baseline']['value']
        if dp.world_size == 1 and not a.no_extras and a.config == 'step':
            # BASELINE configs[4] beside the headline configuration (its own processor; parity: tests/test_gpu_step.py
            # ::test_long_clip_steps_136_frames_match_the_oracle and tests/test_gpu_fullsize.py)
            del pr
            torch.cuda.empty_cache()
            lw = CONFIGS['long']
            pl = build_processor(lw['batch'], not a.no_graph, lw['frames'], lw['audio_len'])
            lb = synthetic_batch(lw['batch'], 0, pl.device, lw['frames'], lw['audio_len'])
            el = timed_steps(pl, pl.dp, lb, 10, 3, sync=False)
            line['long_context_run'] = dict(workload=lw['name'], batch_per_gpu=lw['batch'], frames=lw['frames'],
                                            clips_per_s=lw['batch'] * 10 / el, ms_per_step=el / 10 * 1e3, steps=10,
                                            dtype='f32')
        # RCCL writes a version banner through C stdio, which is flushed at exit -- i.e. AFTER this line; push it out first
        # so that the JSON line is the last thing on stdout
        try:
            import ctypes
            ctypes.CDLL(None).fflush(None)
        except Exception:
            pass
        print(json.dumps(line), flush=True)
    dp.barrier()
    import torch.distributed as dist
    if dist.is_initialized():
        dist.destroy_process_group()


if __name__ == '__main__':
    # No restart supervisor (r02 had one for a once-in-~60-starts death below hipGraphLaunch): 500 fresh-process starts + 12
    # loops of the GPU test suite at r03 produced no death (tools/stress_starts.py, profiles/r03_stress_starts.json).  Should
    # one occur, the native frames are printed (csrc/debug.hip) and the process dies with its signal -- rc != 0, no line.
    os.environ.setdefault('S2AG_CRASH_TRACE', '1')
    main()
