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


class Vocab:                       # duck-typed speaker model (utils/vocab.py): class name must be 'Vocab'
    def __init__(self, n):
        self.n_words = n
        self.word2index = {'v%d' % i: i for i in range(n)}


def synthetic_batch(B, seed, device):
    """CPU generator seeded per rank, then copied."""
    g = torch.Generator().manual_seed(seed)
    text = torch.zeros(B, T, dtype=torch.int64)
    for b in range(B):
        k = int(torch.randint(2, 9, (1,), generator=g))
        pos = torch.randperm(T, generator=g)[:k]
        text[b, pos] = torch.randint(4, N_WORDS, (k,), generator=g)
    audio = (torch.randn(B, AUDIO_LEN, generator=g) * 0.05).clamp_(-1, 1)
    mfcc = torch.randn(B, NUM_MFCC, MFCC_LEN, generator=g) * 0.1
    target = torch.randn(B, T, POSE_DIM, generator=g) * 0.2
    vid = torch.randint(0, N_SPK, (B,), generator=g)
    return [t.to(device) for t in (text, audio, mfcc, target, vid)]


def make_cfg():
    return types.SimpleNamespace(n_pre_poses=4, n_poses=T, input_context='both', hidden_size=300,
                                 hidden_size_s2eg=300, n_layers=4, dropout_prob=0.3, freeze_wordembed=False,
                                 loss_warmup=0, loss_gan_weight=5.0, z_type='speaker', loss_reg_weight=0.05,
                                 loss_regression_weight=500, loss_kld_weight=0.1, wordembed_dim=300,
                                 learning_rate=5e-4, discriminator_lr_weight=0.2)


def build_processor(B, hip_graph):
    from speech2affective_gestures_amd import processor_v2 as P
    lang = types.SimpleNamespace(n_words=N_WORDS, word_embedding_weights=None)
    meta = types.SimpleNamespace(n_poses=T, expected_audio_length=AUDIO_LEN, num_mfcc_combined=NUM_MFCC,
                                 lang_model=lang, speaker_model=Vocab(N_SPK), n_samples=0)
    args = types.SimpleNamespace(batch_size=B, train_s2ag=True, work_dir_s2ag=None, save_log=False, print_log=False,
                                 hip_graph=hip_graph,
                                 overlap_passes=os.environ.get('S2AG_OVERLAP_PASSES', '1') != '0')
    pr = P.Processor(ROOT, args, make_cfg(), {'train_data_s2ag': meta, 'val_data_s2ag': meta, 'test_data_s2ag': meta},
                     POSE_DIM, 3, 16000)
    pr.meta_info['epoch'] = 1            # discriminator branch active (epoch > loss_warmup)
    for m in (pr.s2ag_generator, pr.s2ag_discriminator, pr.trimodal_generator):
        m.train()
    return pr


def matrix_products_mode():
    """How the large matrix products are formed in this run (see the module docstring)."""
    from speech2affective_gestures_amd import _lib as L
    np_ = int(L.load().s2ag_gru_coop_split_pieces())
    return {0: 'f32 MFMA (v_mfma_f32_16x16x4_f32) everywhere',
            2: 'fp32 operands as 2 bf16 pieces (3 products, 16 mantissa bits, fp32 accumulation) on the bf16 matrix pipe for '
               'the GRU recurrence, its input projections / input gradients and the big convs forward; f32 MFMA elsewhere',
            3: 'fp32 operands as 3 bf16 pieces (6 products, fp32-equivalent, fp32 accumulation) on the bf16 matrix pipe for '
               'the GRU recurrence, its input projections / input gradients and the big convs forward; f32 MFMA elsewhere'
            }[np_]


def gru_roofline(B, iters=20):
    """Time the dominant kernel (gru_seq_fwd, H=300, both directions, T=34) with HIP events on its stream."""
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
    for _ in range(3):
        launch()
    evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(iters)]
    for a, b in evs:
        a.record(s)
        launch()
        b.record(s)
    torch.cuda.synchronize()
    ms = sum(a.elapsed_time(b) for a, b in evs) / iters
    flops = 2.0 * B * T * 2 * H * 3 * H           # recurrent mat-vec MACs x2, both directions
    achieved = flops / (ms * 1e-3) / 1e12
    # HBM/fabric bytes per launch from the PMC passes committed under profiles/r01_m_pmc_* (r01_i_pmc_* for the one-slice kernel) (rocprofv3 --pmc FETCH_SIZE
    # and WRITE_SIZE in separate runs, KB units, FETCH doubled as the gfx950 guide prescribes): gi + exchange-cell
    # reads, y / ydrop / saved-gate / exchange-cell writes.  Only meaningful at the profiled shape (B = 128).
    two_slices = coop and int(lib.s2ag_gru_coop_fwd_slices(B)) == 2
    traffic = ((2 * 112755.4 + 59134.1) if two_slices else (2 * 118252.0 + 59134.6)) * 1024 if (coop and B == 128) else None
    np_ = int(lib.s2ag_gru_coop_split_pieces()) if coop else 0
    ns_ = int(lib.s2ag_gru_coop_fwd_slices(B)) if coop else 1
    name = ((f'gru_coop_fwd_sp2_k<300,32,{np_},2>' if ns_ == 2 else f'gru_coop_fwd_sp_k<300,32,{np_}>') if np_
            else 'gru_coop_fwd_k<300,32>') if coop else 'gru_seq_fwd_k<8>'
    pipe = {0: '12 waves x 38 f32 MFMAs (16x16x4) per CU and step',
            2: '12 waves x 15 bf16 MFMAs (16x16x32; the 3 leading piece products of 2-piece bf16 splits of the fp32 operands: products carry 16 mantissa bits, error vs fp64 1.5e-6 against the f32-MFMA kernel\'s 3.8e-7, tools/diag_gru_split.py; S2AG_GRU_SPLIT=3 is fp32-equivalent) per CU and slice step',
            3: '12 waves x 30 bf16 MFMAs (16x16x32; the 6 leading piece products of exact 3-piece splits of the fp32 '
               'operands: fp32-equivalent, error vs fp64 equal to the f32-MFMA kernel, tools/diag_gru_split.py) per CU '
               'and step'}[np_]
    return dict(bound='mfma', kernel=name + ' (H=300, T=34, 2 directions)',
                achieved=achieved, peak=157.3, unit='TFLOP/s', frac=achieved / 157.3, traffic=traffic,
                traffic_source='profiles/r01_m_pmc_FETCH_SIZE.txt + r01_m_pmc_WRITE_SIZE.txt' if two_slices else 'profiles/r01_i_pmc_FETCH_SIZE.txt + r01_i_pmc_WRITE_SIZE.txt', ms_per_launch=ms,
                algorithmic_flops_per_launch=flops,
                note='peak = dense f32 MFMA peak (the arithmetic is fp32); algorithmic FLOPs = 2 x 3H x H per clip, frame '
                     'and direction.  Sequential recurrence, bound by the per-step exchange latency, not by the pipe: per '
                     'time step (tools/diag_coop_trace.py, profiles/r01_m_coop_gru_phase_trace.txt) ~1.3 us for the new '
                     'state to reach the peers (write-through tagged cells, polling loads), ' + pipe +
                     ', 0.4 us gate math, 0.5 us stores; ' + ('a workgroup alternates between two 16-clip slices (one travels while the other is computed): 80 of 256 CUs per launch -- 13 % longer alone than the one-slice kernel (S2AG_GRU_SLICES=1: 0.139 ms, frac 0.215, 160 CUs) but +1.4 % on the step, where passes share the chip' if ns_ == 2 else '160 of 256 CUs hold W_hh in registers') + '; ms_per_launch includes the '
                     '~5 us exchange-buffer clear')


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


def gen_forward_ms(pr, device):
    """BASELINE metric tail 'gen fwd ms': PoseGenerator forward, eval mode, B = 4 and B = 128, median of 100 replays."""
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
    return out


def conv1d_roofline_run(device, B=256, iters=30):
    """BASELINE configs[3] (the 'Conv1d roofline run'): WavEncoder + TextEncoderTCN forward + backward, train mode,
    dropout on, isolated, B = 256, fp32.  HBM-bound by design: algorithmic traffic 5.75 MB/clip (SURVEY.md 8d:
    every layer reads its input and writes its output once forward; backward re-reads x, reads dy, writes dx)."""
    from speech2affective_gestures_amd.net.multimodal_context_net_v2 import TextEncoderTCN, WavEncoder
    from speech2affective_gestures_amd.optim import ParamArena
    cfg = make_cfg()
    wav, txt = WavEncoder().to(device).train(), TextEncoderTCN(cfg, N_WORDS, 300, dropout=cfg.dropout_prob).to(device).train()
    arena = ParamArena(list(wav.parameters()) + list(txt.parameters()))
    text, audio, _, _, _ = synthetic_batch(B, 5, device)

    def fn():
        arena.zero_grad()
        (wav(audio).sum() + txt(text)[0].sum()).backward()
    ms = _graph_timer(fn, iters)
    clips = B / (ms * 1e-3)
    bytes_per_clip, flops_per_clip = 5.75e6, 413.8e6
    return dict(workload='BASELINE configs[3]: WavEncoder + TextEncoderTCN fwd+bwd, B=256, T=34, fp32, dropout on',
                ms_per_iter=ms, clips_per_s=clips,
                roofline=dict(bound='hbm', achieved=clips * bytes_per_clip / 1e9, peak=8000.0, unit='GB/s',
                              frac=clips * bytes_per_clip / 8e12, traffic=None,
                              note='fp32 MFMA (157.3 TF) caps this path at ~27% of the HBM roofline: '
                                   f'{clips * flops_per_clip / 1e12:.1f} TFLOP/s achieved of 157.3'))


def cpu_baseline(B, steps=2):
    """The oracle's gan_step (ATen fused GRU, drawn dropout) on the host cores -- bounded sample."""
    from oracle import s2ag_oracle as O
    try:
        import psutil
        phys = psutil.cpu_count(logical=False) or os.cpu_count()
    except Exception:
        phys = os.cpu_count()
    # the step is a chain of small ops: oversubscribing a many-core host makes it SLOWER, so the baseline uses the
    # fastest of a few thread counts (1 probe step each) -- the number reported is the best the host can do
    cand = sorted({c for c in (8, 16, 32, 64, phys) if c <= phys})
    oc = O.ModelCfg()
    G = O.recipe_state_dict(O.generator_shapes(oc, N_WORDS, N_SPK), 1)
    D = O.recipe_state_dict(O.aff_discriminator_shapes(), 2)
    T3 = O.recipe_state_dict(O.trimodal_shapes(oc, N_WORDS, N_SPK), 3)
    gopt, dopt, scfg = O.AdamState(), O.AdamState(), O.StepCfg()
    inp = O.recipe_inputs(B, T, 7, N_WORDS, N_SPK)

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
                sample=f'{steps} timed GAN steps (+1 warm-up) of the CPU oracle at batch {B}, T=34, fp32, '
                       f'torch {torch.__version__}, {cores} threads (fastest of {cand}); {dt * 1e3:.0f} ms/step')


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=50)
    ap.add_argument('--warmup', type=int, default=10)
    ap.add_argument('--batch', type=int, default=128, help='clips per GPU')
    ap.add_argument('--no-graph', action='store_true')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--no-extras', action='store_true', help='skip gen-forward latency and the Conv1d roofline run')
    a = ap.parse_args()

    if not torch.cuda.is_available():
        raise SystemExit('bench.py needs an MI355X; the product has no CPU path')
    pr = build_processor(a.batch, not a.no_graph)
    dp = pr.dp
    assert dp.world_size == a.gpus, f'--gpus {a.gpus} but WORLD_SIZE={dp.world_size}'
    from speech2affective_gestures_amd import noise
    noise.manual_seed(1234 + dp.rank)
    batch = synthetic_batch(a.batch, dp.rank, pr.device)
    text, audio, mfcc, target, vid = batch

    for _ in range(max(1, a.warmup)):       # also triggers graph capture (3 internal warm-up steps) on the 1st call
        pr.train_step(text, audio, mfcc, target, vid, sync=False)
    torch.cuda.synchronize()
    dp.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        pr.train_step(text, audio, mfcc, target, vid, sync=False)
    torch.cuda.synchronize()
    dp.barrier()
    torch.cuda.synchronize()
    elapsed = dp.max_over_ranks(time.perf_counter() - t0, pr.device)
    metric = pr._finish(pr._graphed['out']['comps'], pr._graphed['out']['dis']) if pr._graphed else None
    ms = elapsed / a.steps * 1e3
    value = a.batch * dp.world_size * a.steps / elapsed

    if dp.rank == 0:
        line = {
            'metric': 'gan_train_step_clips_per_sec', 'value': value, 'unit': 'clips/s', 'n_gpus': dp.world_size,
            'steps': a.steps, 'warmup': a.warmup, 'ms_per_step': ms, 'higher_is_better': True, 'scaling': 'weak',
            'vs_baseline': None, 'dtype': 'f32', 'data': 'synthetic',
            'config': {'workload': 'BASELINE configs[1]: full G+D GAN step (3 G fwd, 3 D fwd, 1 tri-modal fwd, 2 bwd, '
                                   '2 Adam), 34-frame TED-shaped clips', 'batch_per_gpu': a.batch,
                       'global_batch': a.batch * dp.world_size, 'frames': T, 'n_words': N_WORDS, 'n_speakers': N_SPK,
                       'parallelism': f'dp{dp.world_size}', 'hip_graph': not a.no_graph,
                       'matrix_products': matrix_products_mode(),
                       'last_step_losses': pr.last_losses if metric is not None else None},
        }
        line['roofline'] = gru_roofline(a.batch)
        if dp.world_size == 1 and not a.no_extras:
            line['gen_fwd_ms'] = gen_forward_ms(pr, pr.device)
            line['conv1d_roofline_run'] = conv1d_roofline_run(pr.device)
        if dp.world_size == 1 and not a.no_cpu_baseline:
            line['cpu_baseline'] = cpu_baseline(a.batch)
            line['gpu_over_cpu'] = value / line['cpu_baseline']['value']
        print(json.dumps(line), flush=True)
    dp.barrier()
    import torch.distributed as dist
    if dist.is_initialized():
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
