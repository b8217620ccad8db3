"""TEST INFRASTRUCTURE: bench.py on the CPU device model (tests/emu) -- run under torch.distributed.run exactly as the driver
launches the N > 1 bench (one process per rank, RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* from the environment), with gloo
standing in for RCCL and `--dry-width` shrinking the model so that a step takes seconds on the model.  What this executes is
the LAUNCH PATH of the multi-GPU bench -- rank environment, process-group set-up, the four-segment data-parallel step with
its collectives between the segments, barrier + max-over-ranks timing, exactly one JSON line from rank 0 -- not a
measurement.  Used by tests/test_emu_suite.py::test_bench_launch_path_with_two_ranks."""
import os
import runpy
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(HERE, 'emu'))
import harness  # noqa: E402

harness.install()
if '--stub-heavy-extras' in sys.argv:
    # the single-process line WITH its extras (alt_modes, value_fp32_equivalent, cpu_baseline, long_context_run ...): the parts
    # that run full-width kernels for minutes on the model are replaced by stubs, everything that assembles the line is real
    sys.argv.remove('--stub-heavy-extras')
    import bench
    bench.conv1d_roofline_run = lambda *a, **k: {'stub': True}
    bench.gen_forward_ms = lambda *a, **k: {'stub': True}
    bench.epoch_loop_rate = lambda *a, **k: 1.0
    bench.dp_structure_run = lambda *a, **k: {'stub': True}
    _alt, _ts = bench.alt_modes, bench.timed_steps
    bench.alt_modes = lambda pr, dp, batch, B, steps=1: _alt(pr, dp, batch, B, steps=1)
    bench.timed_steps = lambda pr, dp, batch, steps, warmup, sync: _ts(pr, dp, batch, min(steps, 1), min(warmup, 1), sync)
    bench.CONFIGS['long'].update(batch=2, frames=34, audio_len=36267)      # (the second processor of the line, at the small shape)
    sys.argv = [os.path.join(ROOT, 'bench.py')] + sys.argv[1:]
    bench.main()
else:
    sys.argv = [os.path.join(ROOT, 'bench.py')] + sys.argv[1:]
    runpy.run_path(sys.argv[0], run_name='__main__')
