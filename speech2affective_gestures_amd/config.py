"""The ONE registry of run-time switches: name -> default, what it selects, the test that arms the non-default side.

Every switch is read from the environment as ``S2AG_<NAME>`` once, here; no other module of the package (and no file of
the C library) reads the environment for a product decision.  ``tests/test_host_logic.py`` fails on any ``S2AG_*``
environment name in the package that is not registered below, so an A/B experiment cannot leave a hidden flag behind
again (r03 had ~75; the ones whose alternative lost its A/B are deleted -- DESIGN.md sections 3 and 5 keep the verdicts --
and the constants they chose are plain module attributes that tests may set directly).

``get(name)`` returns the current value; ``override(name, value)`` is a context manager that flips a switch IN PROCESS --
for the switches the C library consumes it also calls ``s2ag_set_option`` -- so a test arms a path itself instead of
waiting for the caller's shell to export a variable."""
import contextlib
import os
from dataclasses import dataclass
from typing import Any, Callable, Dict, Optional


@dataclass(frozen=True)
class Switch:
    name: str
    default: Any
    kind: Callable            # bool / int / str parser of the environment string
    doc: str
    test: Optional[str]       # the test (pytest node id, or a file) that runs the non-default side; None: not a code path
    clib: bool = False        # consumed by libs2ag_hip.so: pushed through s2ag_set_option


def _flag(v: str) -> bool:
    return v.strip().lower() not in ('', '0', 'false', 'off', 'no')


_S = [
    # ---- modes a user chooses ------------------------------------------------------------------------------------------
    Switch('PRECISION', 'fp32', str, "fp32 | bf16 (Conv1d path: bf16 activations in HBM) | bf16_step (that + single-piece bf16 "
           "products in the GRU, its projections and the big weight gradients); bf16.precision() is the in-process form",
           'tests/test_gpu_bf16.py'),
    Switch('GRU_SPLIT', 2, int, "bf16 pieces per fp32 operand of the big matrix products: 2 (default, 16 mantissa bits), 3 "
           "(fp32-equivalent), 0 (f32 MFMA everywhere)", 'tests/test_gpu_ops.py::test_gru_forward_backward', clib=True),
    Switch('FORCE_DIST', False, _flag, "open a world-size-1 RCCL group so that the data-parallel schedule (graph segments + "
           "eager collectives) runs on one GPU", 'tests/test_gpu_step.py::test_world_size_1_rccl_between_graph_segments'),
    Switch('DIST_BACKEND', '', str, "torch.distributed backend instead of nccl (= RCCL): 'gloo' lets two ranks share one GPU",
           'tests/test_gpu_step.py::test_two_ranks_on_one_gpu_match_the_averaged_gradient_emulation'),
    Switch('SPARSE_EMBEDDING', True, _flag, "0: the word-embedding gradient travels as a dense all-reduce instead of the "
           "touched-row all-gather", 'tests/test_host_logic.py::test_gradient_exchange_schedule_world_size_2_gloo'),
    Switch('PREFETCH', True, _flag, "0: yield_batch gathers / decodes on the host synchronously (the reference's data path) "
           "instead of data.BatchFeeder", 'tests/test_gpu_step.py::test_prefetching_batch_feeder_matches_the_host_path'),
    Switch('DETERMINISTIC', False, _flag, "debug, needs the det build flavour (S2AG_HIP_LIB=.../libs2ag_hip_det.so): passes of a step on one "
           "stream, accumulating launches ordered by workgroup index: two runs from the same state are bit-identical",
           'tests/test_gpu_zz_det_flavour.py::test_deterministic_mode_two_runs_are_bit_identical'),
    # ---- fused paths with a layer-by-layer fall-back that parity tests compare against ------------------------------------
    Switch('WAVE12', True, _flag, "0: the wave encoder's head (conv1 + BatchNorm + LeakyReLU + conv2) layer by layer instead "
           "of csrc/wave12.hip", 'tests/test_gpu_wave12.py'),
    Switch('WAVE_FUSED', True, _flag, "0 (bf16 mode): the wave encoder's BatchNorms as kernels of their own instead of folded "
           "into the convs (csrc/wave_fused.hip)", 'tests/test_gpu_wave_fused.py'),
    Switch('TCN_FUSED32', True, _flag, "0 (fp32 mode): the text TCN layer by layer instead of the clip-resident launches",
           'tests/test_gpu_modules.py'),
    Switch('BN_FUSED', True, _flag, "0: BatchNorm statistics and apply as two launches instead of one with a grid-wide wait",
           'tests/test_gpu_ops.py::test_batch_norm_fused_statistics_shapes_and_ticket_rearm'),
    Switch('SYNTH_GRAPH', True, _flag, "0: sliding-window synthesis with eager launches instead of one hipGraph replay per "
           "window", 'tests/test_gpu_step.py::test_synthesis_after_training_steps_uses_the_current_weights'),
    # ---- opt-in kernel VARIANTS (r06, written with the GPU closed: never timed).  One .hip file each, default OFF, so that no
    # default binary moves; same results as the default kernel (tests/test_gpu_zy_variants.py); tools/ab_variants.sh times each
    # against its default in one process the day a GPU answers --------------------------------------------------------------
    Switch('WGRAD32_PIPE', 0, int, "1 | 2: fp32-operand weight gradients (GRUs, text TCN) by csrc/wgrad_tr32p.hip -- the split + "
           "LDS stores of step s + 1 run beside the MFMAs of step s, buffer loads with hardware bounds checks (1: three register "
           "sets of loads in flight as the default kernel, 2: two, everything in architectural VGPRs); bit-identical dw",
           'tests/test_gpu_zy_variants.py::test_pipelined_weight_gradient_is_bit_identical', clib=True),
    Switch('TCN32_PAIR', 0, int, "1: the clip-resident text TCN of the fp32 step (forward and data-gradient chain) with TWO clips per "
           "workgroup (csrc/tcn32p.hip): every weight fragment streamed once per two clips, 68 of 80 MFMA rows real instead of 34 "
           "of 48, activations in LDS as bf16 hi / lo planes split ONCE by their producer, one image + residual / running gradient in "
           "registers; 2: the same kernel with ONE clip per workgroup (what the planes buy without the pairing); bit-identical "
           "h1 / h2 / y / gp1 / gp2 / gx",
           'tests/test_gpu_zy_variants.py::test_pair_tcn_is_bit_identical', clib=True),
    Switch('BN_FOLD_APPLY', False, _flag, "1: a training-mode BatchNorm + LeakyReLU behind a conv that left its column sums (21 per "
           "step) folds them and applies in ONE launch (csrc/bn_foldapply.hip: every workgroup folds the small sums itself, fixed "
           "order, no grid-wide wait) instead of bn_fold_k + bn_apply_k", 'tests/test_gpu_zy_variants.py::test_bn_fold_apply_in_one_launch'),
    Switch('EMB_BWD_ROWS', 0, int, "1: the word-embedding gradient by csrc/emb_rows.hip -- 256 rows x 64 columns per workgroup, the PAD row "
           "(85 % of every transcript) summed in registers: one atomic per column and workgroup instead of ~4 runs (r03 micro-timing on "
           "an MI355X: 42 -> 17 us at B = 256; never through the suite on hardware)",
           'tests/test_gpu_zy_variants.py::test_row_block_embedding_backward', clib=True),
    # ---- process plumbing (no kernel is selected by these) --------------------------------------------------------------
    Switch('HIP_LIB', '', str, "path of another build of the same C ABI (debug / asan flavour)", None),
    Switch('CRASH_TRACE', False, _flag, "native back trace on a fatal signal (csrc/debug.hip)", None),
    Switch('BUILD_JOBS', 4, int, "parallel hipcc jobs of python -m speech2affective_gestures_amd.build", None),
]
REGISTRY: Dict[str, Switch] = {s.name: s for s in _S}
_values: Dict[str, Any] = {}


def _parse(sw: Switch, raw: Optional[str]):
    if raw is None:
        return sw.default
    return sw.kind(raw)


def get(name: str):
    sw = REGISTRY[name]                      # KeyError: an unregistered switch is a bug, not a default
    if name not in _values:
        _values[name] = _parse(sw, os.environ.get('S2AG_' + name))
    return _values[name]


def _push(lib, name: str, value) -> None:
    rc = lib.s2ag_set_option(name.encode(), int(value))
    if rc < 0:
        raise RuntimeError(f's2ag_set_option({name!r}, {int(value)}) refused: {rc}')


def push_to_library(lib) -> None:
    """Called once by _lib.load(): the library starts from the registry's values, never from its own getenv."""
    for sw in _S:
        if sw.clib:
            _push(lib, sw.name, get(sw.name))


def set_value(name: str, value):
    """Set a switch in process; returns the previous value.  Module-level mirrors (wave12.ENABLED, ops.BN_FUSED, ...: see
    ``mirror``) follow through ``on_change``; switches the C library consumes are pushed through ``s2ag_set_option``."""
    sw = REGISTRY[name]
    prev = get(name)
    _values[name] = value
    if sw.clib:
        from . import _lib
        _push(_lib.load(), name, value)
    for fn in _listeners.get(name, ()):
        fn(value)
    return prev


_listeners: Dict[str, list] = {}


def on_change(name: str, fn: Callable) -> None:
    assert name in REGISTRY, name
    _listeners.setdefault(name, []).append(fn)


def mirror(name: str, module_globals: dict, attr: str):
    """Keep ``module.attr`` equal to the switch (the modules read their own attribute at call time; tests may also set the
    attribute directly for a finer scope)."""
    module_globals[attr] = get(name)
    on_change(name, lambda v: module_globals.__setitem__(attr, v))
    return module_globals[attr]


@contextlib.contextmanager
def override(name: str, value):
    prev = set_value(name, value)
    try:
        yield
    finally:
        set_value(name, prev)


def table() -> str:
    """Markdown table of the registry (DESIGN.md section 8 is generated from this)."""
    rows = ['| switch | default | selects | armed by |', '|---|---|---|---|']
    for sw in _S:
        rows.append(f'| `S2AG_{sw.name}` | `{sw.default}` | {sw.doc} | {("`" + sw.test + "`") if sw.test else "-"} |')
    return '\n'.join(rows)
