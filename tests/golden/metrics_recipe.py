"""Inputs of the metrics golden (tests/golden/gen_golden_metrics.py) -- numbers only, shared by the generating script and
the tests.  ``MEAN_DIR_VEC`` has the nesting of the reference's config value (a list holding one list of 27 numbers)."""
import numpy as np

N_BATCHES, N_PRE, T = 3, 4, 34
BATCH_SIZES = (5, 16, 3)
MEAN_DIR_VEC = [[round(float(v), 7) for v in np.random.RandomState(7100).standard_normal(27) * 0.3]]


def metrics_inputs(b: int):
    """(generated, target) direction vectors of batch ``b``: float32 (B_b, 34, 27)."""
    rs = np.random.RandomState(7101 + b)
    tgt = (rs.standard_normal((BATCH_SIZES[b], T, 27)) * 0.2).astype(np.float32)
    out = (tgt * 0.7 + rs.standard_normal(tgt.shape) * 0.1 + 0.02).astype(np.float32)
    return out, tgt
