"""numpy restatement of the counter-based noise of libs2ag_hip.so (csrc/s2ag_common.h: mix32, site_key,
rand_u32, keep_scale, normal_dev).  Test infrastructure: lets the golden generator feed the REFERENCE the very
eps / keep masks the GPU kernels derive from (seed, pass counter, site, index), so a full training step of the
product can be compared with the reference's own output.  tests/test_gpu_ops.py checks GPU == this file."""
import numpy as np

M32 = np.uint64(0xFFFFFFFF)


def mix32(x):
    x = np.asarray(x, dtype=np.uint64) & M32
    x ^= x >> np.uint64(16)
    x = (x * np.uint64(0x7feb352d)) & M32
    x ^= x >> np.uint64(15)
    x = (x * np.uint64(0x846ca68b)) & M32
    x ^= x >> np.uint64(16)
    return x


def site_key(seed, counter, site):
    seed, counter, site = np.uint64(seed), np.uint64(counter), np.uint64(site)
    k0 = mix32((seed & M32) ^ mix32(((counter & M32) + np.uint64(0x9e3779b9) * (site + np.uint64(1))) & M32))
    k1 = mix32((seed >> np.uint64(32)) ^ mix32(((counter >> np.uint64(32)) +
                                                np.uint64(0x85ebca6b) * ((site + np.uint64(0x632be5ab)) & M32)) & M32))
    return k0, k1


def rand_u32(key, idx):
    k0, k1 = key
    idx = np.asarray(idx, dtype=np.uint64)
    x = mix32(((idx & M32) * np.uint64(0x9e3779b1) + k0) & M32)
    x = mix32(x ^ k1 ^ (((idx >> np.uint64(32)) * np.uint64(0xc2b2ae35)) & M32))
    return x


def keep_mask(seed, counter, site, p, n):
    """keep mask already scaled by 1/(1-p), flat index order"""
    key = site_key(seed, counter, site)
    u = (rand_u32(key, np.arange(n, dtype=np.uint64)) >> np.uint64(8)).astype(np.float32) * np.float32(1.0 / 16777216.0)
    return np.where(u >= np.float32(p), np.float32(1.0 / (1.0 - p)), np.float32(0.0)).astype(np.float32)


def normal(seed, counter, site, n):
    key = site_key(seed, counter, site)
    i = np.arange(n, dtype=np.uint64)
    a, b = rand_u32(key, 2 * i), rand_u32(key, 2 * i + 1)
    u1 = ((a >> np.uint64(8)).astype(np.float32) + np.float32(1.0)) * np.float32(1.0 / 16777216.0)
    u2 = (b >> np.uint64(8)).astype(np.float32) * np.float32(1.0 / 16777216.0)
    return (np.sqrt(np.float32(-2.0) * np.log(u1)) * np.cos(np.float32(6.283185307179586) * u2)).astype(np.float32)
