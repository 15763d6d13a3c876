"""div_by_inv (hyperseg_amd/csrc/hs_common.h): n / d as floor((float(n) + 0.5f) * (1.0f / d)) -- the image-level training kernels' replacement for
the compiler's 32-bit division by a run-time value.  The HIP build uses -ffp-contract=off, so the device evaluates exactly this float32
expression; it is held to integer division over the whole range the launchers admit (n < 2^21) for every divisor a patch edge, tile edge,
channel count or pair count can take, including the neighbourhoods of every multiple of d where a wrong rounding would show."""
import numpy as np


def div_by_inv(n, d):
    inv = np.float32(1.0) / np.float32(d)
    return ((n.astype(np.float32) + np.float32(0.5)) * inv).astype(np.int64)


def test_div_by_inv_is_exact_below_2_to_21():
    rng = np.random.default_rng(0)
    top = 1 << 21
    for d in list(range(1, 600)) + [1000, 1023, 1024, 1025, 4096, 5000, 65535, 1 << 20]:
        multiples = np.arange(0, top // d + 1, dtype=np.int64) * d
        n = np.concatenate([np.arange(0, min(top, 70000)), rng.integers(0, top, 100000), np.arange(top - 5000, top),
                            multiples - 1, multiples, multiples + 1]).astype(np.int64)
        n = n[(n >= 0) & (n < top)]
        assert np.array_equal(div_by_inv(n, d), n // d), d
