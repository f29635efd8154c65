"""The adaptive mode's stopping rule is decided on the device from a lower bound on the reference's variance
(par_raytracer_amd/csrc/kernels_pool.h, finalise step) and by the reference's loop only where the bound does not decide.  The
bound must never say "go on" where CalculateVariance (main.cpp:190-222) says "stop": checked here in float32 arithmetic, the
device's expressions restated, on sample sets chosen to sit on either side of the threshold and right at it."""
from __future__ import annotations

import numpy as np

f32 = np.float32


def reference_variance(samples: np.ndarray) -> np.float32:
    """CalculateVariance(vals, count), float for float: sequential sums, L1 colour distance, (count - 1) divisor."""
    n = len(samples)
    mean = np.zeros(3, dtype=f32)
    for s in samples:
        mean = (mean + s).astype(f32)
    mean = (mean / f32(n)).astype(f32)
    var = f32(0)
    for s in samples:
        d = f32(f32(abs(f32(s[0] - mean[0])) + abs(f32(s[1] - mean[1]))) + abs(f32(s[2] - mean[2])))
        var = f32(var + f32(d * d))
    with np.errstate(divide="ignore", invalid="ignore"):
        return f32(var / f32(n - 1))


def device_says_go_on(samples: np.ndarray, threshold: float) -> bool:
    """The finalise step's shortcut: running colour sum and running sum of squares, both sequential float32 sums."""
    n = len(samples)
    total = np.zeros(3, dtype=f32)
    sq = f32(0)
    for s in samples:
        total = (total + s).astype(f32)
        sq = f32(sq + f32(f32(f32(s[0] * s[0]) + f32(s[1] * s[1])) + f32(s[2] * s[2])))
    dot = f32(f32(f32(total[0] * total[0]) + f32(total[1] * total[1])) + f32(total[2] * total[2]))
    bound = f32(sq - f32(dot / f32(n)))
    lhs = f32(bound - f32(f32(2e-5) * sq))
    rhs = f32(f32(f32(f32(threshold) * f32(n - 1)) * f32(1.001)) + f32(1e-30))
    return bool(lhs > rhs)


def test_bound_never_overrules_a_stop():
    rng = np.random.default_rng(7)
    checked = decided = stops = 0
    for trial in range(4000):
        n = int(rng.integers(2, 51))
        base = rng.uniform(0.0, 6.0, 3)
        spread = 10.0 ** rng.uniform(-4, 0.8)
        samples = np.maximum(0.0, base + rng.normal(0.0, spread, (n, 3))).astype(f32)
        if trial % 7 == 0:
            samples[:] = samples[0]                                   # sky: every sample the same colour
        if trial % 11 == 0:
            samples[rng.integers(0, n)] *= f32(50.0)                  # one firefly
        exact = reference_variance(samples)
        # thresholds far from, near and exactly at the exact value
        for thr in (0.01, float(exact) * 0.5, float(exact) * 0.999, float(exact), float(exact) * 1.001, float(exact) * 2.0):
            if not (thr > 0.0) or not np.isfinite(thr):
                continue
            checked += 1
            stop = bool(exact <= f32(thr))
            stops += stop
            if device_says_go_on(samples, thr):
                decided += 1
                assert not stop, "bound said go on, the reference stops: n %d threshold %g exact %g" % (n, thr, exact)
    assert decided > 1000 and stops > checked // 10, (checked, decided, stops)            # the test exercised both outcomes


def test_bound_with_one_sample_and_with_non_finite_colours():
    one = np.array([[1.0, 2.0, 3.0]], dtype=f32)
    assert not device_says_go_on(one, 0.01)              # n - 1 = 0 and identical samples: the loop decides (NaN <= t is false there)
    bad = np.array([[1.0, np.inf, 0.0], [0.5, 0.5, 0.5]], dtype=f32)
    with np.errstate(invalid="ignore", over="ignore"):
        assert not device_says_go_on(bad, 0.01)          # inf - inf = NaN fails the comparison: the loop decides
