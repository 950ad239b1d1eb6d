"""The device evaluates include/rt_math.h to the same bits as the host (the premise of
CPU<->GPU bit parity): large random sweeps of every primitive incl. correctly rounded
divide / sqrt, denormals and special values."""
import numpy as np
import pytest

from test_math import OPS, ev

pytestmark = pytest.mark.gpu


def sweep(rng, lo, hi, n=400000):
    return rng.uniform(lo, hi, n).astype(np.float32)


SPECIAL = np.array([0.0, -0.0, 1.0, -1.0, np.inf, -np.inf, np.nan, 1e-45, 1e-40, 1.17549435e-38, 3.4028235e38,
                    2.0 ** -32, 1 - 2.0 ** -24, 0.5, 6.2831855, 88.7, -103.9, 87.3], dtype=np.float32)


@pytest.mark.parametrize("op", ["log", "exp", "sin", "cos", "sqrt", "rsqrt", "rcp"])
def test_unary_bits(api, orc, op):
    rng = np.random.default_rng(OPS[op])
    tr = api.create_tracer(0)
    rngs = {"log": (0, 4), "exp": (-104, 89), "sin": (-7, 7), "cos": (-7, 7), "sqrt": (0, 1e6), "rsqrt": (0, 1e6), "rcp": (-1e4, 1e4)}[op]
    x = np.concatenate([sweep(rng, *rngs), SPECIAL, (rng.integers(0, 2 ** 32, 100000, dtype=np.uint64) / 4294967296.0).astype(np.float32),
                        rng.integers(0, 2 ** 32, 200000, dtype=np.uint64).astype(np.uint32).view(np.float32)])
    got = tr.debug_math_eval(OPS[op], x)
    want = ev(orc, op, x)
    tr.close()
    nan = np.isnan(want)
    assert np.array_equal(np.isnan(got), nan)
    assert np.array_equal(got.view(np.uint32)[~nan], want.view(np.uint32)[~nan]), op


@pytest.mark.parametrize("op", ["pow", "div", "smoothstep"])
def test_binary_bits(api, orc, op):
    rng = np.random.default_rng(10 + OPS[op])
    tr = api.create_tracer(0)
    x = np.concatenate([sweep(rng, 0, 1), sweep(rng, -1e6, 1e6), np.repeat(SPECIAL, len(SPECIAL)),
                        rng.integers(0, 2 ** 32, 200000, dtype=np.uint64).astype(np.uint32).view(np.float32)])
    y = np.concatenate([sweep(rng, 0.1, 4), sweep(rng, -1e3, 1e3), np.tile(SPECIAL, len(SPECIAL)),
                        rng.integers(0, 2 ** 32, 200000, dtype=np.uint64).astype(np.uint32).view(np.float32)])
    got = tr.debug_math_eval(OPS[op], x, y)
    want = ev(orc, op, x, y)
    tr.close()
    nan = np.isnan(want)
    assert np.array_equal(np.isnan(got), nan)
    assert np.array_equal(got.view(np.uint32)[~nan], want.view(np.uint32)[~nan]), op


@pytest.mark.parametrize("op", ["rcp", "sqrt", "log", "exp", "div"])
def test_fast_exact_sequences_equal_the_ieee_ones(api, orc, op):
    """rt_rcp / rt_sqrt / the divisions inside rt_log and rt_exp run shorter instruction sequences on the device
    (v_rcp_f32 / v_rsq_f32 + fused residual corrections, include/rt_math.h) that were proven bit-identical to the
    correctly rounded results over all 2^32 inputs (tools/ubench/exact_math.hip).  Here: every binade x 4096
    mantissas x both signs, the guard boundaries of the fast paths, and 3M random bit patterns, against the host
    (which evaluates the plain IEEE forms)."""
    rng = np.random.default_rng(77 + OPS[op])
    ex = np.arange(0, 256, dtype=np.uint32)
    man = np.concatenate([rng.integers(0, 1 << 23, 4090, dtype=np.uint64).astype(np.uint32), np.array([0, 1, 2, (1 << 23) - 1, (1 << 23) - 2, 1 << 22], dtype=np.uint32)])
    grid = ((ex[:, None] << 23) | man[None, :]).reshape(-1)
    bits = np.concatenate([grid, grid | np.uint32(0x80000000), rng.integers(0, 2 ** 32, 3_000_000, dtype=np.uint64).astype(np.uint32)])
    x = bits.view(np.float32)
    if op == "log":     # every reduced argument class: all mantissas of a few binades + the sweep above
        x = np.concatenate([x, (np.uint32(0x3f000000) + np.arange(0, 1 << 24, 7, dtype=np.uint32)).view(np.float32)])
    if op == "exp":
        x = np.concatenate([x, rng.uniform(-104, 89, 2_000_000).astype(np.float32), rng.uniform(-0.36, 0.36, 2_000_000).astype(np.float32)])
    tr = api.create_tracer(0)
    if op == "div":
        y = np.roll(x, 12345)
        got = tr.debug_math_eval(OPS[op], x, y)
        want = ev(orc, op, x, y)
    else:
        got = tr.debug_math_eval(OPS[op], x)
        want = ev(orc, op, x)
    tr.close()
    nan = np.isnan(want)
    assert np.array_equal(np.isnan(got), nan)
    bad = got.view(np.uint32)[~nan] != want.view(np.uint32)[~nan]
    assert not bad.any(), (op, int(bad.sum()), x[~nan][bad][:5])


@pytest.mark.parametrize("op", ["log", "exp", "sin", "cos", "sqrt", "rsqrt", "rcp", "pow"])
def test_wave_uniform_paths_equal_the_per_lane_ones(api, orc, op):
    """Round 6: on the device every function with special cases has a straight-line path behind RT_WAVE_ALL (include/rt_math.h) — taken when ALL active
    lanes of a wavefront hold ordinary arguments — in front of the unchanged per-lane code.  The sweeps above mix special values into most wavefronts;
    here every block of 64 consecutive arguments is of the fast class only — including the specials that class folds in with selects: log(+-0) = -inf,
    log(1) = +0, exp of anything that underflows to 0 (-inf too), exp of |x| <= 2^-14 — so the straight-line paths are what runs; then the same arguments
    with a NaN in every wavefront (the per-lane code runs) must give the same bits: both against the host."""
    rng = np.random.default_rng(900 + OPS[op])
    n = 64 * 6000
    if op == "log":
        x = np.exp(rng.uniform(-80, 80, n)).astype(np.float32)
        x[rng.random(n) < 0.2] = 0.0
        x[rng.random(n) < 0.05] = -0.0
        x[rng.random(n) < 0.1] = 1.0
    elif op == "exp":
        x = rng.uniform(-87.3, 87.3, n).astype(np.float32)
        x[rng.random(n) < 0.2] = rng.uniform(-1e-4, 1e-4)
        sel = rng.random(n) < 0.15
        x[sel] = rng.uniform(-1e4, -103.98, int(sel.sum())).astype(np.float32)
        x[rng.random(n) < 0.1] = -np.inf
        x[rng.random(n) < 0.05] = 0.0
    elif op in ("sin", "cos"):
        x = rng.uniform(-2.9e4, 2.9e4, n).astype(np.float32)
        x[: n // 2] = rng.uniform(0, 6.2831855, n // 2).astype(np.float32)
    elif op == "rcp":
        x = (np.exp(rng.uniform(-85, 85, n)) * rng.choice([-1.0, 1.0], n)).astype(np.float32)
    elif op in ("sqrt", "rsqrt"):
        x = np.exp(rng.uniform(-60, 60, n)).astype(np.float32)
    else:   # pow: the sky model's calls — pow(smoothstep in [0, 1], 0.35) and pow(max(0, dot), 1000 / sunFocus ...)
        x = rng.uniform(0, 1, n).astype(np.float32)
        x[rng.random(n) < 0.3] = 0.0
        x[rng.random(n) < 0.2] = 1.0
    y = rng.uniform(0.1, 900, n).astype(np.float32) if op == "pow" else None
    tr = api.create_tracer(0)
    for poison in (False, True):
        xx = x.copy()
        if poison:
            xx[::64] = np.nan
        got = tr.debug_math_eval(OPS[op], xx, y) if y is not None else tr.debug_math_eval(OPS[op], xx)
        want = ev(orc, op, xx, y) if y is not None else ev(orc, op, xx)
        nan = np.isnan(want)
        assert np.array_equal(np.isnan(got), nan), (op, poison)
        bad = got.view(np.uint32)[~nan] != want.view(np.uint32)[~nan]
        assert not bad.any(), (op, poison, int(bad.sum()), xx[~nan][bad][:5])
    tr.close()
