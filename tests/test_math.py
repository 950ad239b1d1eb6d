"""include/rt_math.h: the shared fp32 primitives are accurate (vs float64 libm)
and have the edge-case behaviour the tracer relies on.  CPU side (through the
oracle library); tests/test_gpu_math.py checks the device evaluates the same bits."""
import numpy as np
import pytest

OPS = {"log": 0, "exp": 1, "sin": 2, "cos": 3, "sqrt": 4, "pow": 5, "div": 6, "smoothstep": 7, "rsqrt": 8, "rcp": 9}


def ev(orc, op, x, y=None):
    x = np.ascontiguousarray(x, dtype=np.float32)
    y = np.ascontiguousarray(np.zeros_like(x) if y is None else y, dtype=np.float32)
    out = np.empty_like(x)
    orc.math_eval(OPS[op], x.ctypes.data, y.ctypes.data, out.ctypes.data, len(x))
    return out


def ulp_err(got, want64):
    want32 = want64.astype(np.float32)
    ulp = np.spacing(np.abs(want32)).astype(np.float64)
    return np.abs(got.astype(np.float64) - want64) / ulp


def test_log_accuracy(orc):
    rng = np.random.default_rng(1)
    x = np.concatenate([rng.uniform(0, 1, 200000), np.exp(rng.uniform(-80, 80, 200000)),
                        np.arange(1, 4097) / 4294967296.0, 1 - np.arange(0, 4096) / 16777216.0]).astype(np.float32)
    x = x[x > 0]
    e = ulp_err(ev(orc, "log", x), np.log(x.astype(np.float64)))
    assert e.max() < 1.0, e.max()


def test_log_edge_cases(orc):
    r = ev(orc, "log", [0.0, -0.0, 1.0, np.inf, -1.0, np.nan, 1e-45])
    assert r[0] == -np.inf and r[1] == -np.inf and r[2] == 0.0 and r[3] == np.inf
    assert np.isnan(r[4]) and np.isnan(r[5])
    assert abs(r[6] - np.log(np.float64(np.float32(1e-45)))) < 1e-4


def test_exp_accuracy(orc):
    rng = np.random.default_rng(2)
    x = np.concatenate([rng.uniform(-87, 88, 300000), rng.uniform(-1, 1, 100000), rng.uniform(-20, 0, 100000)]).astype(np.float32)
    e = ulp_err(ev(orc, "exp", x), np.exp(x.astype(np.float64)))
    assert e.max() < 1.0, e.max()


def test_exp_edge_cases(orc):
    r = ev(orc, "exp", [0.0, -np.inf, np.inf, 89.0, -104.0, np.nan, -100.0, 1e-8])
    assert r[0] == 1.0 and r[1] == 0.0 and r[2] == np.inf and r[3] == np.inf and r[4] == 0.0 and np.isnan(r[5])
    assert r[6] == np.float32(np.exp(-100.0))  # subnormal result, single rounding
    assert r[7] == np.float32(1.0)


@pytest.mark.parametrize("op,fn", [("sin", np.sin), ("cos", np.cos)])
def test_sincos_accuracy_on_tracer_range(orc, op, fn):
    rng = np.random.default_rng(3)
    # the tracer evaluates sin/cos on [0, 2*pi] only (RC:144,161)
    x = np.concatenate([rng.uniform(0, 6.2831855, 400000), np.linspace(0, 6.2831855, 100001)]).astype(np.float32)
    got = ev(orc, op, x)
    want = fn(x.astype(np.float64))
    assert np.max(np.abs(got - want)) < 1.5e-7
    mask = np.abs(want) > 0.05
    assert ulp_err(got[mask], want[mask]).max() < 2.0


@pytest.mark.parametrize("op,fn", [("sin", np.sin), ("cos", np.cos)])
def test_sincos_wider_range_and_specials(orc, op, fn):
    rng = np.random.default_rng(4)
    x = rng.uniform(-2000, 2000, 200000).astype(np.float32)
    assert np.max(np.abs(ev(orc, op, x) - fn(x.astype(np.float64)))) < 5e-7
    r = ev(orc, op, [np.inf, -np.inf, np.nan])
    assert np.all(np.isnan(r))
    assert ev(orc, "sin", [0.0])[0] == 0.0 and ev(orc, "cos", [0.0])[0] == 1.0


def test_pow_matches_exp_log_and_hlsl_edges(orc):
    rng = np.random.default_rng(5)
    x = rng.uniform(0, 1, 100000).astype(np.float32)
    y = rng.uniform(0.1, 4, 100000).astype(np.float32)
    got = ev(orc, "pow", x, y)
    want = np.power(x.astype(np.float64), y.astype(np.float64))
    assert np.max(np.abs(got - want) / np.maximum(want, 1e-30)) < 2e-6
    r = ev(orc, "pow", [0.0, 1.0, 0.5, 0.0], [0.35, 123.0, 2.0, 2.0])
    assert r[0] == 0.0 and r[1] == 1.0 and abs(r[2] - 0.25) < 1e-7 and r[3] == 0.0


def test_rcp_sqrt_correctly_rounded_and_div_is_mul_by_rcp(orc):
    rng = np.random.default_rng(6)
    a = rng.uniform(-1e3, 1e3, 200000).astype(np.float32)
    b = rng.uniform(1e-3, 1e3, 200000).astype(np.float32)
    rcp = (1.0 / b.astype(np.float64)).astype(np.float32)
    assert np.array_equal(ev(orc, "rcp", b), rcp)
    # HLSL '/' as GPUs execute it: a * rcp(b), each step correctly rounded (<= 1.5 ulp overall)
    assert np.array_equal(ev(orc, "div", a, b), (a.astype(np.float64) * rcp.astype(np.float64)).astype(np.float32))
    assert ulp_err(ev(orc, "div", a, b), a.astype(np.float64) / b.astype(np.float64)).max() <= 1.5
    assert np.array_equal(ev(orc, "sqrt", np.abs(a)), np.sqrt(np.abs(a).astype(np.float64)).astype(np.float32))


def test_rsqrt_newton_accuracy_and_specials(orc):
    rng = np.random.default_rng(8)
    x = np.concatenate([rng.uniform(1e-6, 1e6, 300000), np.exp(rng.uniform(-85, 85, 200000)),
                        [1.0, 4.0, 0.25, 1e-40, 1.2e-38, 3e38]]).astype(np.float32)
    got = ev(orc, "rsqrt", x)
    want = 1.0 / np.sqrt(x.astype(np.float64))
    assert ulp_err(got, want).max() < 1.5, ulp_err(got, want).max()
    r = ev(orc, "rsqrt", [0.0, -0.0, np.inf, -1.0, np.nan])
    assert r[0] == np.inf and r[1] == -np.inf and r[2] == 0.0 and np.isnan(r[3]) and np.isnan(r[4])
    # normalised vectors come out at unit length within a few ulp
    v = rng.normal(size=(100000, 3)).astype(np.float32)
    d = (v[:, 0] * v[:, 0] + v[:, 1] * v[:, 1] + v[:, 2] * v[:, 2]).astype(np.float32)
    n = v * ev(orc, "rsqrt", d)[:, None]
    assert np.max(np.abs(np.linalg.norm(n.astype(np.float64), axis=1) - 1)) < 5e-7


def test_smoothstep(orc):
    x = np.array([-1, 0, 0.1, 0.2, 0.4, 1, np.nan], dtype=np.float32)
    inv = np.full_like(x, np.float32(1.0) / np.float32(0.4))   # the folded constant 1/(b-a)
    r = ev(orc, "smoothstep", x, inv)
    assert inv[0] == np.float32(2.5)
    assert r[0] == 0 and r[1] == 0 and r[4] == 1 and r[5] == 1 and abs(r[3] - 0.5) < 1e-6
    assert r[6] == 0  # saturate(NaN) = 0 (HLSL)
