"""CPU: the GELU epilogue's Phi(-|x|) = exp2(degree-6 polynomial) form (csrc/common.hpp half_erfc2_lp; the only form since round 6)
checked over EVERY finite bf16 input against the erf-GELU of the reference (nn.GELU() default, src/models/utils/modules.py:32).

The coefficients are read out of common.hpp, the arithmetic is restated in numpy with fp32 roundings where the kernel has them
(FMA = one rounding, exp2 = correctly rounded here / 1 ulp on the GPU), the exact value comes from torch's float64 erf.  The
epilogue rounds its result to bf16, so the figure of merit is how often that rounding differs from the rounding of the exact value.
The GPU-side check of the same numbers is tests/test_gemm_gpu.py::test_gelu_poly_epilogue."""
import os
import re

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
HPP = os.path.join(HERE, "..", "jepa_amd", "csrc", "common.hpp")


def _coefficients():
    src = open(HPP).read()
    body = src[src.index("half_erfc2_lp(f32x2_t x, f32x2_t& a) {"):]
    body = body[:body.index("return")]
    cs = {}
    for name, val in re.findall(r"\b(c[0-6]) = \{([-+0-9.eE]+)f,", body):
        cs[name] = np.float32(val)
    assert sorted(cs) == ["c%d" % i for i in range(7)], cs
    clamp = re.search(r"fmed3f\(__builtin_fabsf\(x\[0\]\), 0\.0f, ([0-9.]+)f\)", body)
    return [cs["c%d" % i] for i in range(7)], np.float32(clamp.group(1))


def _all_bf16():
    bits = np.arange(65536, dtype=np.uint32)
    with np.errstate(all="ignore"):
        x = (bits << 16).view(np.float32)
    return x[np.isfinite(x)]


def _bf16_round(v):
    u = np.asarray(v, dtype=np.float32).view(np.uint32).astype(np.uint64)
    u = (u + 0x7FFF + ((u >> 16) & 1)) >> 16
    return (u.astype(np.uint32) << 16).view(np.float32)


def _fma(a, b, c):   # one rounding: the product of two fp32 numbers is exact in float64
    return (a.astype(np.float64) * b.astype(np.float64) + np.float64(c)).astype(np.float32)


def gelu_lp(x, coef, clamp):
    x = x.astype(np.float32)
    a = np.minimum(np.abs(x), clamp)
    p = _fma(np.full_like(a, coef[6]), a, coef[5])
    for k in (4, 3, 2, 1, 0):
        p = _fma(p, a, coef[k])
    q = np.exp2(p.astype(np.float64)).astype(np.float32)
    s = (x + np.abs(x)).astype(np.float32)
    aq = (a * q).astype(np.float32)
    return (s.astype(np.float64) * 0.5 - aq.astype(np.float64)).astype(np.float32), q


def gelu_as(x):   # the Abramowitz-Stegun 7.1.26 form (option gelu_poly = 0), same style of restatement
    f = np.float32
    x = x.astype(f)
    t = (f(1) / _fma(np.abs(x), np.full_like(x, f(0.3275911 * 0.70710678118654752)), f(1))).astype(f)
    with np.errstate(over="ignore"):
        g = np.exp2((((x * f(-0.72134752044448170)).astype(f)) * x).astype(np.float64)).astype(f)
    p = _fma(np.full_like(x, f(0.5 * 1.061405429)), t, f(0.5 * -1.453152027))
    for c in (0.5 * 1.421413741, 0.5 * -0.284496736, 0.5 * 0.254829592):
        p = _fma(p, t, f(c))
    q = (((p * t).astype(f)) * g).astype(f)
    return (np.maximum(x, f(0)) - np.abs((x * q).astype(f))).astype(f)


def test_gelu_poly_against_erf_over_all_bf16_inputs():
    coef, clamp = _coefficients()
    assert clamp == np.float32(5.0)
    x = _all_bf16()
    exact = torch.nn.functional.gelu(torch.from_numpy(x.astype(np.float64))).numpy()   # erf form, float64
    with np.errstate(over="ignore"):
        y, q = gelu_lp(x, coef, clamp)
    # relu(x) is taken as 0.5 * (x + |x|): x + |x| overflows for the 129 bf16 values >= 2^127 (result +inf instead of x);
    # pre-activations of 1.7e38 do not exist in a network whose next operation is a bf16 GEMM
    big = x >= 2.0 ** 127
    assert np.all(np.isinf(y[big])) and big.sum() <= 130
    x, y, q, exact = x[~big], y[~big], q[~big], exact[~big]
    err = np.abs(y.astype(np.float64) - exact)
    # absolute error everywhere (the clamp included: beyond 5 the result is relu(x) - 5 q(5))
    assert err.max() < 2.5e-6, err.max()
    # q = Phi(-a) to 2e-5 RELATIVE over the fitted range, i.e. the negative tail keeps its relative accuracy
    sel = np.abs(x) <= clamp
    qe = 0.5 * torch.special.erfc(torch.from_numpy(np.abs(x[sel]).astype(np.float64)) / np.sqrt(2.0)).numpy()
    rel = np.abs(q[sel].astype(np.float64) - qe) / qe
    assert rel.max() < 2.5e-5, rel.max()
    # after the epilogue's bf16 rounding: at most a handful of inputs differ from the correctly rounded erf-GELU, by one bf16 ulp
    inside = (x > -clamp) & (np.abs(x) > 2.0 ** -30)
    yb, eb = _bf16_round(y), _bf16_round(exact)
    diff = inside & (yb != eb)
    ulps = np.abs(yb.view(np.uint32).astype(np.int64) - eb.view(np.uint32).astype(np.int64)) >> 16
    assert diff.sum() <= 8 and ulps[diff].max(initial=0) <= 1, (int(diff.sum()), int(ulps[diff].max(initial=0)))
    # ... fewer than the Abramowitz-Stegun form it replaces
    ab = _bf16_round(gelu_as(x))
    assert diff.sum() < (inside & (ab != eb)).sum()
    # below the clamp the result is within 1.5e-6 of the exact value (which is itself below 1.5e-6 in magnitude there)
    assert err[x <= -clamp].max() < 1.5e-6
    # exact identities the step relies on: gelu(0) = 0 and no NaN / inf for any finite input
    assert y[x == 0].tolist() == [0.0, 0.0] or np.all(y[x == 0] == 0)
    assert np.isfinite(y).all()


def test_gelu_poly_propagates_nan_of_either_sign():
    coef, clamp = _coefficients()
    x = np.array([np.float32("nan"), -np.float32("nan"), np.float32("inf")], dtype=np.float32)
    y, _ = gelu_lp(x, coef, clamp)
    assert np.isnan(y[0]) and np.isnan(y[1]) and y[2] == np.inf
