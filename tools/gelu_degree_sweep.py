#!/usr/bin/env python
"""CPU: how many bf16 inputs the GELU epilogue would round differently from the correctly rounded erf-GELU if the polynomial of
log2 Phi(-a) (csrc/common.hpp half_erfc2_lp, degree 6) had a lower degree.  Uses the arithmetic restatement and the input mask of
tests/test_gelu_poly.py; the polynomials are minimax fits by Lawson's iteration (plain, and weighted by a * Phi(-a), the term the
polynomial feeds).  Round-6 record: profiles/r06_gelu_degree_sweep.md.       python tools/gelu_degree_sweep.py"""
import os
import sys

import numpy as np
import torch
from scipy.special import log_ndtr

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
import test_gelu_poly as T  # noqa: E402

x = T._all_bf16()
exact = torch.nn.functional.gelu(torch.from_numpy(x.astype(np.float64))).numpy()
ref = T._bf16_round(exact.astype(np.float32))
coef, clamp = T._coefficients()


def count(cf):
    with np.errstate(over="ignore"):
        y, _ = T.gelu_lp(x, [np.float32(c) for c in cf] + [np.float32(0)] * (7 - len(cf)), clamp)
    m = (x > -clamp) & (np.abs(x) > 2.0 ** -30) & (x < 2.0 ** 127)
    return int((T._bf16_round(y)[m] != ref[m]).sum()), int(m.sum())


def lawson(V, f, g, iters=200):
    w = np.ones(len(f))
    for _ in range(iters):
        ww = (w * g * g + 1e-30) ** 0.5
        c = np.linalg.lstsq(V * ww[:, None], f * ww, rcond=None)[0]
        e = np.abs(V @ c - f) * g
        w = w * (e / e.max() + 1e-12)
        w /= w.sum()
    return c, e.max()


print("shipped degree 6: %d of %d" % count(coef))
a = np.linspace(0, float(clamp), 20001)
f = log_ndtr(-a) / np.log(2)
for deg in (6, 5, 4, 3):
    V = np.vander(a, deg + 1, increasing=True)
    c, e = lawson(V, f, np.ones_like(a))
    cw, _ = lawson(V, f, a * np.exp(log_ndtr(-a)))
    print("degree %d: max |dL| %.2e; mismatches plain fit %d, weighted fit %d (of %d)" % (deg, e, count(c)[0], *count(cw)))
