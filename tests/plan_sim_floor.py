"""How the plan-weighted similarity (caching_score's ranking key, disent_models.py:297-304) is held to the reference.

sum_ij exp((f_i + g_j - d_ij) / blur) a_i b_j d_ij with |f|, |g|, |d| ~ 38 and blur = 0.05 turns one fp32 ulp of an exponent
into 8e-5 relative of a plan entry: the reference's OWN fp32 CPU path is several 1e-3 away from the same arithmetic in float64
(test_gpu_scoring.py).  A blanket tolerance against the fp32 oracle would have to be 100 x the north star's 1e-4; instead the
HIP path is measured against the FLOAT64 oracle and must be no further from it than the reference's fp32 path is on the same
inputs (FACTOR x, with a floor for cases where that path happens to land within rounding), and two rankings may differ only
between candidates whose float64 scores are closer than that per-case bound."""
import numpy as np

FACTOR = 1.5
FLOOR = 1e-3


def bound(want32, truth64):
    return max(FACTOR * float(np.abs(np.asarray(want32, dtype=np.float64) - np.asarray(truth64, dtype=np.float64)).max()), FLOOR)


def check(got, want32, truth64, what=''):
    """max |got - float64 oracle| <= FACTOR x max |fp32 oracle - float64 oracle| (>= FLOOR).  Returns the bound."""
    b = bound(want32, truth64)
    err = float(np.abs(np.asarray(got, dtype=np.float64) - np.asarray(truth64, dtype=np.float64)).max())
    assert err <= b, f'{what}: HIP path {err:.2e} from the float64 oracle, the fp32 CPU oracle {b / FACTOR:.2e} (bound {b:.2e})'
    return b


def order_agrees(got_order, truth64, b):
    """got_order (indices, best first) against the float64 scores: positions may differ only between candidates whose float64
    scores are within 2 b of each other (each side is within b of the truth)."""
    truth64 = np.asarray(truth64, dtype=np.float64)
    want_order = sorted(range(len(truth64)), key=lambda i: truth64[i], reverse=True)
    for a, w in zip(got_order, want_order):
        if a != w and abs(truth64[a] - truth64[w]) > 2 * b:
            return False
    return True
