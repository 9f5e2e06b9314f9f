"""Device form of user callables: a verified Chebyshev series.

The reference lets a scene carry arbitrary Python functions — `ASphericLens(f_asphere_1=callable)` (the sag F(r),
component_group.py:1014-1055) and `Material(name, n=callable)` (n(wavelength in metres), material.py:4-21).  The kernels
cannot call Python, so the scene compiler samples such a function on its domain, fits a Chebyshev series and VERIFIES it
against the callable on points the fit has not seen; a function the series does not reproduce to `rel_tol` of its range
(a kink, noise, a singularity) is refused with the reason — there is no host fallback.  The device evaluates the series
by Clenshaw's recurrence (trace_core.h cheb_eval); the oracle reads the same table, so both restate
"the reference evaluates F" with F replaced by a series that agrees with it to the last two or three digits.
"""
import numpy as np

DEGREES = (16, 24, 32, 48, 64, 96, 128)
REL_TOL = 5e-14


class FitError(ValueError):
    pass


def _sample(func, x):
    out = np.empty(len(x))
    for k, v in enumerate(x):  # user callables are not required to take arrays
        out[k] = float(func(float(v)))
    return out


def fit(func, lo, hi, rel_tol=REL_TOL, what="function"):
    """Chebyshev coefficients c[0..N) of `func` on [lo, hi] such that sum_k c_k T_k(t), t = (2x - lo - hi) / (hi - lo),
    reproduces it to rel_tol * max|func| everywhere on the interval (checked on 4N + 57 points between the nodes).
    Returns (coefficients, measured max abs error).  Raises FitError when no series of up to 128 terms does."""
    lo, hi = float(lo), float(hi)
    if not hi > lo:
        raise FitError(f"{what}: empty fit interval [{lo}, {hi}]")
    mid, half = 0.5 * (lo + hi), 0.5 * (hi - lo)
    best = None
    for n in DEGREES:
        nodes = np.cos(np.pi * (np.arange(n) + 0.5) / n)
        try:
            vals = _sample(func, mid + half * nodes)
        except Exception as exc:  # noqa: BLE001 - the callable's own failure is the message
            raise FitError(f"{what}: the callable failed on [{lo}, {hi}]: {exc}") from exc
        if not np.all(np.isfinite(vals)):
            raise FitError(f"{what}: the callable is not finite everywhere on [{lo:.6g}, {hi:.6g}]")
        coef = np.polynomial.chebyshev.chebfit(nodes, vals, n - 1)
        scale = max(float(np.abs(vals).max()), 1e-300)
        # drop the tail that is rounding noise: shorter series, same accuracy
        keep = n
        while keep > 2 and abs(coef[keep - 1]) <= 0.05 * rel_tol * scale:
            keep -= 1
        coef = coef[:keep]
        probe = np.cos(np.pi * (np.arange(4 * n + 57) + 0.31) / (4 * n + 57))
        err = float(np.abs(np.polynomial.chebyshev.chebval(probe, coef) - _sample(func, mid + half * probe)).max())
        if best is None or err < best[1]:
            best = (coef, err, scale)
        if err <= rel_tol * scale:
            return coef, err
    coef, err, scale = best
    raise FitError(f"{what}: no Chebyshev series of up to {DEGREES[-1]} terms reproduces the callable on [{lo:.6g}, {hi:.6g}] "
                   f"(best: {err / scale:.2e} of its range, needed {rel_tol:.0e}): not smooth enough for a device form")


def derivative(coef, lo, hi):
    """Coefficients of d/dx of the series (same interval)."""
    if len(coef) < 2:
        return np.zeros(1)
    return np.polynomial.chebyshev.chebder(coef) * (2.0 / (hi - lo))


def evaluate(coef, lo, hi, x):
    """Host evaluation (tests): Clenshaw, as the device does it."""
    t = (2.0 * np.asarray(x, dtype=float) - (lo + hi)) / (hi - lo)
    b1 = np.zeros_like(t)
    b2 = np.zeros_like(t)
    for c in coef[:0:-1]:
        b1, b2 = c + 2.0 * t * b1 - b2, b1
    return coef[0] + t * b1 - b2


def record(coef, lo, hi, derivatives=0):
    """aux-table record: [N, lo, hi, c[N]] followed by `derivatives` more blocks of N coefficients (d/dx, d2/dx2),
    zero-padded to N."""
    n = len(coef)
    out = [float(n), float(lo), float(hi)] + [float(c) for c in coef]
    cur = np.asarray(coef, dtype=float)
    for _ in range(derivatives):
        cur = derivative(cur, lo, hi)
        out += [float(c) for c in cur] + [0.0] * (n - len(cur))
    return out
