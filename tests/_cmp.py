"""Comparison helper for tables that may hold the reference's own NaNs (the zero-length stance spline of the window's first phase,
SwingTrajectoryPlanner.cpp:253-276): the NaN PATTERN must be identical, the finite entries are compared by value."""
import numpy as np


def maxdiff_nan(got, want) -> float:
    got, want = np.asarray(got, dtype=float), np.asarray(want, dtype=float)
    assert got.shape == want.shape, (got.shape, want.shape)
    gn, wn = np.isnan(got), np.isnan(want)
    assert np.array_equal(gn, wn), f"NaN pattern differs: {int(gn.sum())} vs {int(wn.sum())} NaNs"
    ok = ~wn
    return float(np.abs(got[ok] - want[ok]).max()) if ok.any() else 0.0
