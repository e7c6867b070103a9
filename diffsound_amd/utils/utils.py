"""Glue helpers on the hot path - mirror of reference src/utils/utils.py:80-90 (the audio I/O,
plotting and COMSOL helpers of that file are outside the hot path and not provided)."""
from ..lobpcg import lobpcg_func


def LOBPCG_solver_freq(stiff_matrix, mass_matrix, niter=1000, freq_limit=None, k=100):
    """k+6 lowest pairs of (K, M), optional frequency cut, the first six (rigid) dropped
    (reference utils.py:80-90)."""
    vals, vecs = lobpcg_func(stiff_matrix, mass_matrix, k + 6, niter=niter, tracker=None, largest=False)
    if freq_limit:
        eigenvalue_limit = (freq_limit * 2 * 3.14159) ** 2
        mask = vals < eigenvalue_limit
        vals = vals[mask]
        vecs = vecs[:, mask]
    return vals[6:], vecs[:, 6:]
