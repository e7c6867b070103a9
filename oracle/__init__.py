"""CPU oracle: a restatement of the reference DiffSound hot path (TEST INFRASTRUCTURE ONLY).

This package restates, in plain NumPy / PyTorch-CPU / SciPy, the algorithm of the
reference path  tet-FEM K/M assembly -> generalised eigensolve -> damped-oscillator
bank (forward + backward).  Every function cites the reference file:line it follows.

Rules (see the task brief, section (3)):
  * only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline``
    leg may import this package, and only as the *checker* / reported CPU baseline;
  * the product package ``diffsound_amd`` never imports it and has no CPU fallback;
  * parity pinning: the reference ships NO tests or golden vectors for this path
    (SURVEY.md §4), so the oracle is pinned against outputs of the reference itself,
    generated in the build container by ``tests/golden/make_golden.py`` (which
    imports /root/reference with three harness shims) and committed as ``.npz``
    fixtures under ``tests/golden/``; ``tests/test_oracle_golden.py`` checks every one.

Third-party arithmetic the reference delegates to and which is therefore re-used
(not restated) here: ``scipy.sparse.linalg.eigsh`` (ARPACK shift-invert,
reference pin scipy==1.10.1, call site src/diffelastic/diff_model.py:356-358) and
``torch`` dense/sparse linear algebra.
"""
