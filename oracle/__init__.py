"""CPU oracle for the MAC-VO per-frame hot path (TEST INFRASTRUCTURE — not product code).

This package is a CPU restatement (torch-CPU / numpy, float32 where the reference
computes in float32, float64 where it computes in float64) of the reference
algorithms listed in SURVEY.md §8(a).  Every function cites the reference
file:line it follows (paths relative to the MAC-VO checkout).

Who may import it: only ``tests/``, ``__graft_entry__.smoke()`` and the
``cpu_baseline`` leg of ``bench.py`` — always as the checker / the reported CPU
baseline, never as the thing that is shipped or measured as the product.  The
product path (``mac-vo_amd``) never imports this package and fails loudly when
the HIP library is missing.

Parity pinning status:

* keypoint selectors, MatchCovariance, gaussain_full_kernels, Covariance_2to3_full,
  disparity_to_depth(_cov), retrieve_pixels, filterPointsInRange:
  PINNED — checked against outputs of the real reference modules imported in the
  build container (``tests/golden/make_golden.py``; vectors in ``tests/golden/*.npz``).
* two-frame PGO residuals / analytic Jacobians / LM_analytic.step control flow:
  PINNED for the in-tree code (``Module/Optimization``) run against a minimal
  PyPose shim; the PyPose pieces themselves (Huber, FastTriggs, TrustRegion, PINV,
  StopOnPlateau, SE3 Exp/Act/Inv) are restated from the published PyPose 0.6.8
  algorithm because ``pypose`` is not installable here → "parity unpinned" for
  those pieces.
* FlowFormer all-pairs cost volume + 9x9 lookup: the submodule
  (MAC-VO/S_FlowFormer, un-pinned) is absent from the reference checkout →
  "parity unpinned"; pinned instead to the mathematical definitions
  (``einsum`` / ``grid_sample(align_corners=True, zeros)``) the upstream code calls.
"""
from . import se3, corr, frontend, selector, covariance, pgo, pipeline  # noqa: F401
