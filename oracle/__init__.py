"""CPU oracle for the DispNet training hot path -- TEST INFRASTRUCTURE ONLY.

This package is a plain PyTorch-CPU / numpy restatement of the reference
algorithms on the hot path (zenithfang/supervised_dispnet: models/, loss_functions.py,
inverse_warp.py, layers.py::SSIM, utils.py SID helpers, kitti_eval GT generation).
Every function cites the reference file:line it follows.

Rules (enforced by tests/test_layout.py):
  * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import it.
  * The product package (supervised_dispnet_amd/) never imports, calls or links it.
  * It is the checker, never the thing measured or shipped.

Parity pin: the reference publishes no tests / golden vectors for this path
(SURVEY.md section 4), so the oracle is pinned against outputs of the reference itself,
generated in the build container by tests/golden/make_goldens.py (which imports
/root/reference through a small shim loader) and committed as tests/golden/*.npz.
tests/test_oracle_golden.py checks every oracle function against those vectors.
"""

from . import detgen  # noqa: F401
