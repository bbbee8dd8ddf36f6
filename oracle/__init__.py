"""CPU oracle for the trajectory hot path -- TEST INFRASTRUCTURE, not product.

Every function here restates, on plain torch-CPU / numpy ops, the algorithm of
one reference function on the hot path and cites the
reference file:line it follows.  It exists so that

  * ``tests/`` can compare the hand-written HIP path with a known-good answer,
  * ``__graft_entry__.smoke()`` can check one small invocation on ``cuda:0``,
  * ``bench.py`` can time a CPU baseline (``cpu_baseline.kind == "port"``)
    beside the GPU number.

Nothing under ``distillation_trajectories_amd/`` imports this package: the
product path has no CPU fallback and fails loudly without the HIP library.

Pinning: the reference's own tests hold no numeric expectations,
so the oracle is pinned against outputs of the reference itself, captured in the
build container by ``tests/golden/make_golden.py`` (which imports the reference
from ``/root/reference`` with stub modules for the absent torchvision/umap) and
committed as ``tests/golden/*.npz``.  ``tests/test_oracle_golden.py`` checks the
oracle against those vectors bit-for-bit (same torch build, same CPU kernels).
"""
