"""CPU oracle for the ddpm-torch hot path — TEST INFRASTRUCTURE, NOT PRODUCT.

Plain-PyTorch (fp32/fp64, CPU) restatement of the reference algorithm for the
path SURVEY.md §8 names: UNet forward/backward, GaussianDiffusion / DDIM tables
and step algebra, Trainer.step / EMA.  Every function cites the reference
file:line it follows (paths relative to the upstream repo root).

Who may import this package: ``tests/``, ``__graft_entry__.smoke()`` and the
``cpu_baseline`` leg of ``bench.py`` — as the CHECKER only.  Nothing under
``ddpm-torch_amd/`` imports it; the product path raises when the HIP library
is missing instead of falling back to this code.

Pinning: the upstream repo ships no tests or golden vectors, so
the oracle is pinned against outputs of the reference itself, generated in the
build container by ``tests/golden/make_golden.py`` (which imports the reference
from /root/reference through a stub package) and committed as fixtures under
``tests/golden/``.  ``tests/test_oracle_golden.py`` checks oracle == fixtures.
"""
