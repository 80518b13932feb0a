"""oracle/ -- TEST INFRASTRUCTURE ONLY.

A CPU restatement (plain PyTorch fp32 + numpy) of the reference's DLS_VAE hot path
(oklen/PEVAE: models/enc2dec/*.py, models/nn_lib.py, models/criterions.py and the
hot-path parts of models/dialog_models.py).  Every function cites the reference
file:line it follows.

Who may import this package: `tests/`, `__graft_entry__.smoke()` and the
`cpu_baseline` leg of `bench.py` -- as the checker / the reported CPU baseline,
never as the thing that is measured or shipped.  Nothing under `pevae_amd/`
imports it; the product path raises when the HIP library is missing.

Pinning status: the reference holds NO tests, golden vectors or fixtures for this
path, and its arithmetic lives in PyTorch itself.  The oracle
is therefore pinned by golden vectors generated in the build container by running
the reference's own module classes (`oracle/make_golden.py`, which imports
/root/reference with two in-memory stubs for `nltk` and `numpy.lib.type_check`);
the vectors are committed under `tests/golden/` and `tests/test_oracle_golden.py`
checks the restatement against every one of them.  "parity unpinned by the
reference's own tests; pinned by reference-generated goldens".
"""
