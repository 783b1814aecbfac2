"""CPU oracle for the DiffSinger diffusion-denoiser hot path.

TEST INFRASTRUCTURE ONLY.  This package is a plain numpy (fp32) restatement of
the reference algorithms (hrukalive/DiffSinger, `modules/backbones/*`,
`modules/core/{ddpm,reflow}.py`, `inference/{dpm_solver_pytorch,uni_pc}.py`).
Only `tests/`, `__graft_entry__.smoke()` and the `cpu_baseline` leg of
`bench.py` may import it - and there only as the checker, never as the thing
that is shipped or measured.  The product (`diffsinger_amd`) never imports
it and fails loudly when the HIP library is missing.

Parity pin: the reference has no tests and no golden vectors of its own
(SURVEY.md section 4), so the oracle is pinned by fixtures generated from the
imported reference itself in the build container - `tests/golden/*.npz`, made
by `tests/golden/make_golden.py` - and checked in `tests/test_oracle_golden.py`.
"""
