"""CPU oracle for the SD denoising hot path.

TEST INFRASTRUCTURE ONLY.  Nothing in ``pytorch_stable_diffusion_amd/`` may import this
package; only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline``
leg use it, as the checker / reported CPU baseline, never as the product path.

It restates, in plain functional PyTorch-CPU fp32, the algorithm of the reference's
sd/pipeline.py loop, sd/ddpm.py, sd/diffusion.py and sd/attention.py (each function cites
the reference lines it follows).  It is pinned against golden vectors produced by importing
the reference itself in the build container (tests/golden/make_golden.py ->
tests/golden/*.npz; tests/test_oracle_golden.py).
"""
