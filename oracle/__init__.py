"""CPU oracle for the GP-NeRF per-ray rendering hot path.

TEST INFRASTRUCTURE ONLY.  This package is a CPU (torch-CPU fp32 / numpy) restatement of the
reference algorithm, function by function, each citing the reference file:line it follows.  Only
`tests/`, `__graft_entry__.smoke()` and the `cpu_baseline` leg of `bench.py` may import it, and
there only as the checker or as the timed CPU baseline -- never as part of the product path.
The product (`gp-nerf_amd/`) never imports this package and fails loudly without its HIP library.

Pinning: the oracle is checked against golden vectors produced by importing the *reference itself*
(`/root/reference/gpnerf/{projection,render_ray,render_image,transformer_network}.py`) in the build
container; the generating script and the vectors live in `tests/golden/`.  The one exception is the
semantic head (`sem_head.py`): the reference module needs detectron2/fvcore, which are absent, and
the reference ships no test or fixture for it -> that function is "parity unpinned".
"""
from . import encoders, gnt, projector, render, sample_ray, sampling, sem_head  # noqa: F401
