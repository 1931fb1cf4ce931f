"""gp-nerf_amd: MI355X-native implementation of GP-NeRF's per-ray rendering hot path.

Host side mirrors the reference's Python interface for this path (same names, arguments, return
structures): `projection.Projector`, `render_ray.render_rays` (+ samplers), `render_image.
render_single_image`, `transformer_network.GNT`, `semantic_branch.NeRFSemSegFPNHead`.  All compute
runs in hand-written gfx950 HIP kernels behind the C-ABI library `csrc/libgpnerf_hip.so`
(declared in `include/gpnerf_hip.h`).  There is no CPU fallback: importing works without a GPU,
but every op raises if the library is missing or a tensor is not on a HIP device.
"""
__version__ = "0.1.0"
