"""MI355X-native sampling hot path for periodic atomistic diffusion models.

Drop-in for the predictor-corrector / Langevin sampling loop of
mila-iqia/diffusion_for_multi_scale_molecular_dynamics: same score-network / generator plugin API and YAML
surface, per-step work in hand-written gfx950 HIP kernels behind the C ABI of include/mdx_hip.h.
"""
__version__ = "0.1.0"
