"""Eager launches of the fused predictor / corrector update at the C2 size, for rocprofv3 --pmc passes."""
import sys
sys.path.insert(0, '/root/repo')
import torch
from diffusion_for_multi_scale_molecular_dynamics_amd import kernels
from diffusion_for_multi_scale_molecular_dynamics_amd._hip import MDX_CORRECTOR, MDX_PREDICTOR, PcFlags, Rng
dev = torch.device('cuda:0')
B, N, C = 1024, 8, 2
sched = kernels.noise_schedule_build(1000, "exponential", 1e-5, 1e-4, 0.25, 2e-5, C, dev)
a = torch.full((B, N), C - 1, dtype=torch.int64, device=dev); x = torch.rand(B, N, 3, device=dev)
lat = torch.tensor([5.43] * 3 + [0.0] * 3, device=dev).repeat(B, 1)
logits = torch.randn(B, N, C, device=dev); logits[..., -1] = -torch.inf
score = torch.randn(B, N, 3, device=dev); a_out, x_out = torch.empty_like(a), torch.empty_like(x)
st = torch.zeros(1, dtype=torch.int32, device=dev)
for _ in range(25):
    kernels.pc_step_update(sched, MDX_PREDICTOR, 500, None, PcFlags(1, 1, 1, 1, 1e-8), a, x, lat, logits, score, None, None, None, None, None, Rng(1, 0, 2, 0), a_out, x_out, lat, st)
    kernels.pc_step_update(sched, MDX_CORRECTOR, 499, None, PcFlags(1, 1, 1, 0, 1e-8), None, x, lat, None, score, None, None, None, None, None, Rng(1, 0, 2, 1), None, x_out, lat, st)
torch.cuda.synchronize()
