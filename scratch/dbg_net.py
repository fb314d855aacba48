import sys, os
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tests')
import numpy as np, torch
import nets
from conftest import load_golden
from diffusion_for_multi_scale_molecular_dynamics_amd.namespace import *
print(torch.__version__, torch.backends.cuda.matmul.allow_tf32, torch.get_float32_matmul_precision(), os.environ.get('TORCH_BLAS_PREFER_HIPBLASLT'))
try:
    print('preferred blas', torch.backends.cuda.preferred_blas_library())
except Exception as e: print(e)
g = load_golden('net_mlp_c1.npz')
net = nets.load_fixture_weights(nets.mlp_net(8,1), g)
def batch(dev):
    return {NOISY_AXL_COMPOSITION: AXL(A=torch.from_numpy(g['A']).to(dev),X=torch.from_numpy(g['X']).to(dev),L=torch.from_numpy(g['L']).to(dev)), TIME: torch.from_numpy(g['time']).to(dev), NOISE: torch.from_numpy(g['noise']).to(dev), CARTESIAN_FORCES: torch.zeros(g['X'].shape).to(dev)}
with torch.no_grad():
    o_cpu = net(batch('cpu'), conditional=False)
    net_g = net.to('cuda')
    o_gpu = net_g(batch('cuda'), conditional=False)
print('X maxabs diff', (o_gpu.X.cpu()-o_cpu.X).abs().max().item(), 'scale', o_cpu.X.abs().max().item())
a = torch.randn(64, 256); b = torch.randn(256, 64)
ref = (a.double() @ b.double())
print('cpu mm err', ((a@b).double()-ref).abs().max().item(), 'gpu mm err', ((a.cuda()@b.cuda()).cpu().double()-ref).abs().max().item())
x = torch.rand(1000)*6.28
print('cos err gpu', (x.cuda().cos().cpu().double()-x.double().cos()).abs().max().item(), 'cpu', (x.cos().double()-x.double().cos()).abs().max().item())
lin = torch.nn.Linear(48, 32)
xin = torch.randn(6,48)
with torch.no_grad():
    print('linear diff', (lin(xin) - lin.cuda()(xin.cuda()).cpu()).abs().max().item())
    s = torch.nn.SiLU(); print('silu diff', (s(xin)-s(xin.cuda()).cpu()).abs().max().item())
