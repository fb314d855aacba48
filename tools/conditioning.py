import sys
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tests')
import numpy as np, torch
import cases, nets
from conftest import load_golden, torus_rel_l2
from oracle import reference_sampler as RS
for name in ['traj_mlp_c1','traj_mlp_c3','traj_egnn_fc','traj_egnn_rc']:
    g = load_golden(name + '.npz')
    noise_kw, sampling_kw, netf = cases.TRAJECTORIES[name]
    npar, spar = cases.as_objects(noise_kw, sampling_kw)
    net = nets.load_fixture_weights(netf(nets.oracle_edge_builder), g)
    outs=[]
    for pert in (0.0, 1e-7):
        class P(RS.ReplayNoise):
            first=True
            def rand(self,*s):
                r=super().rand(*s)
                if P.first and pert:
                    r=(r+np.float32(pert)).astype(np.float32); 
                P.first=False
                return r
        P.first=True
        gen = RS.OracleLangevinGenerator(npar, spar, net, noise=P(g))
        outs.append(gen.sample(int(g['batch'])))
    print(name, 'final deviation from a 1e-7 perturbation of X0:', f"{torus_rel_l2(outs[1].X, outs[0].X):.2e}", 'A equal', np.array_equal(outs[0].A, outs[1].A))
