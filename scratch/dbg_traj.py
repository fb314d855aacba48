import sys
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tests')
import numpy as np, torch
import cases, nets
from conftest import load_golden, torus_rel_l2
import test_generator_gpu as T
cuda = torch.device('cuda:0')
name = 'traj_mlp_c1'
g = load_golden(name + '.npz')
gen, npar, spar, net_cpu = T._build(name, cases.TRAJECTORIES, cuda, fixture=g, record_samples=True, record_samples_corrector_steps=True)
gen.noise_source = T._replayed(g)
with torch.no_grad():
    out = gen.sample(int(g['batch']), cuda)
rec = gen.sample_trajectory_recorder._internal_data
for k, e in enumerate(rec['predictor_step']):
    dp = np.abs(e['model_predictions_i'].X.numpy() - g['pred_model_predictions_i_X'][k]).max()
    dci = torus_rel_l2(e['composition_i'].X.numpy(), g['pred_composition_i_X'][k])
    dc = torus_rel_l2(e['composition_im1'].X.numpy(), g['pred_composition_im1_X'][k])
    print('pred', k, e['time_step_index'], 'in', f'{dci:.2e}', 'netX', f'{dp:.2e}', 'out', f'{dc:.2e}')
    if k < len(rec['corrector_step']):
        c = rec['corrector_step'][k]
        dp = np.abs(c['model_predictions_i'].X.numpy() - g['corr_model_predictions_i_X'][k]).max()
        dc = torus_rel_l2(c['corrected_composition_i'].X.numpy(), g['corr_corrected_composition_i_X'][k])
        print('  corr', c['time_step_index'], 'netX', f'{dp:.2e}', 'out', f'{dc:.2e}')
