"""Files crossing between the two packages, on the REFERENCE's side (container-only; run with the reference on PYTHONPATH and
this package NOT importable):  python cross_pickles.py SAMPLES_FOR_REFERENCE.pt  START_PICKLE_OUT.pt  TRAJECTORIES_FOR_REFERENCE.pt
  * reads a `samples.pt` and a `trajectories.pt` this package wrote with --reference_pickles by a PLAIN torch.load and checks they
    hold the reference's AXL (and, for the recorder's schedule tables, the reference's Noise);
  * writes a starting-configuration pickle the way the reference's tools do (generators/trajectory_initializer.py:151-161:
    {"noisy_axl": AXL, "start_time_step_index": int}) for this package to read."""
import json
import sys

import torch
from diffusion_for_multi_scale_molecular_dynamics.namespace import AXL, AXL_COMPOSITION, NOISY_AXL_COMPOSITION

assert not any(name.startswith("diffusion_for_multi_scale_molecular_dynamics_amd") for name in sys.modules)
samples = torch.load(sys.argv[1], weights_only=False)
axl = samples[AXL_COMPOSITION]
assert type(axl) is AXL, type(axl)
from diffusion_for_multi_scale_molecular_dynamics.noise_schedulers.noise_scheduler import Noise
trajectories = torch.load(sys.argv[3], weights_only=False)
assert type(trajectories["noise"]) is Noise and type(trajectories["predictor_step"]["composition_i"]) is AXL
g = torch.Generator().manual_seed(99)
start = AXL(A=torch.randint(0, 2, (3, 8), generator=g), X=torch.rand(3, 8, 3, generator=g), L=torch.rand(3, 6, generator=g))
torch.save({NOISY_AXL_COMPOSITION: start, "start_time_step_index": 4}, sys.argv[2])
print(json.dumps(dict(samples_class=f"{type(axl).__module__}.{type(axl).__name__}", x_sum=float(axl.X.double().sum()),
                      cartesian_shape=list(samples["cartesian_positions"].shape), start_x_sum=float(start.X.double().sum()))))
