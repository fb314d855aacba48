"""MLP score network (plugin of the ScoreNetwork API; PyTorch forward).

Same hyper-parameters, parameter names (state_dict keys) and function as the reference's MLPScoreNetwork
(src/.../models/score_networks/mlp_score_network.py:18-370), so reference checkpoints load with
`load_state_dict`.  The forward is written for the GPU: the (cos, sin) embedding, the one-hot atom-type
embedding and the input concatenation are laid out to be a handful of GEMMs with no host synchronisation.
"""
import itertools
import math
from dataclasses import dataclass
from typing import Any, AnyStr, Dict

import torch
from torch import nn

from ...namespace import AXL, CARTESIAN_FORCES, NOISE, NOISY_AXL_COMPOSITION, TIME
from .score_network import ScoreNetwork, ScoreNetworkParameters


@dataclass(kw_only=True)
class MLPScoreNetworkParameters(ScoreNetworkParameters):
    """Hyper-parameters (:18-51)."""

    architecture: str = "mlp"
    number_of_atoms: int
    n_hidden_dimensions: int
    hidden_dimensions_size: int
    noise_embedding_dimensions_size: int
    relative_coordinates_embedding_dimensions_size: int
    time_embedding_dimensions_size: int
    atom_type_embedding_dimensions_size: int
    lattice_parameters_embedding_dimensions_size: int
    condition_embedding_size: int = 64
    use_time_dependent_prefactor: bool = False
    use_permutation_invariance: bool = False


class MLPScoreNetwork(ScoreNetwork):
    """Flattened-configuration MLP: inputs (cos 2 pi x, sin 2 pi x), sigma, t, one-hot(a), lattice."""

    def __init__(self, hyper_params: MLPScoreNetworkParameters):
        super().__init__(hyper_params)
        hp = hyper_params
        n, d = hp.number_of_atoms, self.spatial_dimension
        self._natoms = n
        self.num_classes = self.num_atom_types + 1
        self.use_time_dependent_prefactor = hp.use_time_dependent_prefactor
        self.use_permutation_invariance = hp.use_permutation_invariance
        if self.use_permutation_invariance:
            perms = torch.tensor(list(itertools.permutations(range(n))))
            self.perm_indices = perms
            self.inverse_perm_indices = perms.argsort(dim=1)
        n_lattice = d * (d + 1) // 2
        hidden = [hp.hidden_dimensions_size] * hp.n_hidden_dimensions

        self.relative_coordinates_embedding_layer = nn.Linear(2 * d * n, hp.relative_coordinates_embedding_dimensions_size)
        self.noise_embedding_layer = nn.Linear(1, hp.noise_embedding_dimensions_size)
        self.time_embedding_layer = nn.Linear(1, hp.time_embedding_dimensions_size)
        self.atom_type_embedding_layer = nn.Linear(self.num_classes, hp.atom_type_embedding_dimensions_size)
        self.lattice_parameters_embedding_layer = nn.Linear(n_lattice, hp.lattice_parameters_embedding_dimensions_size)
        self.condition_embedding_layer = nn.Linear(d * n, hp.condition_embedding_size)

        first = (hp.relative_coordinates_embedding_dimensions_size + hp.noise_embedding_dimensions_size
                 + hp.time_embedding_dimensions_size + n * hp.atom_type_embedding_dimensions_size
                 + hp.lattice_parameters_embedding_dimensions_size)
        self.flatten = nn.Flatten()
        self.mlp_layers = nn.ModuleList()
        self.conditional_layers = nn.ModuleList()
        for n_in, n_out in zip([first] + hidden[:-1], hidden):
            self.mlp_layers.append(nn.Linear(n_in, n_out))
            self.conditional_layers.append(nn.Linear(hp.condition_embedding_size, n_out))
        self.non_linearity = nn.SiLU()

        if self.use_time_dependent_prefactor:
            dims_in = [hp.noise_embedding_dimensions_size + hp.time_embedding_dimensions_size] + hidden
            dims_out = hidden + [1]
            layers = []
            for k, (a, b) in enumerate(zip(dims_in, dims_out)):
                if k > 0:
                    layers.append(self.non_linearity)
                layers.append(nn.Linear(a, b))
            self.prefactor_mlp = nn.Sequential(*layers)

        self.output_A_layer = nn.Linear(hp.hidden_dimensions_size, n * self.num_classes)
        self.output_X_layer = nn.Linear(hp.hidden_dimensions_size, d * n)
        self.output_L_layer = nn.Linear(hp.hidden_dimensions_size, n_lattice)
        self.output_layers = AXL(A=self.output_A_layer, X=self.output_X_layer, L=self.output_L_layer)

    def _check_batch(self, batch: Dict[AnyStr, torch.Tensor]):
        super()._check_batch(batch)
        assert batch[NOISY_AXL_COMPOSITION].X.shape[1] == self._natoms, \
            "The dimension corresponding to the number of atoms is not consistent with the configuration."

    def _forward_unchecked(self, batch: Dict[AnyStr, torch.Tensor], conditional: bool = False) -> AXL:
        if not self.use_permutation_invariance:
            return self._single(batch[NOISY_AXL_COMPOSITION], batch, conditional)
        # s_sym(x) = 1/|G| sum_g g^-1 . s(g . x)   (:228-262)
        comp = batch[NOISY_AXL_COMPOSITION]
        outs = []
        for perm, inv in zip(self.perm_indices, self.inverse_perm_indices):
            o = self._single(AXL(A=comp.A[:, perm], X=comp.X[:, perm], L=comp.L), batch, conditional)
            outs.append(AXL(A=o.A, X=o.X[:, inv], L=o.L))
        return AXL(A=torch.stack([o.A for o in outs]).mean(dim=0), X=torch.stack([o.X for o in outs]).mean(dim=0),
                   L=torch.stack([o.L for o in outs]).mean(dim=0))

    def get_permuted_batch(self, batch, permutation) -> Dict[AnyStr, Any]:
        """The batch with the atoms of its composition re-ordered by `permutation` (atom types and coordinates; the lattice and
        every other entry as they are) (:250-267)."""
        comp = batch[NOISY_AXL_COMPOSITION]
        permuted = dict(batch)
        permuted[NOISY_AXL_COMPOSITION] = AXL(A=comp.A[:, permutation], X=comp.X[:, permutation], L=comp.L)
        return permuted

    def _single(self, comp: AXL, batch, conditional: bool) -> AXL:
        x = comp.X
        bsz = x.shape[0]
        angles = (2.0 * math.pi) * x
        # [B, 2, N, d] flattened: all cosines, then all sines (:299-305)
        circle = torch.stack([angles.cos(), angles.sin()], dim=1).reshape(bsz, -1)
        sigmas = batch[NOISE].to(x.device)
        times = batch[TIME].to(x.device)
        noise_emb = self.noise_embedding_layer(sigmas)
        time_emb = self.time_embedding_layer(times)
        one_hot = torch.nn.functional.one_hot(comp.A.long(), num_classes=self.num_classes).to(x.dtype)
        features = torch.cat([
            self.relative_coordinates_embedding_layer(circle),
            noise_emb,
            time_emb,
            self.atom_type_embedding_layer(one_hot).reshape(bsz, -1),
            self.lattice_parameters_embedding_layer(comp.L),
        ], dim=1)
        forces = self.condition_embedding_layer(batch[CARTESIAN_FORCES].reshape(bsz, -1)) if conditional else None
        h = features
        for k, (layer, cond) in enumerate(zip(self.mlp_layers, self.conditional_layers)):
            if k:
                h = self.non_linearity(h)
            h = layer(h)
            if conditional:
                h = h + cond(forces)
        out_x = self.output_X_layer(h).reshape(x.shape)
        if self.use_time_dependent_prefactor:
            out_x = self.prefactor_mlp(torch.cat([noise_emb, time_emb], dim=1)).unsqueeze(-1) * out_x
        out_a = self.output_A_layer(h).reshape(bsz, self._natoms, self.num_classes)
        return AXL(A=out_a, X=out_x, L=self.output_L_layer(h))
