"""EGNN score network (plugin of the ScoreNetwork API; PyTorch forward, HIP radius graph).

Same hyper-parameters, parameter names and function as the reference's EGNNScoreNetwork
(src/.../models/score_networks/egnn_score_network.py:23-303).  With `edges: radial_cutoff` the graph is built by
the HIP kernel N1 (utils/neighbors.py), including the reference's quirk of searching in a cell clipped to
2.2 x cutoff with the angles zeroed (:236-240).
"""
import itertools
import math
from dataclasses import dataclass
from typing import AnyStr, Callable, Dict, Optional, Union

import torch

from ...namespace import AXL, NOISE, NOISY_AXL_COMPOSITION
from ...utils import neighbors
from ..egnn import EGNN
from .score_network import ScoreNetwork, ScoreNetworkParameters


@dataclass(kw_only=True)
class EGNNScoreNetworkParameters(ScoreNetworkParameters):
    """Hyper-parameters (:23-45)."""

    architecture: str = "egnn"
    number_of_bloch_wave_shells: int = 1
    message_n_hidden_dimensions: int = 1
    message_hidden_dimensions_size: int = 16
    node_n_hidden_dimensions: int = 1
    node_hidden_dimensions_size: int = 32
    coordinate_n_hidden_dimensions: int = 1
    coordinate_hidden_dimensions_size: int = 32
    residual: bool = True
    attention: bool = False
    normalize: bool = False
    tanh: bool = False
    coords_agg: str = "mean"
    message_agg: str = "mean"
    n_layers: int = 4
    edges: str = "fully_connected"
    radial_cutoff: Union[float, None] = None
    drop_duplicate_edges: bool = True


def complete_lattice_shells(number_of_complete_shells: int, spatial_dimension: int):
    """The first shells of integer lattice vectors under the cubic point group, zero excluded: shells by increasing |K|^2 (when
    the last wanted shell shares its length with others, those are kept too), the vectors of a shell in descending lexicographic
    order -- the set and order of get_cubic_point_group_complete_lattice_shells (src/.../utils/lattice_utils.py:66-126)."""
    n = 2 * number_of_complete_shells
    vectors = [v for v in itertools.product(range(-n, n + 1), repeat=spatial_dimension) if any(v)]
    vectors.sort(key=lambda v: (sum(c * c for c in v), tuple(-c for c in v)))
    seen, shells, previous_norm = set(), [], 0
    for v in vectors:
        if v in seen:
            continue
        orbit = set()
        for perm in itertools.permutations(range(spatial_dimension)):
            for signs in itertools.product((-1, 1), repeat=spatial_dimension):
                orbit.add(tuple(signs[k] * v[perm[k]] for k in range(spatial_dimension)))
        seen |= orbit
        shells.append(sorted(orbit, reverse=True))
        norm = sum(c * c for c in v)
        if len(shells) >= number_of_complete_shells and norm > previous_norm:
            break
        previous_norm = norm
    return shells


def positive_bloch_wave_vectors(number_of_complete_shells: int, spatial_dimension: int) -> torch.Tensor:
    """Integer reciprocal-lattice vectors of the first shells of the cubic point group, one per +-K pair.

    Same set and order as get_cubic_point_group_positive_normalized_bloch_wave_vectors
    (src/.../utils/lattice_utils.py:130-177): the shells of complete_lattice_shells, the first of each {K, -K} pair retained."""
    half = []
    for shell in complete_lattice_shells(number_of_complete_shells, spatial_dimension):
        known = set()
        for v in shell:
            if v in known:
                continue
            half.append(v)
            known |= {v, tuple(-c for c in v)}
    return torch.tensor(half, dtype=torch.float32)


class EGNNScoreNetwork(ScoreNetwork):
    """EGNN on the torus-uplifted positions (cos K.x, sin K.x), projected back with the Gamma matrices."""

    def __init__(self, hyper_params: EGNNScoreNetworkParameters, edge_builder: Optional[Callable] = None):
        """edge_builder(relative_coordinates, unit_cell, radial_cutoff) -> (edges [E,2], degree [B*N]) overrides the
        HIP radius graph; it exists so tests can run this module against the CPU oracle's edge list."""
        super().__init__(hyper_params)
        hp = hyper_params
        self.number_of_features_per_node = self.num_atom_types + 2
        self.number_of_bloch_wave_shells = hp.number_of_bloch_wave_shells
        k_vectors = positive_bloch_wave_vectors(hp.number_of_bloch_wave_shells, self.spatial_dimension)
        self.register_parameter("bloch_wave_reciprocal_lattice_vectors",
                                torch.nn.Parameter(k_vectors, requires_grad=False))
        self.register_parameter("projection_matrices",
                                torch.nn.Parameter(self._projection_matrices(k_vectors), requires_grad=False))
        assert hp.edges in ("fully_connected", "radial_cutoff"), \
            f"Edges type should be fully_connected or radial_cutoff. Got {hp.edges}"
        self.edges = hp.edges
        self.radial_cutoff = hp.radial_cutoff
        if self.edges == "fully_connected":
            assert self.radial_cutoff is None, "Specifying a radial cutoff is inconsistent with edges=fully_connected."
        else:
            assert type(self.radial_cutoff) is float, \
                "A floating point value for the radial cutoff is needed for edges=radial_cutoff."
        self.drop_duplicate_edges = hp.drop_duplicate_edges
        self.edge_builder = edge_builder
        self.graph_status = None      # device word that collects MDX_STATUS_* bits of the forward without a sync
        # Radius graph without a host read between count and fill (so that a whole sampler iteration can be captured into a
        # hipGraph): the edge list is sized for the worst case B N (N - 1) -- it cannot overflow.  What that capacity costs:
        # 20 bytes per edge row (the int64 pair + the head's scalar) and the compact piece rows of the in-kernel message
        # sums (H floats per 16 edge rows) -- 0.3 GB at C3, 0.9 GB at C5 with 256 structures per GPU; no [E, H] buffer
        # exists on this path.  It is taken while that stays below this fraction of the device's FREE memory; above it the
        # two-call protocol (one host read per forward, lists sized to the real edge count) is used and the switch is
        # logged.  Applies when every graph layer runs the fused edge chain (nothing else then needs E on the host).
        self.static_edge_list_max_fraction = 0.5
        self._logged_two_call_switch = False
        self.egnn = self._make_egnn(hp)

    def __getstate__(self):
        """Copies and pickles of the module carry no cached device tensors (the fully connected edge lists)."""
        state = dict(self.__dict__)
        state.pop("_fully_connected_edges", None)
        return state

    @property
    def edge_chain_precision(self):
        """"f32" (exact binary32 MFMA), "f16x3" (split-f16, three products) or None (per-layer library GEMMs): the
        arithmetic of the fused per-edge MFMA kernel of every graph layer (models/egnn.py)."""
        return self.egnn.graph_layers[0].edge_chain_precision if len(self.egnn.graph_layers) else None

    @edge_chain_precision.setter
    def edge_chain_precision(self, value):
        for layer in self.egnn.graph_layers:
            layer.edge_chain_precision = value

    def adapt_f16_range(self):
        """What a generator calls after it has recomputed an iteration with the exact-f32 kernels because the split-f16 ones
        reported a value beyond the f16 range: every graph layer turns the activation maxima that pass collected into
        per-position exponents for its split-f16 kernels (kernels.ActivationScales; device-side, no host read)."""
        for layer in self.egnn.graph_layers:
            layer.adapt_f16_range()

    def begin_f16_range_fallback(self):
        """Called by a generator right before it recomputes an iteration with the exact-f32 kernels: the maxima those kernels
        collect start from zero, so adapt_f16_range() afterwards sees this iteration and not every f32 launch since start-up."""
        for layer in self.egnn.graph_layers:
            layer.begin_f16_range_fallback()

    def reset_f16_range(self):
        """Default activation exponents again.  The exponents are the one piece of state a fallback leaves behind: they only go
        down, so after a fallback the split-f16 results of the same (seed, call index) can differ in the last bits from before
        it; reset_f16_range() restores bit-reproducibility with a fresh process."""
        for layer in self.egnn.graph_layers:
            layer.reset_f16_range()

    def check_status(self):
        """Raise for any MDX_STATUS_* bit the forward passes have collected (one host read); clears the word."""
        if self.graph_status is not None:
            try:
                neighbors._raise_if_cutoff_too_large(self.graph_status)
            finally:
                self.graph_status.zero_()

    def _impose_non_mask_atomic_type_prediction(self, output: AXL):
        """The MASK logit is forced to -inf (score_network.py:183-185) -- already done by mdx_egnn_outputs on the fused path."""
        if not getattr(output.A, "_mdx_mask_imposed", False):
            super()._impose_non_mask_atomic_type_prediction(output)

    def _first_projection_of_inputs(self):
        """(P W_emb, P b_emb) for the first graph layer's per-node projection matrix P = [W_src; W_dst]: the projections of
        the embedded inputs are then a linear map of [sigma | one_hot] like the embedding itself (no [n_nodes, H] x [H, 2H]
        product per forward).  None when the first layer does not run the fused edge chain.  Cached on the parameters'
        versions."""
        layers = self.egnn.graph_layers
        if len(layers) == 0 or not layers[0].use_fused_ops:
            return None
        pack = layers[0]._edge_chain_pack()
        emb = self.egnn.embedding_in
        if pack is None or pack.proj_weight.shape[1] != emb.out_features:
            return None
        first = layers[0].message_mlp[0].weight
        stamp = tuple((t.data_ptr(), t._version) for t in (first, emb.weight, emb.bias)) + (layers[0].edge_chain_precision,)
        if getattr(self, "_first_projection", (None, None))[0] != stamp:
            with torch.no_grad():
                w2 = (pack.proj_weight.double() @ emb.weight.double()).float().contiguous()
                b2 = (pack.proj_weight.double() @ emb.bias.double()).float().contiguous()
            self._first_projection = (stamp, (w2, b2))
        return self._first_projection[1]

    def _make_egnn(self, hp):
        return EGNN(
            input_size=self.number_of_features_per_node, num_classes=self.num_atom_types + 1,
            message_n_hidden_dimensions=hp.message_n_hidden_dimensions,
            message_hidden_dimensions_size=hp.message_hidden_dimensions_size,
            node_n_hidden_dimensions=hp.node_n_hidden_dimensions,
            node_hidden_dimensions_size=hp.node_hidden_dimensions_size,
            coordinate_n_hidden_dimensions=hp.coordinate_n_hidden_dimensions,
            coordinate_hidden_dimensions_size=hp.coordinate_hidden_dimensions_size,
            residual=hp.residual, attention=hp.attention, normalize=hp.normalize, tanh=hp.tanh,
            coords_agg=hp.coords_agg, message_agg=hp.message_agg, n_layers=hp.n_layers)

    @staticmethod
    def _projection_matrices(k_vectors: torch.Tensor) -> torch.Tensor:
        """Gamma^alpha = blockdiag_k( K_k[alpha] * [[0,-1],[1,0]] )  (:103-133)"""
        n_k, d = k_vectors.shape
        gamma = torch.zeros(d, 2 * n_k, 2 * n_k)
        for k in range(n_k):
            gamma[:, 2 * k, 2 * k + 1] = -k_vectors[k]
            gamma[:, 2 * k + 1, 2 * k] = k_vectors[k]
        return gamma

    def _build_edges(self, relative_coordinates: torch.Tensor, lattice_parameters: torch.Tensor):
        bsz, n, d = relative_coordinates.shape
        if self.edges == "fully_connected":
            # (the list depends on the batch's shape alone: built once per shape and device, reused by every forward -- and by
            # the forwards of a captured sampler iteration)
            key = (n, bsz, str(relative_coordinates.device))
            cache = self.__dict__.setdefault("_fully_connected_edges", {})
            if key not in cache:
                if len(cache) >= 8:
                    cache.clear()
                cache[key] = (neighbors.get_edges_batch(n, bsz, device=relative_coordinates.device),
                              torch.full((bsz * n,), n - 1, dtype=torch.int64, device=relative_coordinates.device))
            return cache[key]
        if self.edge_builder is not None:
            lengths = lattice_parameters[:, :d].clip(min=2.2 * self.radial_cutoff)   # "avoid box collapse" (:236-239)
            return self.edge_builder(relative_coordinates, torch.diag_embed(lengths), self.radial_cutoff)
        # drop_duplicate_edges (models/egnn_utils.py:138-140) only matters when a pair of atoms is within the cutoff through more
        # than one periodic image; the cell this graph is built in has every length >= 2.2 x cutoff (the clip above), so a pair
        # has at most one such image and the edge MULTISET is the same with and without the de-duplication.  The reference's
        # `False` differs only in the ORDER of the list (by image, then source), which the EGNN's segment sums do not depend on
        # beyond fp32 summation order: both settings take the sorted list of the HIP kernel.
        if self.graph_status is None or self.graph_status.device != relative_coordinates.device:
            self.graph_status = torch.zeros(1, dtype=torch.int32, device=relative_coordinates.device)
        capacity = bsz * n * (n - 1)
        if relative_coordinates.is_cuda and not torch.is_grad_enabled() and self._static_edge_list_fits(bsz, n, relative_coordinates.device):
            # clip, diagonal cell, cartesian positions, hit masks and emission behind one call (two launches)
            edges, degree, offsets, n_edges = neighbors.get_edges_static_clipped(
                relative_coordinates, lattice_parameters.to(torch.float32), 2.2 * self.radial_cutoff, self.radial_cutoff,
                capacity, status=self.graph_status)
            return edges, (degree, offsets, n_edges)
        lengths = lattice_parameters[:, :d].clip(min=2.2 * self.radial_cutoff)       # "avoid box collapse" (:236-239)
        return neighbors.get_edges_with_radial_cutoff(relative_coordinates, torch.diag_embed(lengths), self.radial_cutoff,
                                                      status=self.graph_status, return_degree=True)

    def _static_edge_list_fits(self, bsz: int, n: int, device) -> bool:
        """May the radius graph of a [bsz, n] batch go into a capacity-sized list (no host read per forward)?  Every graph
        layer must run the fused edge chain (nothing else then needs the edge count on the host) and the list must cost less
        than `static_edge_list_max_fraction` of the device's free memory; the answer is kept per shape (hipMemGetInfo is not
        allowed while a stream is capturing: the eager warm-up iterations that precede every capture ask first)."""
        capacity = bsz * n * (n - 1)
        if self.spatial_dimension != 3:      # (1-D / 2-D radius graphs go through the padded two-call search: utils/neighbors.py)
            return False
        if capacity <= 0 or not all(layer.use_fused_ops and layer._edge_chain_pack() is not None
                                    for layer in self.egnn.graph_layers):
            return False
        width = max((layer._edge_chain_pack().hidden for layer in self.egnn.graph_layers), default=0)
        needed = capacity * 20 + (capacity // 16 + bsz * n) * width * 4
        key = (str(device), capacity, width, self.static_edge_list_max_fraction)
        decisions = self.__dict__.setdefault("_static_decisions", {})
        if key not in decisions and not torch.cuda.is_current_stream_capturing():
            # free = what the driver reports + what torch's caching allocator holds but has not handed out
            free = torch.cuda.mem_get_info(device)[0] + torch.cuda.memory_reserved(device) - torch.cuda.memory_allocated(device)
            decisions[key] = (needed <= self.static_edge_list_max_fraction * free, free)
        fits, free = decisions.get(key, (True, 0))
        if not fits and not self._logged_two_call_switch:
            import logging
            logging.getLogger(__name__).warning(
                "EGNN radius graph: a capacity-sized edge list would take %.1f GB of the %.1f GB free on the device; using "
                "the two-call protocol (one host read per forward, the sampler iteration is not captured in a hipGraph)",
                needed / 2 ** 30, free / 2 ** 30)
            self._logged_two_call_switch = True
        return fits

    def capture_safe(self, batch_size: int, number_of_atoms: int, device) -> bool:
        """Can a forward on a [batch_size, number_of_atoms] batch be captured into a hipGraph (no host synchronisation)?
        What a generator asks before it captures its iteration (LangevinGenerator._run_loop): fully connected graphs always;
        radius graphs when the capacity-sized edge list applies; a caller's own edge builder never (it is host code)."""
        if self.edges == "fully_connected":
            return True
        if self.edge_builder is not None:
            return False
        return self._static_edge_list_fits(batch_size, number_of_atoms, torch.device(device))

    def _forward_unchecked(self, batch: Dict[AnyStr, torch.Tensor], conditional: bool = False) -> AXL:
        comp = batch[NOISY_AXL_COMPOSITION]
        x = comp.X
        bsz, n, d = x.shape
        if x.is_cuda and (self.graph_status is None or self.graph_status.device != x.device):
            self.graph_status = torch.zeros(1, dtype=torch.int32, device=x.device)
        for layer in self.egnn.graph_layers:       # where the fused edge chain reports MDX_STATUS_EGNN_F16_RANGE
            layer.status_word = self.graph_status if x.is_cuda else None
        edges, degree = self._build_edges(x, comp.L)

        emb = self.egnn.embedding_in
        sigma_in = batch[NOISE]
        if (x.is_cuda and not torch.is_grad_enabled() and d == 3 and x.dtype == torch.float32 and
                emb.in_features == self.num_atom_types + 2 and sigma_in.numel() == bsz):
            # inputs and outputs around the EGNN as one launch each (mdx_egnn_node_inputs / mdx_egnn_scores) instead of
            # ~25 elementwise passes per forward; the same arithmetic (see include/mdx_hip.h)
            from ... import kernels
            k_vectors = self.bloch_wave_reciprocal_lattice_vectors.to(x)
            second = self._first_projection_of_inputs()
            res = kernels.egnn_node_inputs(x.contiguous(), k_vectors.contiguous(),
                                           sigma_in.to(device=x.device, dtype=torch.float32).reshape(-1).contiguous(),
                                           comp.A.reshape(bsz, n).long().contiguous(), emb.weight.detach().contiguous(),
                                           emb.bias.detach().contiguous(), second=second)
            z, h, first_proj = res if second is not None else (res[0], res[1], None)
            head = self.egnn.node_classification_layer
            if head.out_features <= 8 and head.in_features % 4 == 0 and comp.L.dtype == torch.float32:
                # classification layer (MASK logit at -inf), scores and the zero lattice output: one launch
                out = self.egnn(h=h, edges=edges, x=z, degree=degree, embedded=True, first_proj=first_proj, classify=False)
                scores, logits, zeros = kernels.egnn_outputs(
                    z, out.X.contiguous(), k_vectors.contiguous(), out.A.contiguous(), head.weight.detach().contiguous(),
                    head.bias.detach().contiguous(), self.num_atom_types, comp.L.numel())
                logits = logits.reshape(bsz, n, -1)
                logits._mdx_mask_imposed = True
                return AXL(A=logits, X=scores.reshape(bsz, n, d), L=zeros.reshape(comp.L.shape))
            out = self.egnn(h=h, edges=edges, x=z, degree=degree, embedded=True, first_proj=first_proj)
            scores = kernels.egnn_scores(z, out.X.contiguous(), k_vectors.contiguous())
            return AXL(A=out.A.reshape(bsz, n, -1), X=scores.reshape(bsz, n, d), L=torch.zeros_like(comp.L))

        flat = x.reshape(bsz * n, d)
        k_vectors = self.bloch_wave_reciprocal_lattice_vectors.to(flat)
        if flat.is_cuda:      # K = d = 3: as a library GEMM this costs ~0.45 ms; as a broadcast product it is one small pass
            kr = ((2.0 * math.pi * flat)[:, None, :] * k_vectors[None, :, :]).sum(dim=-1)          # [nodes, n_k]
        else:
            kr = (2.0 * math.pi * flat) @ k_vectors.t()
        if flat.is_cuda:      # correctly rounded, like mdx_egnn.hip::uplift (why: see there); the CPU path is the reference's own op
            kr64 = kr.double()
            z = torch.stack([kr64.cos().to(kr.dtype), kr64.sin().to(kr.dtype)], dim=2).reshape(bsz * n, -1)
        else:
            z = torch.stack([kr.cos(), kr.sin()], dim=2).reshape(bsz * n, -1)                    # (k, two) interleaved

        sigmas = batch[NOISE].to(x.device).repeat_interleave(n, dim=0)
        one_hot = torch.nn.functional.one_hot(comp.A.reshape(-1).long(), self.num_atom_types + 1).to(x.dtype)
        h = torch.cat([sigmas, one_hot], dim=1)

        out = self.egnn(h=h, edges=edges, x=z, degree=degree)
        # S^alpha = z . Gamma^alpha . z_hat   (:283-290)
        gamma = self.projection_matrices.to(z)
        if z.is_cuda:         # the same contraction as broadcast products (2 n_k x 2 n_k terms per direction)
            scores = ((z[:, None, :, None] * gamma[None]) * out.X[:, None, None, :]).sum(dim=(2, 3))
        else:
            scores = torch.einsum("ni,aij,nj->na", z, gamma, out.X)
        return AXL(A=out.A.reshape(bsz, n, -1), X=scores.reshape(bsz, n, d), L=torch.zeros_like(comp.L))
