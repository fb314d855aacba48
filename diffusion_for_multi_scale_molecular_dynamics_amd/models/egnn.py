"""E(n)-equivariant graph network used as a score network backbone (PyTorch forward).

Same architecture and parameter names (state_dict keys) as the reference's EGNN / E_GCL
(src/.../models/egnn.py:20-385; Satorras et al., arXiv:2102.09844), so reference checkpoints load unchanged.
Written for sorted edge lists (edges grouped by source node, which is what the HIP radius-graph kernel and
get_edges_batch produce):
  * segment sums over sorted edges use torch.segment_reduce -- no atomics, run-to-run deterministic;
  * the first message layer is applied per NODE and gathered per edge (W [h_i | h_j | r] = W_i h_i + W_j h_j + w_r r),
    which removes the [E, 2H+1] concatenation and 2/5 of the message-MLP FLOPs;
  * on device tensors the two per-node reductions run as one-wavefront-per-node HIP kernels over the sorted segments
    (kernels.segment_rows for the messages; kernels.egnn_coord_head = last coordinate layer H -> 1, product with
    coord_diff and segment mean in one pass over the [E, H] activations);
  * on device tensors, without autograd, the whole per-edge part of a layer -- first message layer, the message MLP,
    the attention gate, the coordinate MLP and its H -> 1 head -- is ONE hand-written MFMA kernel (kernels.egnn_edge_chain,
    csrc/mdx_egnn_chain.hip) that keeps the [E, H] activations in registers between layers; `tanh` and `normalize` act in
    the per-node kernel that adds the coordinate updates up; `edge_chain_precision` selects exact binary32 MFMA ("f32") or
    the split-f16 three-product form ("f16x3"); None keeps the per-layer PyTorch path (also taken for shapes the kernel does
    not cover: widths above 256, other activations).  Narrow and unequal message / coordinate widths (the reference's defaults
    are 16 / 32) run on the chain zero-padded to its next width.

The HIP calls are invisible to autograd, so they are used only when no gradient can be requested (torch.no_grad(), or
nothing requires grad); with autograd on, the module runs as plain PyTorch.  Callers that pass their own edge list
(no `degree`) get it sorted by source here first: the segment kernels need that order, the reference's
unsorted_segment_sum accepted any.
"""
from typing import Callable, Optional, Tuple

import torch
from torch import nn

from ..namespace import AXL


def run_mlp(layers, x: torch.Tensor, fused: bool = False) -> torch.Tensor:
    """Apply a Sequential of Linear / SiLU / ... modules as plain PyTorch ops (layer shapes the MFMA chains do not cover).
    (`fused` is accepted for the callers' sake and ignored: rounds 1-4 ran every Linear + SiLU pair of this path as one
    hipBLASLt matmul with a SWISH_BIAS epilogue; the library no longer links a vendor GEMM.)"""
    for layer in layers:
        x = layer(x)
    return x


def segment_sum_sorted(data: torch.Tensor, degree: torch.Tensor) -> torch.Tensor:
    """Sum rows of `data` over consecutive segments of lengths `degree` (edges sorted by source node).

    Replaces unsorted_segment_sum (src/.../models/egnn_utils.py:11-38) for sorted edge lists."""
    return torch.segment_reduce(data, "sum", lengths=degree, axis=0, unsafe=True)


class E_GCL(nn.Module):
    """One equivariant convolution layer (egnn.py:20-289)."""

    def __init__(self, input_size: int, output_size: int, message_n_hidden_dimensions: int,
                 message_hidden_dimensions_size: int, node_n_hidden_dimensions: int, node_hidden_dimensions_size: int,
                 coordinate_n_hidden_dimensions: int, coordinate_hidden_dimensions_size: int,
                 act_fn: Callable = nn.SiLU(), residual: bool = True, attention: bool = False,
                 normalize: bool = False, coords_agg: str = "mean", message_agg: str = "mean", tanh: bool = False):
        super().__init__()
        if coords_agg not in ("mean", "sum"):
            raise ValueError(f"coords_agg should be mean or sum. Got {coords_agg}")
        if message_agg not in ("mean", "sum"):
            raise ValueError(f"message_agg should be mean or sum. Got {message_agg}")
        self.residual, self.attention, self.normalize, self.tanh = residual, attention, normalize, tanh
        self.coords_mean = coords_agg == "mean"
        self.message_mean = message_agg == "mean"
        self.epsilon = 1e-8
        self.input_size = input_size
        self.use_fused_ops = True        # device tensors only; the CPU path is plain PyTorch
        # "f16x3" (default: split-f16 MFMA, binary32-level accuracy -- measured against fp64 in tests/test_egnn_chain_gpu.py --
        # at 2.7x the speed; a value beyond the f16 range is reported and the generator recomputes with "f32"),
        # "f32" (exact binary32 MFMA) or None (per-layer library GEMMs)
        self.edge_chain_precision = "f16x3"
        self.status_word = None          # device int32 word for MDX_STATUS_EGNN_F16_RANGE (set by the score network)
        self._chain = (None, None)       # (stamp, kernels.EdgeChainPack)
        self._node_chain = (None, None)  # (stamp, kernels.RowChainPack): the node MLP after its first layer
        self._node_mlp = (None, None)    # (stamp, kernels.NodeMlpPack): the whole node MLP

        mh, nh, ch = message_hidden_dimensions_size, node_hidden_dimensions_size, coordinate_hidden_dimensions_size
        layers = [nn.Linear(2 * input_size + 1, mh), act_fn]
        for _ in range(message_n_hidden_dimensions):
            layers += [nn.Linear(mh, mh), act_fn]
        self.message_mlp = nn.Sequential(*layers)

        layers = [nn.Linear(input_size + mh, nh), act_fn]
        for _ in range(node_n_hidden_dimensions):
            layers += [nn.Linear(nh, nh), act_fn]
        layers.append(nn.Linear(nh, output_size))
        self.node_mlp = nn.Sequential(*layers)

        layers = [nn.Linear(mh, ch), act_fn]
        for _ in range(coordinate_n_hidden_dimensions):
            layers += [nn.Linear(ch, ch), act_fn]
        layers.append(nn.Linear(ch, 1, bias=False))
        if tanh:
            layers.append(nn.Tanh())
        self.coord_mlp = nn.Sequential(*layers)

        if attention:
            self.att_mlp = nn.Sequential(nn.Linear(mh, 1), nn.Sigmoid())

    def __getstate__(self):
        """Copies and pickles of the module (copy.deepcopy, torch.save of a whole model) carry no device images: the packs
        hold raw pointers and are rebuilt on first use."""
        state = dict(self.__dict__)
        state["_chain"], state["_node_chain"], state["status_word"] = (None, None), (None, None), None
        state["_node_mlp"] = (None, None)
        state.pop("_chain_kept", None)
        state.pop("_node_mlp_kept", None)
        state.pop("_node_chain_kept", None)
        state.pop("_activation_scales", None)
        return state

    def _scales(self, kind: str, n_layers: int, device):
        """kernels.ActivationScales of one of the layer's chains ("edge", "node", "rows"): the per-position powers of two of
        the split-f16 kernels and the maxima the exact-f32 kernels collect, shared by the chain's packs of every precision
        and kept across repacks (an exponent only ever goes down)."""
        from .. import kernels
        kept = self.__dict__.setdefault("_activation_scales", {})
        key = (kind, n_layers, str(device))
        if key not in kept:
            kept[key] = kernels.ActivationScales(n_layers, device)
        return kept[key]

    def adapt_f16_range(self):
        """After an exact-f32 pass (the generator's answer to an f16-range report): turn the maxima that pass collected into
        activation exponents for the split-f16 kernels (device-side, no host read)."""
        for scales in self.__dict__.get("_activation_scales", {}).values():
            scales.adapt()

    def begin_f16_range_fallback(self):
        """Before the exact-f32 pass of a fallback: forget the maxima earlier f32 launches have left."""
        for scales in self.__dict__.get("_activation_scales", {}).values():
            scales.maxima.zero_()

    def reset_f16_range(self):
        """Back to the default activation exponents (and no collected maxima): what a freshly built layer has."""
        for scales in self.__dict__.get("_activation_scales", {}).values():
            scales.reset()

    def _messages(self, h: torch.Tensor, edge_index: torch.Tensor, radial: torch.Tensor, fused: bool) -> torch.Tensor:
        first = self.message_mlp[0]
        n_in = self.input_size
        w = first.weight
        mh = w.shape[0]
        # node-level projections h W_src^T | h W_dst^T, combined per edge
        proj = torch.nn.functional.linear(h, torch.cat([w[:, :n_in], w[:, n_in:2 * n_in]], dim=0))
        rest = list(self.message_mlp)[1:]
        if fused and mh % 4 == 0 and isinstance(rest[0], nn.SiLU):
            from .. import kernels
            out = kernels.egnn_message_input(proj.contiguous(), edge_index, radial.reshape(-1).contiguous(), first.bias,
                                             w[:, 2 * n_in].contiguous(), silu=True)
            rest = rest[1:]
        else:
            row, col = edge_index[:, 0], edge_index[:, 1]
            pre = proj[:, :mh].index_select(0, row) + proj[:, mh:].index_select(0, col)
            out = torch.addcmul(pre + first.bias, radial, w[:, 2 * n_in].unsqueeze(0))
        out = run_mlp(rest, out, fused)
        if self.attention:
            out = out * self.att_mlp(out)
        return out

    # ---- the layer's public pieces under the reference's names (src/models/egnn.py:136-262): plain PyTorch, any edge order.
    # forward() below does not come through them on the GPU (one MFMA kernel per layer); composed as the reference's forward
    # composes them they give the same function (tests/test_host_cpu.py::test_e_gcl_public_pieces_compose_to_forward).
    def message_model(self, source: torch.Tensor, target: torch.Tensor, radial: torch.Tensor) -> torch.Tensor:
        """m_ij = phi_e(h_i, h_j, |x_i - x_j|^2), gated by its own attention value when `attention`."""
        out = self.message_mlp(torch.cat([source, target, radial], dim=1))
        return out * self.att_mlp(out) if self.attention else out

    def node_model(self, x: torch.Tensor, edge_index: torch.Tensor, messages: torch.Tensor) -> torch.Tensor:
        """h_i' = phi_h(h_i, sum or mean over the edges of source i of m_ij) (+ h_i when `residual`)."""
        from .egnn_utils import unsorted_segment_mean, unsorted_segment_sum
        reduce = unsorted_segment_mean if self.message_mean else unsorted_segment_sum
        out = self.node_mlp(torch.cat([x, reduce(messages, edge_index[:, 0], num_segments=x.size(0))], dim=1))
        return x + out if self.residual else out

    def coord_model(self, coord: torch.Tensor, edge_index: torch.Tensor, coord_diff: torch.Tensor,
                    messages: torch.Tensor) -> torch.Tensor:
        """x_i += sum or mean over the edges of source i of (x_i - x_j) phi_x(m_ij) -- IN PLACE, as the reference."""
        from .egnn_utils import unsorted_segment_mean, unsorted_segment_sum
        reduce = unsorted_segment_mean if self.coords_mean else unsorted_segment_sum
        coord += reduce(coord_diff * self.coord_mlp(messages), edge_index[:, 0], num_segments=coord.size(0))
        return coord

    def coord2radial(self, edge_index: torch.Tensor, coord: torch.Tensor) -> Tuple[torch.Tensor, torch.Tensor]:
        """(|x_i - x_j|^2 [E, 1], x_i - x_j [E, d]); with `normalize` the difference is scaled by normalize_radial_norm."""
        coord_diff = coord[edge_index[:, 0]] - coord[edge_index[:, 1]]
        radial = torch.sum(coord_diff ** 2, 1).unsqueeze(1)
        if self.normalize:
            coord_diff = self.normalize_radial_norm(radial) * coord_diff
        return radial, coord_diff

    def normalize_radial_norm(self, radial_norm_squared: torch.Tensor) -> torch.Tensor:
        """tanh(r^2) / sqrt(r^2 + epsilon^2): the scaled difference goes to 0 like r^2 at contact and to unit length far away."""
        return torch.tanh(radial_norm_squared) / torch.sqrt(radial_norm_squared + self.epsilon ** 2)

    def _chain_modules(self):
        """(first message layer, message H->H layers, coordinate H->H layers, coordinate head) if the per-edge MLPs have
        the Linear / SiLU alternation the fused kernel implements, else None.  The layer's options ride along: `tanh` is the
        nn.Tanh behind the head (applied where the head's scalar is consumed), `normalize` a per-edge factor there too,
        `attention` the gate inside the kernel (_attention_layer)."""
        message, coordinate = list(self.message_mlp), list(self.coord_mlp)
        if self.tanh:
            if not coordinate or not isinstance(coordinate[-1], nn.Tanh):
                return None
            coordinate = coordinate[:-1]
        if self.attention and self._attention_layer() is None:
            return None
        if len(message) % 2 or len(coordinate) % 2 == 0:
            return None
        pairs = list(zip(message[0::2], message[1::2])) + list(zip(coordinate[0:-1:2], coordinate[1:-1:2]))
        if not all(isinstance(lin, nn.Linear) and isinstance(act, nn.SiLU) for lin, act in pairs):
            return None
        if not isinstance(coordinate[-1], nn.Linear):
            return None
        return message[0], message[2::2], coordinate[0:-1:2], coordinate[-1]

    def _attention_layer(self):
        """att_mlp's nn.Linear(H, 1) when the gate has the reference's form Linear + Sigmoid, else None."""
        att = list(getattr(self, "att_mlp", []))
        if len(att) == 2 and isinstance(att[0], nn.Linear) and isinstance(att[1], nn.Sigmoid) and att[0].out_features == 1 \
                and att[0].bias is not None:
            return att[0]
        return None

    def _coord_flags(self) -> int:
        from .. import kernels
        return kernels.coord_flags(self.normalize, self.tanh)

    def _edge_chain_pack(self):
        """The layer's kernels.EdgeChainPack, rebuilt when a parameter, the device or the precision changed; None when
        the fused kernel does not apply."""
        if self.edge_chain_precision is None:
            return None
        modules = self._chain_modules()
        if modules is None:
            return None
        from .. import kernels
        if not kernels.EdgeChainPack.supported(*modules):
            return None
        attention = self._attention_layer() if self.attention else None
        if attention is not None and attention.in_features != modules[0].out_features:
            return None
        linears = [modules[0], *modules[1], *modules[2], modules[3]] + ([attention] if attention is not None else [])
        stamp = (self.edge_chain_precision,) + tuple((t.data_ptr(), t._version) for lin in linears
                                                      for t in (lin.weight, lin.bias) if t is not None)
        if self._chain[0] != stamp:
            # (one image per precision is kept: the generator's one-call switch to "f32" and back repacks nothing)
            kept = self.__dict__.setdefault("_chain_kept", {})
            if kept.get(self.edge_chain_precision, (None, None))[0] != stamp:
                n_layers = len(list(modules[1])) + len(list(modules[2]))
                kept[self.edge_chain_precision] = (stamp, kernels.EdgeChainPack(
                    *modules, input_size=self.input_size, precision=self.edge_chain_precision,
                    scales=self._scales("edge", n_layers, modules[0].weight.device), attention_layer=attention))
            self._chain = kept[self.edge_chain_precision]
        if attention is not None and not self._chain[1].piece_sums_ok:
            return None          # (the gate is instantiated for the in-kernel message sums only: mdx_egnn_edge_chain)
        return self._chain[1]

    def _node_chain_pack(self):
        """kernels.RowChainPack of the node MLP's H -> H layers (all but its first, 2H -> H, layer), or None when the stack
        is not Linear / SiLU alternation of equal widths ending in a Linear."""
        if self.edge_chain_precision is None:
            return None
        node = list(self.node_mlp)
        if len(node) < 3 or len(node) % 2 == 0:
            return None
        linears, acts = node[0::2], node[1::2]
        if not all(isinstance(lin, nn.Linear) for lin in linears) or not all(isinstance(a, nn.SiLU) for a in acts):
            return None
        rest = linears[1:]
        from .. import kernels
        if linears[0].out_features != rest[0].in_features or not kernels.RowChainPack.supported(rest):
            return None
        stamp = (self.edge_chain_precision,) + tuple((t.data_ptr(), t._version) for lin in rest for t in (lin.weight, lin.bias))
        if self._node_chain[0] != stamp:
            # one image per precision is KEPT, like the other two packs: a captured iteration has the split-f16 image's
            # pointers in its kernel arguments, and the generator's one-iteration switch to "f32" and back must not free it
            kept = self.__dict__.setdefault("_node_chain_kept", {})
            if kept.get(self.edge_chain_precision, (None, None))[0] != stamp:
                kept[self.edge_chain_precision] = (stamp, kernels.RowChainPack(
                    rest, self.edge_chain_precision, scales=self._scales("rows", len(rest), rest[0].weight.device)))
            self._node_chain = kept[self.edge_chain_precision]
        return self._node_chain[1]

    def _node_mlp_pack(self, next_layer=None):
        """kernels.NodeMlpPack of the WHOLE node MLP (Linear(2H, H) first), or None when it does not have that shape; with
        the per-node projections of `next_layer` appended when that layer runs the fused edge chain on the same width."""
        if self.edge_chain_precision is None:
            return None
        node = list(self.node_mlp)
        if len(node) < 3 or len(node) % 2 == 0:
            return None
        linears, acts = node[0::2], node[1::2]
        if not all(isinstance(lin, nn.Linear) for lin in linears) or not all(isinstance(a, nn.SiLU) for a in acts):
            return None
        from .. import kernels
        if not kernels.NodeMlpPack.supported(linears):
            return None
        next_pack = next_layer._edge_chain_pack() if next_layer is not None and next_layer.use_fused_ops else None
        projection = None
        if next_pack is not None and tuple(next_pack.proj_weight.shape) == (2 * linears[0].out_features, linears[0].out_features) \
                and next_pack.hidden == linears[0].out_features and next_pack.message_width == next_pack.hidden:
            projection = next_pack.proj_weight
        first_next = next_layer.message_mlp[0].weight if projection is not None else None
        stamp = (self.edge_chain_precision,) + tuple((t.data_ptr(), t._version) for lin in linears for t in (lin.weight, lin.bias)) + \
            ((first_next.data_ptr(), first_next._version) if first_next is not None else (None,))
        if self._node_mlp[0] != stamp:
            kept = self.__dict__.setdefault("_node_mlp_kept", {})
            if kept.get(self.edge_chain_precision, (None, None))[0] != stamp:
                kept[self.edge_chain_precision] = (stamp, kernels.NodeMlpPack(
                    linears, self.edge_chain_precision, next_projection=projection,
                    scales=self._scales("node", kernels.NodeMlpPack.n_chain_layers(linears), linears[0].weight.device)))
            self._node_mlp = kept[self.edge_chain_precision]
        return self._node_mlp[1]

    def _coord_head_is_plain(self) -> bool:
        last = self.coord_mlp[-1]
        return isinstance(last, nn.Linear) and last.out_features == 1 and last.bias is None and \
            last.in_features % 4 == 0 and len(self.coord_mlp) >= 2 and isinstance(self.coord_mlp[-2], nn.SiLU)

    def forward(self, h: torch.Tensor, edge_index: torch.Tensor, coord: torch.Tensor,
                degree: Optional[torch.Tensor] = None, offsets: Optional[torch.Tensor] = None,
                n_edges: Optional[torch.Tensor] = None, node_proj: Optional[torch.Tensor] = None,
                next_layer: Optional["E_GCL"] = None) -> Tuple[torch.Tensor, torch.Tensor]:
        """node_proj: this layer's per-node projections [n_nodes, 2H], when the previous layer's node kernel has already
        computed them; next_layer: the graph layer that follows (its projections are then computed by this layer's node
        kernel and left in self._next_proj).
        h [n_nodes, F]; edge_index [E, 2] sorted by column 0; coord [n_nodes, D]; degree [n_nodes] edge counts;
        offsets [n_nodes] = exclusive scan of degree (enables the segment kernels on device tensors); n_edges: int64 [1]
        on the device when edge_index is a capacity-sized list whose first n_edges rows are the edges (fused chain only)."""
        row, col = edge_index[:, 0], edge_index[:, 1]
        if degree is None:
            degree = torch.bincount(row, minlength=h.shape[0])
        # the HIP calls are invisible to autograd: only when no gradient can be requested
        needs_grad = torch.is_grad_enabled() and (h.requires_grad or coord.requires_grad or
                                                  any(t.requires_grad for t in self.parameters()))
        fused = self.use_fused_ops and h.is_cuda and not needs_grad
        if fused and offsets is not None and edge_index.shape[0] > 0:
            pack = self._edge_chain_pack()
            if pack is not None:
                return self._forward_edge_chain(pack, h, edge_index, coord, degree, offsets, n_edges, node_proj, next_layer)
        assert n_edges is None, "a capacity-sized edge list needs the fused edge chain in every layer"
        count = degree.clamp(min=1).to(h.dtype).unsqueeze(1)      # (the reference DIVIDES by the count: egnn_utils.py:66-68)

        coord_diff = coord.index_select(0, row) - coord.index_select(0, col)
        radial = (coord_diff ** 2).sum(dim=1, keepdim=True)
        if self.normalize:
            coord_diff = torch.tanh(radial) / torch.sqrt(radial + self.epsilon ** 2) * coord_diff

        messages = self._messages(h, edge_index, radial, fused)

        segments = fused and offsets is not None and messages.shape[1] % 4 == 0
        if segments and self._coord_head_is_plain():
            from .. import kernels
            hidden = run_mlp(list(self.coord_mlp)[:-1], messages, fused)
            coord = coord + kernels.egnn_coord_head(hidden.contiguous(), self.coord_mlp[-1].weight.reshape(-1),
                                                    coord_diff.contiguous(), offsets, degree, self.coords_mean)
        else:
            trans = segment_sum_sorted(coord_diff * run_mlp(self.coord_mlp, messages, fused), degree)
            coord = coord + (trans / count if self.coords_mean else trans)

        if segments:
            from .. import kernels
            agg = kernels.segment_rows(messages.contiguous(), offsets, degree, self.message_mean)
        else:
            agg = segment_sum_sorted(messages, degree)
            if self.message_mean:
                agg = agg / count
        out = run_mlp(self.node_mlp, torch.cat([h, agg], dim=1), fused)
        if self.residual:
            out = h + out
        return out, coord


    def _forward_edge_chain(self, pack, h, edge_index, coord, degree, offsets, n_edges=None, node_proj=None, next_layer=None):
        """E_GCL.forward with the per-edge work in one MFMA kernel: node projections (library GEMM, per node) -> fused
        chain -> the two sorted-segment reductions -> node MLP."""
        from .. import kernels
        proj = node_proj if node_proj is not None else torch.nn.functional.linear(h, pack.proj_weight)
        coord = coord.contiguous()
        # the messages are added up per node inside the kernel (piece sums) and never written out as [E, H]
        in_kernel = pack.piece_sums_ok and h.shape[0] < (1 << 31)
        messages, edge_scalar = kernels.egnn_edge_chain(pack, proj.contiguous(), coord, edge_index, status=self.status_word,
                                                        n_edges_dev=n_edges, piece_sums=in_kernel)
        padded = pack.message_width != pack.hidden      # a narrow / unequal-width layer run zero-padded (kernels.EdgeChainPack)
        if padded:
            # the message sums come out [n_nodes, H] with columns message_width .. H-1 zero: the node MLP takes the real ones
            coord_out = kernels.egnn_coord_aggregate(edge_scalar, coord, edge_index, offsets, degree, self.coords_mean,
                                                     flags=self._coord_flags())
            agg = (kernels.segment_combine(messages, edge_index.shape[0], offsets, degree, self.message_mean) if in_kernel
                   else kernels.segment_rows(messages, offsets, degree, self.message_mean))
            node_in = torch.cat([h, agg[:, :pack.message_width]], dim=1)
        elif in_kernel and h.shape[1] == messages.shape[1] and coord.shape[1] <= 8:
            whole = self._node_mlp_pack(next_layer)
            if whole is not None and whole.hidden == h.shape[1]:
                # everything per node between the edge chain and the node MLP in one pass (the message sums and the updated
                # coordinates), then the whole node MLP (its 2H -> H layer included, reading h and the sums as the two halves of
                # its input row: [h | agg] is never formed), the residual and -- when a graph layer follows -- that layer's
                # per-node projections: one launch on the matrix cores
                h = h.contiguous()
                agg, coord_out = kernels.egnn_node_gather(messages, edge_index.shape[0], offsets, degree, self.message_mean,
                                                          None, edge_scalar, coord, edge_index, self.coords_mean,
                                                          flags=self._coord_flags())
                out = kernels.node_mlp_rows(whole, h, self.residual, status=self.status_word, agg=agg)
                if whole.projects:
                    out, self._next_proj = out
                return out, coord_out
            node_in, coord_out = kernels.egnn_node_gather(messages, edge_index.shape[0], offsets, degree, self.message_mean,
                                                          h.contiguous(), edge_scalar, coord, edge_index, self.coords_mean,
                                                          flags=self._coord_flags())
        else:
            coord_out = kernels.egnn_coord_aggregate(edge_scalar, coord, edge_index, offsets, degree, self.coords_mean,
                                                     flags=self._coord_flags())
            agg = (kernels.segment_combine(messages, edge_index.shape[0], offsets, degree, self.message_mean) if in_kernel
                   else kernels.segment_rows(messages, offsets, degree, self.message_mean))
            node_in = torch.cat([h, agg], dim=1)
        node_pack = self._node_chain_pack()
        if node_pack is not None and (not self.residual or h.shape[1] == node_pack.hidden):
            # first node layer (2H -> H, + SiLU): a PyTorch matmul per node; the other layers and the residual: one MFMA launch
            first = self.node_mlp[0]
            hidden = torch.nn.functional.silu(torch.nn.functional.linear(node_in, first.weight, first.bias))
            out = kernels.mlp_chain_rows(node_pack, hidden, residual=h.contiguous() if self.residual else None,
                                         status=self.status_word)
            return out, coord_out
        out = run_mlp(self.node_mlp, node_in, True)
        if self.residual:
            out = h + out
        return out, coord_out


class EGNN(nn.Module):
    """Stack of E_GCL layers with input embedding and node-classification head (egnn.py:292-385)."""

    def __init__(self, input_size: int, num_classes: int, message_n_hidden_dimensions: int,
                 message_hidden_dimensions_size: int, node_n_hidden_dimensions: int, node_hidden_dimensions_size: int,
                 coordinate_n_hidden_dimensions: int, coordinate_hidden_dimensions_size: int,
                 act_fn: Callable = nn.SiLU(), residual: bool = True, attention: bool = False,
                 normalize: bool = False, tanh: bool = False, coords_agg: str = "mean", message_agg: str = "mean",
                 n_layers: int = 4):
        super().__init__()
        self.n_layers = n_layers
        self.embedding_in = nn.Linear(input_size, node_hidden_dimensions_size)
        self.graph_layers = nn.ModuleList([])
        self.node_classification_layer = nn.Linear(node_hidden_dimensions_size, num_classes)
        for _ in range(n_layers):
            self.graph_layers.append(E_GCL(
                input_size=node_hidden_dimensions_size, output_size=node_hidden_dimensions_size,
                message_n_hidden_dimensions=message_n_hidden_dimensions,
                message_hidden_dimensions_size=message_hidden_dimensions_size,
                node_n_hidden_dimensions=node_n_hidden_dimensions,
                node_hidden_dimensions_size=node_hidden_dimensions_size,
                coordinate_n_hidden_dimensions=coordinate_n_hidden_dimensions,
                coordinate_hidden_dimensions_size=coordinate_hidden_dimensions_size,
                act_fn=act_fn, residual=residual, attention=attention, normalize=normalize, coords_agg=coords_agg,
                message_agg=message_agg, tanh=tanh))

    def forward(self, h: torch.Tensor, edges: torch.Tensor, x: torch.Tensor, degree=None, embedded: bool = False,
                first_proj: Optional[torch.Tensor] = None, classify: bool = True) -> AXL:
        """degree: None (a caller's own edge list, any order), the edge count per node [n_nodes] of a list sorted by source,
        or the triple (degree, offsets, n_edges) of a capacity-sized list (utils/neighbors.get_edges_static).
        embedded: h is already embedding_in(node features) (kernels.egnn_node_inputs); first_proj: the first graph layer's
        per-node projections [n_nodes, 2H] of that h, when the caller has them.  classify=False: A is the last layer's h (the
        caller applies node_classification_layer: kernels.egnn_outputs) and L is None."""
        emb = self.embedding_in
        if embedded:
            assert h.shape[1] == emb.out_features
        elif h.is_cuda and emb.in_features <= 8:
            # [sigma | one-hot type] -> hidden: with 2-4 input features the library GEMM spends 0.45 ms on a K = 3
            # problem; as rank-1 updates it is a few elementwise passes over [n_nodes, hidden]
            out = emb.bias.unsqueeze(0) + h[:, :1] * emb.weight[:, 0].unsqueeze(0)
            for k in range(1, emb.in_features):
                out = torch.addcmul(out, h[:, k:k + 1], emb.weight[:, k].unsqueeze(0))
            h = out
        else:
            h = emb(h)
        if degree is None:
            # a caller's own edge list: any order is accepted (as by the reference's unsorted_segment_sum); the segment
            # kernels need the edges grouped by source, so sort them (stable: the order within a node is kept)
            edges = edges[torch.argsort(edges[:, 0], stable=True)]
            degree = torch.bincount(edges[:, 0], minlength=h.shape[0])
        n_edges = None
        if isinstance(degree, tuple):
            degree, offsets, n_edges = degree
        else:
            offsets = (torch.cumsum(degree, 0) - degree) if h.is_cuda else None
        proj = first_proj
        for k, layer in enumerate(self.graph_layers):
            following = self.graph_layers[k + 1] if k + 1 < len(self.graph_layers) else None
            h, x = layer(h, edges, x, degree, offsets, n_edges, node_proj=proj, next_layer=following)
            proj = layer.__dict__.pop("_next_proj", None)       # left there by the layer's node kernel, if it computed them
        if not classify:
            return AXL(A=h, X=x, L=None)
        return AXL(A=self.node_classification_layer(h), X=x, L=torch.zeros_like(x))
