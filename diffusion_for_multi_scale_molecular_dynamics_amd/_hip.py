"""ctypes binding of csrc/libmdx_hip.so (the C ABI declared in include/mdx_hip.h).

There is NO fallback: if the shared library is missing, or a tensor is not a contiguous device tensor of the
expected dtype, the call raises.  PyTorch is used only as the owner of device memory and streams.
"""
import ctypes as C
import os
import subprocess

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(_HERE, "csrc")
LIB_PATH = os.path.join(CSRC, "libmdx_hip.so")

MDX_OK = 0
MDX_PREDICTOR, MDX_CORRECTOR = 0, 1
STATUS_CUTOFF_TOO_LARGE, STATUS_MASK_AT_LAST_STEP, STATUS_EGNN_F16_RANGE, STATUS_GRAPH_CAPACITY = 1, 2, 4, 8
EGNN_COORD_NORMALIZE, EGNN_COORD_TANH = 1, 2      # MDX_EGNN_COORD_* (coord_flags of mdx_egnn_node_gather / _coord_aggregate)
EGNN_CHAIN_MAX_LAYERS = 16
MAX_CLASSES = 8
TAG_COORD, TAG_GUMBEL, TAG_LATTICE, TAG_INIT, TAG_REPAINT_X0, TAG_BINARY, TAG_REPAINT_Z, TAG_REPAINT_U, \
    TAG_INIT_LATTICE, TAG_RESAMPLE_Z, TAG_RESAMPLE_U = range(11)

ABI_VERSION = 14         # MDX_ABI_VERSION of include/mdx_hip.h
ABI_SYMBOLS = (
    "mdx_abi_version", "mdx_status_string", "mdx_noise_schedule_build", "mdx_index_set", "mdx_index_add",
    "mdx_fill_time_sigma", "mdx_relative_coordinates_update", "mdx_lattice_parameters_update",
    "mdx_relative_coordinates_update_dev", "mdx_lattice_parameters_update_dev",
    "mdx_atom_types_update", "mdx_pc_step_update", "mdx_noise_relative_coordinates", "mdx_noise_atom_types", "mdx_noise_relative_coordinates_sigmas", "mdx_noise_atom_types_per_atom",
    "mdx_noise_lattice_parameters",
    "mdx_repaint_constrained_rows", "mdx_forward_diffusion_step", "mdx_radius_graph_count", "mdx_radius_graph_fill", "mdx_radius_graph_fill_capped", "mdx_egnn_radius_graph", "mdx_egnn_radius_graph_workspace_words", "mdx_mlp_forward",
    "mdx_mlp_pc_sample", "mdx_mlp_pc_sample_variant", "mdx_mlp_pc_sample_workspace_floats", "mdx_mlp_image_floats", "mdx_mlp_pack_image", "mdx_egnn_message_input", "mdx_egnn_coord_head", "mdx_segment_rows",
    "mdx_egnn_chain_image_bytes", "mdx_egnn_chain_pack", "mdx_egnn_chain_adapt_activation_exponents", "mdx_egnn_edge_chain", "mdx_egnn_piece_rows", "mdx_segment_combine", "mdx_egnn_node_gather", "mdx_mlp_chain_rows", "mdx_egnn_coord_aggregate",
    "mdx_egnn_node_inputs", "mdx_egnn_scores", "mdx_egnn_outputs", "mdx_node_mlp_rows", "mdx_node_mlp_rows_split",
    "mdx_rng_fill", "mdx_math_probe",
)
MLP_MAX_HIDDEN = 8
# options of mdx_mlp_pc_sample (include/mdx_hip.h)
MLP_SAMPLE_GENERIC_KERNEL, MLP_SAMPLE_UNFOLDED, MLP_SAMPLE_CALLER_NOISE, MLP_SAMPLE_NO_FIXED_SOFTMAX, \
    MLP_SAMPLE_NO_P2_TABLE, MLP_SAMPLE_DIAG_NO_FORWARD, MLP_SAMPLE_DIAG_NO_UPDATE = 1, 2, 4, 8, 16, 256, 512
MLP_SAMPLE_PADDED_FAMILY = 128


class MdxError(RuntimeError):
    """A C-ABI call returned a negative status."""


class EdgeChainRangeError(MdxError):
    """MDX_STATUS_EGNN_F16_RANGE: the split-f16 edge chain met an activation beyond the f16 range; the call's results are
    invalid and must be recomputed with edge_chain_precision='f32'."""


class Schedule(C.Structure):
    """mdx_schedule_t"""
    _fields_ = [("total_time_steps", C.c_int32), ("num_classes", C.c_int32), ("sigma_min", C.c_double),
                ("time", C.c_void_p), ("sigma", C.c_void_p), ("g", C.c_void_p), ("g_squared", C.c_void_p),
                ("epsilon", C.c_void_p), ("q_matrix", C.c_void_p), ("q_bar_matrix", C.c_void_p),
                ("q_bar_tm1_matrix", C.c_void_p)]


class Rng(C.Structure):
    """mdx_rng_t"""
    _fields_ = [("seed", C.c_uint64), ("call", C.c_uint32), ("draw_stride", C.c_uint32),
                ("draw_offset", C.c_uint32), ("reserved", C.c_uint32), ("call_dev", C.c_void_p)]


class PcFlags(C.Structure):
    """mdx_pc_flags_t"""
    _fields_ = [("atom_type_greedy_sampling", C.c_int32), ("one_atom_type_transition_per_step", C.c_int32),
                ("use_fixed_lattice_parameters", C.c_int32), ("update_atom_types", C.c_int32),
                ("small_epsilon", C.c_float)]


class Mlp(C.Structure):
    """mdx_mlp_t"""
    _fields_ = [(n, C.c_int32) for n in ("number_of_atoms", "spatial_dimension", "num_classes", "hidden_size",
                                         "n_hidden", "e_coordinates", "e_noise", "e_time", "e_atom_type", "e_lattice")] + \
        [(n, C.c_void_p) for n in ("w_coordinates_t", "b_coordinates", "w_noise_t", "b_noise", "w_time_t", "b_time",
                                   "w_atom_type_t", "b_atom_type", "w_lattice_t", "b_lattice")] + \
        [("w_hidden_t", C.c_void_p * 8), ("b_hidden", C.c_void_p * 8)] + \
        [(n, C.c_void_p) for n in ("w_out_a_t", "b_out_a", "w_out_x_t", "b_out_x", "w_out_l_t", "b_out_l",
                                   "packed_image", "folded_input", "folded_output", "folded_padded")]


class EgnnChain(C.Structure):
    """mdx_egnn_chain_t"""
    _fields_ = [(n, C.c_int32) for n in ("hidden", "n_message_layers", "n_coord_layers", "precision", "message_mode",
                                         "reserved")] + \
        [(n, C.c_void_p) for n in ("weight_image", "biases", "bias_in", "w_radial", "weight_exponents",
                                   "activation_exponents", "activation_maxima", "attention_weight", "attention_bias")]


def build(force=False):
    """Compile csrc/mdx_hip.hip for gfx950 with hipcc (cross-compiles without a GPU)."""
    srcs = [os.path.join(CSRC, f) for f in sorted(os.listdir(CSRC)) if f.endswith((".hip", ".hpp", ".h")) or f == "Makefile"]
    srcs.append(os.path.join(os.path.dirname(_HERE), "include", "mdx_hip.h"))
    stale = not os.path.exists(LIB_PATH) or any(os.path.getmtime(LIB_PATH) < os.path.getmtime(s) for s in srcs)
    if force or stale:
        subprocess.check_call(["make", "-s", "-j4", "-C", CSRC, "-B"])
    return LIB_PATH


_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise MdxError(
                f"{LIB_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                f"(or `make -C {CSRC}`). There is no CPU fallback for the sampling hot path.")
        L = C.CDLL(LIB_PATH)
        _declare(L)
        if L.mdx_abi_version() != ABI_VERSION:
            raise MdxError("libmdx_hip.so ABI version mismatch")
        _lib = L
    return _lib


def _declare(L):
    vp, i32, i64, f32, f64, u32, u64 = C.c_void_p, C.c_int, C.c_int64, C.c_float, C.c_double, C.c_uint32, C.c_uint64
    L.mdx_abi_version.restype = i32
    L.mdx_abi_version.argtypes = []
    L.mdx_status_string.restype = C.c_char_p
    L.mdx_status_string.argtypes = [i32]
    L.mdx_noise_schedule_build.restype = i32
    L.mdx_noise_schedule_build.argtypes = [i32, i32, f64, f64, f64, f64, i32] + [vp] * 12 + [vp]
    L.mdx_index_set.restype = i32
    L.mdx_index_set.argtypes = [vp, C.c_int32, vp]
    L.mdx_index_add.restype = i32
    L.mdx_index_add.argtypes = [vp, C.c_int32, vp]
    L.mdx_fill_time_sigma.restype = i32
    L.mdx_fill_time_sigma.argtypes = [C.POINTER(Schedule), i32, i32, vp, vp, vp, i64, vp]
    L.mdx_relative_coordinates_update.restype = i32
    L.mdx_relative_coordinates_update.argtypes = [vp, vp, vp, f32, f32, f32, i64, vp, vp]
    L.mdx_lattice_parameters_update.restype = i32
    L.mdx_lattice_parameters_update.argtypes = [vp, vp, vp, f32, f32, f32, i64, vp, vp]
    L.mdx_relative_coordinates_update_dev.restype = i32
    L.mdx_relative_coordinates_update_dev.argtypes = [vp, vp, vp, vp, i64, vp, vp]
    L.mdx_lattice_parameters_update_dev.restype = i32
    L.mdx_lattice_parameters_update_dev.argtypes = [vp, vp, vp, vp, i64, vp, vp]
    L.mdx_atom_types_update.restype = i32
    L.mdx_atom_types_update.argtypes = [vp] * 7 + [i64, i32, i32, f32, i32, i32, vp, vp, vp]
    L.mdx_pc_step_update.restype = i32
    L.mdx_pc_step_update.argtypes = [C.POINTER(Schedule), i32, i32, vp, C.POINTER(PcFlags)] + [vp] * 10 + \
        [Rng, i64, i32, i32, vp, vp, vp, vp, vp]
    L.mdx_noise_relative_coordinates.restype = i32
    L.mdx_noise_relative_coordinates.argtypes = [vp, vp, f32, i64, vp, vp]
    L.mdx_noise_atom_types.restype = i32
    L.mdx_noise_atom_types.argtypes = [vp, vp, vp, i64, i32, vp, vp]
    L.mdx_noise_atom_types_per_atom.restype = i32
    L.mdx_noise_atom_types_per_atom.argtypes = [vp, vp, vp, i64, i32, vp, vp]
    L.mdx_noise_relative_coordinates_sigmas.restype = i32
    L.mdx_noise_relative_coordinates_sigmas.argtypes = [vp, vp, vp, i64, vp, vp]
    L.mdx_noise_lattice_parameters.restype = i32
    L.mdx_noise_lattice_parameters.argtypes = [vp, vp, vp, i64, vp, vp]
    L.mdx_repaint_constrained_rows.restype = i32
    L.mdx_repaint_constrained_rows.argtypes = [C.POINTER(Schedule), i32, vp, vp, vp, vp, i32, vp, vp, Rng, i64, i32,
                                               i32, vp, vp, vp]
    L.mdx_forward_diffusion_step.restype = i32
    L.mdx_forward_diffusion_step.argtypes = [C.POINTER(Schedule), i32, vp, vp, vp, Rng, i64, i32, i32, vp, vp, vp]
    L.mdx_radius_graph_count.restype = i32
    L.mdx_radius_graph_count.argtypes = [vp, vp, f32, i64, i32, i32, vp, vp, vp]
    L.mdx_radius_graph_fill.restype = i32
    L.mdx_radius_graph_fill.argtypes = [vp, vp, f32, i64, i32, i32, vp, vp, vp, vp, vp]
    L.mdx_radius_graph_fill_capped.restype = i32
    L.mdx_radius_graph_fill_capped.argtypes = [vp, vp, f32, i64, i32, i32, vp, i64, vp, vp, vp, vp, vp]
    L.mdx_egnn_radius_graph.restype = i32
    L.mdx_egnn_radius_graph.argtypes = [vp, vp, i32, f32, f32, i64, i32, i64, vp, vp, vp, vp, vp, vp, i64, vp]
    L.mdx_egnn_radius_graph_workspace_words.restype = i64
    L.mdx_egnn_radius_graph_workspace_words.argtypes = [i64, i32]
    L.mdx_mlp_forward.restype = i32
    L.mdx_mlp_forward.argtypes = [C.POINTER(Mlp), vp, vp, vp, vp, vp, i64, vp, vp, vp, vp]
    L.mdx_mlp_pc_sample.restype = i32
    L.mdx_mlp_pc_sample.argtypes = [C.POINTER(Schedule), C.POINTER(Mlp), C.POINTER(PcFlags), i32, i32, i32, i32, Rng, i64,
                                    vp, vp, vp, vp, i64, u32, vp, vp]
    L.mdx_mlp_pc_sample_variant.restype = i32
    L.mdx_mlp_pc_sample_variant.argtypes = [C.POINTER(Mlp), u32]
    L.mdx_mlp_pc_sample_workspace_floats.restype = i64
    L.mdx_mlp_pc_sample_workspace_floats.argtypes = [C.POINTER(Mlp), i32, i32, i32, i64]
    L.mdx_mlp_image_floats.restype = i64
    L.mdx_mlp_image_floats.argtypes = [C.POINTER(Mlp)]
    L.mdx_mlp_pack_image.restype = i32
    L.mdx_mlp_pack_image.argtypes = [C.POINTER(Mlp), vp, vp]
    L.mdx_egnn_message_input.restype = i32
    L.mdx_egnn_message_input.argtypes = [vp, vp, vp, vp, vp, i64, i32, i32, vp, vp]
    L.mdx_egnn_coord_head.restype = i32
    L.mdx_egnn_coord_head.argtypes = [vp, vp, vp, vp, vp, i64, i32, i32, i32, vp, vp]
    L.mdx_segment_rows.restype = i32
    L.mdx_segment_rows.argtypes = [vp, vp, vp, i64, i32, i32, vp, vp]
    L.mdx_egnn_chain_image_bytes.restype = i64
    L.mdx_egnn_chain_image_bytes.argtypes = [i32, i32]
    L.mdx_egnn_chain_pack.restype = i32
    L.mdx_egnn_chain_pack.argtypes = [C.POINTER(vp), i32, vp, i32, i32, C.c_uint32, vp, vp, vp]
    L.mdx_egnn_edge_chain.restype = i32
    L.mdx_egnn_edge_chain.argtypes = [C.POINTER(EgnnChain), vp, vp, i32, vp, i64, vp, vp, vp, vp, vp]
    L.mdx_node_mlp_rows.restype = i32
    L.mdx_node_mlp_rows.argtypes = [vp, vp, i32, i64, vp, vp, vp, vp, vp]
    L.mdx_node_mlp_rows_split.restype = i32
    L.mdx_node_mlp_rows_split.argtypes = [vp, vp, vp, i32, i64, vp, vp, vp, vp, vp]
    L.mdx_egnn_node_inputs.restype = i32
    L.mdx_egnn_node_inputs.argtypes = [vp, vp, i32, vp, i32, vp, vp, vp, i32, i32, i64, vp, vp, vp, vp, i32, vp, vp]
    L.mdx_egnn_scores.restype = i32
    L.mdx_egnn_scores.argtypes = [vp, vp, vp, i32, i64, vp, vp]
    L.mdx_egnn_outputs.restype = i32
    L.mdx_egnn_outputs.argtypes = [vp, vp, vp, i32, vp, vp, vp, i32, i32, i32, i64, vp, vp, vp, i64, vp]
    L.mdx_segment_combine.restype = i32
    L.mdx_segment_combine.argtypes = [vp, i64, vp, vp, i64, i32, i32, vp, vp, vp]
    L.mdx_egnn_node_gather.restype = i32
    L.mdx_egnn_node_gather.argtypes = [vp, i64, vp, vp, i64, i32, i32, vp, vp, vp, vp, i32, vp, i32, i32, vp, vp]
    L.mdx_egnn_chain_adapt_activation_exponents.restype = i32
    L.mdx_egnn_chain_adapt_activation_exponents.argtypes = [vp, i32, vp, vp]
    L.mdx_egnn_piece_rows.restype = i64
    L.mdx_egnn_piece_rows.argtypes = [i64, i64]
    L.mdx_mlp_chain_rows.restype = i32
    L.mdx_mlp_chain_rows.argtypes = [C.POINTER(EgnnChain), vp, vp, i64, vp, vp, vp, vp]
    L.mdx_egnn_coord_aggregate.restype = i32
    L.mdx_egnn_coord_aggregate.argtypes = [vp, vp, i32, vp, vp, vp, i64, i32, i32, vp, vp]
    L.mdx_rng_fill.restype = i32
    L.mdx_rng_fill.argtypes = [i32, u64, u32, u32, u32, i64, i32, vp, vp]
    L.mdx_math_probe.restype = i32
    L.mdx_math_probe.argtypes = [i32, vp, i64, vp, vp]


def check(status, what):
    if status != MDX_OK:
        msg = lib().mdx_status_string(status).decode()
        raise MdxError(f"{what}: {msg} (status {status})")


def stream_handle():
    """The raw hipStream_t of torch's current stream (so launches are captured by torch.cuda.graph)."""
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def ptr(t, dtype, name):
    """Device pointer of a contiguous device tensor of the given dtype; None stays NULL."""
    if t is None:
        return None
    if not isinstance(t, torch.Tensor):
        raise TypeError(f"{name} must be a torch.Tensor")
    if not t.is_cuda:
        raise MdxError(f"{name} lives on {t.device}: the sampling hot path runs on the GPU only (no CPU fallback)")
    if t.dtype != dtype:
        raise TypeError(f"{name} must have dtype {dtype}, got {t.dtype}")
    if not t.is_contiguous():
        raise ValueError(f"{name} must be contiguous")
    return C.c_void_p(t.data_ptr())
