"""The D3PM helpers a reference-style plugin imports (src/.../utils/d3pm_utils.py:7-150), on device tensors.

The sampler itself never calls them: its posterior p(a_{t-1} | a_t, logits) is computed inside the fused update kernel
(mdx_pc_step_update / mdx_atom_types_update, include/mdx_hip.h).  get_probability_at_previous_time_step routes to that kernel
when its operands are the sampler's (logits, a strict one-hot a_t, ONE matrix triple for the whole batch); any other operands
-- distributions instead of one-hot vectors, matrices that differ from atom to atom -- are evaluated with the reference's
own contractions.
"""
import torch


def class_index_to_onehot(index: torch.Tensor, num_classes: int) -> torch.Tensor:
    """float tensor of 0s and 1s of shape index.shape + (num_classes,)  (:7-20)"""
    return torch.nn.functional.one_hot(index.long(), num_classes=num_classes).to(device=index.device, dtype=torch.float)


def compute_q_at_given_a0(one_hot_a0: torch.Tensor, q_bar_t: torch.Tensor) -> torch.Tensor:
    """q(a_t | a_0) = a_0 Qbar_t  (:23-39)"""
    return torch.einsum("...j,...ji->...i", one_hot_a0.to(q_bar_t), q_bar_t)


def compute_q_at_given_atm1(one_hot_atm1: torch.Tensor, q_tm1: torch.Tensor) -> torch.Tensor:
    """q(a_t | a_{t-1}) = a_{t-1} Q_{t-1}  (:42-61: the product with the transposed matrix's transpose)"""
    return torch.einsum("...j,...ij->...i", one_hot_atm1.to(q_tm1), torch.transpose(q_tm1, -2, -1))


def get_probability_from_logits(logits: torch.Tensor, lowest_probability_value: float) -> torch.Tensor:
    """softmax, every class at least lowest_probability_value, renormalised  (:127-150)"""
    raw_probabilities = torch.nn.functional.softmax(logits, dim=-1)
    clipped_probabilities = raw_probabilities.clip(min=lowest_probability_value)
    return clipped_probabilities / clipped_probabilities.sum(dim=-1).unsqueeze(-1)


def _one_matrix(m: torch.Tensor):
    """The [C, C] matrix every atom shares, or None when the leading dimensions are not a broadcast of one matrix."""
    if m.dim() == 2:
        return m
    lead = m.shape[:-2]
    if all(s == 0 for s in m.stride()[:len(lead)]):            # an expand() of one matrix: what the sampler builds
        return m.reshape(-1, m.shape[-2], m.shape[-1])[0]
    return None


def get_probability_at_previous_time_step(probability_at_zeroth_timestep: torch.Tensor,
                                          one_hot_probability_at_current_timestep: torch.Tensor, q_matrices: torch.Tensor,
                                          q_bar_matrices: torch.Tensor, q_bar_tm1_matrices: torch.Tensor, small_epsilon: float,
                                          probability_at_zeroth_timestep_are_logits: bool = False) -> torch.Tensor:
    """P(a_{t-1} | a_t, gamma_0) = (gamma_0 Qbar_{t-1})_i (Q_t a_t)_i / (gamma_0 Qbar_t a_t)  (:64-124)"""
    p0, a_t = probability_at_zeroth_timestep, one_hot_probability_at_current_timestep
    shared = [_one_matrix(m) for m in (q_matrices, q_bar_matrices, q_bar_tm1_matrices)]
    if (probability_at_zeroth_timestep_are_logits and p0.is_cuda and p0.dim() == 3 and p0.dtype == torch.float32
            and all(m is not None for m in shared) and not torch.is_grad_enabled()):
        from .. import kernels
        indices = a_t.argmax(dim=-1)
        if bool((class_index_to_onehot(indices, a_t.shape[-1]) == a_t.to(torch.float)).all()):
            zeros = torch.zeros_like(p0)
            _, probabilities = kernels.atom_types_update(p0.contiguous(), indices.contiguous(),
                                                         *[m.to(p0).contiguous() for m in shared], zeros, None, small_epsilon,
                                                         False, False, return_probabilities=True)
            return probabilities
    if probability_at_zeroth_timestep_are_logits:
        p0 = get_probability_from_logits(p0, lowest_probability_value=small_epsilon)
    numerator1 = torch.einsum("...j,...ji->...i", p0, q_bar_tm1_matrices)
    numerator2 = torch.einsum("...ij,...j->...i", q_matrices, a_t)
    den1 = torch.einsum("...ij,...j->...i", q_bar_matrices, a_t)
    den2 = torch.einsum("...j,...j->...", p0, den1)
    return numerator1 * numerator2 / den2.unsqueeze(-1)
