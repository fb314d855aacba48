"""src/.../utils/tensor_utils.py: per-structure values or matrices viewed over every other dimension of a batch (views, no copy)."""
from typing import Tuple

import torch


def broadcast_batch_tensor_to_all_dimensions(batch_values: torch.Tensor, final_shape: Tuple[int, ...]) -> torch.Tensor:
    """[B] -> a view of shape final_shape = [B, n1, n2, ...] whose entries depend on the batch index alone (:6-40)."""
    assert batch_values.dim() == 1, "The batch values should be a one-dimensional tensor."
    assert final_shape[0] == batch_values.shape[0], "The final shape should have the batch_size as its first dimension."
    return batch_values.view(-1, *([1] * (len(final_shape) - 1))).expand(*final_shape)


def broadcast_batch_matrix_tensor_to_all_dimensions(batch_values: torch.Tensor, final_shape: Tuple[int, ...]) -> torch.Tensor:
    """[B, m1, m2] -> a view of shape [*final_shape, m1, m2] (final_shape = [B, n1, n2, ...], the matrix dimensions excluded)
    whose matrices depend on the batch index alone (:43-83)."""
    assert batch_values.dim() == 3, "The batch values should be a three-dimensional tensor."
    assert final_shape[0] == batch_values.shape[0], "The final shape should have the batch_size as its first dimension."
    m1, m2 = batch_values.shape[-2:]
    return batch_values.view(-1, *([1] * (len(final_shape) - 1)), m1, m2).expand(*final_shape, m1, m2)
