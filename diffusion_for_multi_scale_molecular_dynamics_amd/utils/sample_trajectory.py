"""Trajectory recorder with the reference's on-disk format (src/.../utils/sample_trajectory.py:7-44)."""
from collections import defaultdict
from typing import Any, Dict, NamedTuple, Union

import torch


class SampleTrajectory:
    def __init__(self):
        self._internal_data = defaultdict(list)

    def reset(self):
        self._internal_data = defaultdict(list)

    def record(self, key: str, entry: Union[Dict[str, Any], NamedTuple]):
        self._internal_data[key].append(entry)

    def write_to_pickle(self, path_to_pickle: str):
        data = dict(self._internal_data)
        for key, value in data.items():
            if len(value) == 1:
                data[key] = value[0]
        self._internal_data = data
        with open(path_to_pickle, "wb") as fd:
            torch.save(data, fd)
