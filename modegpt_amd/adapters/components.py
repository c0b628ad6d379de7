"""What adapters hand to the compressors: the nn.Linear modules of one layer (*Components) or their weight
tensors (*Tensors).  Same names and fields as the reference's carriers (src/adapters/model_adapter.py:19-82)."""
from __future__ import annotations

from dataclasses import dataclass
from typing import Optional

import torch.nn as nn
from torch import Tensor


@dataclass
class MLPTensors:
    up_proj: Tensor
    down_proj: Tensor
    gate_proj: Optional[Tensor]

    def to(self, dtype):
        self.up_proj = self.up_proj.to(dtype=dtype)
        self.down_proj = self.down_proj.to(dtype=dtype)
        if self.gate_proj is not None:
            self.gate_proj = self.gate_proj.to(dtype=dtype)
        return self


@dataclass
class VOTensors:
    v_proj: Tensor
    o_proj: Tensor

    def to(self, dtype):
        self.v_proj, self.o_proj = self.v_proj.to(dtype=dtype), self.o_proj.to(dtype=dtype)
        return self


@dataclass
class QKTensors:
    query_proj: Tensor
    key_proj: Tensor

    def to(self, dtype):
        self.query_proj, self.key_proj = self.query_proj.to(dtype=dtype), self.key_proj.to(dtype=dtype)
        return self


@dataclass
class MLPComponents:
    block: Optional[nn.Module]
    up_proj: nn.Module
    down_proj: nn.Module
    gate_proj: Optional[nn.Module] = None


@dataclass
class QKComponents:
    block: Optional[nn.Module]
    query_proj: nn.Module
    key_proj: nn.Module


@dataclass
class VOComponents:
    block: Optional[nn.Module]
    v_proj: nn.Module
    o_proj: nn.Module


@dataclass
class AttentionComponents:
    block: nn.Module
    q_proj: nn.Module
    k_proj: nn.Module
    v_proj: Optional[nn.Module] = None
    o_proj: Optional[nn.Module] = None


