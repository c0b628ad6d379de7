"""`patch_config(model)` by the reference's module path (src/patchers/patch.py:13-75): stamp the per-layer ranks and the
auto_map entry on model.config and hand back a copy of the config as it was.  The work is the adapter's patch_config."""
from __future__ import annotations

from ..adapters.model_adapter import ModelAdapter


def patch_config(model):
    return ModelAdapter.from_model(model, None).patch_config()
