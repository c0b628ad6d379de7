"""Drop-in shim: `python -m src.run_modegpt ...` (the reference's command line) runs the modegpt_amd engine, and
`from src.calibration import load_calibs` etc. resolve to the modegpt_amd modules of the same name."""
import importlib
import sys

_ALIASES = {
    "src.run_modegpt": "modegpt_amd.run_modegpt",
    "src.calibration": "modegpt_amd.calibration",
    "src.compression_utils": "modegpt_amd.compression_utils",
    "src.model_utils": "modegpt_amd.model_utils",
    "src.eval": "modegpt_amd.eval",
    "src.compression": "modegpt_amd.compression",
    "src.compression.compress_mlp": "modegpt_amd.compression.compress_mlp",
    "src.compression.compress_qk": "modegpt_amd.compression.compress_qk",
    "src.compression.compress_vo": "modegpt_amd.compression.compress_vo",
    "src.adapters": "modegpt_amd.adapters",
    "src.adapters.CompressionConfig": "modegpt_amd.adapters.CompressionConfig",
    "src.adapters.model_adapter": "modegpt_amd.adapters.model_adapter",
    "src.adapters.LlamaAdapter": "modegpt_amd.adapters.LlamaAdapter",
    "src.adapters.QwenAdapter": "modegpt_amd.adapters.QwenAdapter",
    "src.adapters.OPTAdapter": "modegpt_amd.adapters.OPTAdapter",
}


class _AliasFinder:
    def find_spec(self, name, path=None, target=None):
        real = _ALIASES.get(name)
        if real is None:
            return None
        mod = importlib.import_module(real)
        sys.modules[name] = mod
        return importlib.util.spec_from_loader(name, loader=_AliasLoader(mod))


class _AliasLoader:
    def __init__(self, mod):
        self.mod = mod

    def create_module(self, spec):
        return self.mod

    def exec_module(self, module):
        pass


import importlib.util  # noqa: E402

sys.meta_path.insert(0, _AliasFinder())
