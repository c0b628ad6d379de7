"""The run's argument surface.  One table (`_SPEC`) drives three things: the dataclass fields, the argparse flags
(`--<name>` for every public field, booleans as store_true) and the help text.  Names, types and defaults are the
contract a user of the reference relies on (reference: src/adapters/CompressionConfig.py:8-35); the mechanics are
this file's own."""
from __future__ import annotations

import argparse
import dataclasses
from typing import Any, Dict, Optional

# (name, python type, default, optional?, help)
_SPEC = (
    ("model", str, "facebook/opt-6.7b", False, "HF model name or checkpoint directory to compress"),
    ("device", int, 0, False, "CUDA(HIP) device ordinal"),
    ("factorize_src_model", str, "", False, "legacy, unused by the llama/qwen/opt paths"),
    ("nystrom_src_model", str, "", False, "legacy, unused by the llama/qwen/opt paths"),
    ("tokenizer_src", str, "mistralai/Mixtral-8x7B-v0.1", False, "tokenizer to load when the checkpoint has none"),
    ("output_dir", str, "compressed_output", False, "where <output_dir>/model is written"),
    ("temp_storage_dir", str, "./compressed_output/layers/", False, "per-(layer, stage) artefacts layer_<i>_<stage>"),
    ("dataset", str, "wikitext", False, "calibration / perplexity corpus: wikitext | c4 | alpaca | synthetic"),
    ("nystrom_ridge", float, 1e-2, False, "lambda of the MLP ridge-leverage scores (added as its fp32 rounding)"),
    ("order", str, None, True, "mlp,qk,vo  -- <method>,<method>,<method>"),
    ("calib_size", int, 32, False, "number of 2048-token calibration samples"),
    ("calibs_batch_size", int, 4, False, "samples per calibration forward pass"),
    ("compression_ratio", float, 0.5, False, "average fraction of each module removed"),
    ("note", str, "NA", False, "free text stored with the run's metrics"),
    ("max_sparsity", float, 0.8, False, "per-layer cap of the allocated sparsity"),
    ("sparsity_smoothing", float, 0.15, False, "softmax temperature of the Block-Influence allocation"),
    ("ridge_vo", float, 1e-4, False, "eigenvalue ridge of sqrt(Sigma_x) in the VO stage"),
    ("ridge_qk", float, 1e-6, False, "eigenvalue ridge of sqrt(Sigma_k) in the grouped QK stage"),
    ("debug", bool, False, False, "verbose diagnostics"),
)


class _DictLike:
    """config.get('x', d) / config['x'] / 'x' in config / config.to_dict(), as callers of the reference use them."""

    def get(self, key: str, default=None):
        value = getattr(self, key, default)
        return default if value is None else value

    def __getitem__(self, key: str):
        return getattr(self, key)

    def __contains__(self, key: str) -> bool:
        return hasattr(self, key)

    def to_dict(self) -> Dict[str, Any]:
        return {f.name: getattr(self, f.name) for f in dataclasses.fields(self)}


def _add_flags(parser: argparse.ArgumentParser) -> argparse.ArgumentParser:
    for name, tp, default, _optional, text in _SPEC:
        if tp is bool:
            parser.add_argument("--" + name, action="store_true", default=default, help=text)
        else:
            parser.add_argument("--" + name, type=tp, default=default, help=text)
    return parser


def _make_parser(cls, parser=None):
    return _add_flags(parser or argparse.ArgumentParser())


def _from_args(cls, args=None):
    known = vars(_make_parser(cls).parse_args(args))
    return cls(**{name: known[name] for name, *_ in _SPEC})


CompressionConfig = dataclasses.make_dataclass(
    "CompressionConfig",
    [(name, Optional[tp] if optional else tp, dataclasses.field(default=default))
     for name, tp, default, optional, _ in _SPEC]
    + [("_parser_extras", Optional[dict], dataclasses.field(default=None, init=False, repr=False, compare=False))],
    bases=(_DictLike,),
    namespace={"make_parser": classmethod(_make_parser), "from_args": classmethod(_from_args),
               "_FIELD_HELP": {name: text for name, *_rest, text in _SPEC}},
)
CompressionConfig.__module__ = __name__
CompressionConfig.__doc__ = "Compression run configuration (see _SPEC for fields, defaults and flags)."
