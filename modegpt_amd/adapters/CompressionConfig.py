"""CLI / argument surface of the compression run -- field for field the reference's
src/adapters/CompressionConfig.py:6-95 (names, types, defaults, --flag mapping, dict-style access)."""
from __future__ import annotations

import argparse
import dataclasses
import typing
from dataclasses import dataclass, field
from typing import Optional


@dataclass
class CompressionConfig:
    model: str = "facebook/opt-6.7b"
    device: int = 0
    factorize_src_model: str = ""
    nystrom_src_model: str = ""
    tokenizer_src: str = "mistralai/Mixtral-8x7B-v0.1"
    output_dir: str = "compressed_output"
    temp_storage_dir: str = "./compressed_output/layers/"
    dataset: str = "wikitext"
    nystrom_ridge: float = 1e-2
    order: Optional[str] = None
    calib_size: int = 32
    calibs_batch_size: int = 4
    compression_ratio: float = 0.5
    note: str = "NA"
    max_sparsity: float = 0.8
    sparsity_smoothing: float = 0.15
    ridge_vo: float = 1e-4
    ridge_qk: float = 1e-6
    debug: bool = False

    _parser_extras: dict = field(default=None, init=False, repr=False, compare=False)

    _FIELD_HELP = {"order": "mlp,qk,vo  -- <method>,<method>,<method>"}

    @classmethod
    def _resolve_type(cls, tp):
        """Optional[X] -> X; plain types unchanged."""
        if isinstance(tp, str):  # `from __future__ import annotations` keeps annotations as strings
            tp = typing.get_type_hints(cls).get(tp, None) or eval(tp, vars(typing) | {"Optional": Optional})
        if typing.get_origin(tp) is None:
            return tp
        inner = [a for a in typing.get_args(tp) if a is not type(None)]
        return inner[0] if inner else str

    @classmethod
    def make_parser(cls, parser=None):
        parser = parser or argparse.ArgumentParser()
        hints = typing.get_type_hints(cls)
        for f in dataclasses.fields(cls):
            if f.name.startswith("_"):
                continue
            tp = cls._resolve_type(hints[f.name])
            if tp is bool:
                parser.add_argument(f"--{f.name}", action="store_true", default=f.default)
                continue
            kw = {"type": tp}
            if f.default is not dataclasses.MISSING:
                kw["default"] = f.default
            else:
                kw["required"] = True
            if f.name in cls._FIELD_HELP:
                kw["help"] = cls._FIELD_HELP[f.name]
            parser.add_argument(f"--{f.name}", **kw)
        return parser

    @classmethod
    def from_args(cls, args=None):
        ns = cls.make_parser().parse_args(args)
        names = {f.name for f in dataclasses.fields(cls) if f.init}
        return cls(**{k: v for k, v in vars(ns).items() if k in names})

    def get(self, key: str, default=None):
        val = getattr(self, key, default)
        return default if val is None else val

    def __getitem__(self, key: str):
        return getattr(self, key)

    def __contains__(self, key: str):
        return hasattr(self, key)

    def to_dict(self) -> dict:
        return {f.name: getattr(self, f.name) for f in dataclasses.fields(self)}
