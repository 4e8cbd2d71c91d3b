"""YAML <-> nested dataclass glue for the training configuration.

The reference gets this from the third-party ``dataclass_wizard.YAMLWizard`` mixin (TRN:33, ADC:52, LOS:10):
``TrainConfig.from_yaml_file(path)`` / ``config.to_yaml_file(path)``, keys written in kebab-case
(``num-iterations``) and accepted in kebab- or snake_case (config/*.yaml mix both), nested dataclasses as nested
mappings, unknown keys ignored (config/tat_truck_every_8_test.yaml carries a misspelt
``position_learning_rateo`` that the reference silently drops).  This module is a small self-contained
replacement with the same observable behaviour; it warns about the keys it drops.
"""
from __future__ import annotations

import dataclasses
import typing
import warnings
from typing import Any, Dict, Type, TypeVar

import yaml

T = TypeVar("T")


def _unwrap_optional(tp):
    if typing.get_origin(tp) is typing.Union:
        args = [a for a in typing.get_args(tp) if a is not type(None)]
        if len(args) == 1:
            return args[0]
    return tp


def _coerce(value: Any, tp) -> Any:
    tp = _unwrap_optional(tp)
    if value is None:
        return None
    if dataclasses.is_dataclass(tp):
        return from_dict(tp, value)
    if tp is float and isinstance(value, (int, str)) and not isinstance(value, bool):
        return float(value)      # "3e-6" is a string for PyYAML's YAML-1.1 float rule
    if tp is int and isinstance(value, (float, str)) and not isinstance(value, bool):
        as_float = float(value)
        return int(as_float) if as_float.is_integer() else as_float   # TRN:43 annotates int, defaults 1000.
    if tp is bool and isinstance(value, str):
        return value.strip().lower() in ("1", "true", "yes", "on")
    return value


def from_dict(cls: Type[T], data: Dict[str, Any]) -> T:
    if data is None:
        return cls()
    if not isinstance(data, dict):
        raise TypeError(f"{cls.__name__}: expected a mapping, got {type(data).__name__}")
    hints = typing.get_type_hints(cls)
    names = {f.name for f in dataclasses.fields(cls) if f.init}
    kwargs = {}
    for raw_key, value in data.items():
        key = str(raw_key).replace("-", "_")
        if key not in names:
            warnings.warn(f"{cls.__name__}: ignoring unknown configuration key '{raw_key}'")
            continue
        kwargs[key] = _coerce(value, hints.get(key, Any))
    return cls(**kwargs)


def to_dict(obj: Any, kebab: bool = True) -> Dict[str, Any]:
    out = {}
    for f in dataclasses.fields(obj):
        value = getattr(obj, f.name)
        if dataclasses.is_dataclass(value):
            value = to_dict(value, kebab)
        out[f.name.replace("_", "-") if kebab else f.name] = value
    return out


class YAMLConfig:
    """Mixin giving a dataclass the four YAMLWizard entry points the reference scripts use."""

    @classmethod
    def from_yaml(cls, text: str):
        return from_dict(cls, yaml.safe_load(text) or {})

    @classmethod
    def from_yaml_file(cls, path: str):
        with open(path) as fh:
            return cls.from_yaml(fh.read())

    def to_yaml(self) -> str:
        return yaml.safe_dump(to_dict(self), sort_keys=True)

    def to_yaml_file(self, path: str) -> None:
        with open(path, "w") as fh:
            fh.write(self.to_yaml())
