"""A minimal stand-in for ml_collections.ConfigDict (absent from this image): attribute and item
access, `in`, .get, .lock(), .to_dict(), ** unpacking - what the reference's configs/*.py and
train_*.py use (configs/pm_vae_mnist.py:4-50, train_pm_vae.py:46-52,108-109)."""
from __future__ import annotations

from typing import Any, Dict, Iterator


class ConfigDict:
    def __init__(self, initial: Dict[str, Any] = None):
        object.__setattr__(self, "_fields", {})
        object.__setattr__(self, "_locked", False)
        for k, v in (initial or {}).items():
            self[k] = v

    # attribute / item access ------------------------------------------------------------
    def __getattr__(self, name: str) -> Any:
        try:
            return self._fields[name]
        except KeyError:
            raise AttributeError(name)

    def __setattr__(self, name: str, value: Any) -> None:
        self[name] = value

    def __getitem__(self, key: str) -> Any:
        return self._fields[key]

    def __setitem__(self, key: str, value: Any) -> None:
        if self._locked and key not in self._fields:
            raise KeyError(f"config is locked; cannot add new key {key!r}")
        if isinstance(value, dict):
            value = ConfigDict(value)
        self._fields[key] = value

    def __contains__(self, key: str) -> bool:
        return key in self._fields

    def __iter__(self) -> Iterator[str]:
        return iter(self._fields)

    def __len__(self) -> int:
        return len(self._fields)

    def keys(self):
        return self._fields.keys()

    def items(self):
        return self._fields.items()

    def values(self):
        return self._fields.values()

    def get(self, key: str, default: Any = None) -> Any:
        return self._fields.get(key, default)

    def lock(self) -> "ConfigDict":
        object.__setattr__(self, "_locked", True)
        for v in self._fields.values():
            if isinstance(v, ConfigDict):
                v.lock()
        return self

    def to_dict(self) -> Dict[str, Any]:
        return {k: (v.to_dict() if isinstance(v, ConfigDict) else v) for k, v in self._fields.items()}

    def __repr__(self) -> str:
        return f"ConfigDict({self.to_dict()!r})"


def load_config_file(path: str) -> ConfigDict:
    """`--config path/to/file.py`: imports the file and calls its get_config()
    (ml_collections.config_flags.DEFINE_config_file, train_pm_vae.py:25)."""
    import importlib.util

    spec = importlib.util.spec_from_file_location("_pm_config", path)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod.get_config()


def apply_overrides(config: ConfigDict, overrides) -> None:
    """`--config.a.b=value` command-line overrides."""
    import ast

    for item in overrides:
        key, _, raw = item.partition("=")
        parts = key.split(".")
        node = config
        for p in parts[:-1]:
            node = node[p]
        try:
            val = ast.literal_eval(raw)
        except Exception:
            val = raw
        node[parts[-1]] = val
