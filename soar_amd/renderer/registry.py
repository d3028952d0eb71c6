"""Minimal stand-in for the threestudio plugin registry (the `submodules/threestudio` directory of the reference is an
empty submodule).  Used ONLY when the real package cannot be imported; the surface is what the SOAR extension touches:
``threestudio.register(name)``, ``threestudio.find(name)``, ``threestudio.info(msg)`` and a ``BaseObject`` whose
constructor parses ``cfg`` into the nested ``Config`` dataclass and calls ``configure()``."""
from __future__ import annotations

import dataclasses
import logging
from typing import Any, Dict, Optional

try:                                    # pragma: no cover - not installed in this image
    import threestudio as _ts
    register, find, info = _ts.register, _ts.find, _ts.info
    from threestudio.utils.base import BaseObject
    HAVE_THREESTUDIO = True
except Exception:
    HAVE_THREESTUDIO = False
    __modules__: Dict[str, Any] = {}
    _log = logging.getLogger("soar_amd")

    def register(name: str):
        def deco(cls):
            __modules__[name] = cls
            return cls
        return deco

    def find(name: str):
        return __modules__[name]

    def info(msg: str):
        _log.info(msg)

    class BaseObject:
        @dataclasses.dataclass
        class Config:
            pass

        cfg: "BaseObject.Config"

        def __init__(self, cfg: Optional[dict] = None, *args, **kwargs):
            fields = {f.name for f in dataclasses.fields(self.Config)}
            cfg = dict(cfg or {})
            unknown = set(cfg) - fields
            if unknown:
                raise ValueError(f"unknown config keys for {type(self).__name__}: {sorted(unknown)}")
            self.cfg = self.Config(**cfg)
            self.device = kwargs.pop("device", None)
            self.configure(*args, **kwargs)

        def configure(self, *args, **kwargs) -> None:
            pass
