"""Importable alias for the package directory ``psychoacoustic-adverserial-attacks_amd/``.

The product package directory carries the upstream project's (hyphenated) name, which Python
cannot import directly; this alias points its ``__path__`` at that directory, so
``import paa_amd.core.projections`` loads
``psychoacoustic-adverserial-attacks_amd/core/projections.py``.
"""
import os as _os

_REAL = _os.path.join(_os.path.dirname(_os.path.dirname(_os.path.abspath(__file__))),
                      "psychoacoustic-adverserial-attacks_amd")
__path__ = [_REAL]
with open(_os.path.join(_REAL, "__init__.py")) as _f:
    exec(compile(_f.read(), _os.path.join(_REAL, "__init__.py"), "exec"))
