"""Importable alias of the hyphen-named package directory ``autorag-research_amd/``.

Python cannot import a directory whose name contains '-', so this one-file package re-points its
``__path__`` at the real directory and runs that directory's ``__init__``.  All code lives there.
"""

from pathlib import Path as _Path

_real = _Path(__file__).resolve().parent.parent / "autorag-research_amd"
__path__ = [str(_real)]
exec(compile((_real / "__init__.py").read_text(), str(_real / "__init__.py"), "exec"))
