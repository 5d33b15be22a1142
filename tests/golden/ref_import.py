"""Import recipe for the reference tree (build container only).

`import autorag_research` needs third-party modules this image does not have (tiktoken, hydra, langchain_core,
pgvector, tenacity, ...).  `import_reference()` registers empty stand-in modules for them -- stand-ins for ABSENT
DEPENDENCIES, nothing of the reference itself -- and puts /root/reference on sys.path.  Used by make_golden.py and by
the build-container-only tests; nothing here runs on the GPU box (no /root/reference there).
"""

from __future__ import annotations

import sys
import types

REFERENCE_ROOT = "/root/reference"
STUBBED = ["tiktoken", "evaluate", "hydra", "hydra.utils", "langchain_core", "langchain_core.embeddings",
           "langchain_core.language_models", "nltk", "omegaconf", "pgvector", "pgvector.sqlalchemy", "rouge_score",
           "rouge_score.rouge_scorer", "sacrebleu", "sacrebleu.metrics", "sacrebleu.metrics.bleu", "tenacity",
           "psycopg", "dotenv"]


def _stub(name: str) -> None:
    m = types.ModuleType(name)
    m.__path__ = []  # type: ignore[attr-defined]
    m.__getattr__ = lambda n: type(n, (object,), {  # type: ignore[assignment]
        "__init__": lambda s, *a, **k: None,
        "__class_getitem__": classmethod(lambda c, i: c),
    })
    sys.modules[name] = m


def import_reference() -> None:
    """Make `import autorag_research...` work in this process (idempotent)."""
    if getattr(import_reference, "_done", False):
        return
    sys.dont_write_bytecode = True
    try:  # before the stand-ins exist: torch's import walks sys.modules with `inspect`, which trips over their __getattr__
        import torch  # noqa: F401
    except ImportError:
        pass
    try:  # ... and transformers probes optional packages (nltk, evaluate, ...) by name when a model class is first imported
        from transformers import AutoModel, BertModel, ColPaliForRetrieval  # noqa: F401
    except Exception:  # noqa: BLE001 - only the embedding fixtures need it
        pass
    if REFERENCE_ROOT not in sys.path:
        sys.path.insert(0, REFERENCE_ROOT)
    for n in STUBBED:
        if n not in sys.modules:
            _stub(n)
    from sqlalchemy.types import UserDefinedType

    class _Vector(UserDefinedType):
        cache_ok = True

        def __init__(self, dim=None):
            self.dim = dim

        def get_col_spec(self, **kw):
            return f"VECTOR({self.dim})"

    sys.modules["pgvector.sqlalchemy"].Vector = _Vector  # type: ignore[attr-defined]
    import_reference._done = True  # type: ignore[attr-defined]
