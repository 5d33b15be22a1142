"""The slice of AutoRAG-Research's plugin contract this path needs.

When the real `autorag_research` package is importable (a deployment that installs this plugin next
to it) the reference's own classes are re-exported, so `Mi355VectorSearchRetrievalPipeline` IS-A
`autorag_research.pipelines.retrieval.base.BaseRetrievalPipeline` and the unchanged Executor accepts
it.  On a box without the reference (the GPU box, the bench harness) minimal stand-ins with the SAME
names, signatures and behaviour are used instead:

  BasePipelineConfig / BaseRetrievalPipelineConfig   autorag_research/config.py:35-130
  PipelineType                                        autorag_research/config.py:21-25
  BasePipeline                                        autorag_research/pipelines/base.py:7-34
  EmbeddingError                                      autorag_research/exceptions.py:23-29
  require_retrieval_unit / VALID_RETRIEVAL_UNITS      autorag_research/retrieval_units.py:1-33
"""

from __future__ import annotations

from abc import ABC, abstractmethod
from dataclasses import dataclass, field
from enum import Enum
from typing import Any, Literal, cast

try:  # pragma: no cover - exercised only where the reference is installed
    from autorag_research.config import BaseRetrievalPipelineConfig, PipelineType  # type: ignore
    from autorag_research.exceptions import EmbeddingError  # type: ignore
    from autorag_research.pipelines.base import BasePipeline  # type: ignore
    from autorag_research.retrieval_units import (  # type: ignore
        VALID_RETRIEVAL_UNITS,
        RetrievalUnit,
        require_retrieval_unit,
    )

    HAVE_REFERENCE = True
except Exception:  # noqa: BLE001 - any import problem means "reference not usable here"
    HAVE_REFERENCE = False

    RetrievalUnit = Literal["chunk", "image_chunk", "mixed"]  # type: ignore[misc]
    VALID_RETRIEVAL_UNITS: frozenset[str] = frozenset({"chunk", "image_chunk", "mixed"})  # type: ignore[no-redef]

    def require_retrieval_unit(value: object, *, default: "RetrievalUnit | None" = None):  # type: ignore[no-redef]
        """Valid unit -> itself, None -> default, anything else -> ValueError (retrieval_units.py:24-33)."""
        if isinstance(value, str) and value in VALID_RETRIEVAL_UNITS:
            return cast("RetrievalUnit", value)
        if value is None:
            return default
        raise ValueError(f"Invalid retrieval_unit {value!r}. Expected one of: {', '.join(sorted(VALID_RETRIEVAL_UNITS))}.")

    class PipelineType(Enum):  # type: ignore[no-redef]
        RETRIEVAL = "retrieval"
        GENERATION = "generation"

    class EmbeddingError(Exception):  # type: ignore[no-redef]
        """Raised when text retrieval is asked for without an embedding model."""

        def __init__(self):
            super().__init__(
                "We don't know what is wrong, but your Embedding model is criminal. "
                "Scroll up to see the traceback and find your real reason."
            )

    @dataclass
    class BasePipelineConfig(ABC):
        name: str
        description: str = ""
        pipeline_type: PipelineType = field(init=False)
        top_k: int = 10
        batch_size: int = 128
        max_concurrency: int = 16
        max_retries: int = 3
        retry_delay: float = 1.0

        @abstractmethod
        def get_pipeline_class(self) -> type: ...

        @abstractmethod
        def get_pipeline_kwargs(self) -> dict[str, Any]: ...

        @abstractmethod
        def get_run_kwargs(self) -> dict[str, Any]: ...

    @dataclass
    class BaseRetrievalPipelineConfig(BasePipelineConfig, ABC):  # type: ignore[no-redef]
        pipeline_type: PipelineType = field(default=PipelineType.RETRIEVAL, init=False)

        def get_run_kwargs(self) -> dict[str, Any]:
            return {
                "top_k": self.top_k,
                "batch_size": self.batch_size,
                "max_concurrency": self.max_concurrency,
                "max_retries": self.max_retries,
                "retry_delay": self.retry_delay,
            }

    class BasePipeline(ABC):  # type: ignore[no-redef]
        def __init__(self, session_factory: Any, name: str, schema: Any | None = None):
            self.session_factory = session_factory
            self.name = name
            self._schema = schema

        @abstractmethod
        def _get_pipeline_config(self) -> dict[str, Any]: ...

        @abstractmethod
        def run(self, *args, **kwargs) -> dict[str, Any]: ...


__all__ = ["HAVE_REFERENCE", "BasePipeline", "BaseRetrievalPipelineConfig", "PipelineType", "EmbeddingError",
           "require_retrieval_unit", "VALID_RETRIEVAL_UNITS", "RetrievalUnit"]
