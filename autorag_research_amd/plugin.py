"""Entry-point target for `[project.entry-points."autorag_research.pipelines"]`.

The reference's plugin registry (plugin_registry.py:199-255) resolves the entry point to a MODULE and
copies the YAML files it finds in that module's package (flat or under `retrieval/`) into
`configs/pipelines/retrieval/` on `autorag-research plugin sync`.  The YAMLs live in ./retrieval/.
"""

from .gqr import Mi355GQRHybridPipelineConfig, Mi355GQRHybridRetrievalPipeline  # noqa: F401
from .hyde import Mi355HyDEPipelineConfig, Mi355HyDERetrievalPipeline  # noqa: F401
from .hybrid import (  # noqa: F401
    Mi355HybridCCPipelineConfig,
    Mi355HybridCCRetrievalPipeline,
    Mi355HybridRRFPipelineConfig,
    Mi355HybridRRFRetrievalPipeline,
)
from .heaven import Mi355HEAVENPipelineConfig, Mi355HEAVENRetrievalPipeline  # noqa: F401
from .pipelines import (  # noqa: F401
    Mi355ImageVectorSearchPipelineConfig,
    Mi355ImageVectorSearchRetrievalPipeline,
    Mi355VectorSearchPipelineConfig,
    Mi355VectorSearchRetrievalPipeline,
)
