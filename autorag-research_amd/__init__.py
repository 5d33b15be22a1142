"""autorag-research_amd -- MI355X-native dense-retrieval core for AutoRAG-Research's Vector Search hot path.

Import name: ``autorag_research_amd`` (the directory keeps the project's hyphenated name; the
importable alias package next to it points its ``__path__`` here).

Layout: csrc/ (HIP kernels + C ABI -> libmi355dr.so), _native.py (ctypes), index.py (one GPU shard),
store.py / service.py / pipelines.py (host-side mirror of the reference's repository / service /
pipeline interfaces for this path), heaven.py (HEAVEN two-stage caller: cosine top-N then candidate MaxSim),
gqr.py (Guided Query Refinement caller: candidate-pool refinement loops on the GPU), hybrid.py (RRF / convex-combination
fusion of two child pipelines), hyde.py (hypothetical-document retrieval),
metrics.py (retrieval metrics), embeddings.py (embedding interfaces), shards.py (fp32 shard files: memory-mapped,
chunked load), sharded.py (row-sharded multi-GPU search).
"""

__version__ = "0.1.0"

from .index import Mi355Index  # noqa: F401
from ._native import NativeError  # noqa: F401
