"""autorag_research_amd -- MI355X-native dense-retrieval core for AutoRAG-Research's Vector Search hot path.

Distribution name ``autorag-research-amd``; import name and directory ``autorag_research_amd`` (a plain package:
the pipeline YAMLs under ``retrieval/`` and ``configs/`` and the built ``libmi355dr.so`` are package data).

Layout: csrc/ (HIP kernels + C ABI -> libmi355dr.so), _native.py (ctypes), index.py (one GPU shard),
store.py / service.py / pipelines.py (host-side mirror of the reference's repository / service /
pipeline interfaces for this path), heaven.py (HEAVEN two-stage caller: cosine top-N then candidate MaxSim),
gqr.py (Guided Query Refinement caller: candidate-pool refinement loops on the GPU), hybrid.py (RRF / convex-combination
fusion of two child pipelines), hyde.py (hypothetical-document retrieval),
metrics.py (retrieval metrics), embeddings.py (embedding interfaces), shards.py (fp32 shard files: memory-mapped,
chunked load), sharded.py (row-sharded multi-GPU search).
"""

__version__ = "0.3.0"

from .index import Mi355Index  # noqa: F401
from ._native import NativeError  # noqa: F401
