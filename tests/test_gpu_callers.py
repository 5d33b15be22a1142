"""GPU: the fusion / HyDE callers over the real index (the same checks tests/test_host_logic.py runs on the oracle-backed
stand-in): reference `_retrieve_by_id` dicts of tests/golden/hybrid_golden.json and hyde_golden.json."""

import pytest
from helpers import build_golden_stores, load_service_golden

pytestmark = pytest.mark.gpu


@pytest.fixture()
def gpu_env(native_built):
    store, g = build_golden_stores()
    return store, g, load_service_golden()


def test_hybrid_pipelines_on_gpu_match_reference_dicts(gpu_env):
    import test_host_logic as host

    host.test_hybrid_pipelines_match_reference_dicts(gpu_env)


def test_hyde_pipeline_on_gpu_matches_reference_dicts(gpu_env):
    import test_host_logic as host

    host.test_hyde_pipeline_matches_reference_dicts(gpu_env)
