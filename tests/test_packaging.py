"""Packaging of the plugin: entry point, package data, and the reference registry's own scan of the package.

Reference: autorag_research/plugin_registry.py:90-120 (`_scan_module_yamls` scans `importlib.resources.files(module)`,
which needs a PACKAGE), :199-255 (`ep.load()` must return a module).
"""

from __future__ import annotations

import importlib
import importlib.metadata as md
import subprocess
import sys
from importlib.resources import files
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parent.parent
YAMLS = {"mi355_vector_search", "mi355_image_vector_search", "mi355_heaven", "mi355_gqr_hybrid", "mi355_hybrid_rrf",
         "mi355_hybrid_cc", "mi355_hyde"}


def test_package_is_a_plain_importable_package():
    import autorag_research_amd as pkg

    assert Path(pkg.__file__).parent == ROOT / "autorag_research_amd"
    assert list(pkg.__path__) == [str(ROOT / "autorag_research_amd")]          # no __path__ tricks
    assert not (ROOT / "autorag-research_amd").exists()


def test_entry_point_metadata_resolves_to_the_package(tmp_path):
    """Build the distribution's metadata (what `pip install -e .` writes) and read it back with importlib.metadata."""
    r = subprocess.run([sys.executable, "setup.py", "-q", "egg_info", "--egg-base", str(tmp_path)], cwd=ROOT,
                       capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    dists = list(md.distributions(path=[str(tmp_path)]))
    assert [d.metadata["Name"] for d in dists] == ["autorag-research-amd"]
    eps = [e for e in dists[0].entry_points if e.group == "autorag_research.pipelines"]
    assert [(e.name, e.value) for e in eps] == [("mi355_vector_search", "autorag_research_amd")]
    mod = eps[0].load()
    assert mod is importlib.import_module("autorag_research_amd")
    # pyproject.toml (setuptools >= 61) and setup.cfg (this image's 59.6) say the same thing
    text = (ROOT / "pyproject.toml").read_text()
    assert 'mi355_vector_search = "autorag_research_amd"' in text and 'packages = ["autorag_research_amd"]' in text
    assert "[build-system]" in text and "retrieval/*.yaml" in text and "libmi355dr.so" in text


def test_registry_style_scan_finds_every_yaml():
    """What the reference's `_scan_module_yamls` does: files(module).iterdir(), YAMLs flat or one directory down."""
    import autorag_research_amd as pkg

    found = {}
    for res in files(pkg).iterdir():
        if res.name.endswith(".yaml"):
            found[res.name[:-5]] = None
        elif res.is_dir():
            for sub in res.iterdir():
                if sub.name.endswith(".yaml"):
                    found[sub.name[:-5]] = res.name
    assert set(found) == YAMLS and set(found.values()) == {"retrieval"}


@pytest.mark.skipif(not Path("/root/reference/autorag_research/plugin_registry.py").exists(),
                    reason="reference tree only exists in the build container")
def test_reference_registry_scans_this_package():
    """The reference's own function over this package (build container only)."""
    code = (
        "import sys; sys.path.insert(0, %r); sys.path.insert(0, %r)\n"
        "from ref_import import import_reference\n"
        "import_reference()\n"
        "from autorag_research import plugin_registry as pr\n"
        "import autorag_research_amd as pkg\n"
        "infos = pr._scan_module_yamls(pkg, 'mi355_vector_search', 'pipelines')\n"
        "print(sorted((i.config_name, i.subcategory) for i in infos))\n"
    ) % (str(ROOT), str(ROOT / "tests" / "golden"))
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, cwd=ROOT,
                       env={"PYTHONDONTWRITEBYTECODE": "1", "PATH": "/usr/bin:/bin"})
    assert r.returncode == 0, r.stderr[-2000:]
    assert r.stdout.strip() == str(sorted((n, "retrieval") for n in YAMLS))
