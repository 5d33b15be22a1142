"""GPU: a fixed slice of the randomized differential campaign (tools/fuzz_parity.py) -- random shapes, data modes and
option settings vs the CPU oracle, bit for bit (search paths) / to float64 rounding (GQR refinement).  The open-ended form is `python tools/fuzz_parity.py --seconds N`."""

import sys
from pathlib import Path

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
sys.path.insert(0, str(Path(__file__).resolve().parents[1] / "tools"))


@pytest.mark.parametrize("base", [101, 202, 303])
def test_fuzz_slice(native_built, oracle, base):
    import fuzz_parity as fz

    for case in range(12):
        seed = base * 1_000_003 + case
        rng = np.random.default_rng(seed)
        u = rng.random()
        kind = fz.pick_kind(u)
        try:
            fz.CHECKS[kind](rng, case)
        except AssertionError as e:  # pragma: no cover - a failure names the seed to replay
            raise AssertionError(f"seed {seed} ({kind}): {e}") from e
