"""Generated-code check of the screen kernels' software pipeline (runs on CPU: hipcc cross-compiles gfx950).

The screen kernels keep LDS-DMA loads (global_load_lds) of later K-steps in flight while the current step's
operands are read from LDS and multiplied.  That only holds if the compiler's waitcnt insertion does not put a
full `s_waitcnt vmcnt(0)` in front of the ds_reads -- which it does, silently and at a 2x cost, when the LDS reads
are typed differently (see the NOTE in csrc/k_screen.h).  This test pins the property on the generated ISA.
"""

import subprocess
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parents[1]
CSRC = ROOT / "autorag-research_amd" / "csrc"


def _kernel_bodies(asm: str) -> dict[str, list[str]]:
    out, name = {}, None
    for line in asm.split("\n"):
        if line.startswith("_ZN5mi355") and line.split(":")[0].endswith("ScreenArgsE") and ":" in line:
            name = line.split(":")[0]
            out[name] = []
        elif name is not None:
            s = line.strip()
            if s.startswith("s_endpgm"):
                name = None
            elif s and not s.startswith((";", ".")):
                out[name].append(s)
    return out


@pytest.fixture(scope="module")
def screen_asm(tmp_path_factory):
    hipcc = Path("/opt/rocm/bin/hipcc")
    if not hipcc.exists():
        pytest.skip("hipcc not available")
    out = tmp_path_factory.mktemp("asm") / "mi355dr.s"
    cmd = [str(hipcc), "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", f"-I{ROOT / 'include'}",
           f"-I{CSRC}", str(CSRC / "mi355dr.hip"), "-S", "--cuda-device-only", "-o", str(out)]
    subprocess.run(cmd, check=True, capture_output=True, timeout=900)
    return _kernel_bodies(out.read_text())


def _is_vm0(op: str) -> bool:
    return op.startswith("s_waitcnt") and "vmcnt(0)" in op


@pytest.mark.parametrize("i8", [False, True])
def test_screen256_keeps_dma_in_flight(screen_asm, i8):
    name = f"_ZN5mi35511k_screen256ILi0ELb{int(i8)}EEEvNS_10ScreenArgsE"
    ops = screen_asm[name]
    mf = [i for i, o in enumerate(ops) if o.startswith("v_mfma")]
    assert len(mf) == 64  # 4 quadrants x 8 MFMAs, main loop + peeled last K-tile
    want = "v_mfma_i32_32x32x32_i8" if i8 else "v_mfma_f32_32x32x16_bf16"
    assert all(ops[i].startswith(want) for i in mf)
    region = ops[mf[0]:mf[-1]]
    # only the hand-written drain of the LAST K-tile may wait for every outstanding load
    assert sum(_is_vm0(o) for o in region) == 1
    assert sum(o.startswith("s_waitcnt") and "vmcnt(4)" in o for o in region) >= 3
    assert not any(o.startswith("scratch_") for o in ops), "register spill in the screen kernel"


@pytest.mark.parametrize("i8", [False, True])
def test_screen128_double_buffering(screen_asm, i8):
    name = f"_ZN5mi3558k_screenILb{int(i8)}EEEvNS_10ScreenArgsE"
    ops = screen_asm[name]
    mf = [i for i, o in enumerate(ops) if o.startswith("v_mfma")]
    gl = [i for i, o in enumerate(ops) if o.startswith("global_load_lds")]
    assert len(mf) == 16 and len(gl) == 16
    # loop body: [vmcnt(0); barrier] -> issue next step's 8 DMA loads -> ds_reads -> MFMAs; no vm wait after the issue
    last_issue = max(i for i in gl if i < mf[0])
    assert not any("vmcnt" in o for o in ops[last_issue:mf[-1]] if o.startswith("s_waitcnt"))
    assert not any(o.startswith("scratch_") for o in ops)
