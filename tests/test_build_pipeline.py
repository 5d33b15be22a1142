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
CSRC = ROOT / "autorag_research_amd" / "csrc"


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
    # one K-step body (4 quadrants x 8 MFMAs) serves every K-step of every tile; the compiler may peel or unroll it
    assert len(mf) % 32 == 0 and 32 <= len(mf) <= 96
    copies = len(mf) // 32
    want = "v_mfma_i32_32x32x32_i8" if i8 else "v_mfma_f32_32x32x16_bf16"
    assert all(ops[i].startswith(want) for i in mf)
    # the K-step body: from the first LDS-DMA issue after the prologue's barrier to the last MFMA
    gl = [i for i, o in enumerate(ops) if o.startswith("global_load_lds")]
    body_start = max(i for i in gl if i < mf[0]) - 1
    body = ops[body_start:mf[-1]]
    assert not any(_is_vm0(o) for o in body), "the K-step waits for ALL outstanding LDS-DMA loads"
    assert sum(o.startswith("s_waitcnt") and "vmcnt(4)" in o for o in body) == 4 * copies
    assert sum(o.startswith("global_load_lds") for o in body) >= 8 * copies - 1  # (the first issue may sit just above)
    # the hit path's queue stores must not be preceded by a vector-memory wait (the compiler adds one for LDS
    # accesses it can see; they are inline asm for that reason)
    for i, o in enumerate(ops):
        if o.startswith("ds_write_b32"):
            assert not any("vmcnt" in p for p in ops[max(0, i - 6):i] if p.startswith("s_waitcnt")), ops[i - 6:i + 1]
    assert not any(o.startswith("scratch_") for o in ops), "register spill in the screen kernel"
    # ... and must not be followed by a wait for themselves: a hit stalls the whole workgroup for as long as this path
    # takes (DESIGN 4.1 "what a hit costs"); LDS operations of one wave execute in order, the flush reads them later
    writes = [i for i, o in enumerate(ops) if o.startswith("ds_write_b32")]
    assert writes and len(writes) % 3 == 0
    for i in writes[2::3]:  # the last store of every (query, row, value) entry
        assert not any(p.startswith("s_waitcnt") and "lgkmcnt(0)" in p for p in ops[i + 1:i + 5]), ops[i:i + 6]


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
