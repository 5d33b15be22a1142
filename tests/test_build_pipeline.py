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
        if line.startswith("_ZN5mi355") and line.split(":")[0].endswith(("ScreenArgsE", "ScreenArgs2E")) and ":" in line:
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
def lib_asm_text(tmp_path_factory):
    hipcc = Path("/opt/rocm/bin/hipcc")
    if not hipcc.exists():
        pytest.skip("hipcc not available")
    out = tmp_path_factory.mktemp("asm") / "mi355dr.s"
    cmd = [str(hipcc), "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", f"-I{ROOT / 'include'}",
           f"-I{CSRC}", str(CSRC / "mi355dr.hip"), "-S", "--cuda-device-only", "-o", str(out)]
    subprocess.run(cmd, check=True, capture_output=True, timeout=900)
    return out.read_text()


@pytest.fixture(scope="module")
def screen_asm(lib_asm_text):
    return _kernel_bodies(lib_asm_text)


def _is_vm0(op: str) -> bool:
    return op.startswith("s_waitcnt") and "vmcnt(0)" in op


def _loop_blocks(ops: list[str]) -> list[str]:
    """The instructions between the first and the last MFMA (the persistent K loop incl. its cold branches)."""
    mf = [i for i, o in enumerate(ops) if o.startswith("v_mfma")]
    return ops[mf[0]:mf[-1] + 1]


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


@pytest.mark.parametrize("i8", [False, True])
def test_screen256c_structure(screen_asm, i8):
    """Third form (the one the library launches): free-running waves -- one workgroup barrier per K-step, every fragment read
    issued two micro-steps ahead under counted LDS waits, nine (int8: + the row-group records) LDS-DMA pieces per K-step spread
    over the micro-steps, the append path out of line (one copy, reached by a call), no scratch."""
    names = [n for n in screen_asm if f"k_screen256cILb{int(i8)}EEEvNS_11ScreenArgs2E" in n]
    assert len(names) == 1, sorted(screen_asm)
    ops = screen_asm[names[0]]
    want = "v_mfma_i32_32x32x32_i8" if i8 else "v_mfma_f32_32x32x16_bf16"
    mf = [o for o in ops if o.startswith("v_mfma")]
    # 8 micro-steps of 4 MFMAs; the two micro-steps that open a tile exist twice (C = 0 as an inline constant)
    assert len(mf) == 40 and all(o.startswith(want) for o in mf)
    assert sum(o.rstrip().endswith(", 0") for o in mf) == 8
    assert not any(o.startswith("scratch_") for o in ops), "register spill in the screen kernel"
    loop = _loop_blocks(ops)
    assert sum(o.startswith("s_barrier") for o in loop) == 1
    dma = [o for o in loop if o.startswith("global_load_lds_dwordx4")]
    # (8 per K-step; the micro-step that opens a tile exists twice in the text -- with and without the C = 0 start -- and
    # carries two of them)
    assert len(dma) in (8, 10) and all(", s[" in o for o in dma), dma
    assert len([o for o in loop if o.startswith("global_load_lds_dword ")]) == (1 if i8 else 0)
    assert not any(o.startswith("s_load") for o in loop)  # (a scalar load in flight would turn counted LDS waits into lgkmcnt(0))
    counted = [o for o in loop if o.startswith("s_waitcnt") and "lgkmcnt(" in o and "lgkmcnt(0)" not in o]
    assert len(counted) >= 6, counted
    # the hand-over: this wave's pieces of the next K-step have landed, its fragments are in registers
    i_bar = next(i for i, o in enumerate(loop) if o.startswith("s_barrier"))
    before = loop[max(0, i_bar - 12):i_bar]
    assert any(_is_vm0(o) for o in before) and any(o.startswith("s_waitcnt") and "lgkmcnt(0)" in o for o in before)
    # the append path is a call
    assert any(o.startswith("s_swappc_b64") for o in ops)


@pytest.mark.parametrize("ks", [1, 2, 3, 4, 5, 6])
def test_screen_rq_structure(screen_asm, ks):
    """k_screen_rq (round 5; the form the library launches for int8 shadows of at most 768 B per row): the query fragments are
    registers -- loaded by 4 KS global loads BEFORE the loop and waited for there (left to the waitcnt pass their waits land in
    front of the first use inside the tile loop and drain the LDS-DMA ring on every tile) --, every LDS read of the loop's hot
    path is a ROW fragment (one ds_read_b128 per MFMA), two 1-KiB row pieces per K-step and wave, a hand-over (counted vmcnt
    wait + barrier) in every K-step but a tile's last (KS >= 2: the four block tests of a tile then sit between two barriers),
    no full vmcnt(0) between the first and the last MFMA outside the flush of the hit-lane queue, the hit path INLINE (five
    inline-asm ds_write_b128 per test site behind a wave-uniform branch; no call anywhere), no scratch."""
    names = [n for n in screen_asm if f"k_screen_rqILi{ks}ELb1ELb1E" in n]   # <KS, SPLIT = true, int8>
    assert len(names) == 1, sorted(screen_asm)
    ops = screen_asm[names[0]]
    stages = {4: 4, 5: 5}.get(ks, 6)
    skip_last = ks >= 2
    assert not any(o.startswith("scratch_") for o in ops), "register spill in the screen kernel"
    assert not any(o.startswith("s_swappc") for o in ops), "a call in the screen kernel (its entry drains the LDS-DMA ring)"
    mf = [i for i, o in enumerate(ops) if o.startswith("v_mfma")]
    copies = len(mf) // (16 * ks)   # (the compiler may peel the tile loop once: the same body with / without a previous tile)
    assert copies in (1, 2) and len(mf) == 16 * ks * copies and all(ops[i].startswith("v_mfma_i32_32x32x32_i8") for i in mf)
    assert sum(ops[i].rstrip().endswith(", 0") for i in mf) == 4 * copies   # a tile's four blocks start from C = 0 (inline constant)
    pre = ops[:mf[0]]
    assert sum(o.startswith("global_load_dwordx4") for o in pre) == 4 * ks
    assert sum(o.startswith("global_load_lds_dwordx4") for o in pre) == 2 * (stages - 1 if skip_last else stages)  # the prologue fills the ring
    loop = ops[mf[0]:mf[-1] + 1]
    assert not any(o.startswith(("global_load_dwordx4", "s_load")) for o in loop)
    assert sum(o.startswith("s_barrier") for o in loop) == (ks - 1 if skip_last else ks) * copies
    assert sum(o.startswith("global_load_lds_dwordx4") for o in loop) == 2 * ks * copies
    counted_vm = [o for o in loop if o.startswith("s_waitcnt") and "vmcnt(" in o and "vmcnt(0)" not in o]
    assert len(counted_vm) == (ks - 1 if skip_last else ks) * copies, [o for o in loop if "vmcnt" in o]
    assert all(f"vmcnt({2 * (stages - 2)})" in o or f"vmcnt({2 * (stages - 3)})" in o or f"vmcnt({2 * (stages - 4)})" in o for o in counted_vm)
    # full drains inside the loop: only the flush of the hit-lane queue (one site, wave-uniform branch at a tile's start)
    # full drains belong to the flushes of the hit-lane queue (cold blocks, entered by wave-uniform branches): whatever the block
    # layout, there are few of them and every one sits next to the flush's atomics
    assert sum(_is_vm0(o) for o in loop) <= 12 * copies and sum(o.startswith("global_atomic_add") for o in ops) >= 1
    counted = [o for o in loop if o.startswith("s_waitcnt") and "lgkmcnt(" in o and "lgkmcnt(0)" not in o and "vmcnt" not in o]
    assert len(counted) >= 9 * ks * copies, counted
    # the hit path: 4 test sites per copy of the tile loop + 2 behind it (the last tile's row half 1), 5 stores each
    n_st = sum(o.startswith("ds_write_b128") for o in ops)
    assert n_st % 5 == 0 and n_st >= 20, n_st
    # the sibling drift limiter: per tile one agent-scope store of the own progress word and ONE LDS-DMA dword fetch of the
    # siblings' (no register, no wait of its own); the polling loads live in the cold spin block next to an s_sleep
    n_pub = sum(o.startswith("global_store_dword") and o.endswith("sc1") for o in ops)   # (+ the "done" word behind the loop)
    n_get = sum(o.startswith("global_load_lds_dword ") and o.endswith("sc1") for o in ops)
    assert 1 <= n_get <= 2 and n_get <= n_pub <= n_get + 1, (n_pub, n_get)
    polls = [o for o in ops if o.startswith("global_load_dword ") and o.endswith("sc1")]
    assert polls and sum(o.startswith("s_sleep") for o in ops) == len(polls)


def _whole_kernel(asm: str, name: str) -> tuple[list[str], str]:
    """(instructions, .amdhsa descriptor text) of one kernel."""
    a = asm.index(f"\n{name}:")
    b = asm.index(".end_amdhsa_kernel", a)
    body = asm[a:b]
    ops = [ln.strip() for ln in body.split("\n") if ln.strip() and not ln.strip().startswith((";", "."))]
    return ops, body[body.index(".amdhsa_kernel"):]


@pytest.mark.parametrize("name", ["_ZN5mi3557k_pruneILi64ELi1024EEEvNS_9PruneArgsE", "_ZN5mi3556k_scanENS_8ScanArgsE"])
def test_gather_kernels_use_global_loads_and_no_stack(lib_asm_text, name):
    """The re-score gathers rebuild their row pointers from shuffled integers; without the address-space cast of
    dev_common.h (load_gmem_f4) the compiler emits FLAT loads, which count on lgkmcnt as well as vmcnt -- every LDS wait
    of the chain then drains the prefetched pieces.  And k_prune's body must stay inlined: as a function its argument
    struct lives on the stack."""
    ops, desc = _whole_kernel(lib_asm_text, name)
    wide_flat = [o for o in ops if o.startswith("flat_load_dwordx4")]
    assert not wide_flat, f"{len(wide_flat)} FLAT 16-byte loads in {name}"
    assert sum(o.startswith("global_load_dwordx4") for o in ops) >= 32
    assert ".amdhsa_private_segment_fixed_size 0" in desc, "stack use (spills or an out-of-line body)"
    assert not any(o.startswith(("scratch_", "s_swappc")) for o in ops)


@pytest.mark.parametrize("nq", [32, 64])
@pytest.mark.parametrize("i8", [False, True])
def test_screen_stream_ring_stays_in_flight(screen_asm, i8, nq):
    """k_screen_stream's stage loop: the counted wait, one barrier, the four LDS-DMA pieces of the stage five ahead, the
    fragment reads and MFMAs -- and no full vmcnt(0) anywhere between the wait and the last MFMA (the compiler puts one
    in front of the first use of a value it loaded itself: the per-query constants are therefore used before the loop)."""
    name = f"_ZN5mi35515k_screen_streamILb{int(i8)}ELi{nq}EEEvNS_10ScreenArgsE"
    ops = screen_asm[name]
    w = [i for i, o in enumerate(ops) if o.startswith("s_waitcnt") and "vmcnt(16)" in o]
    assert len(w) == 1, "one counted wait: the head of the stage loop"
    mf = [i for i, o in enumerate(ops) if o.startswith("v_mfma") and i > w[0]]
    nj = nq // 32
    assert len(mf) == 4 * nj
    body = ops[w[0]:mf[-1] + 1]
    assert sum(o.startswith("s_barrier") for o in body) == 1
    assert sum(o.startswith("global_load_lds_dwordx4") for o in body) == 4
    assert sum(o.startswith("ds_read_b128") for o in body) == 4 * (1 + nj)
    assert not any(_is_vm0(o) for o in body)


@pytest.fixture(scope="module")
def maxsim_asm_text(tmp_path_factory):
    """Device assembly of the two MaxSim translation units (round 6 split the screens off: mi355dr_maxsim_screen.hip), concatenated."""
    hipcc = Path("/opt/rocm/bin/hipcc")
    if not hipcc.exists():
        pytest.skip("hipcc not available")
    texts = []
    for unit in ("mi355dr_maxsim.hip", "mi355dr_maxsim_screen.hip"):
        out = tmp_path_factory.mktemp("asm") / (Path(unit).stem + ".s")
        cmd = [str(hipcc), "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", f"-I{ROOT / 'include'}",
               f"-I{CSRC}", str(CSRC / unit), "-S", "--cuda-device-only", "-o", str(out)]
        subprocess.run(cmd, check=True, capture_output=True, timeout=900)
        texts.append(out.read_text())
    return "\n".join(texts)


def test_no_kernel_of_the_library_touches_the_stack(lib_asm_text, maxsim_asm_text):
    """Not one `scratch_*` instruction and a zero private segment in EVERY kernel of mi355dr.hip and mi355dr_maxsim.hip.  Round 4's
    k_maxsim -- the exact MaxSim kernel behind every candidate re-score and `mi355dr_maxsim_subset` -- carried 65 of them: it
    wrote to its by-value argument block and indexed the block's arrays with run-time values, so the whole block was copied to
    the stack at entry and read back from there."""
    import re

    for text in (lib_asm_text, maxsim_asm_text):   # (maxsim_asm_text: both MaxSim units)
        kernels = re.findall(r"\n(_ZN5mi355[^\n:]+):[^\n]*\n(.*?)\.end_amdhsa_kernel", text, re.S)
        assert len(kernels) >= 20
        for name, body in kernels:
            assert not re.search(r"^\s*scratch_", body, re.M), name
            assert ".amdhsa_private_segment_fixed_size 0" in body, name


@pytest.mark.parametrize("form", ["bps2", "bps4", "pipelined"])
@pytest.mark.parametrize("ncb", [12, 16])
def test_maxsim_workgroup_screen_keeps_its_ring_in_flight(maxsim_asm_text, ncb, form):
    """k_maxsim16_wg (round 4): the token blocks come through a ring of LDS-DMA stages shared by the workgroup's 8 waves (stages
    of 2 or 4 32-token blocks).  Pinned on the generated code: the query fragments live in registers (no ds_read feeds an MFMA's B
    operand -- 8 fragment reads per block, all of them token fragments), exactly one counted vector-memory wait per stage
    loop and NO full `s_waitcnt vmcnt(0)` between the prologue's wait and the loop's last MFMA (a per-document vector load of the
    block offsets -- or the query fragments left to their first use -- put one there and drained 96 KiB of stream per document),
    block offsets through the scalar cache, no scratch.  The software-pipelined form (the default) has two stage loops (waves
    with one / two column blocks), and the fold of a block sits BETWEEN the MFMAs of the next one."""
    bps = 2 if form == "bps2" else 4
    pipe = form == "pipelined"
    name = f"_ZN5mi35513k_maxsim16_wgILi{ncb}ELb1ELi{bps}ELb{int(pipe)}EEEvNS_8Ms16ArgsEl"
    ops, desc = _whole_kernel(maxsim_asm_text, name)
    assert ".amdhsa_private_segment_fixed_size 0" in desc and not any(o.startswith("scratch_") for o in ops)
    inloop = 12 - bps   # pieces of (stages - 2) stages may still fly at the hand-over
    w12 = [i for i, o in enumerate(ops) if o.startswith("s_waitcnt") and "vmcnt(12)" in o]
    wl = [i for i, o in enumerate(ops) if o.startswith("s_waitcnt") and f"vmcnt({inloop})" in o]
    loops = 2 if pipe else 1
    assert len(w12) == 1 and len(wl) == loops and w12[0] < wl[0]
    mf = [i for i, o in enumerate(ops) if o.startswith("v_mfma_f32_32x32x16_bf16")]
    # per block of a stage: 8 MFMAs for the wave's first column block + 8 for its second (pipelined: a loop with 8 + one with 16)
    assert len(mf) == (24 * bps if pipe else 16 * bps)
    loop = ops[w12[0] + 1:max(mf) + 1]
    assert not any(_is_vm0(o) for o in loop), [o for o in loop if "vmcnt" in o]
    assert sum(o.startswith("global_load_lds_dwordx4") for o in loop) == bps * loops  # the stage one ring ahead: a 1-KiB piece per block and wave
    assert not any(o.startswith(("global_load_dword", "flat_load")) for o in loop)  # (block offsets: s_load)
    # token fragments only: 16 reads for the first two blocks in front of the loop(s) + 8 per block inside (wherever the block
    # layout puts the loop's last eight); the B operands never come from LDS
    n_rd = sum(o.startswith("ds_read_b128") for o in ops)
    assert n_rd == 16 + 8 * bps * loops and sum(o.startswith("ds_read_b128") for o in loop) >= 8 * bps - 8
    if pipe:
        # the fold rides in the shadow of the MFMAs: between the first and the last MFMA of some block's burst sit >= 6 v_max3
        gaps = [sum(o.startswith("v_max3_f32") for o in ops[mf[i]:mf[i + 7]]) for i in range(0, len(mf) - 7, 8)]
        assert max(gaps) >= 6, gaps
    assert any(o.startswith("s_load_dwordx2") for o in loop)


def test_maxsim_packed_screen_folds_and_sums_in_the_mfma_shadow(maxsim_asm_text):
    """k_maxsim16_wg8 (round 6: the granule-packed copy).  Pinned on the generated code: the ring discipline of k_maxsim16_wg's
    pipelined form (one counted wait per stage loop, no full vector-memory wait inside, the stage one ring ahead staged by LDS-DMA,
    token fragments only from LDS), the granule maxima folded BETWEEN the MFMAs of the next block (v_max3 only: no canonicalising
    v_max_f32 x, x, x of raw MFMA results), the deferred per-document sums (the DPP chain) between MFMAs too, and a document cursor
    that is scalar: no v_readfirstlane and no 64-bit VALU compare between the prologue's wait and the loops' last MFMA."""
    name = "_ZN5mi35514k_maxsim16_wg8ENS_8Ms16ArgsENS_8Ms16PackEi"
    ops, desc = _whole_kernel(maxsim_asm_text, name)
    assert ".amdhsa_private_segment_fixed_size 0" in desc and not any(o.startswith("scratch_") for o in ops)
    w12 = [i for i, o in enumerate(ops) if o.startswith("s_waitcnt") and "vmcnt(12)" in o]
    wl = [i for i, o in enumerate(ops) if o.startswith("s_waitcnt") and "vmcnt(8)" in o]
    assert len(w12) == 1 and len(wl) == 2 and w12[0] < wl[0]
    mf = [i for i, o in enumerate(ops) if o.startswith("v_mfma_f32_32x32x16_bf16")]
    assert len(mf) == 96    # 4 blocks per stage x (8 MFMAs in the one-column-block loop + 16 in the other)
    loop = ops[w12[0] + 1:max(mf) + 1]
    assert not any(_is_vm0(o) for o in loop), [o for o in loop if "vmcnt" in o]
    assert sum(o.startswith("global_load_lds_dwordx4") for o in loop) == 8
    assert not any(o.startswith(("global_load_dword", "flat_load")) for o in loop)
    assert sum(o.startswith("ds_read_b128") for o in ops) == 16 + 8 * 4 * 2
    assert any(o.startswith("s_load_dwordx2") for o in loop)            # granule offsets through the scalar cache
    assert not any(o.startswith(("v_readfirstlane", "v_cmp_ge_i64", "v_cmp_gt_i64", "v_cmp_lt_i64", "v_cmp_le_i64")) for o in loop)
    bursts = [ops[mf[i]:mf[i + 7] + 1] for i in range(0, len(mf) - 7, 8)]
    assert max(sum(o.startswith("v_max3_f32") for o in b) for b in bursts) >= 6
    assert max(sum(o.startswith("v_add_f32_dpp") for o in b) for b in bursts) >= 3
    assert not any(o.startswith("v_max_f32") and len(set(o.replace(",", " ").split()[1:])) == 1 for b in bursts for o in b)


def test_the_product_kernels_carry_no_timing_builds():
    """the screens' ablation forms (template parameter ABL: no fragment reads / tests / barrier / LDS-DMA, ...)
    live in tools/k_screen_rq_abl.h / tools/k_screen256c_abl.h for tools/screen_ab.hip; the library's headers carry the kernels
    alone, and the packed MaxSim copy (measured slower in round 4) is gone."""
    for h in ("k_screen_rq.h", "k_screen256c.h", "k_screen256_common.h", "k_prune_wide.h"):
        assert "ABL" not in (CSRC / h).read_text(), h
    assert not (CSRC / "k_maxsim_wgp.h").exists()
    assert (ROOT / "tools" / "k_screen_rq_abl.h").exists() and "ABL" in (ROOT / "tools" / "k_screen_rq_abl.h").read_text()
