"""CPU: ISA-level regression checks of the latency-critical decoder-step kernels (hipcc cross-compiles gfx950 without a GPU).

profiles/r1_skinny_isa_notes.txt records why the first-generation decoder-step kernels sat at 4-17 us per launch: predicated
loads become exec-masked branches and the compiler then waits with vmcnt(0) before every consumer, so a launch pays one
memory round trip per slab.  The second-generation kernels (csrc/k_dstep.hip) issue every load of a wave unconditionally
up front; these tests keep it that way: all buffer loads of the product kernel come before its first matrix instruction,
the waits are counted (vmcnt(N) with N falling), and the attention kernel has its projections' partial sums AND the first
64 keys / values in flight before its first wait."""
import re
import shutil
import subprocess
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parent.parent
CSRC = ROOT / "seamless_communication_amd" / "csrc"
HIPCC = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"

pytestmark = pytest.mark.skipif(not Path(HIPCC).exists(), reason="hipcc not available")


def _isa(src: str, tmp_path: Path) -> str:
    out = tmp_path / "k.o"
    subprocess.run([HIPCC, "--offload-arch=gfx950", "-O3", "-std=c++17", "-x", "hip", "-c", str(CSRC / src), "--save-temps", "-o", str(out)],
                   cwd=tmp_path, check=True, capture_output=True, timeout=600)
    files = list(tmp_path.glob("*gfx950.s"))
    assert len(files) == 1
    return files[0].read_text()


def _function(isa: str, mangled_fragment: str) -> list:
    lines = isa.splitlines()
    starts = [i for i, l in enumerate(lines) if re.match(r"^_ZN2sc.*" + re.escape(mangled_fragment) + r".*:", l)]
    assert len(starts) == 1, (mangled_fragment, len(starts))
    end = next(i for i in range(starts[0], len(lines)) if "s_endpgm" in lines[i])
    return lines[starts[0]: end + 1]


def _innermost_loop_with(fn: list, *needles: str) -> list:
    """The smallest backward-branch region (label .. branch to that label) that contains every needle."""
    labels = {l.split(":")[0]: i for i, l in enumerate(fn) if re.match(r"^\.LBB\d+_\d+:", l)}
    best = None
    for i, l in enumerate(fn):
        m = re.search(r"s_c?branch\w*\s+(\.LBB\d+_\d+)", l)
        if m and m.group(1) in labels and labels[m.group(1)] < i:
            body = fn[labels[m.group(1)]: i + 1]
            if all(any(n in b for b in body) for n in needles) and (best is None or len(body) < len(best)):
                best = body
    assert best is not None, f"no loop containing {needles}"
    return best


@pytest.fixture(scope="module")
def dstep_isa(tmp_path_factory):
    return _isa("k_dstep.hip", tmp_path_factory.mktemp("dstep"))


def _ops(fn):
    return [l.strip() for l in fn if l.startswith("\t") and not l.strip().startswith((";", "."))]


@pytest.mark.parametrize("inst,loads,mfmas", [("gemvp_kernelILi1ELi4ELi0E", 32, 32), ("gemvp_kernelILi1ELi1ELi0E", 12, 8),
                                              ("gemvp_kernelILi2ELi2ELi0E", 40, 32)])
def test_gemvp_loads_are_issued_up_front_with_counted_waits(dstep_isa, inst, loads, mfmas):
    ops = _ops(_function(dstep_isa, inst))
    first_mfma = next(i for i, o in enumerate(ops) if o.startswith("v_mfma_f32_32x32x16"))
    head = ops[:first_mfma]
    assert sum(o.startswith("buffer_load_dwordx4") for o in head) == loads  # W0 A0 W1 A1 W2 W3 before any product
    last_mfma = max(i for i, o in enumerate(ops) if o.startswith("v_mfma_f32_32x32x16"))
    body = ops[first_mfma: last_mfma + 1]
    assert sum(o.startswith("v_mfma_f32_32x32x16") for o in body) == mfmas
    waits = [int(re.search(r"vmcnt\((\d+)\)", o).group(1)) for o in ops[: last_mfma + 1] if "vmcnt(" in o]
    assert waits and waits[0] > 0 and waits[-1] == 0  # the first products start while later fragments are still in flight
    assert not any(o.startswith("s_cbranch") for o in body)  # one basic block from the first product to the last
    assert not any(o.startswith(("v_accvgpr_read", "v_accvgpr_write", "scratch_")) for o in body)


# template arguments: CROSS, ANC (beam search: key / value rows named by the ancestor table), ROWS (decode engine: per-slot
# row state and position - one 8-byte load in front of everything else)
@pytest.mark.parametrize("inst,loads", [("dattn_kernelILb0ELb0ELb0E", 12 + 32), ("dattn_kernelILb1ELb0ELb0E", 4 + 32)])  # the greedy step's two
def test_decoder_attention_has_its_operands_in_flight_before_the_first_wait(dstep_isa, inst, loads):
    ops = _ops(_function(dstep_isa, inst))
    first_wait = next(i for i, o in enumerate(ops) if "vmcnt(" in o)
    assert sum(o.startswith("global_load_dwordx4") for o in ops[:first_wait]) >= loads
    assert not any(o.startswith("scratch_") for o in ops)


@pytest.mark.parametrize("inst,early", [("dattn_kernelILb0ELb0ELb1E", 12), ("dattn_kernelILb1ELb0ELb1E", 0)])
def test_engine_attention_waits_once_for_its_slot_then_has_the_whole_trip_in_flight(dstep_isa, inst, early):
    """dattn_kernel<*, false, true> (decode engine): the slot's {row state, position} pair is ONE load of at most 8 bytes, requested
    BEFORE the live-row test (round 6: the pair and the live-row count are one round trip, not two in a row); self-attention's
    projection partials (indexed by the slot) travel with it, cross-attention's follow the test; then all 32 key / value loads of
    the first trip are issued before the next wait that drains any of them; no scratch."""
    ops = _ops(_function(dstep_isa, inst))
    waits = [i for i, o in enumerate(ops) if "vmcnt(" in o]
    assert sum(o.startswith("global_load_dwordx4") for o in ops[: waits[0]]) >= early
    assert sum(o.startswith("global_load_dwordx2") for o in ops[: waits[0]]) <= 1
    assert sum(o.startswith("global_load_dwordx4") for o in ops[waits[0]: waits[1]]) >= 32
    assert not any(o.startswith("scratch_") for o in ops)


def test_beam_search_attention_fetches_the_table_entries_then_the_whole_trip(dstep_isa):
    """dattn_kernel<false, true>: the 12 projection partials and the 16 ancestor-table entries of the first trip are in
    flight before the first wait, the 32 key / value loads of the trip (addresses = kernel-argument base + a 32-bit offset
    each) before the next one that drains them; no scratch (228 VGPRs: two waves per SIMD like the plain variant)."""
    ops = _ops(_function(dstep_isa, "dattn_kernelILb0ELb1ELb0E"))
    waits = [i for i, o in enumerate(ops) if "vmcnt(" in o]
    head = ops[: waits[0]]
    assert sum(o.startswith("global_load_dwordx4") for o in head) >= 12 and sum(o.startswith("global_load_dword ") or o.startswith("global_load_dword\t") or o == "global_load_dword" or o.startswith("global_load_dword v") for o in head) >= 16
    kv = 0
    for a, b in zip(waits, waits[1:] + [len(ops)]):
        kv = max(kv, sum(o.startswith("global_load_dwordx4") for o in ops[a:b]))
    assert kv >= 32
    assert not any(o.startswith("scratch_") for o in ops)


# --------------------------------------------------------------------------------------------------------------------- #
# round 2: the 8-wave DMA GEMM tile and the fp16-split attention kernel
# --------------------------------------------------------------------------------------------------------------------- #
@pytest.fixture(scope="module")
def gemm_ps_isa(tmp_path_factory):
    return _isa("k_gemm_ps.hip", tmp_path_factory.mktemp("gemmps"))


# template arguments: tile, waves, ILV, SPLIT, CONV, HALF (mid-slab barrier), AMAX (arg-max epilogue), PP (alternating load /
# compute segments: the shipped schedule of the 8-wave tile, linted separately below)
@pytest.mark.parametrize("inst,mfmas_per_slab", [("gemm_ps_kernelILi256ELi256ELi4ELi2ELb1ELb1ELb0ELb1ELb0ELi0E", 32),
                                                 ("gemm_ps_kernelILi256ELi256ELi4ELi2ELb1ELb1ELb1ELb1ELb0ELi0E", 32),
                                                 ("gemm_ps_kernelILi128ELi128ELi2ELi2ELb1ELb1ELb0ELb1ELb0ELi0E", 16),
                                                 ("gemm_ps_kernelILi256ELi256ELi4ELi2ELb1ELb1ELb0ELb0ELb0ELi0E", 32),
                                                 ("gemm_ps_kernelILi128ELi128ELi2ELi2ELb1ELb1ELb0ELb0ELb0ELi0E", 16),
                                                 # the arg-max epilogue (unit projection): same slab loop
                                                 ("gemm_ps_kernelILi256ELi256ELi4ELi2ELb1ELb1ELb0ELb1ELb1ELi0E", 32),
                                                 ("gemm_ps_kernelILi128ELi128ELi2ELi2ELb1ELb1ELb0ELb1ELb1ELi0E", 16)])
def test_dma_gemm_slab_loop_keeps_its_pipeline(gemm_ps_isa, inst, mfmas_per_slab):
    """The K loop of the pre-split GEMM (plain and implicit-conv variant): no scratch, DMAs issued as
    `buffer_load_dwordx4 ... lds` between the matrix instructions, one s_barrier per slab, and the only full drain
    (vmcnt(0)) is the last slab's - a drain per slab would serialise global->LDS copies and MFMAs."""
    fn = _function(gemm_ps_isa, inst)
    assert not any("scratch_" in l for l in fn)
    loop = _innermost_loop_with(fn, "v_mfma_f32_32x32x16", "s_barrier")
    ops = _ops(loop)
    n_mfma = sum(o.startswith("v_mfma_f32_32x32x16") for o in ops)
    n_bar = sum(o.startswith("s_barrier") for o in ops)
    assert n_bar >= 1 and n_mfma == mfmas_per_slab * n_bar  # unrolled by the number of stages the compiler kept in the loop
    n_dma = sum(o.startswith("buffer_load_dwordx4") and o.endswith("lds") for o in ops)
    assert n_dma == 6 * n_bar
    assert sum("vmcnt(0)" in o for o in ops) <= n_bar  # at most the tail branch of each unrolled step drains
    assert any(re.search(r"vmcnt\(6\)", o) for o in ops)  # the counted wait that leaves the next slab's DMAs in flight
    # DMAs really sit between the matrix instructions (the ILV schedule), not in a block in front of them
    first = next(i for i, o in enumerate(ops) if o.startswith("v_mfma"))
    last = max(i for i, o in enumerate(ops) if o.startswith("v_mfma"))
    assert any(o.startswith("buffer_load_dwordx4") for o in ops[first:last])


@pytest.mark.parametrize("inst", ["gemm_ps_kernelILi256ELi256ELi4ELi2ELb1ELb1ELb0ELb0ELb0ELi4E",   # plain
                                  "gemm_ps_kernelILi256ELi256ELi4ELi2ELb1ELb1ELb1ELb0ELb0ELi4E",   # implicit convolution
                                  "gemm_ps_kernelILi256ELi256ELi4ELi2ELb1ELb1ELb0ELb0ELb1ELi4E"])  # arg-max epilogue
def test_dma_gemm_alternating_schedule_keeps_its_segments(gemm_ps_isa, inst):
    """The 8-wave tile's alternating schedule (k_gemm_ps.hip, PP): no scratch (the two roles are two loops - one loop with a
    role branch per segment spilt 900 bytes per lane); a COMPUTE segment is `s_setprio 1`, 16 matrix instructions and nothing
    else, `s_setprio 0`; a LOAD segment is 8 fragment reads, 3 DMAs, lgkmcnt(0); segments end in a barrier; the DMAs of the
    slab two ahead are left in flight by a counted vmcnt(6), and only tail branches drain."""
    fn = _function(gemm_ps_isa, inst)
    ops = _ops(fn)
    assert not any(o.startswith("scratch_") for o in ops)
    prio1 = [i for i, o in enumerate(ops) if o.startswith("s_setprio 1")]
    prio0 = [i for i, o in enumerate(ops) if o.startswith("s_setprio 0")]
    assert len(prio1) == len(prio0) >= 12  # 2 roles x 3 stages x 2 chunks (+ the first / last chunk of waves 4-7)
    for a, b in zip(prio1, prio0):
        body = [o for o in ops[a + 1: b] if not o.startswith("s_nop")]
        assert len(body) == 16 and all(o.startswith("v_mfma_f32_32x32x16") for o in body), body[:3]
    loop = _innermost_loop_with(fn, "v_mfma_f32_32x32x16", "s_barrier", "ds_read_b128")
    lops = _ops(loop)
    n_bar = sum(o.startswith("s_barrier") for o in lops)
    assert n_bar % 4 == 0 and n_bar >= 4
    assert sum(o.startswith("v_mfma_f32_32x32x16") for o in lops) == 8 * n_bar  # 32 per slab, 4 barriers per slab
    assert sum(o.startswith("ds_read_b128") for o in lops) == 4 * n_bar         # 16 per slab
    assert sum(o.startswith("buffer_load_dwordx4") and o.endswith("lds") for o in lops) == 6 * n_bar // 4
    assert any(re.search(r"vmcnt\(6\)", o) for o in lops)
    assert sum("vmcnt(0)" in o for o in lops) <= n_bar // 4  # the tail branch of each unrolled step


@pytest.fixture(scope="module")
def attn_isa(tmp_path_factory):
    return _isa("k_attn.hip", tmp_path_factory.mktemp("attn"))


@pytest.mark.parametrize("inst,max_exec_branches", [("attn_mfma16_kernelILi0E", 2), ("attn_mfma16_kernelILi1E", 4)])
def test_split_attention_loop_has_no_predicated_gathers(attn_isa, inst, max_exec_branches):
    """Round 1's attention kernel spent its time in 16 exec-masked branches per key tile (the Shaw term's LDS gather sunk
    under the key mask, one full LDS wait each).  The fp16-split kernel fetches those terms unconditionally ahead of the
    matrix instructions: the key-tile loop holds 24 MFMAs (12 for K.Q^T, 12 for V^T.P^T), two barriers, the exec-masked
    regions of the K/V prefetch only, and no scratch."""
    fn = _function(attn_isa, inst)
    assert not any("scratch_" in l for l in fn)
    loop = _innermost_loop_with(fn, "v_mfma_f32_32x32x16", "s_barrier", "v_exp_f32")
    ops = _ops(loop)
    assert sum(o.startswith("v_mfma_f32_32x32x16") for o in ops) == 24
    assert sum(o.startswith("s_barrier") for o in ops) == 2
    assert sum(o.startswith("s_cbranch_execz") or o.startswith("s_cbranch_execnz") for o in ops) <= max_exec_branches
    assert sum(o.startswith("v_exp_f32") for o in ops) == 17  # 16 probabilities + the rescale factor, on the hardware exp2
