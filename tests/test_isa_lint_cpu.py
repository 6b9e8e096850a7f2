"""CPU: ISA-level regression checks of the latency-critical loops (hipcc cross-compiles gfx950 without a GPU).

profiles/r1_skinny_isa_notes.txt records why the shipped decoder-step kernels sit at 4-17 us per launch: predicated loads
become exec-masked branches and the compiler then waits with vmcnt(0) before every consumer.  The rewritten kernels avoid
that by construction; these tests keep it that way: the slab loop of skinny2_kernel must stay ONE basic block with no
`s_waitcnt vmcnt(0)`, no exec-masked branch and no accumulator moves, and the prefetching attention loop must not wait
for its prefetch before the matrix instructions of the current tile."""
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
def skinny2_isa(tmp_path_factory):
    return _isa("k_skinny2.hip", tmp_path_factory.mktemp("skinny2"))


@pytest.mark.parametrize("inst,mfmas,loads", [("skinny2_kernelILi1ELi1EE", 16, 24), ("skinny2_kernelILi2ELi1EE", 32, 40),
                                              ("skinny2_kernelILi1ELi4EE", 64, 48)])
def test_skinny2_slab_loop_is_straight_line_and_pipelined(skinny2_isa, inst, mfmas, loads):
    loop = _innermost_loop_with(_function(skinny2_isa, inst), "v_mfma_f32_32x32x16")
    text = "\n".join(loop)
    assert text.count("v_mfma_f32_32x32x16") == mfmas        # two slabs per trip, hi + lo per fragment
    assert text.count("buffer_load_dwordx4") == loads        # both register sets are refilled inside the trip
    assert "vmcnt(0)" not in text                            # never drains the loads of the next slab
    assert "s_cbranch_execz" not in text and "s_cbranch_execnz" not in text   # no predicated loads
    assert sum(1 for l in loop if re.match(r"^\.LBB", l)) == 1                # one basic block
    assert "v_accvgpr" not in text                           # accumulators stay in the accumulation registers
    assert "scratch_" not in text


def test_attention_prefetch_is_not_drained_before_the_matrix_instructions(tmp_path):
    isa = _isa("k_attn.hip", tmp_path)
    for inst in ("attn_mfma_kernelILb1ELb1EE", "attn_mfma_kernelILb0ELb1EE"):
        loop = _innermost_loop_with(_function(isa, inst), "v_mfma_f32_32x32x2", "global_load_dwordx4")
        idx_load = [i for i, l in enumerate(loop) if "global_load_dwordx4" in l]
        idx_mfma = [i for i, l in enumerate(loop) if "v_mfma_f32_32x32x2" in l]
        assert len(idx_load) == 4 and len(idx_mfma) == 64
        assert max(idx_load) < min(idx_mfma)                 # the prefetch is issued before the tile is multiplied
        between = "\n".join(loop[max(idx_load): min(idx_mfma)])
        assert "vmcnt(0)" not in between and not re.search(r"vmcnt\([0-3]\)", between)   # ... and not waited for there
    # the shipped instantiation has no prefetch: its loads are consumed (stored to LDS) before the barrier
    shipped = _innermost_loop_with(_function(isa, "attn_mfma_kernelILb1ELb0EE"), "v_mfma_f32_32x32x2", "global_load_dwordx4")
    first_mfma = min(i for i, l in enumerate(shipped) if "v_mfma_f32_32x32x2" in l)
    last_load = max(i for i, l in enumerate(shipped) if "global_load_dwordx4" in l)
    assert "vmcnt(0)" in "\n".join(shipped[last_load:first_mfma])  # the tile is waited for before it is multiplied
