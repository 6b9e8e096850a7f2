"""CPU: the C-ABI shared library builds for gfx950, loads, and exports every
symbol that include/seamless_hip.h declares; the ctypes binding covers the same
set.  No compute call is made (there is no GPU here)."""
import ctypes
import re
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parent.parent
HEADER = ROOT / "include" / "seamless_hip.h"                    # the drop-in boundary
INTERNAL_HEADER = ROOT / "include" / "seamless_hip_internal.h"  # kernel-level test hooks, dispatch introspection


def _symbols_of(path):
    text = re.sub(r"/\*.*?\*/", "", path.read_text(), flags=re.S)
    return sorted(set(re.findall(r"\b(sc_[a-z0-9_]+)\s*\(", text)))


def declared_symbols():
    return sorted(set(_symbols_of(HEADER)) | set(_symbols_of(INTERNAL_HEADER)))


def test_public_header_holds_the_boundary_only():
    """What a binding of the reference needs and nothing else: no kernel-level hooks, no
    dispatch introspection, none of the experiment switches of earlier rounds."""
    public, internal = _symbols_of(HEADER), _symbols_of(INTERNAL_HEADER)
    assert not [s for s in public if s.startswith("sc_op_")]
    assert not set(public) & set(internal)
    for gone in ("sc_set_cu_partition", "sc_set_decoder_priority", "sc_device_cu_count"):
        assert gone not in public and gone not in internal
    assert "sc_decoder_step_family" in internal and all(s.startswith("sc_op_") or s == "sc_decoder_step_family" for s in internal)
    text = (ROOT / "INTEGRATION.md").read_text()
    assert not [s for s in public if s not in text], "INTEGRATION.md must name every entry of the boundary"


@pytest.fixture(scope="module")
def lib_path():
    from seamless_communication_amd import build

    return build.build()


def test_header_declares_expected_surface():
    syms = declared_symbols()
    for must in ("sc_load", "sc_free", "sc_fbank", "sc_encode_speech", "sc_generate_text", "sc_decode_text",
                 "sc_t2u_nar", "sc_get_units", "sc_vocode", "sc_last_error", "sc_abi_version"):
        assert must in syms


def test_library_exports_every_declared_symbol(lib_path):
    lib = ctypes.CDLL(str(lib_path))
    missing = [s for s in declared_symbols() if not hasattr(lib, s)]
    assert not missing, f"declared in seamless_hip.h but not exported: {missing}"


def test_binding_covers_header_and_abi_version(lib_path):
    from seamless_communication_amd import _lib

    assert sorted(_lib.SIGNATURES) == declared_symbols()
    lib = _lib.load_library()
    assert lib.sc_abi_version() == _lib.SC_ABI_VERSION
    m = re.search(r"#define\s+SC_ABI_VERSION\s+(\d+)", HEADER.read_text())
    assert int(m.group(1)) == _lib.SC_ABI_VERSION


def test_struct_layout_matches_header():
    """sizeof(sc_config)/sizeof(sc_gen_opts)/sizeof(sc_tensor_desc) as a C compiler sees the header."""
    import subprocess
    import tempfile

    from seamless_communication_amd import _lib

    src = '#include <stdio.h>\n#include "seamless_hip.h"\nint main(){printf("%zu %zu %zu\\n", sizeof(sc_config), sizeof(sc_gen_opts), sizeof(sc_tensor_desc));return 0;}\n'
    with tempfile.TemporaryDirectory() as d:
        c = Path(d) / "t.c"
        c.write_text(src)
        subprocess.check_call(["gcc", "-I", str(ROOT / "include"), str(c), "-o", str(Path(d) / "t")])
        out = subprocess.check_output([str(Path(d) / "t")]).decode().split()
    assert [int(x) for x in out] == [ctypes.sizeof(_lib.sc_config), ctypes.sizeof(_lib.sc_gen_opts),
                                     ctypes.sizeof(_lib.sc_tensor_desc)]


def test_product_path_fails_loudly_without_gpu():
    import torch

    if torch.cuda.is_available():
        pytest.skip("a GPU is visible")
    from seamless_communication_amd._lib import SeamlessHipError
    from seamless_communication_amd.inference import Translator

    with pytest.raises(ValueError):
        Translator("seamlessM4T_v2_large", None, device="cpu")  # HIP only, no CPU fallback
    from seamless_communication_amd.config import tiny_config
    from seamless_communication_amd.runtime import HipS2STModel
    from seamless_communication_amd import synthetic as syn

    cfg = tiny_config()
    with pytest.raises(SeamlessHipError):
        HipS2STModel(cfg, syn.make_unity_state_dict(cfg), None, device=0)


def test_host_array_arguments_convert_like_the_runtime_passes_them():
    """The runtime hands numpy arrays / tensors over as c_void_p (runtime._ptr).  Calling the stage entry points with a
    NULL handle exercises exactly that ctypes conversion and returns SC_ERR_INVALID before any device work."""
    import numpy as np

    from seamless_communication_amd import _lib
    from seamless_communication_amd.runtime import _i32, _ptr

    lib = _lib.load_library()
    tok, lens = _i32(np.zeros((1, 4))), _i32([4])
    buf = np.zeros(8, dtype=np.float32)
    o = _lib.sc_gen_opts()
    assert lib.sc_encode_text(None, _ptr(tok), 1, 4, _ptr(lens), _ptr(buf)) == -1
    assert lib.sc_encode_speech(None, _ptr(buf), 1, 2, _ptr(lens), _ptr(buf), _ptr(lens)) == -1
    assert lib.sc_generate_text(None, _ptr(buf), 1, 2, _ptr(lens), ctypes.byref(o), _ptr(tok), 2, _ptr(tok), _ptr(lens),
                                _ptr(buf), _ptr(None)) == -1
    assert lib.sc_decode_text(None, _ptr(buf), 1, 2, _ptr(lens), _ptr(tok), 4, _ptr(buf)) == -1
    assert b"null" in lib.sc_last_error() or b"bad argument" in lib.sc_last_error()


def test_integration_guide_names_every_entry_point():
    """INTEGRATION.md is the reference-side view of the boundary: every exported entry (the kernel-level `sc_op_*` test
    hooks are covered as a family) is named there with the reference call site it replaces."""
    from pathlib import Path

    text = (Path(__file__).resolve().parents[1] / "INTEGRATION.md").read_text()
    missing = [s for s in declared_symbols() if not s.startswith("sc_op_") and s not in text]
    assert not missing, missing
    assert "sc_op_*" in text


def test_switches_that_change_results_need_the_debug_gate(lib_path):
    """One table for every SC_* variable the library reads (csrc/common.cpp), read once per process.  A stray SC_SPLIT_MODE /
    SC_DECODER_GEN1 / ... in the environment of a drop-in library must not change its numbers: those are honoured only together
    with SC_DEBUG_NUMERICS=1; schedule switches (same bits) pass."""
    import os
    import subprocess
    import sys

    code = ("import ctypes; from seamless_communication_amd import _lib; l = _lib.load_library(); "
            "print(l.sc_op_knob(b'SC_SPLIT_MODE', 0), l.sc_op_knob(b'SC_DECODER_GEN1', -1), l.sc_op_knob(b'SC_VOC_STREAMS', 3), l.sc_op_knob(b'SC_PS_TILE', 256))")

    def run(**env):
        e = {k: v for k, v in os.environ.items() if not k.startswith("SC_")}
        e.update(env)
        r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, cwd=str(ROOT), env=e, timeout=120)
        assert r.returncode == 0, r.stderr[-1000:]
        return r.stdout.split()

    assert run() == ["0", "-1", "3", "256"]
    assert run(SC_SPLIT_MODE="1", SC_DECODER_GEN1="1", SC_VOC_STREAMS="1", SC_PS_TILE="128") == ["0", "-1", "1", "128"]
    assert run(SC_SPLIT_MODE="1", SC_DECODER_GEN1="1", SC_DEBUG_NUMERICS="1") == ["1", "1", "3", "256"]


def test_every_switch_the_sources_read_is_in_the_table(lib_path):
    """knob::value / is_set / live abort on a name that is missing from the table in csrc/common.cpp (they are called with
    literals; a miss is a programming error that would otherwise fire the first time that code path runs - possibly inside a
    stream capture).  Every call site of the sources is checked here; the test hook `sc_op_knob` answers an unknown name with
    the default instead of aborting."""
    import re

    csrc = ROOT / "seamless_communication_amd" / "csrc"
    table = set(re.findall(r'\{"(SC_[A-Z0-9_]+)",\s*[012],', (csrc / "common.cpp").read_text()))
    assert len(table) > 20
    used = {}
    for f in sorted(csrc.glob("*.hip")) + sorted(csrc.glob("*.cpp")) + sorted(csrc.glob("*.h")):
        for m in re.finditer(r'(?:knob::(?:value|is_set|live)|env_int|env_set)\(\s*"(SC_[A-Z0-9_]+)"', f.read_text()):
            used.setdefault(m.group(1), f.name)
    missing = {k: v for k, v in used.items() if k not in table}
    assert not missing, missing
    from seamless_communication_amd import _lib

    lib = _lib.load_library()
    assert lib.sc_op_knob(b"SC_NO_SUCH_SWITCH", 7) == 7
