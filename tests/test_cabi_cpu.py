"""CPU: the C-ABI shared library builds for gfx950, loads, and exports every
symbol that include/seamless_hip.h declares; the ctypes binding covers the same
set.  No compute call is made (there is no GPU here)."""
import ctypes
import re
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parent.parent
HEADER = ROOT / "include" / "seamless_hip.h"


def declared_symbols():
    text = re.sub(r"/\*.*?\*/", "", HEADER.read_text(), flags=re.S)
    names = re.findall(r"\b(sc_[a-z0-9_]+)\s*\(", text)
    return sorted(set(names))


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
