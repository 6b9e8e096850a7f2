"""Wall time per launch of each decoder-step kernel as a dependent chain in a replayed hipGraph (sc_op_chain_bench)."""
import ctypes as C
import sys
from pathlib import Path

import torch  # noqa: F401  (same process environment as the product path)

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
from seamless_communication_amd import _lib  # noqa: E402

lib = _lib.load_library()
names = ["add_i32", "reduce_ln S=4", "gemvp 1024x1024 S=4", "gemvp 1024x8192 S=8", "gemvp 8192x1024 planes", "dattn self pos=20",
         "dattn cross 63 keys", "gemvp 1024x1024 + reduce_ln alternating"]
for rows in (1, 32, 64):
    for kind, name in enumerate(names):
        us = C.c_float(0)
        st = lib.sc_op_chain_bench(kind, rows, 96, 10, C.byref(us))
        assert st == 0, lib.sc_last_error().decode()
        print(f"rows={rows:3d} {name:42s} {us.value:7.2f} us per launch", flush=True)
