"""Builds libseamless_hip.so (gfx950 only) in-tree with hipcc.

``python -m seamless_communication_amd.build`` or :func:`build`.  hipcc
cross-compiles without a GPU; the .so lands next to the sources so that it
travels with the tree to the GPU box.
"""
from __future__ import annotations

import hashlib
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor
from pathlib import Path

CSRC = Path(__file__).resolve().parent / "csrc"
LIB = Path(__file__).resolve().parent / "libseamless_hip.so"
OBJ_DIR = CSRC / "build"
SOURCES = [
    "common.cpp", "prof.hip", "k_gemm.hip", "k_gemm2.hip", "k_gemm_ps.hip", "k_resblock.hip", "k_skinny.hip", "k_dstep.hip", "k_dstep3.hip", "k_norm.hip", "k_attn.hip", "k_fbank.hip", "k_misc.hip", "k_beam.hip", "k_engine.hip",
    "model_load.hip", "model_encoder.hip", "model_decoder.hip", "model_t2u.hip", "engine.hip", "api.hip",
]
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wall", "-Wno-unused-function"]


def _hipcc() -> str:
    for cand in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", "hipcc"):
        if cand and (os.path.sep not in cand or os.path.exists(cand)):
            return cand
    raise RuntimeError("hipcc not found")


def _stamp(src: Path) -> str:
    h = hashlib.sha256()
    h.update(src.read_bytes())
    for hdr in sorted(CSRC.glob("*.h")) + sorted((CSRC.parent.parent / "include").glob("*.h")):
        h.update(hdr.read_bytes())
    h.update(" ".join(FLAGS).encode())
    return h.hexdigest()


def _compile(src_name: str, verbose: bool) -> Path:
    src = CSRC / src_name
    obj = OBJ_DIR / (src_name + ".o")
    stamp = OBJ_DIR / (src_name + ".stamp")
    digest = _stamp(src)
    if obj.exists() and stamp.exists() and stamp.read_text() == digest:
        return obj
    cmd = [_hipcc(), *FLAGS, "-x", "hip", "-c", str(src), "-o", str(obj)]
    if verbose:
        print(" ".join(cmd), flush=True)
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError(f"hipcc failed for {src_name}:\n{r.stdout}\n{r.stderr}")
    if r.stderr.strip() and verbose:
        print(r.stderr, file=sys.stderr)
    stamp.write_text(digest)
    return obj


def build(verbose: bool = False, force: bool = False) -> Path:
    OBJ_DIR.mkdir(parents=True, exist_ok=True)
    if force:
        for f in OBJ_DIR.glob("*.stamp"):
            f.unlink()
    with ThreadPoolExecutor(max_workers=min(8, os.cpu_count() or 4)) as ex:
        objs = list(ex.map(lambda s: _compile(s, verbose), SOURCES))
    newest = max(o.stat().st_mtime for o in objs)
    if not LIB.exists() or LIB.stat().st_mtime < newest:
        cmd = [_hipcc(), "--offload-arch=gfx950", "-shared", "-fPIC", *map(str, objs), "-o", str(LIB)]
        if verbose:
            print(" ".join(cmd), flush=True)
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"link failed:\n{r.stdout}\n{r.stderr}")
    return LIB


if __name__ == "__main__":
    print(build(verbose=True, force="--force" in sys.argv))
