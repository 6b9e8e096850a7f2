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


def _stamp(src: Path, flags=FLAGS) -> str:
    h = hashlib.sha256()
    h.update(src.read_bytes())
    for hdr in sorted(CSRC.glob("*.h")) + sorted((CSRC.parent.parent / "include").glob("*.h")):
        h.update(hdr.read_bytes())
    h.update(" ".join(flags).encode())
    return h.hexdigest()


def _compile(src_name: str, verbose: bool, obj_dir: Path = OBJ_DIR, flags=FLAGS) -> Path:
    src = CSRC / src_name
    obj = obj_dir / (src_name + ".o")
    stamp = obj_dir / (src_name + ".stamp")
    digest = _stamp(src, flags)
    if obj.exists() and stamp.exists() and stamp.read_text() == digest:
        return obj
    cmd = [_hipcc(), *flags, "-x", "hip", "-c", str(src), "-o", str(obj)]
    if verbose:
        print(" ".join(cmd), flush=True)
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError(f"hipcc failed for {src_name}:\n{r.stdout}\n{r.stderr}")
    if r.stderr.strip() and verbose:
        print(r.stderr, file=sys.stderr)
    stamp.write_text(digest)
    return obj


def build(verbose: bool = False, force: bool = False, variant: str = "", extra_flags=()) -> Path:
    """`variant` (kernel development): objects under build/<variant>/, library libseamless_hip.<variant>.so, FLAGS + extra_flags;
    loaded with SC_LIB_VARIANT=<variant>."""
    obj_dir = OBJ_DIR / variant if variant else OBJ_DIR
    lib = LIB.with_name(f"libseamless_hip.{variant}.so") if variant else LIB
    flags = [*FLAGS, *extra_flags]
    obj_dir.mkdir(parents=True, exist_ok=True)
    if force:
        for f in obj_dir.glob("*.stamp"):
            f.unlink()
    with ThreadPoolExecutor(max_workers=min(8, os.cpu_count() or 4)) as ex:
        objs = list(ex.map(lambda s: _compile(s, verbose, obj_dir, flags), SOURCES))
    newest = max(o.stat().st_mtime for o in objs)
    LIB_ = lib
    if not LIB_.exists() or LIB_.stat().st_mtime < newest:
        cmd = [_hipcc(), "--offload-arch=gfx950", "-shared", "-fPIC", *map(str, objs), "-o", str(LIB_)]
        if verbose:
            print(" ".join(cmd), flush=True)
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"link failed:\n{r.stdout}\n{r.stderr}")
    return LIB_


if __name__ == "__main__":
    import argparse
    import shlex

    ap = argparse.ArgumentParser()
    ap.add_argument("--force", action="store_true")
    ap.add_argument("--variant", default="")
    ap.add_argument("--flags", default="", help="extra hipcc flags of a --variant build")
    a = ap.parse_args()
    print(build(verbose=True, force=a.force, variant=a.variant, extra_flags=shlex.split(a.flags)))
