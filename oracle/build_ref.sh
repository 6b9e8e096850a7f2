#!/usr/bin/env bash
# ORACLE-SIDE build recipe (test infrastructure).  Compiles, from where they lie under
# /root/reference, the reference's own native sources for this path:
#   * kaldi-native-fbank (ggml/examples/kaldi-native-fbank/csrc) + oracle/knf_ref_wrap.cc
#       -> oracle/_ref/libknf_ref.so     (fbank, a1)
#   * ggml (ggml/src/*.c) + the fairseq2 restatement ggml/examples/unity/fairseq2.cpp +
#     oracle/ggml_ref_wrap.cc
#       -> oracle/_ref/libggml_ref.so    (LayerNorm / Linear / FFN / MHA / encoder + decoder layers /
#                                          adaptor layer / embedding frontend / beam search)
# No reference source is copied into this repository; oracle/_ref/ is git-ignored but travels to
# the GPU box with the tree.  The reference's cmake build is NOT used.
set -euo pipefail
HERE="$(cd "$(dirname "${BASH_SOURCE[0]}")" && pwd)"
GGML="${REFERENCE_ROOT:-/root/reference}/ggml"
REF="$GGML/examples"
KNF="$REF/kaldi-native-fbank/csrc"
OUT="$HERE/_ref"
if [ ! -d "$KNF" ]; then
    echo "build_ref.sh: $KNF not found (no reference tree on this machine) - keeping prebuilt files" >&2
    exit 0
fi
mkdir -p "$OUT"

if [ ! -f "$OUT/libknf_ref.so" ] || [ "$HERE/knf_ref_wrap.cc" -nt "$OUT/libknf_ref.so" ]; then
    O="$OUT/obj_knf"; mkdir -p "$O"
    for f in feature-fbank feature-functions feature-window mel-computations rfft log; do
        g++ -O2 -fPIC -std=c++14 -I"$REF" -c "$KNF/$f.cc" -o "$O/$f.o"
    done
    gcc -O2 -fPIC -c "$KNF/fftsg.c" -o "$O/fftsg.o"
    g++ -O2 -fPIC -std=c++14 -I"$REF" -c "$HERE/knf_ref_wrap.cc" -o "$O/knf_ref_wrap.o"
    g++ -shared -o "$OUT/libknf_ref.so" "$O"/*.o -lm
    rm -rf "$O"
    echo "built $OUT/libknf_ref.so"
fi

if [ ! -f "$OUT/libggml_ref.so" ] || [ "$HERE/ggml_ref_wrap.cc" -nt "$OUT/libggml_ref.so" ]; then
    O="$OUT/obj_ggml"; mkdir -p "$O"
    INC="-I$GGML/include -I$GGML/include/ggml -I$GGML/src -I$REF -I$REF/unity"
    pids=()
    for f in ggml ggml-alloc ggml-backend ggml-quants; do
        gcc -O2 -fPIC -mavx2 -mfma -mf16c -D_GNU_SOURCE $INC -c "$GGML/src/$f.c" -o "$O/$f.o" 2>/dev/null &
        pids+=($!)
    done
    for f in feature-fbank feature-functions feature-window mel-computations rfft log; do
        g++ -O2 -fPIC -std=c++14 -I"$REF" -c "$KNF/$f.cc" -o "$O/knf-$f.o" &
        pids+=($!)
    done
    gcc -O2 -fPIC -c "$KNF/fftsg.c" -o "$O/fftsg.o" &
    pids+=($!)
    g++ -O2 -fPIC -std=c++14 $INC -c "$REF/unity/fairseq2.cpp" -o "$O/fairseq2.o" 2>/dev/null &
    pids+=($!)
    g++ -O2 -fPIC -std=c++14 $INC -c "$HERE/ggml_ref_wrap.cc" -o "$O/ggml_ref_wrap.o" &
    pids+=($!)
    for p in "${pids[@]}"; do wait "$p"; done
    g++ -shared -o "$OUT/libggml_ref.so" "$O"/*.o -lpthread -lm
    rm -rf "$O"
    echo "built $OUT/libggml_ref.so"
fi
