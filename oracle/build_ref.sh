#!/usr/bin/env bash
# ORACLE-SIDE build recipe (test infrastructure).  Compiles the reference's own
# kaldi-native-fbank C/C++ sources, from where they lie under /root/reference,
# plus oracle/knf_ref_wrap.cc into oracle/_ref/libknf_ref.so.  No reference
# source is copied into this repository; oracle/_ref/ is git-ignored but travels
# to the GPU box with the tree.  The reference's cmake build is NOT used.
set -euo pipefail
HERE="$(cd "$(dirname "${BASH_SOURCE[0]}")" && pwd)"
REF="${REFERENCE_ROOT:-/root/reference}/ggml/examples"
KNF="$REF/kaldi-native-fbank/csrc"
OUT="$HERE/_ref"
if [ ! -d "$KNF" ]; then
    echo "build_ref.sh: $KNF not found (no reference tree on this machine) - keeping prebuilt files" >&2
    exit 0
fi
mkdir -p "$OUT/obj"
if [ -f "$OUT/libknf_ref.so" ] && [ "$OUT/libknf_ref.so" -nt "$HERE/knf_ref_wrap.cc" ]; then
    exit 0
fi
for f in feature-fbank feature-functions feature-window mel-computations rfft log; do
    g++ -O2 -fPIC -std=c++14 -I"$REF" -c "$KNF/$f.cc" -o "$OUT/obj/$f.o"
done
gcc -O2 -fPIC -c "$KNF/fftsg.c" -o "$OUT/obj/fftsg.o"
g++ -O2 -fPIC -std=c++14 -I"$REF" -c "$HERE/knf_ref_wrap.cc" -o "$OUT/obj/knf_ref_wrap.o"
g++ -shared -o "$OUT/libknf_ref.so" "$OUT"/obj/*.o -lm
rm -rf "$OUT/obj"
echo "built $OUT/libknf_ref.so"
