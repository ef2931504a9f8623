#!/bin/bash
# A/B of decoder builds on the GPU box: tools/ab_decode.sh out_file lib1 lib2 ...   (libraries under ab/, or "product")
# Each library decodes the same synthetic frames (tools/decode_sweep.py: round trip checked) at the library's own launch rule.
out=$1; shift
mkdir -p "$(dirname "$out")"
: > "$out"
for lib in "$@"; do
    arg=""
    [ "$lib" != "product" ] && arg="--lib $lib"
    echo "=== $lib: 1024 frames in flight (and 64, 256)" >> "$out"
    python tools/decode_sweep.py --frames 1024 --sizes 64,256,1024 --groups -1 --repeat 1 $arg >> "$out" 2>&1
    echo "=== $lib: 4096 frames in flight (64 distinct)" >> "$out"
    python tools/decode_sweep.py --frames 4096 --distinct 64 --sizes 4096 --groups -1 --repeat 1 $arg >> "$out" 2>&1
    echo "=== $lib: natural image (tulips tiled), 1024 and 4096 frames in flight" >> "$out"
    python tools/decode_sweep.py --frames 4096 --kind tulips --sizes 1024,4096 --groups -1 --repeat 1 $arg >> "$out" 2>&1
done
