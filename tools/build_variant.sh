#!/bin/bash
# tools/build_variant.sh <name> "<extra hipcc flags>": a second build of the library for same-box A/B runs —
# splat_renderer_amd/_variants/libsplat_<name>.so, selected with SPLAT_LIB_PATH (splat_renderer_amd/_lib.py).
set -e
name=$1; flags=$2
root=$(cd "$(dirname "$0")/.." && pwd)
mkdir -p "$root/splat_renderer_amd/_variants"
make -s -C "$root/splat_renderer_amd/csrc" -j8 BUILD=_obj_$name OUT=../_variants/libsplat_$name.so EXTRA="$flags" lib
rm -rf "$root/splat_renderer_amd/csrc/_obj_$name"
echo "built splat_renderer_amd/_variants/libsplat_$name.so ($flags)"
