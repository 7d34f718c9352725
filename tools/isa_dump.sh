#!/bin/bash
# Disassemble one HIP source's gfx950 code object: tools/isa_dump.sh csrc-file.hip out.s [extra hipcc flags]
# (per-kernel VGPR / scratch figures are in the .s metadata at the end; used to compare codegen before / after a refactor)
set -e
src=$1; out=$2; shift 2
/opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 -std=c++17 -fPIC -Wno-unused-result "$@" --cuda-device-only -S -o "$out" "$src"
