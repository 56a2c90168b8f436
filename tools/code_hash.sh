#!/bin/bash
# Developer tool: hash of the .text section of the gfx950 code object inside a library -- two builds whose device code is
# identical bit for bit (a refactoring that must not move the ISA) print the same 16 hex digits.   usage: tools/code_hash.sh lib.so
d=$(mktemp -d); /opt/rocm/lib/llvm/bin/llvm-objcopy -O binary --only-section=.hip_fatbin $1 $d/fat.bin && /opt/rocm/lib/llvm/bin/clang-offload-bundler --unbundle --type=o --input=$d/fat.bin --targets=hipv4-amdgcn-amd-amdhsa--gfx950 --output=$d/k.co && /opt/rocm/lib/llvm/bin/llvm-objcopy -O binary --only-section=.text $d/k.co $d/text.bin && sha256sum $d/text.bin | cut -c1-16; rm -rf $d
