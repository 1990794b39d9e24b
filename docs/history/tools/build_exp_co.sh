#!/bin/bash
# code object with the numbered experiment variants of the assembly kernels -> build/gemm_asm_exp.co (loaded through PSAM_GEMM_ASM_CO;
# build/ is git-ignored but travels to the GPU box). Extra arguments go to the generator.
set -e
cd "$(dirname "$0")/../protosam_amd/csrc"
CLANG=/opt/rocm/lib/llvm/bin/clang
mkdir -p ../../build
python3 gemm_asm_gen.py --experiments "$@" > ../../build/gemm_asm_exp.s
$CLANG -x assembler -target amdgcn-amd-amdhsa -mcpu=gfx950 -c ../../build/gemm_asm_exp.s -o ../../build/gemm_asm_exp.o
$CLANG -target amdgcn-amd-amdhsa -mcpu=gfx950 ../../build/gemm_asm_exp.o -o ../../build/gemm_asm_exp.co
rm -f ../../build/gemm_asm_exp.o
ls -la ../../build/gemm_asm_exp.co
