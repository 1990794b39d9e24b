// Host-side carrier of the gfx950 code object assembled from gemm_asm.s (see gemm_asm_gen.py): the C-ABI library loads it with
// hipModuleLoadData at the first GEMM that selects the assembly kernels (csrc/gemm.hip, tile 15).
    .section .rodata
    .globl psam_gemm_asm_co
    .globl psam_gemm_asm_co_end
    .balign 4096
psam_gemm_asm_co:
    .incbin "gemm_asm.co"
psam_gemm_asm_co_end:
    .byte 0
    .section .note.GNU-stack,"",@progbits
