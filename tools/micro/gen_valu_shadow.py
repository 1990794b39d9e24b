#!/usr/bin/env python3
"""Generates tools/micro/valu_shadow_bench.hip: how many VALU instructions does ONE wave per SIMD hide behind its MFMAs?
A loop of 64 back-to-back MFMAs on 8 independent accumulators, each followed by N independent fillers (v_fma_f32 / v_exp_f32 /
v_pk_fma_f32); prints shader cycles per MFMA for v_mfma_f32_32x32x16_f16 (32 cycles of the matrix pipe) and v_mfma_f32_16x16x32_f16
(16 cycles). One workgroup of 256 threads per CU (160 KB of LDS requested), 256 workgroups."""
variants = []
for kind in ("32", "16", "16v", "16a", "16va"):
    for fill in ("fma", "exp", "pk", "salu", "mix"):
        if kind != "32" and fill in ("salu", "mix"):
            continue
        if kind in ("16v", "16a", "16va") and fill != "fma":
            continue
        for n in range(0, 9 if kind == "32" else 6 if kind == "16" else 5):
            if n == 0 and fill != "fma":
                continue
            variants.append((kind, fill, n))

out = ['#include <hip/hip_runtime.h>', '#include <cstdio>', '#include <vector>', '#include <algorithm>', '']
for (kind, fill, n) in variants:
    body = []
    fi = 0
    for m in range(64):
        blk = m % 8
        if kind == "32":
            body.append("v_mfma_f32_32x32x16_f16 a[%d:%d], v[4:7], v[8:11], a[%d:%d]" % (blk * 16, blk * 16 + 15, blk * 16, blk * 16 + 15))
        elif kind == "16":
            body.append("v_mfma_f32_16x16x32_f16 a[%d:%d], v[4:7], v[8:11], a[%d:%d]" % (blk * 4, blk * 4 + 3, blk * 4, blk * 4 + 3))
        else:
            # the global attention kernel's forms: scores accumulate in VGPRs (v64..), operands come from AGPRs (a96.. / a100..)
            acc = ("v[%d:%d]" % (64 + blk * 4, 64 + blk * 4 + 3)) if "v" in kind[2:] else ("a[%d:%d]" % (blk * 4, blk * 4 + 3))
            ab = "a[96:99], a[100:103]" if "a" in kind[2:] else "v[4:7], v[8:11]"
            body.append("v_mfma_f32_16x16x32_f16 %s, %s, %s" % (acc, ab, acc))
        for j in range(n):
            r = 16 + 2 * (fi % 12)
            fi += 1
            if fill == "fma":
                body.append("v_fma_f32 v%d, v%d, v12, v13" % (r, r))
            elif fill == "exp":
                body.append("v_exp_f32 v%d, v%d" % (r, r))
            elif fill == "salu":
                body.append("s_add_u32 s%d, s%d, 1" % (40 + fi % 8, 40 + fi % 8))
            elif fill == "mix":        # 4 plain VALU + (n - 4) scalar instructions per MFMA
                if j < 4:
                    body.append("v_fma_f32 v%d, v%d, v12, v13" % (r, r))
                else:
                    body.append("s_add_u32 s%d, s%d, 1" % (40 + fi % 8, 40 + fi % 8))
            else:
                body.append("v_pk_fma_f32 v[%d:%d], v[%d:%d], v[12:13], v[14:15]" % (r, r + 1, r, r + 1))
    name = "k_%s_%s_%d" % (kind, fill, n)
    clob = ", ".join('"v%d"' % i for i in list(range(4, 40)) + list(range(64, 96))) + ", " + ", ".join('"a%d"' % i for i in range(128))
    asm = "\\n\\t".join(
        ["s_memtime %0", "s_waitcnt lgkmcnt(0)"] +
        ["v_mov_b32 v%d, 0x3c003c00" % i for i in range(4, 12)] + ["v_mov_b32 v%d, 0x3f000000" % i for i in range(12, 16)] +
        ["v_mov_b32 v%d, 0" % i for i in list(range(16, 40)) + list(range(64, 96))] + ["v_accvgpr_write_b32 a%d, 0" % i for i in range(128)] +
        ["v_accvgpr_write_b32 a%d, v4" % i for i in range(96, 104)] +
        ["L_loop_%=:"] + body + ["s_sub_u32 %2, %2, 1", "s_cmp_lg_u32 %2, 0", "s_cbranch_scc1 L_loop_%=", "s_nop 7", "s_nop 7", "s_memtime %1", "s_waitcnt lgkmcnt(0)"])
    out.append('__global__ __launch_bounds__(512, 1) void %s(unsigned long long* out, int iters) {' % name)
    out.append('  extern __shared__ char lds[];')
    out.append('  unsigned long long c0, c1; int it = iters;')
    out.append('  asm volatile("%s" : "=s"(c0), "=s"(c1), "+s"(it) : : %s, "memory", "scc", "s40", "s41", "s42", "s43", "s44", "s45", "s46", "s47");' % (asm, clob))
    out.append('  if ((threadIdx.x & 63) == 0) out[blockIdx.x * 8 + (threadIdx.x >> 6)] = c1 - c0;')
    out.append('  if (iters < 0) lds[threadIdx.x] = 1;')
    out.append('}')
out.append('int main() {')
out.append('  unsigned long long* d; hipMalloc(&d, 256 * 8 * 8); std::vector<unsigned long long> h(2048);')
out.append('  const int iters = 200;')
for (kind, fill, n) in variants:
    name = "k_%s_%s_%d" % (kind, fill, n)
    out.append('  hipFuncSetAttribute((const void*)%s, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);' % name)
    out.append('  for (int rep = 0; rep < 2; ++rep) { hipLaunchKernelGGL(%s, dim3(256), dim3(256), 160 * 1024, 0, d, iters); hipDeviceSynchronize(); }' % name)
    out.append('  hipMemcpy(h.data(), d, 1024 * 8, hipMemcpyDeviceToHost); std::sort(h.begin(), h.end());')
    out.append('  printf("%s x%s: %d fillers per MFMA: %%.1f cycles per MFMA (median wave)\\n", (double)h[512] / (iters * 64.0));' % ("v_mfma_32x32x16" if kind == "32" else "v_mfma_16x16x32" + {"16": "", "16v": " (C/D in VGPRs)", "16a": " (A/B from AGPRs)", "16va": " (C/D in VGPRs, A/B from AGPRs)"}[kind], fill, n))
for (kind, fill, n) in variants:
    if kind != "16" or fill == "pk":
        continue
    name = "k_%s_%s_%d" % (kind, fill, n)
    out.append('  for (int rep = 0; rep < 2; ++rep) { hipLaunchKernelGGL(%s, dim3(256), dim3(512), 160 * 1024, 0, d, iters); hipDeviceSynchronize(); }' % name)
    out.append('  hipMemcpy(h.data(), d, 2048 * 8, hipMemcpyDeviceToHost); std::sort(h.begin(), h.end());')
    out.append('  printf("TWO WAVES PER SIMD v_mfma_16x16x32 x%s: %d fillers per MFMA: %%.1f cycles per MFMA and wave = %%.1f per MFMA of the SIMD (median wave)\\n", (double)h[1024] / (iters * 64.0), (double)h[1024] / (iters * 128.0));' % (fill, n))
out.append('  return 0; }')
open("tools/micro/valu_shadow_bench.hip", "w").write("\n".join(out) + "\n")
