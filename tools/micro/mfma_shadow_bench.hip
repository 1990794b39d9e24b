// Micro-benchmark: what does ONE wave per SIMD (4 waves per CU, 256 workgroups) pay for memory instructions placed
// between back-to-back v_mfma_f32_32x32x16_f16 (32 cycles each)? Per group of 4 MFMAs: nothing / 2 ds_read_b128 /
// 1 global_load_lds_dwordx4 / both. Prints shader cycles per MFMA (median over workgroups).
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <vector>
typedef _Float16 half8_t __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f4 __attribute__((ext_vector_type(4)));

template <int READS, int DMA>
__global__ __launch_bounds__(256, 1) void k(const char* __restrict__ src, int iters, unsigned long long* cyc, float* sink) {
  extern __shared__ __attribute__((aligned(16))) char lds[];
  const int t = threadIdx.x, lane = t & 63;
  const int wv = __builtin_amdgcn_readfirstlane(t >> 6);
  f32x16 acc[8];
  for (int i = 0; i < 8; ++i) for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
  half8_t fa[4], fb[4], nf[4];
  f4 stage[8];
  for (int i = 0; i < 8; ++i) stage[i] = f4{0.f, 0.f, 0.f, 0.f};
  for (int i = 0; i < 4; ++i) for (int e = 0; e < 8; ++e) nf[i][e] = (_Float16)0.f;
  for (int i = 0; i < 4; ++i) for (int e = 0; e < 8; ++e) { fa[i][e] = (_Float16)(lane * 0.001f + i); fb[i][e] = (_Float16)(0.5f - i); }
  const char* base = src + (size_t)(blockIdx.x & 7) * (1 << 20);
  __syncthreads();
  const unsigned long long c0 = __builtin_amdgcn_s_memtime();
  for (int it2 = 0; it2 < iters; it2 += 2) {
#pragma unroll
  for (int hh = 0; hh < 2; ++hh) {
    const int it = it2 + hh;
#pragma unroll
    for (int g = 0; g < 4; ++g) {     // 4 groups of 4 MFMAs = 16 MFMAs per iteration
#pragma unroll
      for (int m = 0; m < 4; ++m) {
        acc[(g & 1) * 4 + m] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa[m], fb[g], acc[(g & 1) * 4 + m], 0, 0, 0);
        if (READS && m < READS) {
          // fragments of the NEXT group / iteration: a second register set, consumed 4+ MFMAs later
          nf[(g * READS + m) & 3] = *reinterpret_cast<const half8_t*>(lds + ((it + g * 2 + m) & 63) * 1024 + lane * 16);
        }
        if (DMA == 2 && m == 3) {   // through registers: the load of this group, the LDS write of the previous group's data
          const char* gp = base + (size_t)(((it * 4 + g) * 4 + wv) & 1023) * 1024 + lane * 16;
          // 8 loads (two iterations) in flight: the write takes the register loaded two iterations ago
          *reinterpret_cast<f4*>(lds + 65536 + ((g * 4 + wv) & 31) * 1024 + lane * 16) = stage[hh * 4 + g];
          stage[hh * 4 + g] = *reinterpret_cast<const f4*>(gp);
        }
        if (DMA == 1 && m == 3) {
          const char* gp = base + (size_t)(((it * 4 + g) * 4 + wv) & 1023) * 1024 + lane * 16;
          __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)gp,
                                           (__attribute__((address_space(3))) void*)(lds + 65536 + ((g * 4 + wv) & 31) * 1024), 16, 0, 0);
        }
        __builtin_amdgcn_sched_barrier(0);
      }
    }
    if (READS) {   // rotate the fragment sets (the reads of this iteration feed the next one)
#pragma unroll
      for (int i = 0; i < 4; ++i) { half8_t tmp = fa[i]; fa[i] = nf[i]; nf[i] = tmp; }
    }
    if (DMA == 1) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
  }
  }
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
  const unsigned long long c1 = __builtin_amdgcn_s_memtime();
  if (t == 0) cyc[blockIdx.x] = c1 - c0;
  float s = 0.f;
  for (int i = 0; i < 8; ++i) s += acc[i][lane & 15];
  for (int i = 0; i < 8; ++i) s += stage[i][i & 3];
  if (s == 123.456f) sink[0] = s;
}

template <int READS, int DMA>
static void run(const char* name, const char* src, unsigned long long* dcyc, float* sink) {
  const int iters = 2000;
  (void)hipFuncSetAttribute((const void*)k<READS, DMA>, hipFuncAttributeMaxDynamicSharedMemorySize, 131072);
  for (int rep = 0; rep < 2; ++rep) {
    hipLaunchKernelGGL((k<READS, DMA>), dim3(256), dim3(256), 131072, 0, src, iters, dcyc, sink);
    (void)hipDeviceSynchronize();
  }
  std::vector<unsigned long long> h(256);
  (void)hipMemcpy(h.data(), dcyc, 256 * 8, hipMemcpyDeviceToHost);
  std::sort(h.begin(), h.end());
  printf("%-52s %6.1f cycles per MFMA\n", name, (double)h[128] / iters / 16.0);
}

int main() {
  char* src; unsigned long long* dcyc; float* sink;
  (void)hipMalloc(&src, 8u << 20); (void)hipMemset(src, 1, 8u << 20);
  (void)hipMalloc(&dcyc, 256 * 8); (void)hipMalloc(&sink, 64);
  run<0, 0>("MFMA only", src, dcyc, sink);
  run<2, 0>("+ 2 ds_read_b128 per 4 MFMAs", src, dcyc, sink);
  run<4, 0>("+ 4 ds_read_b128 per 4 MFMAs", src, dcyc, sink);
  run<0, 1>("+ 1 global_load_lds_dwordx4 per 4 MFMAs", src, dcyc, sink);
  run<2, 1>("+ 2 ds_read_b128 + 1 LDS-DMA per 4 MFMAs", src, dcyc, sink);
  run<0, 2>("+ 1 global_load_dwordx4 + 1 ds_write_b128 per 4 MFMAs", src, dcyc, sink);
  run<2, 2>("+ 2 ds_read_b128 + 1 load + 1 ds_write_b128 per 4 MFMAs", src, dcyc, sink);
  return 0;
}
