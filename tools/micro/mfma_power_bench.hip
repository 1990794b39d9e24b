// What does the chip sustain on v_mfma_f32_32x32x16_f16 alone? One workgroup of 4 waves per CU (160 KiB of LDS requested so that
// exactly one is resident), each wave issues back-to-back MFMAs on 8 independent accumulator blocks from registers - no memory
// traffic at all - for ~40 ms. Reports TFLOP/s from HIP events and the shader clock from s_memtime / s_memrealtime (100 MHz).
// Operands: random fp16 (|x| < 1), or zeros (no toggling), and the number of busy CUs is varied.
//   hipcc --offload-arch=gfx950 -O2 tools/micro/mfma_power_bench.hip -o tools/micro/mfma_power_bench && ./tools/micro/mfma_power_bench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>

typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float float16v __attribute__((ext_vector_type(16)));

__global__ __launch_bounds__(256, 1) void mfma_loop(const half8* __restrict__ src, float* __restrict__ sink, unsigned long long* __restrict__ stamps, int iters) {
  extern __shared__ char lds[];
  const int lane = threadIdx.x;
  half8 a[4], b[4];
  for (int i = 0; i < 4; ++i) { a[i] = src[(blockIdx.x * 8 + i) * 256 + lane]; b[i] = src[(blockIdx.x * 8 + 4 + i) * 256 + lane]; }
  float16v acc[8];
  for (int i = 0; i < 8; ++i) for (int j = 0; j < 16; ++j) acc[i][j] = 0.f;
  unsigned long long c0 = __builtin_readcyclecounter(), r0 = wall_clock64();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < 4; ++u) {
#pragma unroll
      for (int i = 0; i < 8; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[(i + u) & 3], b[i & 3], acc[i], 0, 0, 0);
    }
  }
  unsigned long long c1 = __builtin_readcyclecounter(), r1 = wall_clock64();
  float s = 0.f;
  for (int i = 0; i < 8; ++i) for (int j = 0; j < 16; ++j) s += acc[i][j];
  if (s == 123.456f) sink[0] = s;
  if (lane == 0) { stamps[blockIdx.x * 2] = c1 - c0; stamps[blockIdx.x * 2 + 1] = r1 - r0; }
  if (iters < 0) lds[lane] = 1;
}

int main() {
  const int maxwg = 256;
  std::vector<_Float16> h((size_t)maxwg * 8 * 256 * 8);
  half8* d; float* sink; unsigned long long* st;
  hipMalloc(&d, h.size() * 2); hipMalloc(&sink, 4); hipMalloc(&st, maxwg * 16);
  hipFuncSetAttribute((const void*)mfma_loop, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int data = 0; data < 2; ++data) {
    srand(1);
    for (auto& v : h) v = data ? (_Float16)((rand() / (float)RAND_MAX) * 2.f - 1.f) : (_Float16)0.f;
    hipMemcpy(d, h.data(), h.size() * 2, hipMemcpyHostToDevice);
    for (int wgs : {32, 64, 128, 192, 256}) {
      const int iters = 120000;   // x 32 MFMAs of 32 cycles = 123 M cycles per wave
      for (int rep = 0; rep < 2; ++rep) {
        hipEventRecord(e0);
        hipLaunchKernelGGL(mfma_loop, dim3(wgs), dim3(256), 160 * 1024, 0, d, sink, st, iters);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        std::vector<unsigned long long> s(wgs * 2);
        hipMemcpy(s.data(), st, wgs * 16, hipMemcpyDeviceToHost);
        double cyc = 0, rt = 0;
        for (int i = 0; i < wgs; ++i) { cyc += s[2 * i]; rt += s[2 * i + 1]; }
        const double flop = (double)wgs * 4 * iters * 32.0 * 2 * 32 * 32 * 16;
        if (rep) printf("%s operands, %3d workgroups (4 waves, one per SIMD): %7.1f ms  %7.1f TFLOP/s  shader clock %5.0f MHz  %.2f cycles per MFMA\n", data ? "random" : "zero  ", wgs,
                        ms, flop / ms / 1e9, cyc / rt * 100.0, cyc / wgs / (iters * 32.0));
      }
    }
  }
  return 0;
}
