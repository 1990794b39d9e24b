// Issue rate of the fp32-input MFMAs on gfx950: cycles per instruction with 4 independent accumulators, 1 / 2 / 4 waves per SIMD.
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
template <int KIND>
__global__ void k(float* out, long long* cyc, int iters) {
  f32x16 a0 = {0}, a1 = {0}, a2 = {0}, a3 = {0};
  f32x4 b0 = {0}, b1 = {0}, b2 = {0}, b3 = {0};
  float x = threadIdx.x * 1e-3f, y = 1.0f + threadIdx.x * 1e-4f;
  long long t0 = __builtin_readcyclecounter();
  for (int i = 0; i < iters; ++i) {
    if (KIND == 0) {
      a0 = __builtin_amdgcn_mfma_f32_32x32x2f32(x, y, a0, 0, 0, 0);
      a1 = __builtin_amdgcn_mfma_f32_32x32x2f32(x, y, a1, 0, 0, 0);
      a2 = __builtin_amdgcn_mfma_f32_32x32x2f32(x, y, a2, 0, 0, 0);
      a3 = __builtin_amdgcn_mfma_f32_32x32x2f32(x, y, a3, 0, 0, 0);
    } else if (KIND == 2) {   // runs of four on the SAME accumulator, three accumulators (the ALP similarity kernel's order)
      a0 = __builtin_amdgcn_mfma_f32_32x32x2f32(x, y, a0, 0, 0, 0);
      a0 = __builtin_amdgcn_mfma_f32_32x32x2f32(y, x, a0, 0, 0, 0);
      a0 = __builtin_amdgcn_mfma_f32_32x32x2f32(x, y, a0, 0, 0, 0);
      a0 = __builtin_amdgcn_mfma_f32_32x32x2f32(y, x, a0, 0, 0, 0);
      a1 = __builtin_amdgcn_mfma_f32_32x32x2f32(x, y, a1, 0, 0, 0);
      a1 = __builtin_amdgcn_mfma_f32_32x32x2f32(y, x, a1, 0, 0, 0);
      a1 = __builtin_amdgcn_mfma_f32_32x32x2f32(x, y, a1, 0, 0, 0);
      a1 = __builtin_amdgcn_mfma_f32_32x32x2f32(y, x, a1, 0, 0, 0);
      a2 = __builtin_amdgcn_mfma_f32_32x32x2f32(x, y, a2, 0, 0, 0);
      a2 = __builtin_amdgcn_mfma_f32_32x32x2f32(y, x, a2, 0, 0, 0);
      a2 = __builtin_amdgcn_mfma_f32_32x32x2f32(x, y, a2, 0, 0, 0);
      a2 = __builtin_amdgcn_mfma_f32_32x32x2f32(y, x, a2, 0, 0, 0);
    } else {
      b0 = __builtin_amdgcn_mfma_f32_16x16x4f32(x, y, b0, 0, 0, 0);
      b1 = __builtin_amdgcn_mfma_f32_16x16x4f32(x, y, b1, 0, 0, 0);
      b2 = __builtin_amdgcn_mfma_f32_16x16x4f32(x, y, b2, 0, 0, 0);
      b3 = __builtin_amdgcn_mfma_f32_16x16x4f32(x, y, b3, 0, 0, 0);
    }
  }
  long long t1 = __builtin_readcyclecounter();
  float s = 0;
  for (int r = 0; r < 16; ++r) s += a0[r] + a1[r] + a2[r] + a3[r];
  for (int r = 0; r < 4; ++r) s += b0[r] + b1[r] + b2[r] + b3[r];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if (threadIdx.x == 0 && blockIdx.x == 0) *cyc = t1 - t0;
}
int main() {
  float* out; long long* cyc; hipMalloc(&out, 1 << 24); hipMalloc(&cyc, 8);
  const int iters = 20000;
  for (int kind = 0; kind < 3; ++kind)
    for (int waves = 4; waves <= 16; waves *= 2) {   // waves per CU (block = waves * 64 threads, one block per CU)
      hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
      hipEventRecord(e0);
      if (kind == 0) hipLaunchKernelGGL(k<0>, dim3(256), dim3(waves * 64), 0, 0, out, cyc, iters);
      else if (kind == 2) hipLaunchKernelGGL(k<2>, dim3(256), dim3(waves * 64), 0, 0, out, cyc, iters);
      else hipLaunchKernelGGL(k<1>, dim3(256), dim3(waves * 64), 0, 0, out, cyc, iters);
      hipEventRecord(e1); hipEventSynchronize(e1);
      float ms; hipEventElapsedTime(&ms, e0, e1);
      long long c; hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
      const int per = kind == 2 ? 12 : 4;
      const double flops = (kind == 1 ? 1024.0 : 4096.0) * per * iters * waves * 256;
      printf("%s  %2d waves/CU: %.1f clock-counter ticks per MFMA per wave, %.1f TFLOP/s\n",
             kind == 0 ? "32x32x2f32 (4 independent)" : kind == 2 ? "32x32x2f32 (runs of 4 dependent)" : "16x16x4f32", waves,
             (double)c / ((double)per * iters), flops / (ms * 1e-3) / 1e12);
    }
  return 0;
}
