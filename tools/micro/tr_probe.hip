#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>
typedef __fp16 fp16x4 __attribute__((__vector_size__(4 * sizeof(__fp16))));
__global__ void k(const _Float16* in, _Float16* out, int stride) {
  __shared__ __attribute__((aligned(16))) _Float16 s[4096];
  for (int i = threadIdx.x; i < 4096; i += 64) s[i] = in[i];
  __syncthreads();
  int l = threadIdx.x;
  int i = l & 15, g = l >> 4;
  // 16-lane group g reads the block rows g*4..g*4+3 (row stride `stride` halfs), 16 columns; lane i supplies row i/4, cols (i%4)*4..+3
  const _Float16* a = s + (g * 4 + (i >> 2)) * stride + (i & 3) * 4;
  fp16x4 v = __builtin_amdgcn_ds_read_tr16_b64_v4f16((__attribute__((address_space(3))) fp16x4*)a);
  for (int j = 0; j < 4; ++j) out[l * 4 + j] = (_Float16)v[j];
}
int main() {
  std::vector<_Float16> h(4096), o(256);
  const int stride = 80;
  for (int i = 0; i < 4096; ++i) h[i] = (_Float16)(float)((i / stride) * 100 + (i % stride));   // row*100 + col
  _Float16 *di, *dout;
  hipMalloc(&di, 8192); hipMalloc(&dout, 512);
  hipMemcpy(di, h.data(), 8192, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, di, dout, stride);
  hipMemcpy(o.data(), dout, 512, hipMemcpyDeviceToHost);
  for (int l = 0; l < 64; ++l) { printf("lane %2d:", l); for (int j = 0; j < 4; ++j) printf(" %6.0f", (float)o[l * 4 + j]); printf("\n"); }
  return 0;
}
