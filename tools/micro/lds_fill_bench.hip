// Micro-benchmark: how fast can ONE CU (8 waves, one workgroup per CU, 256 workgroups) move 64 KiB per iteration from an
// L2-resident source into LDS, (1) with global_load_lds_dwordx4 (LDS-DMA), (2) through registers (global_load_dwordx4 +
// ds_write_b128), each alone and with the GEMM's fragment-read load beside it (24 ds_read_b128 per wave per iteration)?
// Prints shader cycles per iteration (s_memtime, median over workgroups). Build: hipcc --offload-arch=gfx950 -O3
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <vector>
typedef _Float16 half_t;
typedef float f4 __attribute__((ext_vector_type(4)));

template <int N> __device__ __forceinline__ void wait_vmcnt() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

template <int MODE, int READS>
__global__ __launch_bounds__(512) void fill_kernel(const char* __restrict__ src, size_t region, int iters,
                                                   unsigned long long* __restrict__ cyc, float* __restrict__ sink) {
  extern __shared__ __attribute__((aligned(16))) char lds[];   // 128 KiB: two 64 KiB buffers
  const int t = threadIdx.x, lane = t & 63;
  const int wv = __builtin_amdgcn_readfirstlane(t >> 6);
  // every XCD (blockIdx & 7) streams the same `region` bytes, so the source is L2-resident after the first pass
  const char* base = src + (size_t)(blockIdx.x & 7) * region;
  f4 accv = {0.f, 0.f, 0.f, 0.f};
  __syncthreads();
  const unsigned long long c0 = __builtin_amdgcn_s_memtime();
  size_t off = ((size_t)(blockIdx.x >> 3) * 65536) % region;
  for (int it = 0; it < iters; ++it) {
    char* buf = lds + (it & 1) * 65536;
    // 64 KiB per iteration = 8 waves x 8 pieces x 1 KiB
    if (MODE == 0) {
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const char* g = base + off + (size_t)(j * 8 + wv) * 1024 + lane * 16;
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g,
                                         (__attribute__((address_space(3))) void*)(buf + (j * 8 + wv) * 1024), 16, 0, 0);
      }
    } else {
      f4 r[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) r[j] = *reinterpret_cast<const f4*>(base + off + (size_t)(j * 8 + wv) * 1024 + lane * 16);
#pragma unroll
      for (int j = 0; j < 8; ++j) *reinterpret_cast<f4*>(buf + (j * 8 + wv) * 1024 + lane * 16) = r[j];
    }
    if (READS) {   // the other buffer, conflict-free linear reads
      const char* rb = lds + ((it + 1) & 1) * 65536;
#pragma unroll
      for (int j = 0; j < READS; ++j) {
        const f4 v = *reinterpret_cast<const f4*>(rb + ((j * 8 + wv) & 63) * 1024 + lane * 16);
        accv += v;
      }
    }
    wait_vmcnt<0>();
    __syncthreads();
    off += 65536 * 32;
    if (off >= region) off -= region;
  }
  const unsigned long long c1 = __builtin_amdgcn_s_memtime();
  if (t == 0) cyc[blockIdx.x] = c1 - c0;
  if (accv[0] + accv[1] + accv[2] + accv[3] == 123.456f) sink[0] = accv[0];
}

template <int MODE, int READS>
static void run(const char* name, const char* src, size_t region, unsigned long long* dcyc, float* sink) {
  const int iters = 400;
  hipFuncSetAttribute((const void*)fill_kernel<MODE, READS>, hipFuncAttributeMaxDynamicSharedMemorySize, 131072);
  for (int rep = 0; rep < 2; ++rep) {
    hipLaunchKernelGGL((fill_kernel<MODE, READS>), dim3(256), dim3(512), 131072, 0, src, region, iters, dcyc, sink);
    hipDeviceSynchronize();
  }
  std::vector<unsigned long long> h(256);
  hipMemcpy(h.data(), dcyc, 256 * 8, hipMemcpyDeviceToHost);
  std::sort(h.begin(), h.end());
  const double per = (double)h[128] / iters;
  printf("%-44s %8.0f cycles / 64 KiB  = %5.1f B/clk/CU\n", name, per, 65536.0 / per);
}

int main() {
  const size_t region = 2u << 20;   // 2 MiB per XCD
  char* src; unsigned long long* dcyc; float* sink;
  hipMalloc(&src, region * 8); hipMemset(src, 1, region * 8);
  hipMalloc(&dcyc, 256 * 8); hipMalloc(&sink, 64);
  run<0, 0>("LDS-DMA alone", src, region, dcyc, sink);
  run<1, 0>("global_load + ds_write_b128 alone", src, region, dcyc, sink);
  run<0, 24>("LDS-DMA + 24 ds_read_b128 per wave", src, region, dcyc, sink);
  run<1, 24>("registers + 24 ds_read_b128 per wave", src, region, dcyc, sink);
  run<0, 48>("LDS-DMA + 48 ds_read_b128 per wave", src, region, dcyc, sink);
  run<1, 48>("registers + 48 ds_read_b128 per wave", src, region, dcyc, sink);
  return 0;
}
