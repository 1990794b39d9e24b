// Shared device helpers for the protosam_amd HIP kernels (gfx950 / CDNA4 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef _Float16 half_t;
typedef _Float16 half2_t __attribute__((ext_vector_type(2)));
typedef _Float16 half4_t __attribute__((ext_vector_type(4)));
typedef _Float16 half8_t __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

#define PSAM_OK 0
#define PSAM_ERR_ARG 1
#define PSAM_ERR_LAUNCH 2

#define WAVE 64

static inline int psam_launch_status() {
  hipError_t e = hipGetLastError();
  return e == hipSuccess ? PSAM_OK : PSAM_ERR_LAUNCH;
}

// CU / XCD counts of the current device, read once per translation unit (persistent grids, XCD-aware maps)
static inline void psam_device_geometry(int* cus, int* xcds) {
  static int g_cus = 0, g_xcds = 0;
  if (g_cus == 0) {
    int dev = 0, v = 0;
    (void)hipGetDevice(&dev);
    g_cus = (hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && v > 0) ? v : 256;
    g_xcds = (hipDeviceGetAttribute(&v, hipDeviceAttributeNumberOfXccs, dev) == hipSuccess && v > 0) ? v : 8;
    (void)hipGetLastError();
  }
  *cus = g_cus;
  *xcds = g_xcds;
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}
__device__ __forceinline__ float wave_min(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fminf(v, __shfl_xor(v, o, 64));
  return v;
}

// erf-form GELU (torch.nn.GELU() default). erf by Abramowitz & Stegun 7.1.26 (|error| <= 1.5e-7, far below the fp16
// rounding of every consumer) on the hardware exp2 / rcp, with the 1 / sqrt(2) of erf(x / sqrt(2)) folded into the constants:
// 14 VALU operations per element (the assembly GEMM epilogues of csrc/gemm_asm*_gen.py issue exactly this sequence, so every
// GEMM kernel produces the same bits; the epilogue of the 256x256 assembly tile is VALU-issue bound on it).
__device__ __forceinline__ float gelu_erf(float x) {
  const float ax = fabsf(x);
  const float t = __builtin_amdgcn_rcpf(fmaf(ax, 0.23164189f, 1.0f));               // 1 / (1 + p |x| / sqrt(2))
  float poly = fmaf(1.061405429f, t, -1.453152027f);
  poly = fmaf(poly, t, 1.421413741f);
  poly = fmaf(poly, t, -0.284496736f);
  poly = fmaf(poly, t, 0.254829592f);
  const float e = __builtin_amdgcn_exp2f((x * x) * -0.72134752044448170f);          // exp(-x^2 / 2)
  const float y = fmaf(-poly * t, e, 1.0f);                                         // |erf(x / sqrt(2))|
  const float h = 0.5f * x;
  return fmaf(h, copysignf(y, x), h);
}

// XCD-aware bijective block remap (8 XCDs; block b runs on XCD b % 8): gives each XCD a
// contiguous chunk of the logical tile space so neighbouring tiles share an L2.
__host__ __device__ __forceinline__ int xcd_remap(int bid, int nwg) {
  const int nx = 8;
  int xcd = bid % nx, idx = bid / nx;
  int q = nwg / nx, r = nwg % nx;
  int base = xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
  return base + idx;
}
