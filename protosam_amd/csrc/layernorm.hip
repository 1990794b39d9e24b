// Row LayerNorm (eps inside sqrt, affine), fp32 in -> fp16 or fp32 out. One wavefront per row,
// whole row held in registers (float4 per lane), two-pass mean/variance with 64-lane butterflies.
//
// Replaces torch.nn.LayerNorm on the hot path:
//   * DINOv2 Block.norm1/norm2 and final norm (eps 1e-6)   [external hub model; call site
//     /root/reference/models/grid_proto_fewshot.py:90-91]
//   * SAM Block.norm1/norm2 (models/segment_anything/modeling/image_encoder.py:174-193, eps 1e-6 via
//     build_sam.py:72)
//   * TwoWayAttentionBlock.norm4 on the 4096 image tokens (modeling/transformer.py:176-180, eps 1e-5)
//   * LayerNorm2d of the SAM neck when the map is kept token-major [HW, C]
//     (modeling/common.py:31-43; image_encoder.py:97,105)
// HBM-bound: reads 4*D bytes, writes 2*D (fp16) or 4*D (fp32) per row.
#include "common.h"
#include <stdlib.h>

#define LN_MAXV 8  // supports D <= 64 * 4 * 8 = 2048

template <typename OutT>
__global__ __launch_bounds__(256) void layernorm_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                        const float* __restrict__ b, OutT* __restrict__ y,
                                                        float* __restrict__ y2, int M, int D, int ldx, int ldy,
                                                        float eps, int zero_tail_rows, int rev) {
  // rev: walk the rows from the last to the first. The residual stream this kernel reads was written front to back by the GEMM
  // before it and (at 16 slices: 335 MB) does not fit the 256 MB memory-side cache: reading it back to front meets the most
  // recently written rows first, while they are still resident
  const int row = (rev ? (int)gridDim.x - 1 - (int)blockIdx.x : (int)blockIdx.x) * 4 + (threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  const int nv = D >> 2;
  if (row >= M) {
    // optional zero rows appended after the M real rows (window-attention pad token)
    if (row < M + zero_tail_rows) {
      for (int i = lane; i < nv; i += 64) {
        OutT* yr = y + (size_t)row * ldy + i * 4;
        yr[0] = (OutT)0.f; yr[1] = (OutT)0.f; yr[2] = (OutT)0.f; yr[3] = (OutT)0.f;
      }
    }
    return;
  }
  const float4* xr = reinterpret_cast<const float4*>(x + (size_t)row * ldx);
  const float4* w4 = reinterpret_cast<const float4*>(w);
  const float4* b4 = reinterpret_cast<const float4*>(b);
  float4 v[LN_MAXV], wv[LN_MAXV], bv[LN_MAXV];
  float s = 0.f;
#pragma unroll
  for (int k = 0; k < LN_MAXV; ++k) {
    int i = lane + 64 * k;
    if (i < nv) v[k] = xr[i];
  }
  // the affine parameters are requested right behind the row (they return in order, after it): one exposed latency per wave
  // instead of two
#pragma unroll
  for (int k = 0; k < LN_MAXV; ++k) {
    int i = lane + 64 * k;
    if (i < nv) { wv[k] = w4[i]; bv[k] = b4[i]; }
  }
#pragma unroll
  for (int k = 0; k < LN_MAXV; ++k) {
    int i = lane + 64 * k;
    if (i < nv) s += (v[k].x + v[k].y) + (v[k].z + v[k].w);
  }
  const float mean = wave_sum(s) / (float)D;
  float q = 0.f;
#pragma unroll
  for (int k = 0; k < LN_MAXV; ++k) {
    int i = lane + 64 * k;
    if (i < nv) {
      float a = v[k].x - mean, c = v[k].y - mean, d = v[k].z - mean, e = v[k].w - mean;
      q += (a * a + c * c) + (d * d + e * e);
    }
  }
  const float var = wave_sum(q) / (float)D;
  const float rstd = 1.0f / sqrtf(var + eps);
#pragma unroll
  for (int k = 0; k < LN_MAXV; ++k) {
    int i = lane + 64 * k;
    if (i < nv) {
      float4 ww = wv[k], bb = bv[k];
      float o0 = (v[k].x - mean) * rstd * ww.x + bb.x;
      float o1 = (v[k].y - mean) * rstd * ww.y + bb.y;
      float o2 = (v[k].z - mean) * rstd * ww.z + bb.z;
      float o3 = (v[k].w - mean) * rstd * ww.w + bb.w;
      OutT* yr = y + (size_t)row * ldy + i * 4;
      if (sizeof(OutT) == 2) {
        half4_t h = {(half_t)o0, (half_t)o1, (half_t)o2, (half_t)o3};
        *reinterpret_cast<half4_t*>(yr) = h;
      } else {
        *reinterpret_cast<float4*>(yr) = make_float4(o0, o1, o2, o3);
      }
      if (y2) *reinterpret_cast<float4*>(y2 + (size_t)row * ldy + i * 4) = make_float4(o0, o1, o2, o3);
    }
  }
}

// out_dtype: 0 = fp16, 1 = fp32. y2 (optional, fp32) receives a second copy when out is fp16.
extern "C" int psam_layernorm(const float* x, const float* w, const float* b, void* y, float* y2, int M, int D,
                              int ldx, int ldy, float eps, int out_dtype, int zero_tail_rows, void* stream) {
  if (M <= 0 || D <= 0 || (D & 3) || D > 64 * 4 * LN_MAXV || (ldx & 3) || (ldy & 3)) return PSAM_ERR_ARG;
  dim3 grid((M + zero_tail_rows + 3) / 4), block(256);
  hipStream_t s = (hipStream_t)stream;
  static int rev = -1;
  if (rev < 0) { const char* e = getenv("PSAM_LN_REVERSE"); rev = e ? atoi(e) : 0; }
  if (out_dtype == 0)
    hipLaunchKernelGGL(layernorm_kernel<half_t>, grid, block, 0, s, x, w, b, (half_t*)y, y2, M, D, ldx, ldy, eps,
                       zero_tail_rows, rev);
  else
    hipLaunchKernelGGL(layernorm_kernel<float>, grid, block, 0, s, x, w, b, (float*)y, (float*)nullptr, M, D, ldx,
                       ldy, eps, zero_tail_rows, rev);
  return psam_launch_status();
}


// Folded LayerNorm (psam_gemm_f16_ln): per-row partial (sum, sum of squares) over 64-column groups, written by the epilogue of
// the residual-stream GEMM, -> (mean, rstd) per row for the consuming GEMM's epilogue. Deterministic (fixed summation order,
// no atomics); biased variance as nn.LayerNorm, E[x^2] - mean^2 in fp32 (|mean| << std on a ViT residual stream; clamped at 0).
__global__ void ln_finalize_kernel(const float* __restrict__ stats, int M, int parts, float inv_d, float eps,
                                   float* __restrict__ mr) {
  // four lanes per row (round 5: one thread per row read its 160 bytes with 64 different lines per load instruction; 5.9 -> ~3.5 us per
  // 65536-row call, 64 calls per step): lane `sub` adds parts sub, sub + 4, ...; two shuffles; a fixed order either way
  const int gid = blockIdx.x * blockDim.x + threadIdx.x;
  const int m = gid >> 2, sub = gid & 3;
  const bool live = m < M;
  const float2* s = reinterpret_cast<const float2*>(stats) + (size_t)(live ? m : 0) * parts;
  float s1 = 0.f, s2 = 0.f;
  for (int i = sub; i < parts; i += 4) {
    const float2 v = s[i];
    s1 += v.x;
    s2 += v.y;
  }
  s1 += __shfl_xor(s1, 1); s2 += __shfl_xor(s2, 1);
  s1 += __shfl_xor(s1, 2); s2 += __shfl_xor(s2, 2);
  if (!live || sub != 0) return;
  const float mean = s1 * inv_d;
  const float var = fmaxf(s2 * inv_d - mean * mean, 0.f);
  reinterpret_cast<float2*>(mr)[m] = make_float2(mean, 1.0f / sqrtf(var + eps));
  // behind the M (mean, rstd) pairs: -mean as the MFMA operand of the assembly GEMM's rank-1 correction, fp16 {hi, hi, lo, 0 x 5}
  // in the k slots 0..7 (multiplied with the {hi, lo, hi, 0 x 5} fragment of s: hi hi + lo hi + hi lo, fp32-accurate)
  const half_t hi = (half_t)(-mean);
  const half_t lo = (half_t)(-mean - (float)hi);
  half8_t f;
  f[0] = hi; f[1] = hi; f[2] = lo;
#pragma unroll
  for (int e = 3; e < 8; ++e) f[e] = (half_t)0.f;
  reinterpret_cast<half8_t*>(mr + 2 * (size_t)M)[m] = f;
}

extern "C" int psam_ln_finalize(const float* stats, int M, int D, float eps, float* mr, void* stream) {
  if (M <= 0 || D <= 0 || (D % 64) != 0) return PSAM_ERR_ARG;
  hipLaunchKernelGGL(ln_finalize_kernel, dim3((M + 63) / 64), dim3(256), 0, (hipStream_t)stream, stats, M, D / 64,
                     1.0f / (float)D, eps, mr);
  return psam_launch_status();
}
