// SAM prompt encoder + two-way mask decoder kernels (token side in fp32; the 4096-token image side uses the
// fp16/fp32-acc GEMM of gemm.hip for its projections and the kernels below for everything else).
//
// Replaces models/segment_anything/modeling/prompt_encoder.py (PositionEmbeddingRandom :171-214, _embed_points
// :73-92, _embed_boxes :94-101, get_dense_pe :62-71), modeling/transformer.py (Attention :185-240,
// TwoWayAttentionBlock :109-182, TwoWayTransformer :16-106), modeling/mask_decoder.py (predict_masks :112-149,
// MLP :154-176) and the mask post-processing of modeling/sam.py (:133-161, :292-321) / models/ProtoSAM.py:669-676.
#include "common.h"

// =====================================================================================================
// small_linear: y[g,m,n] = act(sum_k (x[g,m,k] [+ x2[g,m,k]]) W[g,n,k] + b[g,n]) (+ resid[g,m,n]);  fp32, few rows.
// One wave per output column n: its W row sits in registers (KV = K/64 values per lane, coalesced), every x row
// is streamed from L1/L2 and reduced with a 64-lane butterfly.
template <int KV>
__global__ __launch_bounds__(256) void small_linear_kernel(const float* __restrict__ x, const float* __restrict__ x2,
                                                           const float* __restrict__ W,
                                                           const float* __restrict__ b, const float* __restrict__ resid,
                                                           float* __restrict__ y, int M, int N, long long xg, long long wg,
                                                           long long bg, long long yg, int ldx, int ldy, int act) {
  const int n = blockIdx.x * 4 + (threadIdx.x >> 6);
  const int g = blockIdx.y;
  const int lane = threadIdx.x & 63;
  if (n >= N) return;
  const float* wr = W + (size_t)g * wg + (size_t)n * (KV * 64);
  float w[KV];
#pragma unroll
  for (int i = 0; i < KV; ++i) w[i] = wr[lane + 64 * i];
  const float bias = b ? b[(size_t)g * bg + n] : 0.f;
  // RB rows at a time (round 5: one row per iteration was a chain of M dependent load -> 6-step butterfly -> store round trips, 8.7 us for
  // the 18 rows of a one-slice call): their loads are in flight together, the butterflies interleave, lane r stores row r
  constexpr int RB = KV <= 4 ? 8 : 2;
  for (int m0 = 0; m0 < M; m0 += RB) {
    float s[RB];
#pragma unroll
    for (int r = 0; r < RB; ++r) {
      const int m = min(m0 + r, M - 1);
      const float* xr = x + (size_t)g * xg + (size_t)m * ldx;
      float a = 0.f;
      if (x2) {  // fused `queries + query_pe` (transformer.py:157,163,177,98)
        const float* x2r = x2 + (size_t)g * xg + (size_t)m * ldx;
#pragma unroll
        for (int i = 0; i < KV; ++i) a += (xr[lane + 64 * i] + x2r[lane + 64 * i]) * w[i];
      } else {
#pragma unroll
        for (int i = 0; i < KV; ++i) a += xr[lane + 64 * i] * w[i];
      }
      s[r] = a;
    }
#pragma unroll
    for (int r = 0; r < RB; ++r) s[r] = wave_sum(s[r]);
    float mine = s[0];
#pragma unroll
    for (int r = 1; r < RB; ++r) mine = lane == r ? s[r] : mine;
    if (lane < RB && m0 + lane < M) {
      mine += bias;
      if (act == 1) mine = fmaxf(mine, 0.f);
      const size_t o = (size_t)g * yg + (size_t)(m0 + lane) * ldy + n;
      if (resid) mine += resid[o];
      y[o] = mine;
    }
  }
}

// Many rows (the automatic mask generator decodes hundreds of prompt sets at once): the same contraction as a 64x64 tile
// per block on the fp32 MFMA (v_mfma_f32_32x32x2f32), W as the A operand so each lane ends up with 4 consecutive output
// columns of one row (16 B stores). x (+ x2) and W tiles are staged through LDS in 32-wide k slabs.
#define SLM_LD 33
// VEC: every row of x (+ x2) and W starts on a 16-byte boundary - the k slab is requested as two float4 per operand and thread. (The
// element-wise form compiled to sixteen single-dword loads per slab with a branch and, for x + x2, a full vmcnt(0) wait per element:
// 23-31 us per call of the 243-row token-side linears of a 16-slice step whatever their grid.)
template <bool VEC>
__global__ __launch_bounds__(256) void small_linear_mfma_kernel(const float* __restrict__ x, const float* __restrict__ x2,
                                                                const float* __restrict__ W, const float* __restrict__ b,
                                                                const float* __restrict__ resid, float* __restrict__ y,
                                                                int M, int N, int K, long long xg, long long wg,
                                                                long long bg, long long yg, int ldx, int ldy, int act, int ldw) {
  __shared__ float Xs[64 * SLM_LD], Ws[64 * SLM_LD];
  const int g = blockIdx.z, m0 = blockIdx.x * 64, n0 = blockIdx.y * 64;
  const int t = threadIdx.x, wave = t >> 6, lane = t & 63, lr = lane & 31, lk = lane >> 5;
  const int wn = wave & 1, wm = wave >> 1;
  const int lrow = t >> 2, lcol = (t & 3) * 8;
  const float* xr = x + (size_t)g * xg + (size_t)(m0 + lrow) * ldx + lcol;
  const float* x2r = x2 ? x2 + (size_t)g * xg + (size_t)(m0 + lrow) * ldx + lcol : nullptr;
  const float* wr = W + (size_t)g * wg + (size_t)(n0 + lrow) * ldw + lcol;
  const bool xin = m0 + lrow < M, win = n0 + lrow < N;
  f32x16 acc = {0};
  // the k slab AFTER the one being multiplied is requested before its MFMAs (round 5): with sixteen workgroups on the chip - nine
  // tokens of 27 prompt sets against a 256 x 256 weight - nothing else hides the ~2 us of a global load, and the two-way block's
  // 2048 -> 256 projection walked its 64 slabs at that pace (186 us for 0.25 GFLOP)
  float xv[8], wv[8];
  auto fetch = [&](int k0) {
    if (VEC) {
      float4 a0 = make_float4(0.f, 0.f, 0.f, 0.f), a1 = a0, w0 = a0, w1 = a0;
      if (xin) {
        a0 = *reinterpret_cast<const float4*>(xr + k0);
        a1 = *reinterpret_cast<const float4*>(xr + k0 + 4);
        if (x2r) {
          const float4 p0 = *reinterpret_cast<const float4*>(x2r + k0), p1 = *reinterpret_cast<const float4*>(x2r + k0 + 4);
          a0.x += p0.x; a0.y += p0.y; a0.z += p0.z; a0.w += p0.w;
          a1.x += p1.x; a1.y += p1.y; a1.z += p1.z; a1.w += p1.w;
        }
      }
      if (win) {
        w0 = *reinterpret_cast<const float4*>(wr + k0);
        w1 = *reinterpret_cast<const float4*>(wr + k0 + 4);
      }
      xv[0] = a0.x; xv[1] = a0.y; xv[2] = a0.z; xv[3] = a0.w; xv[4] = a1.x; xv[5] = a1.y; xv[6] = a1.z; xv[7] = a1.w;
      wv[0] = w0.x; wv[1] = w0.y; wv[2] = w0.z; wv[3] = w0.w; wv[4] = w1.x; wv[5] = w1.y; wv[6] = w1.z; wv[7] = w1.w;
      return;
    }
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      xv[i] = xin ? xr[k0 + i] : 0.f;
      if (x2r && xin) xv[i] += x2r[k0 + i];
      wv[i] = win ? wr[k0 + i] : 0.f;
    }
  };
  fetch(0);
  for (int k0 = 0; k0 < K; k0 += 32) {
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      Xs[lrow * SLM_LD + lcol + i] = xv[i];
      Ws[lrow * SLM_LD + lcol + i] = wv[i];
    }
    __syncthreads();
    if (k0 + 32 < K) fetch(k0 + 32);
#pragma unroll
    for (int s2 = 0; s2 < 16; ++s2) {
      const float a = Ws[(wn * 32 + lr) * SLM_LD + 2 * s2 + lk];
      const float bq = Xs[(wm * 32 + lr) * SLM_LD + 2 * s2 + lk];
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, bq, acc, 0, 0, 0);
    }
  }
  // acc[r]: n = n0 + wn*32 + (r&3) + 8*(r>>2) + 4*lk ; m = m0 + wm*32 + lr
  const int m = m0 + wm * 32 + lr;
  if (m >= M) return;
  if (VEC && (N & 3) == 0) {      // (the launcher checked: bias / resid / y rows on 16-byte boundaries) four columns per 16-byte access
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int n = n0 + wn * 32 + 8 * q + 4 * lk;
      if (n >= N) break;
      float4 v = make_float4(acc[q * 4 + 0], acc[q * 4 + 1], acc[q * 4 + 2], acc[q * 4 + 3]);
      if (b) {
        const float4 bb = *reinterpret_cast<const float4*>(b + (size_t)g * bg + n);
        v.x += bb.x; v.y += bb.y; v.z += bb.z; v.w += bb.w;
      }
      if (act == 1) { v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f); }
      const size_t o = (size_t)g * yg + (size_t)m * ldy + n;
      if (resid) {
        const float4 r = *reinterpret_cast<const float4*>(resid + o);
        v.x += r.x; v.y += r.y; v.z += r.z; v.w += r.w;
      }
      *reinterpret_cast<float4*>(y + o) = v;
    }
    return;
  }
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const int n = n0 + wn * 32 + 8 * q + 4 * lk;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      if (n + i >= N) break;
      float v = acc[q * 4 + i] + (b ? b[(size_t)g * bg + n + i] : 0.f);
      if (act == 1) v = fmaxf(v, 0.f);
      const size_t o = (size_t)g * yg + (size_t)m * ldy + n + i;
      if (resid) v += resid[o];
      y[o] = v;
    }
  }
}

// y = sum over `ks` planes of parts[ks][M][N] + bias (+ resid): the second half of psam_small_linear_splitk
__global__ void sum_planes_kernel(const float* __restrict__ parts, int ks, const float* __restrict__ b, const float* __restrict__ resid,
                                  float* __restrict__ y, int M, int N, int ldy) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= M * N) return;
  const int m = i / N, n = i % N;
  float v = b ? b[n] : 0.f;
  for (int k = 0; k < ks; ++k) v += parts[(size_t)k * M * N + i];
  const size_t o = (size_t)m * ldy + n;
  if (resid) v += resid[o];
  y[o] = v;
}
// A token-side linear with a LONG contraction and few rows (the two-way block's MLP output, 2048 -> 256 on 9 tokens per prompt set,
// transformer.py:170-171 / common.py:13-26): sixteen 64x64 tiles walking 64 k slabs each leave 240 CUs idle (156 us for 0.25 GFLOP).
// Here K is cut into `ks` ranges that run as the groups of one small_linear_mfma launch (fp32 partial planes in `parts`, [ks][M][N],
// caller-owned: no hidden state), then one pass adds them in a fixed order with the bias and the residual. K % (64 ks) == 0.
extern "C" int psam_small_linear_splitk(const float* x, const float* W, const float* b, const float* resid, float* y, float* parts,
                                        int M, int N, int K, int ks, int ldx, int ldy, void* stream) {
  if (M <= 0 || N <= 0 || ks < 2 || (K % (64 * ks)) != 0 || !parts) return PSAM_ERR_ARG;
  hipStream_t s = (hipStream_t)stream;
  const int kr = K / ks;
  const bool vec = ((ldx | K | kr) & 3) == 0 && ((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(W) | reinterpret_cast<uintptr_t>(parts)) & 15) == 0 &&
                   (((long long)M * N) & 3) == 0;      // (the planes of `parts` are the groups' outputs, rows of N floats)
  if (vec)
    hipLaunchKernelGGL(small_linear_mfma_kernel<true>, dim3((M + 63) / 64, (N + 63) / 64, ks), dim3(256), 0, s, x, (const float*)nullptr, W,
                       (const float*)nullptr, (const float*)nullptr, parts, M, N, kr, (long long)kr, (long long)kr, 0LL, (long long)M * N, ldx, N, 0, K);
  else
    hipLaunchKernelGGL(small_linear_mfma_kernel<false>, dim3((M + 63) / 64, (N + 63) / 64, ks), dim3(256), 0, s, x, (const float*)nullptr, W,
                       (const float*)nullptr, (const float*)nullptr, parts, M, N, kr, (long long)kr, (long long)kr, 0LL, (long long)M * N, ldx, N, 0, K);
  hipLaunchKernelGGL(sum_planes_kernel, dim3((M * N + 255) / 256), dim3(256), 0, s, parts, ks, b, resid, y, M, N, ldy);
  return psam_launch_status();
}

extern "C" int psam_small_linear(const float* x, const float* x2, const float* W, const float* b, const float* resid,
                                 float* y, int G,
                                 int M, int N, int K, long long xg, long long wg, long long bg, long long yg, int ldx,
                                 int ldy, int act, void* stream) {
  if (G <= 0 || M <= 0 || N <= 0 || (K % 64) != 0) return PSAM_ERR_ARG;
  hipStream_t s = (hipStream_t)stream;
  if (M >= 32 && (K % 32) == 0) {
    static const int vec_on = [] { const char* e = getenv("PSAM_SMALL_LINEAR_VEC"); return e ? atoi(e) : 1; }();      // (0: element-wise loads, A/B)
    const bool vec = vec_on && ((ldx | K) & 3) == 0 && ((xg | wg) & 3) == 0 &&
                     ((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(W) | reinterpret_cast<uintptr_t>(x2)) & 15) == 0 &&
                     ((N & 3) != 0 || (((ldy | yg | bg) & 3) == 0 &&      // (N % 4 == 0: the epilogue moves four columns per access)
                                       ((reinterpret_cast<uintptr_t>(y) | reinterpret_cast<uintptr_t>(b) | reinterpret_cast<uintptr_t>(resid)) & 15) == 0));
    if (vec)
      hipLaunchKernelGGL(small_linear_mfma_kernel<true>, dim3((M + 63) / 64, (N + 63) / 64, G), dim3(256), 0, s, x, x2, W, b,
                         resid, y, M, N, K, xg, wg, bg, yg, ldx, ldy, act, K);
    else
      hipLaunchKernelGGL(small_linear_mfma_kernel<false>, dim3((M + 63) / 64, (N + 63) / 64, G), dim3(256), 0, s, x, x2, W, b,
                         resid, y, M, N, K, xg, wg, bg, yg, ldx, ldy, act, K);
    return psam_launch_status();
  }
  dim3 grid((N + 3) / 4, G), block(256);
#define SL(KV) hipLaunchKernelGGL(small_linear_kernel<KV>, grid, block, 0, s, x, x2, W, b, resid, y, M, N, xg, wg, bg, yg, ldx, ldy, act)
  switch (K / 64) {
    case 1: SL(1); break;
    case 2: SL(2); break;
    case 4: SL(4); break;
    case 32: SL(32); break;
    default: return PSAM_ERR_ARG;
  }
#undef SL
  return psam_launch_status();
}

// =====================================================================================================
// gemm_f32: out[m,n] = sum_k (a[m,k] [+ a2[m % a2_mod, k]]) * w[n,k] + bias[n] [+ resid[m,n]], everything fp32, on the
// exact-fp32 matrix instruction (v_mfma_f32_32x32x2_f32: a k-ordered fmaf chain, 157 TFLOP/s peak).
// The image side of the two-way decoder (transformer.py:163-167,176-180,98-103: k/v/q projections of the 4096 image
// tokens, the i2t out-projection; mask_decoder.py:137 ConvTranspose #1 as a GEMM) runs here so that sigmoid(low_res_masks)
// stays inside the 1e-3 parity budget: with fp16 operands (the encoder's GEMM) this stage alone measured 0.8e-3..1.2e-3.
// 3.2 GFLOP per prompt set = ~25 us at this rate; the encoder's 6 TFLOP per slice stay on the fp16 MFMA.
// Tile BM x BN (4 waves as 2 x 2, wave tile BM/2 x BN/2 in 32x32 blocks), K slabs of 32 staged through LDS ([row][33]
// floats: the one-dword fragment reads of a 32-lane half hit 32 different banks), next slab prefetched into registers
// under the MFMAs. W is the MFMA's A operand, so a lane ends up with 4 consecutive output columns of one row (16-byte stores).
#define GF_LD 33
template <int BM, int BN>
__global__ __launch_bounds__(256) void gemm_f32_kernel(const float* __restrict__ a, const float* __restrict__ a2, int a2_mod,
                                                       const float* __restrict__ w, const float* __restrict__ bias,
                                                       const float* __restrict__ resid, float* __restrict__ out, int M,
                                                       int N, int K, int lda, int ldw, int ldo, int hm_nk, int hm_hd) {
  constexpr int MI = BM / 64, NI = BN / 64;       // 32x32 blocks per wave in m / n
  constexpr int AV = BM * 8 / 256, WV = BN * 8 / 256;   // float4 loads per thread and slab
  __shared__ float As[BM * GF_LD], Ws[BN * GF_LD];
  const int t = threadIdx.x, wave = t >> 6, lane = t & 63, lr = lane & 31, lk = lane >> 5;
  const int wm = wave >> 1, wn = wave & 1;
  const int m0 = blockIdx.x * BM, n0 = blockIdx.y * BN;
  f32x16 acc[MI][NI];
#pragma unroll
  for (int i = 0; i < MI; ++i)
#pragma unroll
    for (int j = 0; j < NI; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
  float4 av[AV], wv[WV];
  auto load_slab = [&](int k0) {
#pragma unroll
    for (int i = 0; i < AV; ++i) {
      const int idx = t + 256 * i, row = idx >> 3, c4 = (idx & 7) * 4;
      const int m = m0 + row;
      float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
      if (m < M) {
        v = *reinterpret_cast<const float4*>(a + (size_t)m * lda + k0 + c4);
        if (a2) {
          const float4 p = *reinterpret_cast<const float4*>(a2 + (size_t)(m % a2_mod) * lda + k0 + c4);
          v.x += p.x; v.y += p.y; v.z += p.z; v.w += p.w;
        }
      }
      av[i] = v;
    }
#pragma unroll
    for (int i = 0; i < WV; ++i) {
      const int idx = t + 256 * i, row = idx >> 3, c4 = (idx & 7) * 4;
      wv[i] = *reinterpret_cast<const float4*>(w + (size_t)(n0 + row) * ldw + k0 + c4);   // N % BN == 0
    }
  };
  load_slab(0);
  for (int k0 = 0; k0 < K; k0 += 32) {
    __syncthreads();   // every wave is done reading the previous slab
#pragma unroll
    for (int i = 0; i < AV; ++i) {
      const int idx = t + 256 * i, row = idx >> 3, c4 = (idx & 7) * 4;
      float* d = As + row * GF_LD + c4;
      d[0] = av[i].x; d[1] = av[i].y; d[2] = av[i].z; d[3] = av[i].w;
    }
#pragma unroll
    for (int i = 0; i < WV; ++i) {
      const int idx = t + 256 * i, row = idx >> 3, c4 = (idx & 7) * 4;
      float* d = Ws + row * GF_LD + c4;
      d[0] = wv[i].x; d[1] = wv[i].y; d[2] = wv[i].z; d[3] = wv[i].w;
    }
    __syncthreads();
    if (k0 + 32 < K) load_slab(k0 + 32);
#pragma unroll
    for (int s2 = 0; s2 < 16; ++s2) {
      float xf[MI], wf[NI];
#pragma unroll
      for (int i = 0; i < MI; ++i) xf[i] = As[(wm * (BM / 2) + i * 32 + lr) * GF_LD + 2 * s2 + lk];
#pragma unroll
      for (int j = 0; j < NI; ++j) wf[j] = Ws[(wn * (BN / 2) + j * 32 + lr) * GF_LD + 2 * s2 + lk];
#pragma unroll
      for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < NI; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(wf[j], xf[i], acc[i][j], 0, 0, 0);
    }
  }
  // acc[i][j][r]: m = m0 + wm*BM/2 + i*32 + lr ; n = n0 + wn*BN/2 + j*32 + (r&3) + 8*(r>>2) + 4*lk
#pragma unroll
  for (int i = 0; i < MI; ++i) {
    const int m = m0 + wm * (BM / 2) + i * 32 + lr;
    if (m >= M) continue;
#pragma unroll
    for (int j = 0; j < NI; ++j)
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int n = n0 + wn * (BN / 2) + j * 32 + 8 * q + 4 * lk;
        float4 v = make_float4(acc[i][j][4 * q], acc[i][j][4 * q + 1], acc[i][j][4 * q + 2], acc[i][j][4 * q + 3]);
        if (bias) {
          const float4 b = *reinterpret_cast<const float4*>(bias + n);
          v.x += b.x; v.y += b.y; v.z += b.z; v.w += b.w;
        }
        if (resid) {
          const float4 r = *reinterpret_cast<const float4*>(resid + (size_t)m * ldo + n);
          v.x += r.x; v.y += r.y; v.z += r.z; v.w += r.w;
        }
        if (hm_nk > 0) {   // head-major: out[image = m / nk][head = n / hd][m % nk][n % hd] (psam_gemm_f32_heads)
          const size_t o = ((size_t)(m / hm_nk) * (N / hm_hd) + n / hm_hd) * hm_nk * hm_hd + (size_t)(m % hm_nk) * hm_hd + n % hm_hd;
          *reinterpret_cast<float4*>(out + o) = v;
        } else {
          *reinterpret_cast<float4*>(out + (size_t)m * ldo + n) = v;
        }
      }
  }
}

// gemm_f32x3: the same product at fp32 accuracy on the fp16 matrix pipe. a = ah + al and w = wh + wl (h = fp16(.), l = fp16(. - h): 22
// significant bits each), out = sum_k ah wh + ah wl + al wh in fp32 accumulation - the dropped al wl is 2^-22 of a term, like the
// rounding of the split itself (measured against a float64 product: tests/test_sam_gpu.py). Three v_mfma_f32_32x32x16_f16 (3 x 16 k per
// 32 cycles) replace eight v_mfma_f32_32x32x2_f32 (8 x 2 k per 64 cycles each): the matrix time of a tile drops 5x and the launch is
// bound by its bytes (the fp32 form: 82 us for 170 MB at 27 prompt sets, 0.45 of the fp32-MFMA rate). The image-token operand is split in
// registers on its way to LDS (never in HBM), the weights are split once by the caller (psam_split_f16). 64 x BN tiles, k slabs of 32,
// W as the MFMA's A operand (a lane ends up with 4 consecutive output columns of one row, as gemm_f32_kernel).
#define G3_LD 40      // halfs per LDS row: 80 bytes - the 16-byte fragment reads of 32 consecutive rows touch every bank once
template <int BN>
__global__ __launch_bounds__(256) void gemm_f32x3_kernel(const float* __restrict__ a, const float* __restrict__ a2, int a2_mod,
                                                         const half_t* __restrict__ wh, const half_t* __restrict__ wl,
                                                         const float* __restrict__ bias, const float* resid,
                                                         float* out, int M, int N, int K, int lda, int ldw, int ldo,
                                                         int hm_nk, int hm_hd, float acc_scale) {      // (resid may be out: the in-place residual update)
  constexpr int BM = 64, NI = BN / 64;                 // 32x32 blocks per wave along n (waves 2 x 2: 32 rows x BN / 2 columns each)
  constexpr int WV = BN * 32 / 8 / 256;                // 16-byte pieces of a W slab (hi or lo) per thread
  __shared__ __attribute__((aligned(16))) half_t Ah[BM * G3_LD], Al[BM * G3_LD], Wh[BN * G3_LD], Wl[BN * G3_LD];
  const int t = threadIdx.x, wave = t >> 6, lane = t & 63, lr = lane & 31, lk = lane >> 5;
  const int wm = wave >> 1, wn = wave & 1;
  const int m0 = blockIdx.x * BM, n0 = blockIdx.y * BN;
  f32x16 acc[NI];
#pragma unroll
  for (int j = 0; j < NI; ++j)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
  const int arow = t >> 2, ac = (t & 3) * 8;           // this thread's 8 consecutive k of one A row
  const int am = m0 + arow;
  const float* ap = a + (size_t)(am < M ? am : 0) * lda + ac;
  const float* a2p = a2 ? a2 + (size_t)((am < M ? am : 0) % a2_mod) * lda + ac : nullptr;
  float4 av0, av1;
  half8_t wvh[WV], wvl[WV];
  auto load_slab = [&](int k0) {
    av0 = make_float4(0.f, 0.f, 0.f, 0.f); av1 = av0;
    if (am < M) {
      av0 = *reinterpret_cast<const float4*>(ap + k0);
      av1 = *reinterpret_cast<const float4*>(ap + k0 + 4);
      if (a2p) {
        const float4 p0 = *reinterpret_cast<const float4*>(a2p + k0), p1 = *reinterpret_cast<const float4*>(a2p + k0 + 4);
        av0.x += p0.x; av0.y += p0.y; av0.z += p0.z; av0.w += p0.w;
        av1.x += p1.x; av1.y += p1.y; av1.z += p1.z; av1.w += p1.w;
      }
    }
#pragma unroll
    for (int i = 0; i < WV; ++i) {
      const int idx = t + 256 * i, row = idx >> 2, c8 = (idx & 3) * 8;
      wvh[i] = *reinterpret_cast<const half8_t*>(wh + (size_t)(n0 + row) * ldw + k0 + c8);   // N % BN == 0
      wvl[i] = *reinterpret_cast<const half8_t*>(wl + (size_t)(n0 + row) * ldw + k0 + c8);
    }
  };
  load_slab(0);
  for (int k0 = 0; k0 < K; k0 += 32) {
    __syncthreads();   // every wave is done reading the previous slab
    {
      const float v[8] = {av0.x, av0.y, av0.z, av0.w, av1.x, av1.y, av1.z, av1.w};
      half8_t h, l;
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        h[i] = (half_t)v[i];
        l[i] = (half_t)(v[i] - (float)h[i]);
      }
      *reinterpret_cast<half8_t*>(Ah + arow * G3_LD + ac) = h;
      *reinterpret_cast<half8_t*>(Al + arow * G3_LD + ac) = l;
    }
#pragma unroll
    for (int i = 0; i < WV; ++i) {
      const int idx = t + 256 * i, row = idx >> 2, c8 = (idx & 3) * 8;
      *reinterpret_cast<half8_t*>(Wh + row * G3_LD + c8) = wvh[i];
      *reinterpret_cast<half8_t*>(Wl + row * G3_LD + c8) = wvl[i];
    }
    __syncthreads();
    if (k0 + 32 < K) load_slab(k0 + 32);
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      const int xo = (wm * 32 + lr) * G3_LD + ks * 16 + lk * 8;
      const half8_t xh = *reinterpret_cast<const half8_t*>(Ah + xo), xl = *reinterpret_cast<const half8_t*>(Al + xo);
#pragma unroll
      for (int j = 0; j < NI; ++j) {
        const int wo = (wn * (BN / 2) + j * 32 + lr) * G3_LD + ks * 16 + lk * 8;
        const half8_t fh = *reinterpret_cast<const half8_t*>(Wh + wo), fl = *reinterpret_cast<const half8_t*>(Wl + wo);
        acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fl, xh, acc[j], 0, 0, 0);     // (the two small products first)
        acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fh, xl, acc[j], 0, 0, 0);
        acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fh, xh, acc[j], 0, 0, 0);
      }
    }
  }
  // acc[j][r]: m = m0 + wm*32 + lr ; n = n0 + wn*BN/2 + j*32 + (r&3) + 8*(r>>2) + 4*lk
  const int m = m0 + wm * 32 + lr;
  if (m >= M) return;
#pragma unroll
  for (int j = 0; j < NI; ++j)
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int n = n0 + wn * (BN / 2) + j * 32 + 8 * q + 4 * lk;
      float4 v = make_float4(acc[j][4 * q] * acc_scale, acc[j][4 * q + 1] * acc_scale, acc[j][4 * q + 2] * acc_scale, acc[j][4 * q + 3] * acc_scale);
      if (bias) {
        const float4 b = *reinterpret_cast<const float4*>(bias + n);
        v.x += b.x; v.y += b.y; v.z += b.z; v.w += b.w;
      }
      if (resid) {
        const float4 r = *reinterpret_cast<const float4*>(resid + (size_t)m * ldo + n);
        v.x += r.x; v.y += r.y; v.z += r.z; v.w += r.w;
      }
      if (hm_nk > 0) {   // head-major: out[image = m / nk][head = n / hd][m % nk][n % hd] (psam_gemm_f32_heads)
        const size_t o = ((size_t)(m / hm_nk) * (N / hm_hd) + n / hm_hd) * hm_nk * hm_hd + (size_t)(m % hm_nk) * hm_hd + n % hm_hd;
        *reinterpret_cast<float4*>(out + o) = v;
      } else {
        *reinterpret_cast<float4*>(out + (size_t)m * ldo + n) = v;
      }
    }
}

// acc_scale: the weights may be split after a scaling by a power of two (wh + wl = w * 2^s: a weight of 0.05 has a SUBNORMAL fp16 lo half
// - absolute resolution 2^-25 instead of 2^-22 relative; times 256 it is normal down to |w| = 5e-4) - acc_scale = 2^-s undoes it, exactly.
// out[M,N] = acc_scale * (a [+ a2[m % a2_mod]]) @ (wh + wl)^T + bias [+ resid]: a / bias / resid / out fp32, wh / wl the fp16 halves of the fp32
// weight (psam_split_f16). nk > 0: head-major output [M / nk][N / hd][nk][hd] (no residual), as psam_gemm_f32_heads.
extern "C" int psam_gemm_f32x3(const float* a, const float* a2, int a2_mod, const void* wh, const void* wl, const float* bias,
                               const float* resid, float* out, int M, int N, int K, int lda, int ldw, int ldo, int nk, int hd,
                               float acc_scale, void* stream) {
  if (M <= 0 || N <= 0 || K <= 0 || (K % 32) || (N % 64) || (lda % 4) || (ldw % 8) || (ldo % 4) || (a2 && a2_mod <= 0) || !wh || !wl)
    return PSAM_ERR_ARG;
  if (nk > 0 && (hd <= 0 || (hd % 4) || (N % hd) || (M % nk) || resid)) return PSAM_ERR_ARG;
  if ((reinterpret_cast<uintptr_t>(a) | reinterpret_cast<uintptr_t>(a2) | reinterpret_cast<uintptr_t>(wh) | reinterpret_cast<uintptr_t>(wl) |
       reinterpret_cast<uintptr_t>(bias) | reinterpret_cast<uintptr_t>(resid) | reinterpret_cast<uintptr_t>(out)) & 15)
    return PSAM_ERR_ARG;
  hipStream_t s = (hipStream_t)stream;
  if (N % 128 == 0)
    hipLaunchKernelGGL((gemm_f32x3_kernel<128>), dim3((M + 63) / 64, N / 128), dim3(256), 0, s, a, a2, a2_mod, (const half_t*)wh,
                       (const half_t*)wl, bias, resid, out, M, N, K, lda, ldw, ldo, nk, hd, acc_scale);
  else
    hipLaunchKernelGGL((gemm_f32x3_kernel<64>), dim3((M + 63) / 64, N / 64), dim3(256), 0, s, a, a2, a2_mod, (const half_t*)wh,
                       (const half_t*)wl, bias, resid, out, M, N, K, lda, ldw, ldo, nk, hd, acc_scale);
  return psam_launch_status();
}

static int gemm_f32_launch(const float* a, const float* a2, int a2_mod, const float* w, const float* bias,
                           const float* resid, float* out, int M, int N, int K, int lda, int ldw, int ldo, int hm_nk, int hm_hd, void* stream) {
  if (M <= 0 || N <= 0 || K <= 0 || (K % 32) || (N % 64) || (lda % 4) || (ldw % 4) || (ldo % 4) || (a2 && a2_mod <= 0))
    return PSAM_ERR_ARG;
  hipStream_t s = (hipStream_t)stream;
  // Tile shape by measurement (tools/gemm_f32_shapes.py, us per call at N x K = 128 x 256 for 1 / 2 / 16 / 26 / 40 prompt sets of 4096
  // tokens; a 32x32x2 fp32 MFMA holds the matrix pipe for 64 cycles, so small tiles cost nothing in LDS traffic and fill the CUs'
  // last round better): 128x128 24.9 / 23.4 / 47.9 / 90.5 / 119.5 - 64x128 13.7 / 14.1 / 47.8 / 79.7 / 108.5 - 64x64 10.5 / 10.9 /
  // 50.1 / 85.5 / 116.2.
  struct Shape { int bm, bn; };
  const Shape shapes[3] = {{128, 128}, {64, 128}, {64, 64}};
  int best = (M > 16384 && (N % 128) == 0) ? 1 : 2;
  {   // (A/B: PSAM_GEMM_F32_SHAPE=0..2, parsed once, anything else ignored)
    static const int forced = [] { const char* fe = getenv("PSAM_GEMM_F32_SHAPE"); const int v = fe ? atoi(fe) : -1; return (v >= 0 && v <= 2) ? v : -1; }();
    if (forced >= 0 && N % shapes[forced].bn == 0) best = forced;
  }
#define PSAM_GF32(BM_, BN_)                                                                                                     \
  hipLaunchKernelGGL((gemm_f32_kernel<BM_, BN_>), dim3((M + BM_ - 1) / BM_, N / BN_), dim3(256), 0, s, a, a2, a2_mod, w, bias, \
                     resid, out, M, N, K, lda, ldw, ldo, hm_nk, hm_hd)
  if (best == 0) PSAM_GF32(128, 128);
  else if (best == 1) PSAM_GF32(64, 128);
  else PSAM_GF32(64, 64);
#undef PSAM_GF32
  return psam_launch_status();
}
extern "C" int psam_gemm_f32(const float* a, const float* a2, int a2_mod, const float* w, const float* bias,
                             const float* resid, float* out, int M, int N, int K, int lda, int ldw, int ldo, void* stream) {
  return gemm_f32_launch(a, a2, a2_mod, w, bias, resid, out, M, N, K, lda, ldw, ldo, 0, 0, stream);
}
// The same product written HEAD-MAJOR: out fp32 [M / nk images][N / hd heads][nk][hd] - the K / V projections of the token-to-image
// attention (transformer.py:228-230 on the 4096-token operand), whose kernel then reads a head's keys as contiguous 64-byte rows
// (token-major, a (key, head) slice is 64 bytes of a 512-byte row: every 16-byte load of a wave touches 64 different lines and the
// kernel ran at the L1's line rate, 128-328 us for 113 MB). M % nk == 0, hd % 4 == 0, N % hd == 0; no residual.
extern "C" int psam_gemm_f32_heads(const float* a, const float* a2, int a2_mod, const float* w, const float* bias, float* out, int M,
                                   int N, int K, int lda, int ldw, int nk, int hd, void* stream) {
  if (nk <= 0 || hd <= 0 || (hd % 4) || (N % hd) || (M % nk)) return PSAM_ERR_ARG;
  return gemm_f32_launch(a, a2, a2_mod, w, bias, nullptr, out, M, N, K, lda, ldw, N, nk, hd, stream);
}

// =====================================================================================================
// small_attention: softmax(q k^T / sqrt(hd)) v with <= 16 keys; one thread per (batch, query row, head).
// Used for the token self-attention (Tq = Tk = T, hd 32) and for image->token cross attention
// (Tq = 4096 image tokens, q fp16 from the GEMM, hd 16; transformer.py:176-180).
template <int HD, typename QT, typename OT>
__global__ void small_attention_kernel(const QT* __restrict__ q, const float* __restrict__ k,
                                       const float* __restrict__ v, OT* __restrict__ out, int B, int Tq, int Tk, int NH,
                                       int ldq, int ldk, int ldv, int ldo) {
  const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  const long long total = (long long)B * Tq * NH;
  if (idx >= total) return;
  const int h = (int)(idx % NH);
  const long long row = idx / NH;  // b*Tq + qi
  const int b = (int)(row / Tq);
  const QT* qp = q + (size_t)row * ldq + h * HD;
  float qv[HD];
#pragma unroll
  for (int d = 0; d < HD; ++d) qv[d] = (float)qp[d];
  const float inv = 1.0f / sqrtf((float)HD);
  float s[16];
  float mx = -INFINITY;
#pragma unroll
  for (int j = 0; j < 16; ++j) {
    s[j] = -INFINITY;
    if (j < Tk) {
      const float* kp = k + ((size_t)b * Tk + j) * ldk + h * HD;
      float a = 0.f;
#pragma unroll
      for (int d = 0; d < HD; ++d) a += qv[d] * kp[d];
      s[j] = a * inv;
      mx = fmaxf(mx, s[j]);
    }
  }
  float l = 0.f;
  float o[HD];
#pragma unroll
  for (int d = 0; d < HD; ++d) o[d] = 0.f;
#pragma unroll
  for (int j = 0; j < 16; ++j) {
    if (j < Tk) {
      const float p = expf(s[j] - mx);
      l += p;
      const float* vp = v + ((size_t)b * Tk + j) * ldv + h * HD;
#pragma unroll
      for (int d = 0; d < HD; ++d) o[d] += p * vp[d];
    }
  }
  OT* op = out + (size_t)row * ldo + h * HD;
#pragma unroll
  for (int d = 0; d < HD; ++d) op[d] = (OT)(o[d] / l);
}

// q_f16 = 1: q and out are fp16 (image side); 0: fp32.
extern "C" int psam_small_attention(const void* q, const float* k, const float* v, void* out, int B, int Tq, int Tk,
                                    int NH, int hd, int ldq, int ldk, int ldv, int ldo, int q_f16, void* stream) {
  if (B <= 0 || Tq <= 0 || Tk <= 0 || Tk > 16) return PSAM_ERR_ARG;
  const long long total = (long long)B * Tq * NH;
  dim3 grid((unsigned)((total + 255) / 256)), block(256);
  hipStream_t s = (hipStream_t)stream;
  if (hd == 16 && q_f16)
    hipLaunchKernelGGL((small_attention_kernel<16, half_t, half_t>), grid, block, 0, s, (const half_t*)q, k, v,
                       (half_t*)out, B, Tq, Tk, NH, ldq, ldk, ldv, ldo);
  else if (hd == 16)
    hipLaunchKernelGGL((small_attention_kernel<16, float, float>), grid, block, 0, s, (const float*)q, k, v,
                       (float*)out, B, Tq, Tk, NH, ldq, ldk, ldv, ldo);
  else if (hd == 32 && !q_f16)
    hipLaunchKernelGGL((small_attention_kernel<32, float, float>), grid, block, 0, s, (const float*)q, k, v,
                       (float*)out, B, Tq, Tk, NH, ldq, ldk, ldv, ldo);
  else
    return PSAM_ERR_ARG;
  return psam_launch_status();
}

// =====================================================================================================
// t2i_attention: token -> image cross attention (transformer.py:163-167, 98-103). q fp32 [B,T,NH*16];
// K, V fp16 [B*Nk, NH*16] (outputs of the k_proj / v_proj GEMMs); out fp32 [B,T,NH*16].
// One block per (t, head, b); 256 threads x up to 16 keys each; two passes over register-resident scores.
template <typename KT>
__device__ __forceinline__ void load16(const KT* __restrict__ p, float* v);
template <>
__device__ __forceinline__ void load16<half_t>(const half_t* __restrict__ p, float* v) {
  const half8_t* hp = reinterpret_cast<const half8_t*>(p);
  const half8_t k0 = hp[0], k1 = hp[1];
#pragma unroll
  for (int d = 0; d < 8; ++d) { v[d] = (float)k0[d]; v[8 + d] = (float)k1[d]; }
}
template <>
__device__ __forceinline__ void load16<float>(const float* __restrict__ p, float* v) {
  const float4* fp = reinterpret_cast<const float4*>(p);
#pragma unroll
  for (int d = 0; d < 4; ++d) {
    const float4 x = fp[d];
    v[4 * d] = x.x; v[4 * d + 1] = x.y; v[4 * d + 2] = x.z; v[4 * d + 3] = x.w;
  }
}
template <typename KT>
__global__ __launch_bounds__(256) void t2i_attention_kernel(const float* __restrict__ q, const KT* __restrict__ K,
                                                            const KT* __restrict__ V, float* __restrict__ out, int T,
                                                            int Nk, int NH) {
  constexpr int HD = 16;
  const int t = blockIdx.x, h = blockIdx.y, b = blockIdx.z;
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  __shared__ float red[4][HD + 1];
  __shared__ float bmax;
  const int C = NH * HD;
  const float* qp = q + ((size_t)b * T + t) * C + h * HD;
  float qv[HD];
#pragma unroll
  for (int d = 0; d < HD; ++d) qv[d] = qp[d];
  const float inv = 1.0f / sqrtf((float)HD);
  float s[16];
  float mx = -INFINITY;
#pragma unroll
  for (int j = 0; j < 16; ++j) {
    const int key = tid + 256 * j;
    s[j] = -INFINITY;
    if (key < Nk) {
      float kv[HD];
      load16<KT>(K + ((size_t)b * Nk + key) * C + h * HD, kv);
      float a = 0.f;
#pragma unroll
      for (int d = 0; d < HD; ++d) a += qv[d] * kv[d];
      s[j] = a * inv;
      mx = fmaxf(mx, s[j]);
    }
  }
  mx = wave_max(mx);
  if (lane == 0) red[wv][0] = mx;
  __syncthreads();
  if (tid == 0) bmax = fmaxf(fmaxf(red[0][0], red[1][0]), fmaxf(red[2][0], red[3][0]));
  __syncthreads();
  const float M = bmax;
  float l = 0.f, o[HD];
#pragma unroll
  for (int d = 0; d < HD; ++d) o[d] = 0.f;
#pragma unroll
  for (int j = 0; j < 16; ++j) {
    const int key = tid + 256 * j;
    if (key < Nk) {
      const float p = expf(s[j] - M);
      l += p;
      float vv[HD];
      load16<KT>(V + ((size_t)b * Nk + key) * C + h * HD, vv);
#pragma unroll
      for (int d = 0; d < HD; ++d) o[d] += p * vv[d];
    }
  }
  l = wave_sum(l);
#pragma unroll
  for (int d = 0; d < HD; ++d) o[d] = wave_sum(o[d]);
  __syncthreads();
  if (lane == 0) {
    red[wv][HD] = l;
#pragma unroll
    for (int d = 0; d < HD; ++d) red[wv][d] = o[d];
  }
  __syncthreads();
  if (tid < HD) {
    const float lt = red[0][HD] + red[1][HD] + red[2][HD] + red[3][HD];
    const float ot = red[0][tid] + red[1][tid] + red[2][tid] + red[3][tid];
    out[((size_t)b * T + t) * C + h * HD + tid] = ot / lt;
  }
}

// Round 5: ALL tokens of a (prompt set, head) in one workgroup, one WAVE per token - lane kl walks keys kl, kl + 64, ... with an online
// softmax, the 64 lanes merge at the end (no LDS, no barrier). The kernel above gives every (token, head, prompt set) its own
// workgroup, each of which streams the head's whole K / V slices (512 KB): nine tokens read them nine times (1 GB through the L2
// per launch at 27 prompt sets; 173-281 us for 113 MB of K / V); here the T waves of a workgroup share them in the CU's caches.
// (A first form with sixteen key lanes per token - 256 serial keys per thread - was latency-bound at the old kernel's 221 us.)
template <typename KT>
__global__ __launch_bounds__(1024) void t2i_attention_all_kernel(const float* __restrict__ q, const KT* __restrict__ K,
                                                                const KT* __restrict__ V, float* __restrict__ out, int T,
                                                                int Nk, int NH, int head_major, int S,
                                                                float* __restrict__ part) {
  constexpr int HD = 16;
  const int h = blockIdx.x, b = blockIdx.y, z = blockIdx.z;
  const int tid = threadIdx.x, t = tid >> 6, kl = tid & 63;
  const int C = NH * HD;
  const bool active = t < T;
  const float inv = 1.0f / sqrtf((float)HD);
  float qv[HD];
  {
    const float* qp = q + ((size_t)b * T + (active ? t : 0)) * C + h * HD;
#pragma unroll
    for (int d = 0; d < HD; ++d) qv[d] = qp[d] * inv;
  }
  float m = -INFINITY, l = 0.f, o[HD];
#pragma unroll
  for (int d = 0; d < HD; ++d) o[d] = 0.f;
  // token-major [B][Nk][NH * 16] (psam_gemm_f32 / psam_gemm_f16) or head-major [B][NH][Nk][16] (psam_gemm_f32_heads: a wave's 64 keys are
  // 4 KiB of contiguous rows)
  const int rs = head_major ? HD : C;
  const KT* kp = head_major ? K + ((size_t)b * NH + h) * Nk * HD : K + (size_t)b * Nk * C + h * HD;
  const KT* vp = head_major ? V + ((size_t)b * NH + h) * Nk * HD : V + (size_t)b * Nk * C + h * HD;
  // S > 1 (psam_t2i_attention_split): workgroup z of the S that share (b, h) takes the keys [z * chunk, (z + 1) * chunk) - with one or
  // two prompt sets a launch is 8 ... 16 workgroups whose waves each walk 4096 keys in 64 dependent round trips (67 us, latency);
  // split, every wave has a few and the chip has ~256 workgroups
  const int chunk = S > 1 ? ((Nk + S * 64 - 1) / (S * 64)) * 64 : Nk;
  const int k0 = z * chunk, k1 = min(Nk, k0 + chunk);
#pragma unroll 4
  for (int key = k0 + kl; key < k1; key += 64) {
    float kv[HD], vv[HD];
    load16<KT>(kp + (size_t)key * rs, kv);
    load16<KT>(vp + (size_t)key * rs, vv);
    float a = 0.f;
#pragma unroll
    for (int d = 0; d < HD; ++d) a += qv[d] * kv[d];
    const float mn = fmaxf(m, a);
    const float sc = expf(m - mn), p = expf(a - mn);
    l = l * sc + p;
#pragma unroll
    for (int d = 0; d < HD; ++d) o[d] = o[d] * sc + p * vv[d];
    m = mn;
  }
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) {     // the 64 key lanes of a token are one wave
    const float m2 = __shfl_xor(m, off), l2 = __shfl_xor(l, off);
    const float mn = fmaxf(m, m2);
    const float a1 = m == -INFINITY ? 0.f : expf(m - mn), a2 = m2 == -INFINITY ? 0.f : expf(m2 - mn);   // (a lane without keys: split ranges)
    l = l * a1 + l2 * a2;
#pragma unroll
    for (int d = 0; d < HD; ++d) o[d] = o[d] * a1 + __shfl_xor(o[d], off) * a2;
    m = mn;
  }
  if (S <= 1) {
    if (active && kl == 0) {
      float* op = out + ((size_t)b * T + t) * C + h * HD;
#pragma unroll
      for (int d = 0; d < HD; ++d) op[d] = o[d] / l;
    }
    return;
  }
  // partial (m, l, o[16]) of this key range -> part[(b, h)][z][t]; t2i_combine_kernel merges the S partials in the order of z.
  // (A single-launch form - the last workgroup of (b, h) to arrive combines, found through a device-scope counter - was measured first:
  // its release / acquire fences write back and invalidate the XCD's L2 in every workgroup: 27 prompt sets 2350 -> 2800 us per decoder
  // call, two sets 737 -> 710 only.)
  if (active && kl == 0) {
    float* pp = part + ((((size_t)b * NH + h) * S + z) * T + t) * 18;
    pp[0] = m; pp[1] = l;
#pragma unroll
    for (int d = 0; d < HD; ++d) pp[2 + d] = o[d];
  }
}

// one wave per (b, h, t): lane z < S holds the partial of key range z; fixed-order butterfly; out[b][t][h * 16 ..]
__global__ __launch_bounds__(256) void t2i_combine_kernel(const float* __restrict__ part, float* __restrict__ out, int B, int T, int NH,
                                                          int S) {
  constexpr int HD = 16;
  const int w = blockIdx.x * 4 + (threadIdx.x >> 6), kl = threadIdx.x & 63;
  if (w >= B * NH * T) return;
  const int t = w % T, bh = w / T, h = bh % NH, b = bh / NH;
  const bool have = kl < S;
  const float* pp = part + (((size_t)bh * S + (have ? kl : 0)) * T + t) * 18;
  float m = have ? pp[0] : -INFINITY, l = have ? pp[1] : 0.f, o[HD];
#pragma unroll
  for (int d = 0; d < HD; ++d) o[d] = have ? pp[2 + d] : 0.f;
#pragma unroll
  for (int off = 8; off >= 1; off >>= 1) {       // S <= 16
    const float m2 = __shfl_xor(m, off), l2 = __shfl_xor(l, off);
    const float mn = fmaxf(m, m2);
    const float a1 = m == -INFINITY ? 0.f : expf(m - mn), a2 = m2 == -INFINITY ? 0.f : expf(m2 - mn);
    l = l * a1 + l2 * a2;
#pragma unroll
    for (int d = 0; d < HD; ++d) o[d] = o[d] * a1 + __shfl_xor(o[d], off) * a2;
    m = mn;
  }
  if (kl == 0) {
    float* op = out + ((size_t)b * T + t) * (NH * HD) + h * HD;
#pragma unroll
    for (int d = 0; d < HD; ++d) op[d] = o[d] / l;
  }
}

// kv_f32 = 1: K / V are fp32 (outputs of psam_gemm_f32, the default decoder path); 0: fp16 (outputs of psam_gemm_f16).
static int t2i_launch(const float* q, const void* K, const void* V, float* out, int B, int T, int Nk, int NH, int kv_f32, int S,
                      float* part, void* stream);
extern "C" int psam_t2i_attention(const float* q, const void* K, const void* V, float* out, int B, int T, int Nk, int NH,
                                  int kv_f32, void* stream) {
  return t2i_launch(q, K, V, out, B, T, Nk, NH, kv_f32, 1, nullptr, stream);
}
// The same with the keys of every (prompt set, head) split over S workgroups (1 <= S <= 16) and a second small launch that merges the
// partials in the order of the split index: `part` fp32 scratch of at least B * NH * S * T * 18 elements.
extern "C" int psam_t2i_attention_split(const float* q, const void* K, const void* V, float* out, int B, int T, int Nk, int NH,
                                        int kv_f32, int S, float* part, void* stream) {
  if (S < 1 || S > 16 || T > 16 || Nk < 64 || (S > 1 && !part)) return PSAM_ERR_ARG;
  return t2i_launch(q, K, V, out, B, T, Nk, NH, kv_f32, S, part, stream);
}
static int t2i_launch(const float* q, const void* K, const void* V, float* out, int B, int T, int Nk, int NH, int kv_f32, int S,
                      float* part, void* stream) {
  if (B <= 0 || T <= 0 || Nk <= 0 || Nk > 4096) return PSAM_ERR_ARG;
  const int head_major = (kv_f32 >> 1) & 1;      // bit 1: K / V head-major [B][NH][Nk][16] (psam_gemm_f32_heads)
  kv_f32 &= 1;
  if (head_major && !(T <= 16 && Nk >= 64)) return PSAM_ERR_ARG;
  static const int all_tokens = [] { const char* e = getenv("PSAM_T2I_ALL"); return e ? atoi(e) : 1; }();      // (0: the round-1 kernel, A/B)
  if ((all_tokens || head_major || S > 1) && T <= 16 && Nk >= 64) {
    if (kv_f32)
      hipLaunchKernelGGL(t2i_attention_all_kernel<float>, dim3(NH, B, S), dim3(64 * T), 0, (hipStream_t)stream, q, (const float*)K,
                         (const float*)V, out, T, Nk, NH, head_major, S, part);
    else
      hipLaunchKernelGGL(t2i_attention_all_kernel<half_t>, dim3(NH, B, S), dim3(64 * T), 0, (hipStream_t)stream, q, (const half_t*)K,
                         (const half_t*)V, out, T, Nk, NH, head_major, S, part);
    if (S > 1)
      hipLaunchKernelGGL(t2i_combine_kernel, dim3((B * NH * T + 3) / 4), dim3(256), 0, (hipStream_t)stream, part, out, B, T, NH, S);
    return psam_launch_status();
  }
  if (kv_f32)
    hipLaunchKernelGGL(t2i_attention_kernel<float>, dim3(T, NH, B), dim3(256), 0, (hipStream_t)stream, q, (const float*)K,
                       (const float*)V, out, T, Nk, NH);
  else
    hipLaunchKernelGGL(t2i_attention_kernel<half_t>, dim3(T, NH, B), dim3(256), 0, (hipStream_t)stream, q, (const half_t*)K,
                       (const half_t*)V, out, T, Nk, NH);
  return psam_launch_status();
}

// =====================================================================================================
// ln_pe: image-token row op (C = 256, one wave per row):  y = [LayerNorm](x[row % in_mod] + add_vec)
//   y32 (fp32, may alias x), y16 = fp16(y), ype16 = fp16(y + pe[row % pe_mod])
// Serves `src = image_embeddings + dense` (mask_decoder.py:126-127), `keys + key_pe` (transformer.py:164,178,99)
// and norm4 (transformer.py:180).
__global__ __launch_bounds__(256) void ln_pe_kernel(const float* __restrict__ x, const float* __restrict__ add_vec,
                                                    const float* __restrict__ w, const float* __restrict__ b,
                                                    const float* __restrict__ pe, float* __restrict__ y32,
                                                    half_t* __restrict__ y16, half_t* __restrict__ ype16, int M,
                                                    int in_mod, int pe_mod, float eps, int do_ln,
                                                    const int* __restrict__ img_of_prompt) {
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  if (row >= M) return;
  // in_mod > 0: the input holds one [in_mod,256] embedding per image; prompt (row / in_mod) reads image
  // img_of_prompt[prompt] (or image 0): `torch.repeat_interleave(image_embeddings, B)` of mask_decoder.py:126
  int irow = row;
  if (in_mod) irow = (img_of_prompt ? img_of_prompt[row / in_mod] : 0) * in_mod + row % in_mod;
  float4 v = reinterpret_cast<const float4*>(x + (size_t)irow * 256)[lane];
  if (add_vec) {
    float4 a = reinterpret_cast<const float4*>(add_vec)[lane];
    v.x += a.x; v.y += a.y; v.z += a.z; v.w += a.w;
  }
  if (do_ln) {
    const float mean = wave_sum((v.x + v.y) + (v.z + v.w)) / 256.f;
    const float a0 = v.x - mean, a1 = v.y - mean, a2 = v.z - mean, a3 = v.w - mean;
    const float var = wave_sum((a0 * a0 + a1 * a1) + (a2 * a2 + a3 * a3)) / 256.f;
    const float rstd = 1.0f / sqrtf(var + eps);
    const float4 ww = reinterpret_cast<const float4*>(w)[lane], bb = reinterpret_cast<const float4*>(b)[lane];
    v.x = a0 * rstd * ww.x + bb.x;
    v.y = a1 * rstd * ww.y + bb.y;
    v.z = a2 * rstd * ww.z + bb.z;
    v.w = a3 * rstd * ww.w + bb.w;
  }
  if (y32) reinterpret_cast<float4*>(y32 + (size_t)row * 256)[lane] = v;
  if (y16) {
    half4_t hh = {(half_t)v.x, (half_t)v.y, (half_t)v.z, (half_t)v.w};
    reinterpret_cast<half4_t*>(y16 + (size_t)row * 256)[lane] = hh;
  }
  if (ype16) {
    const float4 p = reinterpret_cast<const float4*>(pe + (size_t)(row % pe_mod) * 256)[lane];
    half4_t hh = {(half_t)(v.x + p.x), (half_t)(v.y + p.y), (half_t)(v.z + p.z), (half_t)(v.w + p.w)};
    reinterpret_cast<half4_t*>(ype16 + (size_t)row * 256)[lane] = hh;
  }
}

extern "C" int psam_ln_pe(const float* x, const float* add_vec, const float* w, const float* b, const float* pe,
                          float* y32, void* y16, void* ype16, int M, int in_mod, int pe_mod, float eps, int do_ln,
                          const int* img_of_prompt, void* stream) {
  if (M <= 0 || pe_mod <= 0) return PSAM_ERR_ARG;
  hipLaunchKernelGGL(ln_pe_kernel, dim3((M + 3) / 4), dim3(256), 0, (hipStream_t)stream, x, add_vec, w, b, pe, y32,
                     (half_t*)y16, (half_t*)ype16, M, in_mod, pe_mod, eps, do_ln, img_of_prompt);
  return psam_launch_status();
}

// =====================================================================================================
// Random-Fourier positional encoding (prompt_encoder.py:186-193): c in [0,1]^2 -> [sin | cos](2*pi*((2c-1) @ G))
__device__ __forceinline__ void pe_pair(float cx, float cy, const float* __restrict__ G, int k, float* sn, float* cs) {
  const float a = (2.f * cx - 1.f) * G[k] + (2.f * cy - 1.f) * G[128 + k];
  const float arg = 6.283185307179586f * a;
  *sn = sinf(arg);
  *cs = cosf(arg);
}

// dense PE of the 64x64 grid, token-major [gh*gw, 256] (get_dense_pe, prompt_encoder.py:62-71,195-206)
__global__ void dense_pe_kernel(const float* __restrict__ G, int gh, int gw, float* __restrict__ pe) {
  const int pix = blockIdx.x, k = threadIdx.x;  // 128 threads
  const float cy = ((float)(pix / gw) + 0.5f) / (float)gh, cx = ((float)(pix % gw) + 0.5f) / (float)gw;
  float sn, cs;
  pe_pair(cx, cy, G, k, &sn, &cs);
  pe[(size_t)pix * 256 + k] = sn;
  pe[(size_t)pix * 256 + 128 + k] = cs;
}
extern "C" int psam_dense_pe(const float* G, int gh, int gw, float* pe, void* stream) {
  hipLaunchKernelGGL(dense_pe_kernel, dim3(gh * gw), dim3(128), 0, (hipStream_t)stream, G, gh, gw, pe);
  return psam_launch_status();
}

// tokens[b, 0:5] = out_tok (iou_token ++ mask_tokens); tokens[b, 5+j] = PE((coords[b,j] + 0.5)/img) + type_emb
// labels: -1 not-a-point (PE zeroed), 0 negative, 1 positive, 2 / 3 box corners; type table row = label + 1
// (prompt_encoder.py:73-101). coords are in the 1024-frame (predictor.apply_coords already applied).
__global__ void prompt_tokens_kernel(const float* __restrict__ coords, const int* __restrict__ labels,
                                     const float* __restrict__ G, const float* __restrict__ type_emb,
                                     const float* __restrict__ out_tok, int Ns, float img_size,
                                     float* __restrict__ tokens) {
  const int j = blockIdx.x, b = blockIdx.y, k = threadIdx.x;  // 128 threads
  const int T = 5 + Ns;
  float* o = tokens + ((size_t)b * T + j) * 256;
  if (j < 5) {
    o[k] = out_tok[j * 256 + k];
    o[128 + k] = out_tok[j * 256 + 128 + k];
    return;
  }
  const int sj = j - 5;
  const int lab = labels[b * Ns + sj];
  float sn = 0.f, cs = 0.f;
  if (lab >= 0) {
    const float cx = (coords[((size_t)b * Ns + sj) * 2 + 0] + 0.5f) / img_size;
    const float cy = (coords[((size_t)b * Ns + sj) * 2 + 1] + 0.5f) / img_size;
    pe_pair(cx, cy, G, k, &sn, &cs);
  }
  const float* te = type_emb + (size_t)(lab + 1) * 256;
  o[k] = sn + te[k];
  o[128 + k] = cs + te[128 + k];
}
extern "C" int psam_prompt_tokens(const float* coords, const int* labels, const float* G, const float* type_emb,
                                  const float* out_tok, int B, int Ns, float img_size, float* tokens, void* stream) {
  if (B <= 0 || Ns < 0) return PSAM_ERR_ARG;
  hipLaunchKernelGGL(prompt_tokens_kernel, dim3(5 + Ns, B), dim3(128), 0, (hipStream_t)stream, coords, labels, G,
                     type_emb, out_tok, Ns, img_size, tokens);
  return psam_launch_status();
}

// =====================================================================================================
// upscale_tail (mask_decoder.py:53-59,137-144): u1 = ConvT(256->64,2,2)(keys) as a GEMM [B*4096, 4*64] (+bias);
// per mid pixel (token, dy, dx): LayerNorm2d(64) -> GELU -> ConvT(64->32,2,2) -> GELU -> dot with the 4
// hyper-network vectors -> masks[b, 0:4, 4*ty+2*dy+dy2, 4*tx+2*dx+dx2]. `upscaled_embedding` is never written.
template <bool MFMA>
__global__ __launch_bounds__(256) void upscale_tail_kernel(const float* __restrict__ u1, const float* __restrict__ lnw,
                                                           const float* __restrict__ lnb, const float* __restrict__ W2r,
                                                           const float* __restrict__ b2, const float* __restrict__ hyper,
                                                           float* __restrict__ masks, int g) {
  extern __shared__ __attribute__((aligned(16))) float sm[];
  float* w2s = sm;                 // [64][128]
  float* hs = w2s + 64 * 128;      // [4][32]
  float* b2s = hs + 128;           // [32]
  float* mids = b2s + 32;          // [256][65]
  const int b = blockIdx.y;
  for (int i = threadIdx.x; i < 64 * 128; i += 256) w2s[i] = W2r[i];
  if (threadIdx.x < 128) hs[threadIdx.x] = hyper[(size_t)b * 128 + threadIdx.x];
  if (threadIdx.x < 32) b2s[threadIdx.x] = b2[threadIdx.x];
  const int tok = blockIdx.x * 64 + (threadIdx.x >> 2);
  const int dd = threadIdx.x & 3, dy = dd >> 1, dx = dd & 1;
  const int ty = tok / g, tx = tok % g;
  const float4* up = reinterpret_cast<const float4*>(u1 + ((size_t)b * g * g + tok) * 256 + dd * 64);
  float* mymid = mids + threadIdx.x * 65;
  {
    float mid[64];
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      float4 v = up[i];
      mid[4 * i] = v.x; mid[4 * i + 1] = v.y; mid[4 * i + 2] = v.z; mid[4 * i + 3] = v.w;
      s += (v.x + v.y) + (v.z + v.w);
    }
    const float mean = s / 64.f;
    float q = 0.f;
#pragma unroll
    for (int c = 0; c < 64; ++c) {
      mid[c] -= mean;
      q += mid[c] * mid[c];
    }
    const float rstd = 1.0f / sqrtf(q / 64.f + 1e-6f);
#pragma unroll
    for (int c = 0; c < 64; ++c) mymid[c] = gelu_erf(mid[c] * rstd * lnw[c] + lnb[c]);
  }
  __syncthreads();
  const int W = 4 * g;
  if (MFMA) {
    // Round 5: the second transposed convolution as [256 mid pixels] x [4 x 32 outputs] x [64 channels] on the exact-fp32 MFMA
    // (v_mfma_f32_32x32x2_f32: W2 as the A operand, the mid pixels as B, both straight from the LDS images above) instead of 8192
    // scalar FMAs per thread fed by one 16-byte LDS read per four of them (365-592 us per 27 prompt sets: LDS-issue bound). A wave
    // owns 64 mid pixels = two 32-pixel blocks x four sub-positions d2; after the GELU a lane holds 16 of a pixel's 32 channels,
    // the partner lane (+32) the others: four dot products with the hyper-network vectors, one cross-lane add.
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, lr = lane & 31, lk = lane >> 5;
#pragma unroll 1
    for (int pb = 0; pb < 2; ++pb) {
      const int px = wave * 64 + pb * 32 + lr;                 // this lane's mid pixel (B operand row, output column)
      const float* mrow = mids + px * 65 + lk;
#pragma unroll 1
      for (int d2 = 0; d2 < 4; ++d2) {
        f32x16 acc;
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] = b2s[(r & 3) + 8 * (r >> 2) + 4 * lk];
        const float* wcol = w2s + lk * 128 + d2 * 32 + lr;
#pragma unroll 8
        for (int s2 = 0; s2 < 32; ++s2) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(wcol[s2 * 256], mrow[2 * s2], acc, 0, 0, 0);
        float o0 = 0.f, o1 = 0.f, o2 = 0.f, o3 = 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int c2 = (r & 3) + 8 * (r >> 2) + 4 * lk;
          const float a = gelu_erf(acc[r]);
          o0 += hs[c2] * a;
          o1 += hs[32 + c2] * a;
          o2 += hs[64 + c2] * a;
          o3 += hs[96 + c2] * a;
        }
        o0 += __shfl_xor(o0, 32); o1 += __shfl_xor(o1, 32); o2 += __shfl_xor(o2, 32); o3 += __shfl_xor(o3, 32);
        if (lk == 0) {
          const int tk = blockIdx.x * 64 + (px >> 2), pdd = px & 3;
          const int y = 4 * (tk / g) + 2 * (pdd >> 1) + (d2 >> 1), x = 4 * (tk % g) + 2 * (pdd & 1) + (d2 & 1);
          float* mp = masks + (size_t)b * 4 * W * W + (size_t)y * W + x;
          mp[0] = o0;
          mp[(size_t)W * W] = o1;
          mp[(size_t)2 * W * W] = o2;
          mp[(size_t)3 * W * W] = o3;
        }
      }
    }
    return;
  }
#pragma unroll 1
  for (int d2 = 0; d2 < 4; ++d2) {
    float acc[32];
#pragma unroll
    for (int c2 = 0; c2 < 32; ++c2) acc[c2] = b2s[c2];
#pragma unroll 2
    for (int c = 0; c < 64; ++c) {
      const float m = mymid[c];
      const float4* wr = reinterpret_cast<const float4*>(&w2s[c * 128 + d2 * 32]);
#pragma unroll
      for (int c4 = 0; c4 < 8; ++c4) {
        const float4 wv = wr[c4];
        acc[4 * c4] += m * wv.x;
        acc[4 * c4 + 1] += m * wv.y;
        acc[4 * c4 + 2] += m * wv.z;
        acc[4 * c4 + 3] += m * wv.w;
      }
    }
    float o0 = 0.f, o1 = 0.f, o2 = 0.f, o3 = 0.f;
#pragma unroll
    for (int c2 = 0; c2 < 32; ++c2) {
      const float a = gelu_erf(acc[c2]);
      o0 += hs[c2] * a;
      o1 += hs[32 + c2] * a;
      o2 += hs[64 + c2] * a;
      o3 += hs[96 + c2] * a;
    }
    const int y = 4 * ty + 2 * dy + (d2 >> 1), x = 4 * tx + 2 * dx + (d2 & 1);
    float* mp = masks + (size_t)b * 4 * W * W + (size_t)y * W + x;
    mp[0] = o0;
    mp[(size_t)W * W] = o1;
    mp[(size_t)2 * W * W] = o2;
    mp[(size_t)3 * W * W] = o3;
  }
}

#define UPS_LDS ((64 * 128 + 128 + 32 + 256 * 65) * 4)
extern "C" int psam_upscale_tail(const float* u1, const float* lnw, const float* lnb, const float* W2r, const float* b2,
                                 const float* hyper, float* masks, int B, int g, void* stream) {
  if (B <= 0 || (g * g) % 64) return PSAM_ERR_ARG;
  static bool attr_set = false;
  if (!attr_set) {
    (void)hipFuncSetAttribute((const void*)upscale_tail_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, UPS_LDS);
    (void)hipFuncSetAttribute((const void*)upscale_tail_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, UPS_LDS);
    attr_set = true;
  }
  static const int mfma = [] { const char* e = getenv("PSAM_UPSCALE_MFMA"); return e ? atoi(e) : 1; }();      // (0: the scalar round-1 form, A/B)
  if (mfma)
    hipLaunchKernelGGL(upscale_tail_kernel<true>, dim3(g * g / 64, B), dim3(256), UPS_LDS, (hipStream_t)stream, u1, lnw, lnb, W2r, b2,
                       hyper, masks, g);
  else
    hipLaunchKernelGGL(upscale_tail_kernel<false>, dim3(g * g / 64, B), dim3(256), UPS_LDS, (hipStream_t)stream, u1, lnw, lnb, W2r, b2,
                       hyper, masks, g);
  return psam_launch_status();
}

// =====================================================================================================
// Mask post-processing. variant 0: bilinear align_corners=False (pip segment_anything 1.0 `Sam`),
// 1: bilinear align_corners=True (vendored `SamBatched`, sam.py:313-320), 2: nearest (vendored `Sam`, sam.py:154-160),
// 3: sigmoid, then bilinear align_corners=False (MedSAM inference, models/ProtoMedSAM.py:49-60; threshold 0.5).
struct Lin2 {
  int i0, i1;
  float l0, l1;
};
__device__ __forceinline__ Lin2 lin2(int dst, int in_size, int out_size, int align) {
  float s;
  if (align) {
    const float sc = out_size > 1 ? (float)(in_size - 1) / (float)(out_size - 1) : 0.f;
    s = sc * (float)dst;
  } else {
    const float sc = (float)in_size / (float)out_size;
    s = sc * ((float)dst + 0.5f) - 0.5f;
    s = s < 0.f ? 0.f : s;
  }
  int i0 = (int)s;
  if (i0 > in_size - 1) i0 = in_size - 1;
  Lin2 r;
  r.i0 = i0;
  r.i1 = i0 + (i0 < in_size - 1 ? 1 : 0);
  float l1 = s - (float)i0;
  l1 = l1 < 0.f ? 0.f : (l1 > 1.f ? 1.f : l1);
  r.l1 = l1;
  r.l0 = 1.f - l1;
  return r;
}
__device__ __forceinline__ float up_sample(const float* __restrict__ p, int IN, int MID, int y, int x, int variant) {
  // value of interpolate(p[IN,IN] -> [MID,MID]) at (y, x)
  if (variant == 2) {
    const float sc = (float)IN / (float)MID;
    int sy = (int)floorf((float)y * sc), sx = (int)floorf((float)x * sc);
    sy = sy < IN - 1 ? sy : IN - 1;
    sx = sx < IN - 1 ? sx : IN - 1;
    return p[(size_t)sy * IN + sx];
  }
  Lin2 ly = lin2(y, IN, MID, variant == 1), lx = lin2(x, IN, MID, variant == 1);
  const float* r0 = p + (size_t)ly.i0 * IN;
  const float* r1 = p + (size_t)ly.i1 * IN;
  float a = r0[lx.i0], b = r0[lx.i1], c = r1[lx.i0], d = r1[lx.i1];
  if (variant == 3) {  // torch.sigmoid(low_res_logits) BEFORE the bilinear resize (models/ProtoMedSAM.py:49-56)
    a = 1.f / (1.f + expf(-a));
    b = 1.f / (1.f + expf(-b));
    c = 1.f / (1.f + expf(-c));
    d = 1.f / (1.f + expf(-d));
  }
  return ly.l0 * (lx.l0 * a + lx.l1 * b) + ly.l1 * (lx.l0 * c + lx.l1 * d);
}

// masks [B, C, IN, IN] logits -> logits at [B, C, MID, MID] (the predictor's `masks` before thresholding when the image
// handed to SAM is MID x MID, so the second interpolate of postprocess_masks is the identity)
__global__ void mask_upsample_kernel(const float* __restrict__ low, int IN, int MID, int variant,
                                     float* __restrict__ out) {
  const int x = blockIdx.x * blockDim.x + threadIdx.x, y = blockIdx.y, pl = blockIdx.z;
  if (x >= MID) return;
  out[((size_t)pl * MID + y) * MID + x] = up_sample(low + (size_t)pl * IN * IN, IN, MID, y, x, variant);
}
extern "C" int psam_mask_upsample(const float* low, int planes, int IN, int MID, int variant, float* out, void* stream) {
  if (planes <= 0 || variant < 0 || variant > 3) return PSAM_ERR_ARG;
  hipLaunchKernelGGL(mask_upsample_kernel, dim3((MID + 255) / 256, MID, planes), dim3(256), 0, (hipStream_t)stream, low,
                     IN, MID, variant, out);
  return psam_launch_status();
}

// pred[y, x] (fp32 {0,1}, [OUT, OUT]) = OR_b ( upsample(low[b, sel])[ny(y), nx(x)] > thr ) with the final
// F.interpolate(mode='nearest') MID -> OUT folded in (models/ProtoSAM.py:669-676).
__global__ void mask_union_kernel(const float* __restrict__ low, int B, int C, int sel, int IN, int MID, int OUT,
                                  int variant, float thr, float* __restrict__ pred) {
  const int x = blockIdx.x * blockDim.x + threadIdx.x, y = blockIdx.y;
  if (x >= OUT) return;
  const float sc = (float)MID / (float)OUT;
  int sy = (int)floorf((float)y * sc), sx = (int)floorf((float)x * sc);
  sy = sy < MID - 1 ? sy : MID - 1;
  sx = sx < MID - 1 ? sx : MID - 1;
  int any = 0;
  for (int b = 0; b < B; ++b) {
    const float v = up_sample(low + ((size_t)b * C + sel) * IN * IN, IN, MID, sy, sx, variant);
    any |= (v > thr);
  }
  pred[(size_t)y * OUT + x] = any ? 1.f : 0.f;
}
extern "C" int psam_mask_union(const float* low, int B, int C, int sel, int IN, int MID, int OUT, int variant, float thr,
                               float* pred, void* stream) {
  if (B <= 0 || sel < 0 || sel >= C || variant < 0 || variant > 3) return PSAM_ERR_ARG;
  hipLaunchKernelGGL(mask_union_kernel, dim3((OUT + 255) / 256, OUT), dim3(256), 0, (hipStream_t)stream, low, B, C, sel,
                     IN, MID, OUT, variant, thr, pred);
  return psam_launch_status();
}

// ---- automatic mask generation: statistics and binarisation straight from the low-res logits -------------------------
// The reference materialises every candidate at full resolution ([64*3, H, W] fp32 per batch) and then reduces it to
// three counts and a box (automatic_mask_generator.py:293-310; utils/amg.py calculate_stability_score :156-176,
// batched_mask_to_box :303-346). Here the up-sampled value is recomputed on the fly from the 256x256 logits (256 KB per
// plane, L2 resident) and only the statistics leave the chip.
// plane p of the selection -> prompt p / nsel, channel first + p % nsel of low [B, C, IN, IN].
// stats int32 [planes, 8] = {count(v > thr + off), count(v > thr - off), count(v > thr), min_x, min_y, max_x, max_y, 0}
// over y < H, x < W of the MID x MID up-sampling (H, W = predictor.input_size: the un-padded part of the model input).
__global__ void mask_stats_init_kernel(int* __restrict__ stats, int planes) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= planes * 8) return;
  const int f = i & 7;
  stats[i] = (f == 3 || f == 4) ? 0x7fffffff : ((f == 5 || f == 6) ? -1 : 0);
}
#define MS_ROWS 32
__global__ __launch_bounds__(256) void mask_stats_kernel(const float* __restrict__ low, int C, int first, int nsel,
                                                         int IN, int MID, int H, int W, int variant, float thr,
                                                         float off, int* __restrict__ stats) {
  const int p = blockIdx.y;
  const float* src = low + ((size_t)(p / nsel) * C + first + p % nsel) * IN * IN;
  const int y0 = blockIdx.x * MS_ROWS, y1 = min(y0 + MS_ROWS, H);
  int hi = 0, lo = 0, ar = 0, mnx = 0x7fffffff, mny = 0x7fffffff, mxx = -1, mxy = -1;
  for (int x = threadIdx.x; x < W; x += blockDim.x) {
    for (int y = y0; y < y1; ++y) {
      const float v = up_sample(src, IN, MID, y, x, variant);
      hi += v > thr + off;
      lo += v > thr - off;
      if (v > thr) {
        ++ar;
        mnx = min(mnx, x); mxx = max(mxx, x);
        mny = min(mny, y); mxy = max(mxy, y);
      }
    }
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    hi += __shfl_xor(hi, o, 64); lo += __shfl_xor(lo, o, 64); ar += __shfl_xor(ar, o, 64);
    mnx = min(mnx, __shfl_xor(mnx, o, 64)); mny = min(mny, __shfl_xor(mny, o, 64));
    mxx = max(mxx, __shfl_xor(mxx, o, 64)); mxy = max(mxy, __shfl_xor(mxy, o, 64));
  }
  if ((threadIdx.x & 63) == 0 && lo) {  // v > thr + off or v > thr implies v > thr - off (off >= 0)
    int* s = stats + (size_t)p * 8;
    if (hi) atomicAdd(s + 0, hi);
    atomicAdd(s + 1, lo);
    if (ar) {
      atomicAdd(s + 2, ar);
      atomicMin(s + 3, mnx); atomicMin(s + 4, mny); atomicMax(s + 5, mxx); atomicMax(s + 6, mxy);
    }
  }
}
extern "C" int psam_mask_stats(const float* low, int B, int C, int first, int nsel, int IN, int MID, int H, int W,
                               int variant, float thr, float off, int* stats, void* stream) {
  if (B <= 0 || first < 0 || nsel <= 0 || first + nsel > C || variant < 0 || variant > 2 || H <= 0 || W <= 0 ||
      H > MID || W > MID || off < 0.f)
    return PSAM_ERR_ARG;
  const int planes = B * nsel;
  hipLaunchKernelGGL(mask_stats_init_kernel, dim3((planes * 8 + 255) / 256), dim3(256), 0, (hipStream_t)stream, stats,
                     planes);
  hipLaunchKernelGGL(mask_stats_kernel, dim3((H + MS_ROWS - 1) / MS_ROWS, planes), dim3(256), 0, (hipStream_t)stream,
                     low, C, first, nsel, IN, MID, H, W, variant, thr, off, stats);
  return psam_launch_status();
}

// out[i] (uint8 {0,1} [n, H, W]) = upsample(low plane idx[i]) > thr for the candidates that survived the filters; with a
// label (uint8 {0,1} [H, W]) also counts[i] = {tp, fp, fn} against it (models/SamWrapper.py:8-13 get_iou).
__global__ __launch_bounds__(256) void mask_binarize_kernel(const float* __restrict__ low, const int* __restrict__ idx,
                                                            int IN, int MID, int H, int W, int variant, float thr,
                                                            uint8_t* __restrict__ out,
                                                            const uint8_t* __restrict__ label,
                                                            unsigned long long* __restrict__ counts) {
  const int i = blockIdx.y;
  const float* src = low + (size_t)idx[i] * IN * IN;
  const int y0 = blockIdx.x * MS_ROWS, y1 = min(y0 + MS_ROWS, H);
  int tp = 0, fp = 0, fn = 0;
  for (int y = y0; y < y1; ++y) {
    for (int x = threadIdx.x; x < W; x += blockDim.x) {
      const int m = up_sample(src, IN, MID, y, x, variant) > thr;
      out[((size_t)i * H + y) * W + x] = (uint8_t)m;
      if (label) {
        const int l = label[(size_t)y * W + x] != 0;
        tp += m & l; fp += m & (!l); fn += (!m) & l;
      }
    }
  }
  if (!label) return;
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    tp += __shfl_xor(tp, o, 64); fp += __shfl_xor(fp, o, 64); fn += __shfl_xor(fn, o, 64);
  }
  if ((threadIdx.x & 63) == 0) {
    if (tp) atomicAdd(counts + (size_t)i * 3 + 0, (unsigned long long)tp);
    if (fp) atomicAdd(counts + (size_t)i * 3 + 1, (unsigned long long)fp);
    if (fn) atomicAdd(counts + (size_t)i * 3 + 2, (unsigned long long)fn);
  }
}
extern "C" int psam_mask_binarize(const float* low, const int* idx, int n, int IN, int MID, int H, int W, int variant,
                                  float thr, uint8_t* out, const uint8_t* label, long long* counts, void* stream) {
  if (n <= 0 || variant < 0 || variant > 2 || H <= 0 || W <= 0 || H > MID || W > MID || (label && !counts))
    return PSAM_ERR_ARG;
  if (label) (void)hipMemsetAsync(counts, 0, sizeof(long long) * 3 * n, (hipStream_t)stream);
  hipLaunchKernelGGL(mask_binarize_kernel, dim3((H + MS_ROWS - 1) / MS_ROWS, n), dim3(256), 0, (hipStream_t)stream, low,
                     idx, IN, MID, H, W, variant, thr, out, label, (unsigned long long*)counts);
  return psam_launch_status();
}

// Statistics (and the binary mask) of MATERIALISED candidate planes: the crop layers of the automatic mask generator
// (automatic_mask_generator.py:221-316 with crop_n_layers > 0) and images that are not at the model's input size take the second
// resize of postprocess_masks, so their candidates exist at full resolution before they are reduced. planes fp32 [n, H, W]
// -> stats int32 [n, 8] as psam_mask_stats; out (optional) uint8 [n, H, W] = plane > thr.
__global__ __launch_bounds__(256) void plane_stats_kernel(const float* __restrict__ planes, int H, int W, float thr, float off,
                                                          int* __restrict__ stats, uint8_t* __restrict__ out) {
  const int p = blockIdx.y;
  const float* src = planes + (size_t)p * H * W;
  const int y0 = blockIdx.x * MS_ROWS, y1 = min(y0 + MS_ROWS, H);
  int hi = 0, lo = 0, ar = 0, mnx = 0x7fffffff, mny = 0x7fffffff, mxx = -1, mxy = -1;
  for (int y = y0; y < y1; ++y) {
    for (int x = threadIdx.x; x < W; x += blockDim.x) {
      const float v = src[(size_t)y * W + x];
      hi += v > thr + off;
      lo += v > thr - off;
      const int m = v > thr;
      if (out) out[((size_t)p * H + y) * W + x] = (uint8_t)m;
      if (m) {
        ++ar;
        mnx = min(mnx, x); mxx = max(mxx, x);
        mny = min(mny, y); mxy = max(mxy, y);
      }
    }
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    hi += __shfl_xor(hi, o, 64); lo += __shfl_xor(lo, o, 64); ar += __shfl_xor(ar, o, 64);
    mnx = min(mnx, __shfl_xor(mnx, o, 64)); mny = min(mny, __shfl_xor(mny, o, 64));
    mxx = max(mxx, __shfl_xor(mxx, o, 64)); mxy = max(mxy, __shfl_xor(mxy, o, 64));
  }
  if ((threadIdx.x & 63) == 0 && (lo || ar)) {
    int* s = stats + (size_t)p * 8;
    if (hi) atomicAdd(s + 0, hi);
    if (lo) atomicAdd(s + 1, lo);
    if (ar) {
      atomicAdd(s + 2, ar);
      atomicMin(s + 3, mnx); atomicMin(s + 4, mny); atomicMax(s + 5, mxx); atomicMax(s + 6, mxy);
    }
  }
}
extern "C" int psam_plane_stats(const float* planes, int n, int H, int W, float thr, float off, int* stats, void* out,
                                void* stream) {
  if (n <= 0 || H <= 0 || W <= 0 || off < 0.f) return PSAM_ERR_ARG;
  hipLaunchKernelGGL(mask_stats_init_kernel, dim3((n * 8 + 255) / 256), dim3(256), 0, (hipStream_t)stream, stats, n);
  hipLaunchKernelGGL(plane_stats_kernel, dim3((H + MS_ROWS - 1) / MS_ROWS, n), dim3(256), 0, (hipStream_t)stream, planes, H, W,
                     thr, off, stats, (uint8_t*)out);
  return psam_launch_status();
}

// ---- mask prompts: PromptEncoder.mask_downscaling (prompt_encoder.py:51-59,102-105) as one kernel -----------------------
// Conv2d(1->4, k2 s2) -> LayerNorm2d(4) -> GELU -> Conv2d(4->16, k2 s2) -> LayerNorm2d(16) -> GELU -> Conv2d(16->256, k1)
// on masks fp32 [n, 4g, 4g] -> token-major dense embeddings fp32 [n, g*g, 256]. One workgroup = 16 tokens x 256 output
// channels; the two small stages (a 4x4 input patch per token) are computed by 16 lanes per token into LDS.
// packed weights (floats): c1w[4][4] c1b[4] n1w[4] n1b[4] c2w[16][4][2][2] c2b[16] n2w[16] n2b[16] c3w[256][16] c3b[256]
#define MD_C1W 0
#define MD_C1B 16
#define MD_N1W 20
#define MD_N1B 24
#define MD_C2W 28
#define MD_C2B (28 + 256)
#define MD_N2W (MD_C2B + 16)
#define MD_N2B (MD_N2W + 16)
#define MD_C3W (MD_N2B + 16)
#define MD_C3B (MD_C3W + 4096)
#define MD_TOTAL (MD_C3B + 256)
__global__ __launch_bounds__(256) void mask_downscale_kernel(const float* __restrict__ masks, const float* __restrict__ wts,
                                                             int g, float eps, float* __restrict__ out) {
  __shared__ float w[MD_C3W];       // everything but the 1x1 conv
  __shared__ float h1[16][4][4];    // [token][c1][py*2+px] after LN + GELU
  __shared__ float h2[16][16];      // [token][c2] after LN + GELU
  const int t = threadIdx.x, n = blockIdx.y;
  const int tok0 = blockIdx.x * 16;
  for (int i = t; i < MD_C3W; i += 256) w[i] = wts[i];
  __syncthreads();
  const int S = 4 * g;
  const float* m = masks + (size_t)n * S * S;
  {  // stage 1: thread (token tl = t/16, position pp = (t%16)/4 ... ) -> 64 (token, position) pairs, 4 channels each
    const int tl = t >> 4, sub = t & 15;
    if (sub < 4) {
      const int tok = tok0 + tl, ty = tok / g, tx = tok % g;
      const int py = sub >> 1, px = sub & 1;
      const float* ip = m + (size_t)(4 * ty + 2 * py) * S + 4 * tx + 2 * px;
      const float i00 = ip[0], i01 = ip[1], i10 = ip[S], i11 = ip[S + 1];
      float c[4], mu = 0.f;
#pragma unroll
      for (int c1 = 0; c1 < 4; ++c1) {
        c[c1] = w[MD_C1B + c1] + w[MD_C1W + c1 * 4 + 0] * i00 + w[MD_C1W + c1 * 4 + 1] * i01 +
                w[MD_C1W + c1 * 4 + 2] * i10 + w[MD_C1W + c1 * 4 + 3] * i11;
        mu += c[c1];
      }
      mu *= 0.25f;
      float var = 0.f;
#pragma unroll
      for (int c1 = 0; c1 < 4; ++c1) var += (c[c1] - mu) * (c[c1] - mu);
      const float inv = 1.0f / sqrtf(var * 0.25f + eps);
#pragma unroll
      for (int c1 = 0; c1 < 4; ++c1) {
        const float v = w[MD_N1W + c1] * ((c[c1] - mu) * inv) + w[MD_N1B + c1];
        h1[tl][c1][sub] = 0.5f * v * (1.0f + erff(v * 0.70710678118654752440f));
      }
    }
  }
  __syncthreads();
  {  // stage 2: thread (token t/16, channel c2 = t%16); LayerNorm over the 16 channels = the 16 lanes of the group
    const int tl = t >> 4, c2 = t & 15;
    float v = w[MD_C2B + c2];
#pragma unroll
    for (int c1 = 0; c1 < 4; ++c1)
#pragma unroll
      for (int q = 0; q < 4; ++q) v += w[MD_C2W + (c2 * 4 + c1) * 4 + q] * h1[tl][c1][q];
    float mu = v;
#pragma unroll
    for (int o = 8; o > 0; o >>= 1) mu += __shfl_xor(mu, o, 64);
    mu *= (1.0f / 16.0f);
    float var = (v - mu) * (v - mu);
#pragma unroll
    for (int o = 8; o > 0; o >>= 1) var += __shfl_xor(var, o, 64);
    const float inv = 1.0f / sqrtf(var * (1.0f / 16.0f) + eps);
    const float y = w[MD_N2W + c2] * ((v - mu) * inv) + w[MD_N2B + c2];
    h2[tl][c2] = 0.5f * y * (1.0f + erff(y * 0.70710678118654752440f));
  }
  __syncthreads();
  // stage 3: thread = output channel, all 16 tokens
  float w3[16];
#pragma unroll
  for (int c2 = 0; c2 < 16; ++c2) w3[c2] = wts[MD_C3W + t * 16 + c2];
  const float b3 = wts[MD_C3B + t];
  for (int tl = 0; tl < 16; ++tl) {
    float v = b3;
#pragma unroll
    for (int c2 = 0; c2 < 16; ++c2) v += w3[c2] * h2[tl][c2];
    out[((size_t)n * g * g + tok0 + tl) * 256 + t] = v;
  }
}
extern "C" int psam_mask_downscale(const float* masks, const float* wts, int n, int g, float eps, float* out, void* stream) {
  if (n <= 0 || g <= 0 || (g * g) % 16 != 0) return PSAM_ERR_ARG;
  hipLaunchKernelGGL(mask_downscale_kernel, dim3(g * g / 16, n), dim3(256), 0, (hipStream_t)stream, masks, wts, g, eps, out);
  return psam_launch_status();
}
