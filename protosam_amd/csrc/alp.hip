// ALP (adaptive local prototype) module kernels: prototype bank + fused cosine-similarity map.
//
// Replaces models/alpmodule.py `MultiProtoAsConv.forward` (:161-198) = `get_prototypes` (:97-159) +
// `get_prediction_from_prototypes` (:57-94) + `safe_norm` (:14-18), and the mask handling / fg-mode
// decision of `FewShotSeg.forward` (models/grid_proto_fewshot.py:228-263), for the inference
// configuration (isval=True, val_wsize pooling, n_ways = n_shots = 1).
//
// Data layout: feature maps stay TOKEN-MAJOR [h*w, C] fp32 exactly as the ViT's final LayerNorm
// emits them (the reference's permute/view to NCHW, grid_proto_fewshot.py:92-95, is never
// materialised). The bank is a fixed-capacity buffer [2*cap, C]: rows [0, n_bg) background
// prototypes, rows [cap, cap + n_fg) foreground prototypes (local cells in row-major cell order, the
// global masked-average prototype LAST, alpmodule.py:155-158), all already `safe_norm`ed. Counts
// live in a device-side int meta[8] = {n_bg, n_fg, fg_mode(0 = 'mask', 1 = 'gridconv+'), ...} so that
// no host sync is needed between bank construction and matching.
//
// 'mask' mode note: F.cosine_similarity(q, p, eps=1e-4)*20 == 20 * safe_norm(q).safe_norm(p) (both
// clamp the norm at 1e-4 and normalise before the dot), and sum(softmax(d)*d) over a single prototype is
// d itself; so 'mask' mode is the P = 1 case of the grid path with only the global prototype in the bank.
//
// All arithmetic is fp32 (the x20 logit scale leaves no room for fp16 operands): the similarity GEMM
// runs on v_mfma_f32_32x32x2_f32, which is bit-wise an fp32 fma chain.
#include "common.h"
#include <stdlib.h>
#include <type_traits>

#define META_NBG 0
#define META_NFG 1
#define META_FGMODE 2
#define META_NCELL_FG 3

__device__ __forceinline__ int nearest_src(int dst, float scale, int in_size) {
  // ATen nearest_neighbor_compute_source_index: min(floor(dst * scale), in - 1), scale = in/out in fp32
  int s = (int)floorf((float)dst * scale);
  return s < in_size - 1 ? s : in_size - 1;
}

// ---- kernel A: per-cell selection flags, slots (exclusive scan), fg mode ------------------------------
// mask: fp32 [MH, MW] (1 = foreground). One block of 256 threads. cells = (h/pw) x (w/pw).
__global__ __launch_bounds__(256) void alp_flags_kernel(const float* __restrict__ mask,
                                                        const float* __restrict__ bmask, int MH, int MW, int h, int w,
                                                        int pw, int ks, float thresh, int* __restrict__ slot_bg,
                                                        int* __restrict__ slot_fg, float* __restrict__ mres,
                                                        int* __restrict__ meta, int force_mode) {
  __shared__ int sbg[256], sfg[256];
  __shared__ int any_full;
  const int t = threadIdx.x;
  const float sh = (float)MH / (float)h, sw = (float)MW / (float)w;
  if (t == 0) any_full = 0;
  // nearest-resized fg mask (F.interpolate(mode='nearest'), grid_proto_fewshot.py:228-231)
  for (int i = t; i < h * w; i += 256) {
    int y = i / w, x = i % w;
    const size_t src = (size_t)nearest_src(y, sh, MH) * MW + nearest_src(x, sw, MW);
    mres[i] = mask[src];
    mres[h * w + i] = bmask ? bmask[src] : 1.f - mask[src];  // back_mask = 1 - fore_mask (ProtoSAM.py:63)
  }
  __syncthreads();
  // fg mode: avg_pool2d(fg, ks).max() >= thresh (grid_proto_fewshot.py:253-256)
  const int kh = h / ks, kwn = w / ks;
  for (int c = t; c < kh * kwn; c += 256) {
    int cy = c / kwn, cx = c % kwn;
    float s = 0.f;
    for (int dy = 0; dy < ks; ++dy)
      for (int dx = 0; dx < ks; ++dx) s += mres[(cy * ks + dy) * w + cx * ks + dx];
    if (s / (float)(ks * ks) >= thresh) atomicOr(&any_full, 1);
  }
  // cell coverage for val_wsize pooling (alpmodule.py:115,131 / :140,153), strict >
  const int ch = h / pw, cw = w / pw, nc = ch * cw;
  const int per = (nc + 255) / 256;
  int cb = 0, cf = 0;
  for (int k = 0; k < per; ++k) {
    int c = t * per + k;
    if (c < nc) {
      int cy = c / cw, cx = c % cw;
      float sf = 0.f, sb = 0.f;
      for (int dy = 0; dy < pw; ++dy)
        for (int dx = 0; dx < pw; ++dx) {
          const int pi = (cy * pw + dy) * w + cx * pw + dx;
          sf += mres[pi];
          sb += mres[h * w + pi];
        }
      const float inv = (float)(pw * pw);
      cf += (sf / inv > thresh);
      cb += (sb / inv > thresh);
    }
  }
  sbg[t] = cb;
  sfg[t] = cf;
  __syncthreads();
  // exclusive scan over 256 partial counts (serial by thread 0: 256 adds, negligible)
  if (t == 0) {
    int ab = 0, af = 0;
    for (int i = 0; i < 256; ++i) {
      int b = sbg[i], f = sfg[i];
      sbg[i] = ab;
      sfg[i] = af;
      ab += b;
      af += f;
    }
    // force_mode: -1 = reference FewShotSeg rule; 0 = 'mask'; 1 = 'gridconv+'; 2 = 'gridconv' (no global proto)
    const int mode = force_mode < 0 ? (any_full ? 1 : 0) : force_mode;
    meta[META_NBG] = ab;
    meta[META_NCELL_FG] = mode ? af : 0;
    meta[META_NFG] = (mode ? af : 0) + (mode == 2 ? 0 : 1);  // + global prototype (last)
    meta[META_FGMODE] = mode;
  }
  __syncthreads();
  const int mode = force_mode < 0 ? (any_full ? 1 : 0) : force_mode;
  int ob = sbg[t], of = sfg[t];
  for (int k = 0; k < per; ++k) {
    int c = t * per + k;
    if (c < nc) {
      int cy = c / cw, cx = c % cw;
      float sf = 0.f, sb = 0.f;
      for (int dy = 0; dy < pw; ++dy)
        for (int dx = 0; dx < pw; ++dx) {
          const int pi = (cy * pw + dy) * w + cx * pw + dx;
          sf += mres[pi];
          sb += mres[h * w + pi];
        }
      const float inv = (float)(pw * pw);
      bool f = sf / inv > thresh, b = sb / inv > thresh;
      slot_fg[c] = (f && mode) ? of : -1;
      slot_bg[c] = b ? ob : -1;
      of += f;
      ob += b;
    }
  }
}

// ---- kernel B: un-normalised prototypes ------------------------------------------------------------
// grid.x = ncell + 1. Blocks [0, ncell): avg-pool of the support features over the cell, written to the
// bg and/or fg slot. Block ncell: global masked-average prototype sum(x*m)/(sum(m)+1e-5).
__global__ __launch_bounds__(256) void alp_protos_kernel(const float* __restrict__ sup, int ld, int h, int w, int C,
                                                         int pw, const float* __restrict__ mres,
                                                         const int* __restrict__ slot_bg,
                                                         const int* __restrict__ slot_fg, const int* __restrict__ meta,
                                                         float* __restrict__ bank, int cap) {
  const int ch = h / pw, cw = w / pw, nc = ch * cw;
  const int c = blockIdx.x;
  if (c < nc) {
    const int sb = slot_bg[c], sf = slot_fg[c];
    if (sb < 0 && sf < 0) return;
    const int cy = c / cw, cx = c % cw;
    const float cnt = (float)(pw * pw);
    for (int k = threadIdx.x; k < C; k += 256) {
      float s = 0.f;
      for (int dy = 0; dy < pw; ++dy)
        for (int dx = 0; dx < pw; ++dx) s += sup[(size_t)((cy * pw + dy) * w + cx * pw + dx) * ld + k];
      s = s / cnt;
      if (sb >= 0 && sb < cap) bank[(size_t)sb * C + k] = s;
      if (sf >= 0 && sf < cap - 1) bank[(size_t)(cap + sf) * C + k] = s;
    }
  } else {
    if (meta[META_FGMODE] == 2) return;
    __shared__ float msum_s;
    if (threadIdx.x < 64) {
      float s = 0.f;
      for (int i = threadIdx.x; i < h * w; i += 64) s += mres[i];
      s = wave_sum(s);
      if (threadIdx.x == 0) msum_s = s;
    }
    __syncthreads();
    const float den = msum_s + 1e-5f;
    int row = meta[META_NCELL_FG];
    if (row > cap - 1) row = cap - 1;
    // sum over all h*w pixels of x * m, per channel: the four waves take the pixels i = wave (mod 4), each lane a channel,
    // four pixels in flight per step; partial sums are combined in a fixed order (wave 0..3) through LDS. (One thread walking all
    // 1296 pixels of its channels in sequence took 836 us - one dependent global load after the other.)
    __shared__ float psum[4][64];
    const int wv = threadIdx.x >> 6, lane = threadIdx.x & 63, npx = h * w;
    if ((C & 255) == 0 && (ld & 3) == 0 && (reinterpret_cast<uintptr_t>(sup) & 15) == 0) {
      // four channels per lane (float4 loads): a quarter of the steps
      __shared__ float4 psum4[4][64];
      for (int k0 = 0; k0 < C; k0 += 256) {
        const int k = k0 + lane * 4;
        float4 s0 = make_float4(0.f, 0.f, 0.f, 0.f), s1 = s0, s2 = s0, s3 = s0;
        auto fma4 = [](float4& a, const float4 v, float m) { a.x += v.x * m; a.y += v.y * m; a.z += v.z * m; a.w += v.w * m; };
        int i = wv;
        for (; i + 12 < npx; i += 16) {
          const float4 a0 = *reinterpret_cast<const float4*>(sup + (size_t)i * ld + k);
          const float4 a1 = *reinterpret_cast<const float4*>(sup + (size_t)(i + 4) * ld + k);
          const float4 a2 = *reinterpret_cast<const float4*>(sup + (size_t)(i + 8) * ld + k);
          const float4 a3 = *reinterpret_cast<const float4*>(sup + (size_t)(i + 12) * ld + k);
          fma4(s0, a0, mres[i]); fma4(s1, a1, mres[i + 4]); fma4(s2, a2, mres[i + 8]); fma4(s3, a3, mres[i + 12]);
        }
        for (; i < npx; i += 4) fma4(s0, *reinterpret_cast<const float4*>(sup + (size_t)i * ld + k), mres[i]);
        psum4[wv][lane] = make_float4((s0.x + s1.x) + (s2.x + s3.x), (s0.y + s1.y) + (s2.y + s3.y), (s0.z + s1.z) + (s2.z + s3.z),
                                      (s0.w + s1.w) + (s2.w + s3.w));
        __syncthreads();
        if (wv == 0) {
          const float4 p0 = psum4[0][lane], p1 = psum4[1][lane], p2 = psum4[2][lane], p3 = psum4[3][lane];
          *reinterpret_cast<float4*>(bank + (size_t)(cap + row) * C + k) =
              make_float4((((p0.x + p1.x) + p2.x) + p3.x) / den, (((p0.y + p1.y) + p2.y) + p3.y) / den,
                          (((p0.z + p1.z) + p2.z) + p3.z) / den, (((p0.w + p1.w) + p2.w) + p3.w) / den);
        }
        __syncthreads();
      }
      return;
    }
    for (int k0 = 0; k0 < C; k0 += 64) {
      const int k = k0 + lane;
      float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
      int i = wv;
      for (; i + 12 < npx; i += 16) {
        const float a0 = k < C ? sup[(size_t)i * ld + k] : 0.f, a1 = k < C ? sup[(size_t)(i + 4) * ld + k] : 0.f;
        const float a2 = k < C ? sup[(size_t)(i + 8) * ld + k] : 0.f, a3 = k < C ? sup[(size_t)(i + 12) * ld + k] : 0.f;
        s0 += a0 * mres[i]; s1 += a1 * mres[i + 4]; s2 += a2 * mres[i + 8]; s3 += a3 * mres[i + 12];
      }
      for (; i < npx; i += 4) s0 += (k < C ? sup[(size_t)i * ld + k] : 0.f) * mres[i];
      psum[wv][lane] = (s0 + s1) + (s2 + s3);
      __syncthreads();
      if (wv == 0 && k < C) bank[(size_t)(cap + row) * C + k] = (((psum[0][lane] + psum[1][lane]) + psum[2][lane]) + psum[3][lane]) / den;
      __syncthreads();
    }
  }
}

// ---- kernel C: safe_norm rows in place (one wave per row) -----------------------------------------------
__global__ __launch_bounds__(256) void alp_norm_kernel(float* __restrict__ bank, const int* __restrict__ meta, int cap,
                                                       int C, float eps) {
  const int r = blockIdx.x * 4 + (threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  if (r >= 2 * cap) return;
  const int n = r < cap ? meta[META_NBG] : meta[META_NFG];
  if ((r < cap ? r : r - cap) >= n) return;
  float* p = bank + (size_t)r * C;
  float s = 0.f;
  for (int k = lane; k < C; k += 64) s += p[k] * p[k];
  s = wave_sum(s);
  const float nrm = fmaxf(sqrtf(s), eps);
  for (int k = lane; k < C; k += 64) p[k] = p[k] / nrm;
}

// ---- kernel D: similarity GEMM + partial softmax-weighted sum ------------------------------------------
// grid (pixel tiles of 64, proto tiles of 64, 2*B): z = b*2 + bank. Block = 4 waves (2 proto x 2 pixel halves).
// part[((z*npt + ptile)*npix_pad + pix)*3 + {m, Z, W}]
#define SIM_LD 33
__global__ __launch_bounds__(256) void alp_sim_kernel(const float* __restrict__ qry, size_t q_bstride, int ld, int npix,
                                                      int C, const float* __restrict__ bank, int cap,
                                                      const int* __restrict__ meta, float eps, float sim_scale,
                                                      float* __restrict__ part, int npt, int npix_pad,
                                                      int which_only) {
  __shared__ float Ps[64 * SIM_LD];
  __shared__ float Qs[64 * SIM_LD];
  __shared__ float red[2][64][3];
  __shared__ float qn2[64];
  const int z = blockIdx.z, b = z >> 1, which = z & 1;
  if (which_only >= 0 && which != which_only) return;
  const int n = which ? meta[META_NFG] : meta[META_NBG];
  const int p0 = blockIdx.y * 64;
  if (p0 >= n) return;
  const int x0 = blockIdx.x * 64;
  const float* Q = qry + (size_t)b * q_bstride;
  const float* P = bank + (size_t)(which * cap) * C;
  const int t = threadIdx.x, lane = t & 63, wv = t >> 6;
  const int wp = wv >> 1, wx = wv & 1;  // proto half, pixel half
  const int lr = lane & 31, lk = lane >> 5;

  f32x16 acc;
#pragma unroll
  for (int r = 0; r < 16; ++r) acc[r] = 0.f;
  float qsq = 0.f;  // threads 0..63: |q|^2 of pixel x0 + t

  // stage mapping: thread loads 8 floats of one row: row = t >> 2, cols (t&3)*8 .. +7
  const int srow = t >> 2, scol = (t & 3) * 8;
  const int prow = p0 + srow, xrow = x0 + srow;
  for (int k0 = 0; k0 < C; k0 += 32) {
    float pv[8], qv[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      pv[e] = prow < n ? P[(size_t)prow * C + k0 + scol + e] : 0.f;
      qv[e] = xrow < npix ? Q[(size_t)xrow * ld + k0 + scol + e] : 0.f;
    }
    __syncthreads();
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      Ps[srow * SIM_LD + scol + e] = pv[e];
      Qs[srow * SIM_LD + scol + e] = qv[e];
    }
    __syncthreads();
    if (t < 64) {
#pragma unroll
      for (int e = 0; e < 32; ++e) {
        float v = Qs[t * SIM_LD + e];
        qsq += v * v;
      }
    }
#pragma unroll
    for (int s = 0; s < 16; ++s) {
      float a = Ps[(wp * 32 + lr) * SIM_LD + 2 * s + lk];
      float bq = Qs[(wx * 32 + lr) * SIM_LD + 2 * s + lk];
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, bq, acc, 0, 0, 0);
    }
  }
  if (t < 64) qn2[t] = qsq;
  __syncthreads();
  // acc[r]: proto row = p0 + wp*32 + (r&3) + 8*(r>>2) + 4*lk ; pixel col = x0 + wx*32 + lr
  const int pxl = wx * 32 + lr;
  const float qinv = sim_scale / fmaxf(sqrtf(qn2[pxl]), eps);
  float d[16];
  float m = -INFINITY;
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const int pr = p0 + wp * 32 + (r & 3) + 8 * (r >> 2) + 4 * lk;
    d[r] = pr < n ? acc[r] * qinv : -INFINITY;
    m = fmaxf(m, d[r]);
  }
  m = fmaxf(m, __shfl_xor(m, 32, 64));
  float Z = 0.f, W = 0.f;
  if (m > -INFINITY) {
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      if (d[r] > -INFINITY) {
        float e = expf(d[r] - m);
        Z += e;
        W += e * d[r];
      }
    }
  }
  Z += __shfl_xor(Z, 32, 64);
  W += __shfl_xor(W, 32, 64);
  if (lk == 0) {
    red[wp][pxl][0] = m;
    red[wp][pxl][1] = Z;
    red[wp][pxl][2] = W;
  }
  __syncthreads();
  if (t < 64 && x0 + t < npix) {
    float m0 = red[0][t][0], m1 = red[1][t][0];
    float mm = fmaxf(m0, m1);
    float e0 = m0 > -INFINITY ? expf(m0 - mm) : 0.f;
    float e1 = m1 > -INFINITY ? expf(m1 - mm) : 0.f;
    float* o = part + (((size_t)z * npt + blockIdx.y) * npix_pad + x0 + t) * 3;
    o[0] = mm;
    o[1] = red[0][t][1] * e0 + red[1][t][1] * e1;
    o[2] = red[0][t][2] * e0 + red[1][t][2] * e1;
  }
}

// ---- kernel E: merge proto-tile partials -> pred[b, bank, pix] = sum(softmax(d) * d) --------------------
__global__ void alp_combine_kernel(const float* __restrict__ part, const int* __restrict__ meta, int npt, int npix,
                                   int npix_pad, float* __restrict__ pred, int which_only, int gsize) {
  const int z = blockIdx.y;  // b*2 + bank
  if (which_only >= 0 && (z & 1) != which_only) return;
  const int x = blockIdx.x * blockDim.x + threadIdx.x;
  if (x >= npix) return;
  const int n = (z & 1) ? meta[META_NFG] : meta[META_NBG];
  // (an empty bank: kernel D2's first group still writes (-inf, 0, 0); kernel D writes nothing; either way pred = 0 / 0)
  const int nt = n > 0 ? (n + gsize - 1) / gsize : (gsize == 96 ? 1 : 0);
  float m = -INFINITY, Z = 0.f, W = 0.f;
  for (int i = 0; i < nt && i < npt; ++i) {
    const float* p = part + (((size_t)z * npt + i) * npix_pad + x) * 3;
    float mi = p[0];
    float mm = fmaxf(m, mi);
    float ea = m > -INFINITY ? expf(m - mm) : 0.f;
    float eb = mi > -INFINITY ? expf(mi - mm) : 0.f;
    Z = Z * ea + p[1] * eb;
    W = W * ea + p[2] * eb;
    m = mm;
  }
  pred[(size_t)z * npix + x] = W / Z;
}

// ---- kernel D2: similarity GEMM + softmax-weighted sum in ONE pass (the default; D + E above are kept for A/B) -------------
// One workgroup = 128 query pixels of one (batch, bank) against one GROUP of 96 prototypes (three 32 x 32 MFMA tiles per wave),
// four waves of 32 pixels each sharing the group's LDS image (64.5 KiB: two workgroups = two waves per SIMD per CU); grid.y walks the groups (validation pools the 36 x 36 support map 2 x 2: up to 325 prototypes
// per bank, typically ~300 background and a few dozen foreground ones - a first version that looped over the groups inside
// the workgroup ran at the MFMA rate per wave but left the background workgroups four times as long as the foreground ones:
// 343 us). The K loop (C = 768 in stages of 32) is double-buffered through registers with one barrier per stage; operands are
// fetched as float4 (kernel D: 16 scalar loads and 16 scalar LDS writes per thread and stage, two barriers, no overlap of loads
// and MFMAs: 384 us per 16-slice step for ~100 us of fp32-MFMA work). LDS rows are 36 floats: a ds_read_b128 of the 32 rows of a fragment is conflict-free, and
// MFMA s of a stage consumes k = {s, s + 16}, so one float4 per lane feeds four MFMAs. |q|^2 comes from the fragments already in
// registers. A bank of one group writes pred = W / Z directly; otherwise the per-pixel (max, Z, W) of the group goes to `part`
// and kernel E merges the groups.
#define S2_LD 36
template <int NWV>   // waves per workgroup = 32-pixel strips sharing one prototype group's LDS image
__global__ __launch_bounds__(NWV * 64) void alp_sim2_kernel(const float* __restrict__ qry, size_t q_bstride, int ld, int npix, int C,
                                                       const float* __restrict__ bank, int cap, const int* __restrict__ meta,
                                                       float eps, float sim_scale, float* __restrict__ pred, int which_only, int dbg,
                                                       float* __restrict__ part, int npt, int npix_pad) {
  constexpr int PX = NWV * 32, NT = NWV * 64;
  constexpr int QI = PX * 8 / NT, PI = 96 * 8 / NT, RSTEP = NT / 8;   // float4 loads per thread and stage, row step between them
  __shared__ __attribute__((aligned(16))) float Qs[2][PX * S2_LD];
  __shared__ __attribute__((aligned(16))) float Ps[2][96 * S2_LD];
  const int z = blockIdx.z, b = z >> 1, which = z & 1;
  if (which_only >= 0 && which != which_only) return;
  const int n = which ? meta[META_NFG] : meta[META_NBG];
  const int g0 = blockIdx.y * 96;
  if (g0 >= n && g0 > 0) return;
  const int x0 = blockIdx.x * PX;
  const float* Q = qry + (size_t)b * q_bstride;
  const float* P = bank + (size_t)(which * cap) * C;
  const int t = threadIdx.x, lane = t & 63, wv = t >> 6;
  const int lr = lane & 31, lk = lane >> 5;
  const int srow = t >> 3, sc4 = (t & 7) * 4;           // staging: rows srow + 16 i, floats sc4 .. sc4 + 3 of the stage
  const int nst = C / 32;

  float m = -INFINITY, Z = 0.f, W = 0.f, qsq = 0.f;
  // global -> register prefetch TWO stages ahead (two register sets, the stage loop unrolled by two so that the sets are
  // static): with 46 KiB of LDS only three workgroups = 1.5 waves per SIMD are resident, so a stage's loads have to be hidden by
  // the wave's own MFMAs - one stage (48 MFMAs, ~1.5 us) is shorter than the load latency under load, two are not
  float4 qreg[2][QI], preg[2][PI];
  {
    f32x16 acc[3];
#pragma unroll
    for (int pt = 0; pt < 3; ++pt)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[pt][r] = 0.f;
    auto load_stage = [&](int st, auto set_c) {
      constexpr int SET = decltype(set_c)::value;
      const int k0 = st * 32 + sc4;
#pragma unroll
      for (int i = 0; i < QI; ++i) {
        const int x = x0 + srow + RSTEP * i;
        qreg[SET][i] = x < npix ? *reinterpret_cast<const float4*>(Q + (size_t)x * ld + k0) : make_float4(0.f, 0.f, 0.f, 0.f);
      }
#pragma unroll
      for (int i = 0; i < PI; ++i) {
        const int pr = g0 + srow + RSTEP * i;
        preg[SET][i] = pr < n ? *reinterpret_cast<const float4*>(P + (size_t)pr * C + k0) : make_float4(0.f, 0.f, 0.f, 0.f);
      }
    };
    auto store_stage = [&](int buf, auto set_c) {
      constexpr int SET = decltype(set_c)::value;
#pragma unroll
      for (int i = 0; i < QI; ++i) *reinterpret_cast<float4*>(&Qs[buf][(srow + RSTEP * i) * S2_LD + sc4]) = qreg[SET][i];
#pragma unroll
      for (int i = 0; i < PI; ++i) *reinterpret_cast<float4*>(&Ps[buf][(srow + RSTEP * i) * S2_LD + sc4]) = preg[SET][i];
    };
    using C0 = std::integral_constant<int, 0>;
    using C1 = std::integral_constant<int, 1>;
    load_stage(0, C0{});
    if (nst > 1) load_stage(1, C1{});
    store_stage(0, C0{});
    __syncthreads();
    // stage st lives in LDS buffer st & 1 and was carried in register set st & 1
    auto stage = [&](auto par_c, int st) {
      constexpr int PAR = decltype(par_c)::value;
      using CP = std::integral_constant<int, PAR>;
      using CN = std::integral_constant<int, PAR ^ 1>;
      if (st + 2 < nst && !(dbg & 1)) load_stage(st + 2, CP{});      // set PAR is free: stage st went to LDS an iteration ago
      if (!(dbg & 2))
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const float4 qv = *reinterpret_cast<const float4*>(&Qs[PAR][(wv * 32 + lr) * S2_LD + 16 * lk + 4 * j]);
        qsq += (qv.x * qv.x + qv.y * qv.y) + (qv.z * qv.z + qv.w * qv.w);
        const float qe[4] = {qv.x, qv.y, qv.z, qv.w};
#pragma unroll
        for (int pt = 0; pt < 3; ++pt) {
          const float4 pv = *reinterpret_cast<const float4*>(&Ps[PAR][(pt * 32 + lr) * S2_LD + 16 * lk + 4 * j]);
          const float pe[4] = {pv.x, pv.y, pv.z, pv.w};
#pragma unroll
          for (int e = 0; e < 4; ++e) acc[pt] = __builtin_amdgcn_mfma_f32_32x32x2f32(pe[e], qe[e], acc[pt], 0, 0, 0);
        }
      }
      if (st + 1 < nst && !(dbg & 4)) store_stage(PAR ^ 1, CN{});     // stage st + 1, requested one stage ago
      if (!(dbg & 8)) __syncthreads();
    };
    for (int st = 0; st < nst; st += 2) {
      stage(C0{}, st);
      if (st + 1 < nst) stage(C1{}, st + 1);
    }
    // acc[pt][r]: prototype g0 + pt*32 + (r&3) + 8*(r>>2) + 4*lk, pixel x0 + wv*32 + lr
    qsq += __shfl_xor(qsq, 32, 64);
    const float qinv = sim_scale / fmaxf(sqrtf(qsq), eps);
    float d[3][16];
    float mg = -INFINITY;
#pragma unroll
    for (int pt = 0; pt < 3; ++pt)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int pr = g0 + pt * 32 + (r & 3) + 8 * (r >> 2) + 4 * lk;
        d[pt][r] = pr < n ? acc[pt][r] * qinv : -INFINITY;
        mg = fmaxf(mg, d[pt][r]);
      }
    if (mg > -INFINITY) {
      const float mm = fmaxf(m, mg);
      const float ea = m > -INFINITY ? expf(m - mm) : 0.f;
      float Zg = 0.f, Wg = 0.f;
#pragma unroll
      for (int pt = 0; pt < 3; ++pt)
#pragma unroll
        for (int r = 0; r < 16; ++r)
          if (d[pt][r] > -INFINITY) {
            const float e = expf(d[pt][r] - mm);
            Zg += e;
            Wg += e * d[pt][r];
          }
      Z = Z * ea + Zg;
      W = W * ea + Wg;
      m = mm;
    }
  }
  // the two lane halves hold disjoint prototype subsets of the same pixel
  const float mo = __shfl_xor(m, 32, 64), Zo = __shfl_xor(Z, 32, 64), Wo = __shfl_xor(W, 32, 64);
  const float mm = fmaxf(m, mo);
  const float ea = m > -INFINITY ? expf(m - mm) : 0.f, eb = mo > -INFINITY ? expf(mo - mm) : 0.f;
  const float Zt = Z * ea + Zo * eb, Wt = W * ea + Wo * eb;
  const int x = x0 + wv * 32 + lr;
  if (lk == 0 && x < npix) {
    if (npt == 1) {
      pred[(size_t)z * npix + x] = Wt / Zt;
    } else {
      float* o = part + (((size_t)z * npt + blockIdx.y) * npix_pad + x) * 3;
      o[0] = mm;
      o[1] = Zt;
      o[2] = Wt;
    }
  }
}

// mres: scratch fp32 [2*h*w] (nearest-resized fg mask, then bg mask). bmask may be null (= 1 - mask).
extern "C" int psam_alp_bank(const float* sup, int ld, int h, int w, int C, const float* mask, const float* bmask,
                             int MH, int MW,
                             int pool_w, int kernel_size, float thresh, float eps, float* bank, int cap, int* meta,
                             int* slot_bg, int* slot_fg, float* mres, int force_mode, void* stream) {
  if (h <= 0 || w <= 0 || C <= 0 || pool_w <= 0 || kernel_size <= 0 || cap < 2) return PSAM_ERR_ARG;
  const int nc = (h / pool_w) * (w / pool_w);
  if (nc + 1 > cap) return PSAM_ERR_ARG;  // capacity must hold every cell + the global prototype
  hipStream_t s = (hipStream_t)stream;
  hipLaunchKernelGGL(alp_flags_kernel, dim3(1), dim3(256), 0, s, mask, bmask, MH, MW, h, w, pool_w, kernel_size,
                     thresh,
                     slot_bg, slot_fg, mres, meta, force_mode);
  hipLaunchKernelGGL(alp_protos_kernel, dim3(nc + 1), dim3(256), 0, s, sup, ld, h, w, C, pool_w, mres, slot_bg,
                     slot_fg, meta, bank, cap);
  hipLaunchKernelGGL(alp_norm_kernel, dim3((2 * cap + 3) / 4), dim3(256), 0, s, bank, meta, cap, C, eps);
  return psam_launch_status();
}

// qry: fp32 token-major, batch b at qry + b*q_bstride, row stride ld. pred: fp32 [B, 2, npix] (bg, fg).
// part: scratch fp32 [2*B * ceil(cap/64) * npix_pad * 3], npix_pad = ceil(npix/64)*64.
extern "C" int psam_alp_sim(const float* qry, long long q_bstride, int ld, int B, int npix, int C, const float* bank,
                            int cap, const int* meta, float eps, float sim_scale, float* part, float* pred,
                            int which_only, void* stream) {
  if (B <= 0 || npix <= 0 || (C % 32) != 0 || cap < 2) return PSAM_ERR_ARG;
  const int npt = (cap + 63) / 64, nxt = (npix + 63) / 64, npix_pad = nxt * 64;
  hipStream_t s = (hipStream_t)stream;
  static int v2 = -1;
  if (v2 < 0) { const char* e = getenv("PSAM_ALP_SIM2"); v2 = e ? atoi(e) : 1; }
  if (v2 && (ld % 4) == 0 && (reinterpret_cast<uintptr_t>(qry) & 15) == 0 && (q_bstride % 4) == 0) {
    static int dbg = -1;
    if (dbg < 0) { const char* e = getenv("PSAM_ALP_DBG"); dbg = e ? atoi(e) : 0; }
    const int ng = (cap + 95) / 96;          // (<= ceil(cap / 64): `part` as sized for kernel D is large enough)
    static int nwv = -1;
    if (nwv < 0) { const char* e = getenv("PSAM_ALP_WAVES"); nwv = e ? atoi(e) : 4; }   // measured: 246 us with two-wave workgroups, 192 us with four (16-slice step)
    if (nwv == 4)
      hipLaunchKernelGGL(alp_sim2_kernel<4>, dim3((npix + 127) / 128, ng, 2 * B), dim3(256), 0, s, qry, (size_t)q_bstride, ld, npix, C,
                         bank, cap, meta, eps, sim_scale, pred, which_only, dbg, part, ng, npix_pad);
    else
      hipLaunchKernelGGL(alp_sim2_kernel<2>, dim3(nxt, ng, 2 * B), dim3(128), 0, s, qry, (size_t)q_bstride, ld, npix, C, bank, cap,
                         meta, eps, sim_scale, pred, which_only, dbg, part, ng, npix_pad);
    if (ng > 1)
      hipLaunchKernelGGL(alp_combine_kernel, dim3((npix + 255) / 256, 2 * B), dim3(256), 0, s, part, meta, ng, npix, npix_pad, pred,
                         which_only, 96);
    return psam_launch_status();
  }
  hipLaunchKernelGGL(alp_sim_kernel, dim3(nxt, npt, 2 * B), dim3(256), 0, s, qry, (size_t)q_bstride, ld, npix, C, bank,
                     cap, meta, eps, sim_scale, part, npt, npix_pad, which_only);
  hipLaunchKernelGGL(alp_combine_kernel, dim3((npix + 255) / 256, 2 * B), dim3(256), 0, s, part, meta, npt, npix,
                     npix_pad, pred, which_only, 64);
  return psam_launch_status();
}
