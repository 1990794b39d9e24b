// HBM-bound resampling / packing kernels around the two ViTs.
//
//  psam_patchify_bilinear : F.interpolate(imgs, (S,S), 'bilinear') (models/grid_proto_fewshot.py:88-89)
//                           fused with the im2col of DINOv2's PatchEmbed conv (k=14, s=14) -> fp16 rows.
//  psam_bilinear_nchw     : F.interpolate(x, size, 'bilinear', align_corners=False) on fp32 NCHW
//                           (grid_proto_fewshot.py:272-273; models/ProtoSAM.py:592-594).
//  psam_prob_argmax       : [optional bilinear to (OH,OW)] -> softmax(dim=1) -> argmax for the 2-class
//                           coarse logits (models/ProtoSAM.py:592-602), writing output_p and uint8 pred.
//  psam_minmax / psam_sam_patchify : per-image min/max, then ((x-min)/(max-min)*255).astype(uint8)
//                           (ProtoSAM.py:660) -> (u8 - pixel_mean)/pixel_std (modeling/sam.py:163-173)
//                           -> im2col of SAM's PatchEmbed conv (k=16, s=16) -> fp16 rows.
//  psam_broadcast_rows    : writes one fp32 row into out[b*stride + off] (cls token + pos_embed[0]).
//
// Bilinear source index math is ATen's area_pixel_compute_source_index (align_corners=False):
//   src = max(scale*(dst+0.5)-0.5, 0), scale = in/out in fp32; i0 = (int)src, i1 = i0 + (i0 < in-1),
//   l1 = src - i0, l0 = 1 - l1;  out = h0*(w0*p00 + w1*p01) + h1*(w0*p10 + w1*p11).
#include "common.h"

struct Lin {
  int i0, i1;
  float l0, l1;
};
__device__ __forceinline__ Lin lin_src(int dst, float scale, int in_size) {
  float s = scale * ((float)dst + 0.5f) - 0.5f;
  s = s < 0.f ? 0.f : s;
  int i0 = (int)s;
  if (i0 > in_size - 1) i0 = in_size - 1;
  Lin r;
  r.i0 = i0;
  r.i1 = i0 + (i0 < in_size - 1 ? 1 : 0);
  float l1 = s - (float)i0;
  l1 = l1 < 0.f ? 0.f : (l1 > 1.f ? 1.f : l1);
  r.l1 = l1;
  r.l0 = 1.f - l1;
  return r;
}
__device__ __forceinline__ float bilerp(const float* __restrict__ p, int W, const Lin& y, const Lin& x) {
  const float* r0 = p + (size_t)y.i0 * W;
  const float* r1 = p + (size_t)y.i1 * W;
  return y.l0 * (x.l0 * r0[x.i0] + x.l1 * r0[x.i1]) + y.l1 * (x.l0 * r1[x.i0] + x.l1 * r1[x.i1]);
}

// out[(b*np + py*npw + px)*Kpad + c*P*P + ky*P + kx] = resize(img)[b,c,py*P+ky,px*P+kx]; zero for k >= 3*P*P
__global__ void patchify_bilinear_kernel(const float* __restrict__ img, int B, int C, int H, int W, int S, int P,
                                         int Kpad, half_t* __restrict__ out) {
  const int npw = S / P;
  const int patch = blockIdx.x, b = blockIdx.y;
  const int py = patch / npw, px = patch % npw;
  const float sh = (float)H / (float)S, sw = (float)W / (float)S;
  half_t* o = out + ((size_t)b * npw * npw + patch) * Kpad;
  const int K = C * P * P;
  const bool same = (H == S && W == S);
  for (int k = threadIdx.x; k < Kpad; k += blockDim.x) {
    float v = 0.f;
    if (k < K) {
      int c = k / (P * P), r = k % (P * P);
      int y = py * P + r / P, x = px * P + r % P;
      const float* pl = img + ((size_t)b * C + c) * H * W;
      if (same) {
        v = pl[(size_t)y * W + x];
      } else {
        Lin ly = lin_src(y, sh, H), lx = lin_src(x, sw, W);
        v = bilerp(pl, W, ly, lx);
      }
    }
    o[k] = (half_t)v;
  }
}

extern "C" int psam_patchify_bilinear(const float* img, int B, int C, int H, int W, int S, int P, int Kpad, void* out,
                                      void* stream) {
  if (B <= 0 || S % P || Kpad < C * P * P) return PSAM_ERR_ARG;
  const int np = (S / P) * (S / P);
  hipLaunchKernelGGL(patchify_bilinear_kernel, dim3(np, B), dim3(256), 0, (hipStream_t)stream, img, B, C, H, W, S, P,
                     Kpad, (half_t*)out);
  return psam_launch_status();
}

__global__ void bilinear_nchw_kernel(const float* __restrict__ in, int planes, int IH, int IW, int OH, int OW,
                                     float* __restrict__ out) {
  const int x = blockIdx.x * blockDim.x + threadIdx.x;
  const int y = blockIdx.y;
  const int pl = blockIdx.z;
  if (x >= OW) return;
  const float sh = (float)IH / (float)OH, sw = (float)IW / (float)OW;
  Lin ly = lin_src(y, sh, IH), lx = lin_src(x, sw, IW);
  out[((size_t)pl * OH + y) * OW + x] = bilerp(in + (size_t)pl * IH * IW, IW, ly, lx);
}

extern "C" int psam_bilinear_nchw(const float* in, int planes, int IH, int IW, int OH, int OW, float* out,
                                  void* stream) {
  if (planes <= 0 || IH <= 0 || IW <= 0 || OH <= 0 || OW <= 0) return PSAM_ERR_ARG;
  hipLaunchKernelGGL(bilinear_nchw_kernel, dim3((OW + 255) / 256, OH, planes), dim3(256), 0, (hipStream_t)stream, in,
                     planes, IH, IW, OH, OW, out);
  return psam_launch_status();
}

// F.interpolate(x, (OH,OW)) of fp32 planes in the two other conventions the vendored SAM copies use for the second stage of
// postprocess_masks: mode 1 = bilinear align_corners=True (SamBatched, modeling/sam.py:313-320: src = dst*(in-1)/(out-1)),
// mode 2 = nearest (vendored Sam, modeling/sam.py:154-160: src = min(floor(dst*in/out), in-1)); mode 0 = psam_bilinear_nchw.
__global__ void resize2d_kernel(const float* __restrict__ in, int IH, int IW, int OH, int OW, int mode,
                                float* __restrict__ out) {
  const int x = blockIdx.x * blockDim.x + threadIdx.x;
  const int y = blockIdx.y;
  const int pl = blockIdx.z;
  if (x >= OW) return;
  const float* p = in + (size_t)pl * IH * IW;
  float v;
  if (mode == 2) {
    const float sh = (float)IH / (float)OH, sw = (float)IW / (float)OW;
    int sy = (int)floorf((float)y * sh), sx = (int)floorf((float)x * sw);
    sy = sy < IH - 1 ? sy : IH - 1;
    sx = sx < IW - 1 ? sx : IW - 1;
    v = p[(size_t)sy * IW + sx];
  } else {
    Lin ly, lx;
    const float sh = OH > 1 ? (float)(IH - 1) / (float)(OH - 1) : 0.f, sw = OW > 1 ? (float)(IW - 1) / (float)(OW - 1) : 0.f;
    const float fy = sh * (float)y, fx = sw * (float)x;
    ly.i0 = min((int)fy, IH - 1); ly.i1 = ly.i0 + (ly.i0 < IH - 1 ? 1 : 0); ly.l1 = fy - (float)ly.i0; ly.l0 = 1.f - ly.l1;
    lx.i0 = min((int)fx, IW - 1); lx.i1 = lx.i0 + (lx.i0 < IW - 1 ? 1 : 0); lx.l1 = fx - (float)lx.i0; lx.l0 = 1.f - lx.l1;
    v = bilerp(p, IW, ly, lx);
  }
  out[((size_t)pl * OH + y) * OW + x] = v;
}
extern "C" int psam_resize2d(const float* in, int planes, int IH, int IW, int OH, int OW, int mode, float* out,
                             void* stream) {
  if (planes <= 0 || IH <= 0 || IW <= 0 || OH <= 0 || OW <= 0 || mode < 0 || mode > 2) return PSAM_ERR_ARG;
  if (mode == 0) return psam_bilinear_nchw(in, planes, IH, IW, OH, OW, out, stream);
  hipLaunchKernelGGL(resize2d_kernel, dim3((OW + 255) / 256, OH, planes), dim3(256), 0, (hipStream_t)stream, in, IH, IW, OH,
                     OW, mode, out);
  return psam_launch_status();
}

// logits [B,2,IH,IW] -> (bilinear to OH,OW unless equal) -> softmax -> prob [B,2,OH,OW], pred u8 [B,OH,OW]
// fg_sum[b] (optional): number of foreground pixels (int32, atomically accumulated; caller zeroes it).
// Each thread produces four consecutive pixels of a row (16-byte probability stores, one 4-byte label store); the
// foreground count is reduced over the workgroup before the single atomic (it was one contended atomic per wave).
__global__ __launch_bounds__(256) void prob_argmax_kernel(const float* __restrict__ logits, int IH, int IW, int OH, int OW,
                                                          float* __restrict__ prob, uint8_t* __restrict__ pred,
                                                          int* __restrict__ fg_sum) {
  __shared__ int wsum[4];
  const int x0 = (blockIdx.x * blockDim.x + threadIdx.x) * 4;
  const int y = blockIdx.y, b = blockIdx.z;
  int fgc = 0;
  if (x0 < OW) {
    const float* l0p = logits + ((size_t)b * 2 + 0) * IH * IW;
    const float* l1p = l0p + (size_t)IH * IW;
    const bool same = (IH == OH && IW == OW);
    const float sh = (float)IH / (float)OH, sw = (float)IW / (float)OW;
    Lin ly = lin_src(y, sh, IH);
    float p0v[4], p1v[4];
    uint8_t fgv[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const int x = min(x0 + k, OW - 1);
      float l0, l1;
      if (same) {
        l0 = l0p[(size_t)y * IW + x];
        l1 = l1p[(size_t)y * IW + x];
      } else {
        Lin lx = lin_src(x, sw, IW);
        l0 = bilerp(l0p, IW, ly, lx);
        l1 = bilerp(l1p, IW, ly, lx);
      }
      const float m = fmaxf(l0, l1);
      const float e0 = expf(l0 - m), e1 = expf(l1 - m);
      const float s = e0 + e1;
      p0v[k] = e0 / s;
      p1v[k] = e1 / s;
      fgv[k] = p1v[k] > p0v[k] ? 1 : 0;  // argmax returns the first maximum on ties
      if (x0 + k < OW) fgc += fgv[k];
    }
    const size_t o = ((size_t)b * 2) * OH * OW + (size_t)y * OW + x0;
    if (x0 + 3 < OW && (OW & 3) == 0) {
      *reinterpret_cast<float4*>(prob + o) = make_float4(p0v[0], p0v[1], p0v[2], p0v[3]);
      *reinterpret_cast<float4*>(prob + o + (size_t)OH * OW) = make_float4(p1v[0], p1v[1], p1v[2], p1v[3]);
      *reinterpret_cast<uchar4*>(pred + ((size_t)b * OH + y) * OW + x0) = make_uchar4(fgv[0], fgv[1], fgv[2], fgv[3]);
    } else {
      for (int k = 0; k < 4 && x0 + k < OW; ++k) {
        prob[o + k] = p0v[k];
        prob[o + k + (size_t)OH * OW] = p1v[k];
        pred[((size_t)b * OH + y) * OW + x0 + k] = fgv[k];
      }
    }
  }
  if (fg_sum) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) fgc += __shfl_xor(fgc, o, 64);
    if ((threadIdx.x & 63) == 0) wsum[threadIdx.x >> 6] = fgc;
    __syncthreads();
    if (threadIdx.x == 0) {
      const int t = wsum[0] + wsum[1] + wsum[2] + wsum[3];
      if (t) atomicAdd(&fg_sum[b], t);
    }
  }
}

extern "C" int psam_prob_argmax(const float* logits, int B, int IH, int IW, int OH, int OW, float* prob, void* pred,
                                int* fg_sum, void* stream) {
  if (B <= 0) return PSAM_ERR_ARG;
  hipLaunchKernelGGL(prob_argmax_kernel, dim3((OW + 1023) / 1024, OH, B), dim3(256), 0, (hipStream_t)stream, logits, IH,
                     IW, OH, OW, prob, (uint8_t*)pred, fg_sum);
  return psam_launch_status();
}

__global__ void broadcast_rows_kernel(const float* __restrict__ row, int D, float* __restrict__ out, long long stride,
                                      long long off) {
  float* o = out + (size_t)blockIdx.x * stride + off;
  for (int k = threadIdx.x; k < D; k += blockDim.x) o[k] = row[k];
}
extern "C" int psam_broadcast_rows(const float* row, int D, float* out, int B, long long stride, long long off,
                                   void* stream) {
  if (B <= 0 || D <= 0) return PSAM_ERR_ARG;
  hipLaunchKernelGGL(broadcast_rows_kernel, dim3(B), dim3(256), 0, (hipStream_t)stream, row, D, out, stride, off);
  return psam_launch_status();
}

// ---- image hand-off to SAM ------------------------------------------------------------------------------
// mm[2*b] = min, mm[2*b+1] = max over the whole [3,H,W] image b (ProtoSAM.py:660: min/max over all channels).
// Encoded as order-preserving uint32 so atomicMin/Max work; caller initialises with psam_minmax_init.
__device__ __forceinline__ uint32_t f2ord(float f) {
  uint32_t u = __float_as_uint(f);
  return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__device__ __forceinline__ float ord2f(uint32_t u) {
  return __uint_as_float((u & 0x80000000u) ? (u & 0x7fffffffu) : ~u);
}
__global__ void minmax_init_kernel(uint32_t* mm, int B) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < B) {
    mm[2 * i] = 0xffffffffu;
    mm[2 * i + 1] = 0u;
  }
}
__global__ __launch_bounds__(256) void minmax_kernel(const float* __restrict__ x, size_t n_per_img,
                                                     uint32_t* __restrict__ mm) {
  const int b = blockIdx.y;
  const float* p = x + (size_t)b * n_per_img;
  float lo = INFINITY, hi = -INFINITY;
  const size_t n4 = ((reinterpret_cast<uintptr_t>(p) & 15) == 0) ? n_per_img / 4 : 0;   // 16-byte loads where aligned
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (size_t)gridDim.x * blockDim.x) {
    const float4 v = reinterpret_cast<const float4*>(p)[i];
    lo = fminf(fminf(lo, v.x), fminf(fminf(v.y, v.z), v.w));
    hi = fmaxf(fmaxf(hi, v.x), fmaxf(fmaxf(v.y, v.z), v.w));
  }
  for (size_t i = n4 * 4 + (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n_per_img; i += (size_t)gridDim.x * blockDim.x) {
    float v = p[i];
    lo = fminf(lo, v);
    hi = fmaxf(hi, v);
  }
  lo = wave_min(lo);
  hi = wave_max(hi);
  __shared__ float red[2][4];      // one atomic pair per workgroup (the 32 K per-wave atomics on 2 B addresses were the cost)
  if ((threadIdx.x & 63) == 0) {
    red[0][threadIdx.x >> 6] = lo;
    red[1][threadIdx.x >> 6] = hi;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    lo = fminf(fminf(red[0][0], red[0][1]), fminf(red[0][2], red[0][3]));
    hi = fmaxf(fmaxf(red[1][0], red[1][1]), fmaxf(red[1][2], red[1][3]));
    atomicMin(&mm[2 * b], f2ord(lo));
    atomicMax(&mm[2 * b + 1], f2ord(hi));
  }
}

// img fp32 [B,3,IH,IW] (the 1024^2 bilinear-upsampled query) -> u8 quantise -> normalise -> im2col fp16
// If IH != S the image is first bilinearly resized to SxS (ProtoSAM.py:592-593) on the fly; min/max must then
// have been taken over the RESIZED image (use psam_bilinear_nchw + psam_minmax, or pass resized input).
__global__ void sam_patchify_kernel(const float* __restrict__ img, const uint32_t* __restrict__ mm, int S, int P,
                                    float m0, float m1, float m2, float s0, float s1, float s2, int quantise,
                                    half_t* __restrict__ out, uint8_t* __restrict__ u8out) {
  const int npw = S / P;
  const int patch = blockIdx.x, b = blockIdx.y;
  const int py = patch / npw, px = patch % npw;
  const float lo = ord2f(mm[2 * b]), hi = ord2f(mm[2 * b + 1]);
  const float rng = hi - lo;
  const int K = 3 * P * P;
  half_t* o = out + ((size_t)b * npw * npw + patch) * K;
  for (int k = threadIdx.x; k < K; k += blockDim.x) {
    int c = k / (P * P), r = k % (P * P);
    int y = py * P + r / P, x = px * P + r % P;
    float v = img[(((size_t)b * 3 + c) * S + y) * S + x];
    float q = (v - lo) / rng;
    if (quantise) {
      q = q * 255.0f;
      // numpy astype(uint8) of a float in [0,255]: truncation toward zero
      q = (float)(int)q;
      if (u8out) u8out[(((size_t)b * 3 + c) * S + y) * S + x] = (uint8_t)q;
    }
    const float mean = c == 0 ? m0 : (c == 1 ? m1 : m2);
    const float sd = c == 0 ? s0 : (c == 1 ? s1 : s2);
    o[k] = (half_t)((q - mean) / sd);
  }
}

extern "C" int psam_minmax(const float* x, int B, long long n_per_img, void* mm, void* stream) {
  if (B <= 0 || n_per_img <= 0) return PSAM_ERR_ARG;
  hipStream_t s = (hipStream_t)stream;
  hipLaunchKernelGGL(minmax_init_kernel, dim3((B + 63) / 64), dim3(64), 0, s, (uint32_t*)mm, B);
  int nb = (int)((n_per_img + 256 * 16 - 1) / (256 * 16));
  if (nb > 128) nb = 128;
  hipLaunchKernelGGL(minmax_kernel, dim3(nb, B), dim3(256), 0, s, x, (size_t)n_per_img, (uint32_t*)mm);
  return psam_launch_status();
}

// quantise = 1: ProtoSAM (uint8 image, SAM pixel_mean/std). quantise = 0: ProtoMedSAM ([0,1] float image,
// mean 0 / std 1, models/ProtoMedSAM.py:203-205).
extern "C" int psam_sam_patchify(const float* img, const void* mm, int B, int S, int P, const float* mean3,
                                 const float* std3, int quantise, void* out, void* u8out, void* stream) {
  if (B <= 0 || S % P) return PSAM_ERR_ARG;
  const int np = (S / P) * (S / P);
  hipLaunchKernelGGL(sam_patchify_kernel, dim3(np, B), dim3(256), 0, (hipStream_t)stream, img, (const uint32_t*)mm, S,
                     P, mean3[0], mean3[1], mean3[2], std3[0], std3[1], std3[2], quantise, (half_t*)out,
                     (uint8_t*)u8out);
  return psam_launch_status();
}

// ---- SAM neck: im2col for the 3x3 / pad 1 conv on a token-major map --------------------------------------------
// in fp16 [B, H*W, C] -> out fp16 [B*H*W, 9*C], column (ky*3+kx)*C + c = in[b, (y+ky-1)*W + (x+kx-1), c] or 0.
// (models/segment_anything/modeling/image_encoder.py:98-104: Conv2d(out_chans, out_chans, 3, padding=1, bias=False))
__global__ void im2col3x3_kernel(const half_t* __restrict__ in, int H, int W, int C, half_t* __restrict__ out) {
  const int pix = blockIdx.x, b = blockIdx.y;
  const int y = pix / W, x = pix % W;
  const int cv = C / 8;  // 16-byte chunks per tap
  uint4* o = reinterpret_cast<uint4*>(out + ((size_t)b * H * W + pix) * 9 * C);
  for (int i = threadIdx.x; i < 9 * cv; i += blockDim.x) {
    int tap = i / cv, c = i % cv;
    int yy = y + tap / 3 - 1, xx = x + tap % 3 - 1;
    uint4 v = make_uint4(0, 0, 0, 0);
    if (yy >= 0 && yy < H && xx >= 0 && xx < W)
      v = reinterpret_cast<const uint4*>(in + ((size_t)b * H * W + (size_t)yy * W + xx) * C)[c];
    o[i] = v;
  }
}
extern "C" int psam_im2col3x3(const void* in, int B, int H, int W, int C, void* out, void* stream) {
  if (B <= 0 || (C % 8) != 0) return PSAM_ERR_ARG;
  hipLaunchKernelGGL(im2col3x3_kernel, dim3(H * W, B), dim3(256), 0, (hipStream_t)stream, (const half_t*)in, H, W, C,
                     (half_t*)out);
  return psam_launch_status();
}

// fp32 -> fp16 cast (residual stream into the neck's 1x1-conv GEMM operand), 8 elements per thread
__global__ void cast_f16_kernel(const float* __restrict__ x, half_t* __restrict__ y, size_t n8) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n8) return;
  const float4 a = reinterpret_cast<const float4*>(x)[2 * i], b = reinterpret_cast<const float4*>(x)[2 * i + 1];
  half8_t h = {(half_t)a.x, (half_t)a.y, (half_t)a.z, (half_t)a.w, (half_t)b.x, (half_t)b.y, (half_t)b.z, (half_t)b.w};
  reinterpret_cast<half8_t*>(y)[i] = h;
}
extern "C" int psam_cast_f16(const float* x, void* y, long long n, void* stream) {
  if (n <= 0 || (n & 7)) return PSAM_ERR_ARG;
  size_t n8 = (size_t)n / 8;
  hipLaunchKernelGGL(cast_f16_kernel, dim3((unsigned)((n8 + 255) / 256)), dim3(256), 0, (hipStream_t)stream, x,
                     (half_t*)y, n8);
  return psam_launch_status();
}

// The two element-wise passes of the reference-width mode of the SAM image encoder (ImageEncoderViT.gemm_x3: every Linear of the blocks at
// fp32 accuracy through psam_gemm_f32x3, whose operands and results are fp32): the attention output (fp16) back to fp32 for the
// projection, and nn.GELU (erf form, modeling/common.py:13-26) between lin1 and lin2, in place. HBM-bound, 16 bytes per lane.
__global__ void cast_f32_kernel(const half_t* __restrict__ x, float* __restrict__ y, size_t n8) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n8) return;
  const half8_t h = reinterpret_cast<const half8_t*>(x)[i];
  reinterpret_cast<float4*>(y)[2 * i] = make_float4((float)h[0], (float)h[1], (float)h[2], (float)h[3]);
  reinterpret_cast<float4*>(y)[2 * i + 1] = make_float4((float)h[4], (float)h[5], (float)h[6], (float)h[7]);
}
extern "C" int psam_cast_f32(const void* x, float* y, long long n, void* stream) {
  if (n <= 0 || (n & 7)) return PSAM_ERR_ARG;
  size_t n8 = (size_t)n / 8;
  hipLaunchKernelGGL(cast_f32_kernel, dim3((unsigned)((n8 + 255) / 256)), dim3(256), 0, (hipStream_t)stream, (const half_t*)x, y, n8);
  return psam_launch_status();
}
__global__ void gelu_f32_kernel(float* __restrict__ x, size_t n4) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n4) return;
  float4 v = reinterpret_cast<float4*>(x)[i];
  v.x = 0.5f * v.x * (1.0f + erff(v.x * 0.70710678118654752440f));
  v.y = 0.5f * v.y * (1.0f + erff(v.y * 0.70710678118654752440f));
  v.z = 0.5f * v.z * (1.0f + erff(v.z * 0.70710678118654752440f));
  v.w = 0.5f * v.w * (1.0f + erff(v.w * 0.70710678118654752440f));
  reinterpret_cast<float4*>(x)[i] = v;
}
extern "C" int psam_gelu_f32(float* x, long long n, void* stream) {
  if (n <= 0 || (n & 3)) return PSAM_ERR_ARG;
  size_t n4 = (size_t)n / 4;
  hipLaunchKernelGGL(gelu_f32_kernel, dim3((unsigned)((n4 + 255) / 256)), dim3(256), 0, (hipStream_t)stream, x, n4);
  return psam_launch_status();
}

// fp32 -> (hi, lo) fp16 pair: hi = half(x) (written when `hi_out`, else read: the folded-LayerNorm GEMM already wrote it), lo = half(x -
// float(hi)). A GEMM on [hi | lo] against [W_hi | W_lo] (three products: hi W_hi + lo W_hi + hi W_lo) then carries ~22 mantissa bits of
// both operands: used for the neck of the SAM image encoder, whose fp16-operand GEMMs alone were 4.0e-4 of the 6.4e-4 embedding error
// (image_encoder.py:90-106).
__global__ void split_f16_kernel(const float* __restrict__ x, half_t* __restrict__ hi, half_t* __restrict__ lo, int write_hi, size_t n8) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n8) return;
  const float4 a = reinterpret_cast<const float4*>(x)[2 * i], b = reinterpret_cast<const float4*>(x)[2 * i + 1];
  const float v[8] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w};
  half8_t h, l;
  if (write_hi) {
#pragma unroll
    for (int e = 0; e < 8; ++e) h[e] = (half_t)v[e];
    reinterpret_cast<half8_t*>(hi)[i] = h;
  } else {
    h = reinterpret_cast<const half8_t*>(hi)[i];
  }
#pragma unroll
  for (int e = 0; e < 8; ++e) l[e] = (half_t)(v[e] - (float)h[e]);
  reinterpret_cast<half8_t*>(lo)[i] = l;
}
extern "C" int psam_split_f16(const float* x, void* hi, void* lo, long long n, int write_hi, void* stream) {
  if (n <= 0 || (n & 7) || !x || !hi || !lo) return PSAM_ERR_ARG;
  size_t n8 = (size_t)n / 8;
  hipLaunchKernelGGL(split_f16_kernel, dim3((unsigned)((n8 + 255) / 256)), dim3(256), 0, (hipStream_t)stream, x, (half_t*)hi,
                     (half_t*)lo, write_hi, n8);
  return psam_launch_status();
}

// Sam.preprocess normalisation (modeling/sam.py:163-168): y[b,c,:,:] = (x[b,c,:,:] - mean[c]) / std[c].
// in_u8 = 1: x is uint8 (the predictor's image tensor, predictor.py:57-58), else fp32. 3 channels.
__global__ void normalize_chw_kernel(const void* __restrict__ x, int in_u8, size_t plane, float m0, float m1, float m2,
                                     float s0, float s1, float s2, float* __restrict__ y, size_t total) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= total) return;
  const int c = (int)((i / plane) % 3);
  const float v = in_u8 ? (float)reinterpret_cast<const uint8_t*>(x)[i] : reinterpret_cast<const float*>(x)[i];
  const float mean = c == 0 ? m0 : (c == 1 ? m1 : m2);
  const float sd = c == 0 ? s0 : (c == 1 ? s1 : s2);
  y[i] = (v - mean) / sd;
}
extern "C" int psam_normalize_chw(const void* x, int in_u8, int B, long long plane, const float* mean3,
                                  const float* std3, float* y, void* stream) {
  if (B <= 0 || plane <= 0) return PSAM_ERR_ARG;
  const size_t total = (size_t)B * 3 * (size_t)plane;
  hipLaunchKernelGGL(normalize_chw_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, x,
                     in_u8, (size_t)plane, mean3[0], mean3[1], mean3[2], std3[0], std3[1], std3[2], y, total);
  return psam_launch_status();
}

// Token-major bilinear resize of a feature map: in fp32 [B][ih*iw, C] (batch stride in_bstride, row stride ld) ->
// out fp32 [B][oh*ow, C] contiguous. F.interpolate(img_fts, size=(32,32), mode='bilinear') of
// models/grid_proto_fewshot.py:96-98 (taken when the encoder yields fewer than 32x32 patches) without the NCHW detour.
__global__ void bilinear_tokens_kernel(const float* __restrict__ in, size_t in_bstride, int ld, int ih, int iw, int C,
                                       int oh, int ow, float* __restrict__ out) {
  const int opix = blockIdx.x, b = blockIdx.y;
  const int oy = opix / ow, ox = opix % ow;
  const float sh = (float)ih / (float)oh, sw = (float)iw / (float)ow;
  const Lin ly = lin_src(oy, sh, ih), lx = lin_src(ox, sw, iw);
  const float* base = in + (size_t)b * in_bstride;
  const float* p00 = base + (size_t)(ly.i0 * iw + lx.i0) * ld;
  const float* p01 = base + (size_t)(ly.i0 * iw + lx.i1) * ld;
  const float* p10 = base + (size_t)(ly.i1 * iw + lx.i0) * ld;
  const float* p11 = base + (size_t)(ly.i1 * iw + lx.i1) * ld;
  float* o = out + ((size_t)b * oh * ow + opix) * C;
  for (int c = threadIdx.x; c < C; c += blockDim.x)
    o[c] = ly.l0 * (lx.l0 * p00[c] + lx.l1 * p01[c]) + ly.l1 * (lx.l0 * p10[c] + lx.l1 * p11[c]);
}
extern "C" int psam_bilinear_tokens(const float* in, long long in_bstride, int ld, int B, int ih, int iw, int C, int oh,
                                    int ow, float* out, void* stream) {
  if (B <= 0 || ih <= 0 || iw <= 0 || oh <= 0 || ow <= 0 || C <= 0) return PSAM_ERR_ARG;
  hipLaunchKernelGGL(bilinear_tokens_kernel, dim3(oh * ow, B), dim3(256), 0, (hipStream_t)stream, in, (size_t)in_bstride,
                     ld, ih, iw, C, oh, ow, out);
  return psam_launch_status();
}

// ---- slice hand-off: scan volume -> normalised, resized, z-tiled query / support slices ------------------------------
// Replaces the host chain of dataloaders/ManualAnnoDatasetv2.py:165-187,317-327 (np.float32 -> norm_func -> cv2.resize
// INTER_LINEAR per slice -> repeat(tile_z_dim)) and dataset_utils.py:101-108 (MR_normalize / CT_normalize).
// vol_dtype: 0 int16, 1 float32, 2 uint8, 3 int32 (raw voxels as stored in the NIfTI file, [Z, H, W], x fastest).
__device__ __forceinline__ float vox(const void* __restrict__ v, int dt, size_t i) {
  switch (dt) {
    case 0: return (float)reinterpret_cast<const short*>(v)[i];
    case 1: return reinterpret_cast<const float*>(v)[i];
    case 2: return (float)reinterpret_cast<const unsigned char*>(v)[i];
    default: return (float)reinterpret_cast<const int*>(v)[i];
  }
}

// out[0] = sum(x), out[1] = sum(x^2) over the whole volume in fp64 (x after slope / intercept scaling)
__global__ __launch_bounds__(256) void volume_stats_kernel(const void* __restrict__ vol, int dt, size_t n, float slope,
                                                           float inter, double* __restrict__ out) {
  double s = 0.0, q = 0.0;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
    const double x = (double)(vox(vol, dt, i) * slope + inter);
    s += x;
    q += x * x;
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    s += __shfl_xor(s, o, 64);
    q += __shfl_xor(q, o, 64);
  }
  __shared__ double rs[4], rq[4];
  if ((threadIdx.x & 63) == 0) { rs[threadIdx.x >> 6] = s; rq[threadIdx.x >> 6] = q; }
  __syncthreads();
  if (threadIdx.x == 0) {
    atomicAdd(out + 0, rs[0] + rs[1] + rs[2] + rs[3]);
    atomicAdd(out + 1, rq[0] + rq[1] + rq[2] + rq[3]);
  }
}
extern "C" int psam_volume_stats(const void* vol, int vol_dtype, long long n, float slope, float inter, double* out,
                                 void* stream) {
  if (n <= 0 || vol_dtype < 0 || vol_dtype > 3) return PSAM_ERR_ARG;
  (void)hipMemsetAsync(out, 0, 2 * sizeof(double), (hipStream_t)stream);
  const int blocks = (int)((n + 256 * 16 - 1) / (256 * 16) < 2048 ? (n + 256 * 16 - 1) / (256 * 16) : 2048);
  hipLaunchKernelGGL(volume_stats_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, vol, vol_dtype, (size_t)n, slope,
                     inter, out);
  return psam_launch_status();
}

// out [Z, tile, S, S] fp32 = tile copies of resize_linear((x*slope + inter - mean) * inv_std) with cv2.INTER_LINEAR's
// float rule: source coordinate (d + 0.5) * in/out - 0.5, floor, clamp (weight collapses onto the edge pixel), horizontal
// pass then vertical pass. mode 1 = cv2.INTER_NEAREST for label volumes (floor(d * in/out), no normalisation).
__global__ __launch_bounds__(256) void volume_slices_kernel(const void* __restrict__ vol, int dt, int H, int W, float slope,
                                                            float inter, float mean, float inv_std, int S, int tile,
                                                            int mode, float* __restrict__ out) {
  const int x = blockIdx.x * 256 + threadIdx.x, y = blockIdx.y, z = blockIdx.z;
  if (x >= S) return;
  const size_t base = (size_t)z * H * W;
  float v;
  if (mode == 1) {
    int sx = (int)floor((double)x * ((double)W / (double)S)), sy = (int)floor((double)y * ((double)H / (double)S));
    sx = sx < W - 1 ? sx : W - 1;
    sy = sy < H - 1 ? sy : H - 1;
    v = vox(vol, dt, base + (size_t)sy * W + sx) * slope + inter;
  } else {
    const double scx = (double)W / (double)S, scy = (double)H / (double)S;
    float fx = (float)(((double)x + 0.5) * scx - 0.5), fy = (float)(((double)y + 0.5) * scy - 0.5);
    int sx = (int)floorf(fx), sy = (int)floorf(fy);
    fx -= (float)sx;
    fy -= (float)sy;
    int sx1 = sx + 1, sy1 = sy + 1;
    if (sx < 0) { sx = 0; sx1 = 0; fx = 0.f; }
    if (sx >= W - 1) { sx = W - 1; sx1 = W - 1; fx = 0.f; }
    if (sy < 0) { sy = 0; sy1 = 0; fy = 0.f; }
    if (sy >= H - 1) { sy = H - 1; sy1 = H - 1; fy = 0.f; }
    auto nv = [&](int yy, int xx) { return ((vox(vol, dt, base + (size_t)yy * W + xx) * slope + inter) - mean) * inv_std; };
    const float r0 = nv(sy, sx) * (1.f - fx) + nv(sy, sx1) * fx;
    const float r1 = nv(sy1, sx) * (1.f - fx) + nv(sy1, sx1) * fx;
    v = r0 * (1.f - fy) + r1 * fy;
  }
  for (int c = 0; c < tile; ++c) out[(((size_t)z * tile + c) * S + y) * S + x] = v;
}
extern "C" int psam_volume_slices(const void* vol, int vol_dtype, int Z, int H, int W, float slope, float inter, float mean,
                                  float inv_std, int S, int tile, int mode, float* out, void* stream) {
  if (Z <= 0 || H <= 0 || W <= 0 || S <= 0 || tile <= 0 || vol_dtype < 0 || vol_dtype > 3 || mode < 0 || mode > 1)
    return PSAM_ERR_ARG;
  hipLaunchKernelGGL(volume_slices_kernel, dim3((S + 255) / 256, S, Z), dim3(256), 0, (hipStream_t)stream, vol, vol_dtype, H,
                     W, slope, inter, mean, inv_std, S, tile, mode, out);
  return psam_launch_status();
}

// ---- convolution front-end for the ResNet-101 encoder (models/backbone/torchvision_backbones.py:12-52) -----------------
// General im2col on token-major (NHWC) half maps: in [B, H*W, C] -> out [B*Ho*Wo, ldo] with column (ky*kw + kx)*C + c =
// in[b, (y*stride - pad + ky*dil), (x*stride - pad + kx*dil), c] or 0 outside; columns kh*kw*C .. ldo-1 are zero (K padding
// for the GEMM). C % 8 == 0. The 1x1 / stride-2 "downsample" conv is the same kernel with kh = kw = 1.
__global__ void im2col_kernel(const half_t* __restrict__ in, int H, int W, int C, int kh, int kw, int stride, int dil,
                              int pad, int Ho, int Wo, int ldo, half_t* __restrict__ out) {
  const int pix = blockIdx.x, b = blockIdx.y;
  const int y = pix / Wo, x = pix % Wo;
  const int cv = C / 8, taps = kh * kw, nv = ldo / 8;
  uint4* o = reinterpret_cast<uint4*>(out + ((size_t)b * Ho * Wo + pix) * ldo);
  for (int i = threadIdx.x; i < nv; i += blockDim.x) {
    uint4 v = make_uint4(0, 0, 0, 0);
    const int tap = i / cv, c = i % cv;
    if (tap < taps) {
      const int yy = y * stride - pad + (tap / kw) * dil, xx = x * stride - pad + (tap % kw) * dil;
      if (yy >= 0 && yy < H && xx >= 0 && xx < W)
        v = reinterpret_cast<const uint4*>(in + ((size_t)b * H * W + (size_t)yy * W + xx) * C)[c];
    }
    o[i] = v;
  }
}
extern "C" int psam_im2col(const void* in, int B, int H, int W, int C, int kh, int kw, int stride, int dil, int pad, int ldo,
                           void* out, void* stream) {
  if (B <= 0 || (C % 8) != 0 || (ldo % 8) != 0 || ldo < kh * kw * C || stride <= 0 || dil <= 0) return PSAM_ERR_ARG;
  const int Ho = (H + 2 * pad - dil * (kh - 1) - 1) / stride + 1, Wo = (W + 2 * pad - dil * (kw - 1) - 1) / stride + 1;
  if (Ho <= 0 || Wo <= 0) return PSAM_ERR_ARG;
  hipLaunchKernelGGL(im2col_kernel, dim3(Ho * Wo, B), dim3(256), 0, (hipStream_t)stream, (const half_t*)in, H, W, C, kh, kw,
                     stride, dil, pad, Ho, Wo, ldo, (half_t*)out);
  return psam_launch_status();
}

// Stem: im2col of the 7x7 / stride 2 / pad 3 conv straight from the fp32 NCHW image: out [B*Ho*Wo, ldo] half with
// column (c*7 + ky)*7 + kx (the weight's own [Cout, Cin, 7, 7] order), zero beyond 147.
__global__ void im2col_stem_kernel(const float* __restrict__ img, int H, int W, int Ho, int Wo, int ldo,
                                   half_t* __restrict__ out) {
  const int pix = blockIdx.x, b = blockIdx.y;
  const int y = pix / Wo, x = pix % Wo;
  half_t* o = out + ((size_t)b * Ho * Wo + pix) * ldo;
  for (int i = threadIdx.x; i < ldo; i += blockDim.x) {
    float v = 0.f;
    if (i < 147) {
      const int c = i / 49, ky = (i % 49) / 7, kx = i % 7;
      const int yy = y * 2 - 3 + ky, xx = x * 2 - 3 + kx;
      if (yy >= 0 && yy < H && xx >= 0 && xx < W) v = img[(((size_t)b * 3 + c) * H + yy) * W + xx];
    }
    o[i] = (half_t)v;
  }
}
extern "C" int psam_im2col_stem(const float* img, int B, int H, int W, int ldo, void* out, void* stream) {
  if (B <= 0 || ldo < 147 || (ldo % 8) != 0) return PSAM_ERR_ARG;
  const int Ho = (H + 6 - 7) / 2 + 1, Wo = (W + 6 - 7) / 2 + 1;
  hipLaunchKernelGGL(im2col_stem_kernel, dim3(Ho * Wo, B), dim3(64), 0, (hipStream_t)stream, img, H, W, Ho, Wo, ldo,
                     (half_t*)out);
  return psam_launch_status();
}

// MaxPool2d(3, stride 2, padding 1) on a token-major half map [B, H*W, C] -> [B, Ho*Wo, C]
__global__ void maxpool3x3s2_kernel(const half_t* __restrict__ in, int H, int W, int C, int Ho, int Wo,
                                    half_t* __restrict__ out) {
  const int pix = blockIdx.x, b = blockIdx.y;
  const int y = pix / Wo, x = pix % Wo;
  for (int c = threadIdx.x; c < C; c += blockDim.x) {
    float m = -INFINITY;
    for (int ky = 0; ky < 3; ++ky)
      for (int kx = 0; kx < 3; ++kx) {
        const int yy = 2 * y - 1 + ky, xx = 2 * x - 1 + kx;
        if (yy >= 0 && yy < H && xx >= 0 && xx < W) m = fmaxf(m, (float)in[((size_t)b * H * W + (size_t)yy * W + xx) * C + c]);
      }
    out[((size_t)b * Ho * Wo + pix) * C + c] = (half_t)m;
  }
}
extern "C" int psam_maxpool3x3s2(const void* in, int B, int H, int W, int C, void* out, void* stream) {
  if (B <= 0 || C <= 0) return PSAM_ERR_ARG;
  const int Ho = (H + 2 - 3) / 2 + 1, Wo = (W + 2 - 3) / 2 + 1;
  hipLaunchKernelGGL(maxpool3x3s2_kernel, dim3(Ho * Wo, B), dim3(128), 0, (hipStream_t)stream, (const half_t*)in, H, W, C,
                     Ho, Wo, (half_t*)out);
  return psam_launch_status();
}

// =====================================================================================================
// Rotation test-time augmentation (ProtoSAM.forward(..., degrees_rotate != 0), models/ProtoSAM.py:544-556 through
// util/utils.py:40-83). The reference calls torchvision 0.15.2's tensor `rotate` (affine grid + grid_sample, NEAREST,
// zero padding, align_corners = False) and `resize(..., BILINEAR, antialias = True)` (aten `_upsample_bilinear2d_aa`);
// torchvision is absent from /root/reference, so both are restated from their published algorithms (oracle/rotate.py).
//
// psam_rotate_nearest: planes [C, H, W] fp32 -> [C, outH, outW]: output pixel (y, x) samples the source at the affine-grid
// point built from xg[x + crop_x], yg[y + crop_y] (the host passes torchvision's base-grid `linspace` values, so the
// expanded canvas and `reverse_tensor`'s centre crop are both expressed by crop offsets) and the 3x2 `rescaled_theta`
// rt (row-major), un-normalised as grid_sample does and rounded half-to-even.
__global__ void rotate_nearest_kernel(const float* __restrict__ src, float* __restrict__ dst, const float* __restrict__ xg,
                                      const float* __restrict__ yg, float r00, float r01, float r10, float r11, float r20,
                                      float r21, int H, int W, int crop_y, int crop_x, int outH, int outW) {
  const int x = blockIdx.x * blockDim.x + threadIdx.x, y = blockIdx.y, c = blockIdx.z;
  if (x >= outW) return;
  const float bx = xg[x + crop_x], by = yg[y + crop_y];
  // base_grid[p, :] . rescaled_theta[:, k] over (x, y, 1)
  const float gx = __fadd_rn(__fadd_rn(__fmul_rn(bx, r00), __fmul_rn(by, r10)), r20);
  const float gy = __fadd_rn(__fadd_rn(__fmul_rn(bx, r01), __fmul_rn(by, r11)), r21);
  const float ix = __fdiv_rn(__fadd_rn(__fmul_rn(__fadd_rn(gx, 1.f), (float)W), -1.f), 2.f);
  const float iy = __fdiv_rn(__fadd_rn(__fmul_rn(__fadd_rn(gy, 1.f), (float)H), -1.f), 2.f);
  const float fx = nearbyintf(ix), fy = nearbyintf(iy);
  float v = 0.f;
  if (fx >= 0.f && fx <= (float)(W - 1) && fy >= 0.f && fy <= (float)(H - 1))
    v = src[((size_t)c * H + (int)fy) * W + (int)fx];
  dst[((size_t)c * outH + y) * outW + x] = v;
}
extern "C" int psam_rotate_nearest(const float* src, float* dst, const float* xg, const float* yg, const float* rt6, int C,
                                   int H, int W, int crop_y, int crop_x, int outH, int outW, void* stream) {
  if (C <= 0 || H <= 0 || W <= 0 || outH <= 0 || outW <= 0 || crop_y < 0 || crop_x < 0 || !rt6) return PSAM_ERR_ARG;
  hipLaunchKernelGGL(rotate_nearest_kernel, dim3((outW + 255) / 256, outH, C), dim3(256), 0, (hipStream_t)stream, src, dst,
                     xg, yg, rt6[0], rt6[1], rt6[2], rt6[3], rt6[4], rt6[5], H, W, crop_y, crop_x, outH, outW);
  return psam_launch_status();
}

// psam_resize_aa: anti-aliased bilinear resize of fp32 planes [C, H, W] -> [C, OH, OW], separable, width first with an
// fp32 intermediate [C, H, OW] (aten UpSampleKernel.cpp, `_compute_indices_min_size_weights_aa` with the triangle filter):
// scale = in / out, support = max(scale, 1), centre = scale * (i + 0.5), taps [int(centre - support + 0.5), int(centre +
// support + 0.5)) clipped to the input, weights triangle((j - centre + 0.5) / max(scale, 1)) normalised to sum 1.
__global__ void resize_aa_pass_kernel(const float* __restrict__ src, float* __restrict__ dst, int in_len, int out_len,
                                      int in_stride, int out_stride, int lines, int line_in_stride, int line_out_stride,
                                      size_t plane_in, size_t plane_out) {
  // one thread per (output index i along the resized axis, line l along the other axis); blockIdx.z = plane
  const int i = blockIdx.x * blockDim.x + threadIdx.x, l = blockIdx.y;
  if (i >= out_len || l >= lines) return;
  const float scale = (float)in_len / (float)out_len;
  const float support = scale >= 1.f ? scale : 1.f;
  const float invscale = scale >= 1.f ? 1.f / scale : 1.f;
  const float center = scale * ((float)i + 0.5f);
  int xmin = (int)(center - support + 0.5f);
  xmin = xmin > 0 ? xmin : 0;
  int xmax = (int)(center + support + 0.5f);
  xmax = xmax < in_len ? xmax : in_len;
  const int xsize = xmax - xmin;
  const float* s = src + blockIdx.z * plane_in + (size_t)l * line_in_stride;
  float total = 0.f;
  for (int j = 0; j < xsize; ++j) {
    float a = fabsf(((float)(j + xmin) - center + 0.5f) * invscale);
    total += a < 1.f ? 1.f - a : 0.f;
  }
  float acc = 0.f;
  for (int j = 0; j < xsize; ++j) {
    float a = fabsf(((float)(j + xmin) - center + 0.5f) * invscale);
    float w = a < 1.f ? 1.f - a : 0.f;
    if (total != 0.f) w /= total;
    acc += w * s[(size_t)(j + xmin) * in_stride];
  }
  dst[blockIdx.z * plane_out + (size_t)l * line_out_stride + (size_t)i * out_stride] = acc;
}
extern "C" int psam_resize_aa(const float* src, float* tmp, float* dst, int C, int H, int W, int OH, int OW, void* stream) {
  if (C <= 0 || H <= 0 || W <= 0 || OH <= 0 || OW <= 0 || !tmp) return PSAM_ERR_ARG;
  hipStream_t s = (hipStream_t)stream;
  // width: lines = rows of the input
  hipLaunchKernelGGL(resize_aa_pass_kernel, dim3((OW + 127) / 128, H, C), dim3(128), 0, s, src, tmp, W, OW, 1, 1, H, W, OW,
                     (size_t)H * W, (size_t)H * OW);
  // height: lines = columns of the intermediate
  hipLaunchKernelGGL(resize_aa_pass_kernel, dim3((OH + 127) / 128, OW, C), dim3(128), 0, s, tmp, dst, H, OH, OW, OW, OW, 1, 1,
                     (size_t)H * OW, (size_t)OH * OW);
  return psam_launch_status();
}
