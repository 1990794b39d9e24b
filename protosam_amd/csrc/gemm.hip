// fp16-operand / fp32-accumulate NT GEMM on MFMA (v_mfma_f32_32x32x16_f16), gfx950.
//
//   out[M,N] = epilogue( A[M,K] . W[N,K]^T + bias[N] )
//
// This is the one contraction engine behind every nn.Linear / 1x1-conv / patch-embed /
// ConvTranspose(2,2,s2) on the hot path:
//   * DINOv2 `Attention.qkv/proj`, `Mlp.fc1/fc2`, `PatchEmbed.proj`  (external hub model; call
//     sites /root/reference/models/grid_proto_fewshot.py:88-91)
//   * SAM `Attention.qkv/proj` (models/segment_anything/modeling/image_encoder.py:223-249),
//     `MLPBlock.lin1/lin2` (modeling/common.py:13-26), `PatchEmbed.proj` (image_encoder.py:375-406),
//     neck 1x1 conv (image_encoder.py:90-96)
//   * mask-decoder image-side projections (modeling/transformer.py:218-240)
//
// Layout: A and W are K-contiguous (nn.Linear's [out,in] weight is used as-is, no transpose).
// Tile 128x128x64, 256 threads = 4 waves in 2x2, each wave owns a 64x64 sub-tile =
// 2x2 MFMA 32x32 accumulators (64 acc VGPRs). A/W tiles go global -> LDS by direct 16-byte DMA
// (`global_load_lds_dwordx4`, no VGPR staging), double-buffered: the DMA of k-tile t+1 is issued
// before the MFMAs of k-tile t, one barrier per k-tile. The 16-byte-slot XOR swizzle
// (slot ^= (row>>1)&7) that makes the ds_read_b128 fragment reads bank-conflict-free is applied on
// the DMA's per-lane SOURCE address, because the LDS side of the DMA is lane-linear.
#include "common.h"

enum { EPI_F16 = 0, EPI_GELU_F16 = 1, EPI_F32 = 2 };

struct GemmArgs {
  const half_t* A;
  const half_t* W;
  const float* bias;   // [N] or null
  void* out;           // half or float [*, ldo]
  const float* resid;  // float [*, ldr] or null (EPI_F32 only)
  const float* gamma;  // [N] or null (EPI_F32 only): out = resid + gamma * (acc + bias)
  int M, N, K;
  int lda, ldw, ldo, ldr;
  int resid_mod;       // resid row = resid_mod ? m % resid_mod : m
  int out_seg;         // out row = out_seg ? (m / out_seg) * out_seg_stride + out_seg_off + m % out_seg : m
  int out_seg_stride;
  int out_seg_off;
};

#define BM 128
#define BN 128
#define BK 64

__device__ __forceinline__ int lds_off(int row, int chunk) {
  // half-element offset inside a [128][64] fp16 tile, 16-byte slots XOR-swizzled
  return row * BK + ((chunk ^ ((row >> 1) & 7)) << 3);
}

// one 16-byte global -> LDS DMA per lane: LDS destination = wave-uniform base + lane*16 (the hardware adds the lane
// offset), source address per lane. The XOR swizzle therefore lives on the SOURCE side (which 16-byte chunk of its
// 128-byte row a lane fetches) and on the fragment reads; the LDS image itself is written linearly.
__device__ __forceinline__ void glds16(const half_t* g, half_t* lds_wave_base) {
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g,
                                   (__attribute__((address_space(3))) void*)lds_wave_base, 16, 0, 0);
}

template <int EPI>
__global__ __launch_bounds__(256) void gemm_f16_kernel(GemmArgs p) {
  __shared__ __attribute__((aligned(16))) half_t smem[2][2][BM * BK];  // [buf][A|W] 64 KiB

  const int ntn = p.N / BN;
  const int ntm = (p.M + BM - 1) / BM;
  const int tile = xcd_remap(blockIdx.x, ntm * ntn);
  const int m0 = (tile / ntn) * BM;
  const int n0 = (tile % ntn) * BN;

  const int t = threadIdx.x;
  const int lane = t & 63;
  const int wv = __builtin_amdgcn_readfirstlane(t >> 6);
  const int wm = wv >> 1, wn = wv & 1;
  const int lr = lane & 31, lg = lane >> 5;

  // staging: wave wv moves 1-KiB pieces q = wv*4 + j (8 tile rows each) of the A tile and of the W tile.
  // lane -> (row = q*8 + lane/8, LDS slot = lane%8) fetches source chunk = slot ^ ((row>>1)&7) of that row.
  const half_t* ag[4];
  const half_t* wg[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int row = (wv * 4 + j) * 8 + (lane >> 3);
    const int chunk = (lane & 7) ^ ((row >> 1) & 7);
    int am = m0 + row;
    am = am < p.M ? am : p.M - 1;
    ag[j] = p.A + (size_t)am * p.lda + chunk * 8;
    wg[j] = p.W + (size_t)(n0 + row) * p.ldw + chunk * 8;
  }

  f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  auto stage = [&](int buf, int kt) {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      glds16(ag[j] + kt * BK, &smem[buf][0][(wv * 4 + j) * 8 * BK]);
      glds16(wg[j] + kt * BK, &smem[buf][1][(wv * 4 + j) * 8 * BK]);
    }
  };

  const int nk = p.K / BK;
  stage(0, 0);
  __syncthreads();  // carries the vmcnt(0) for the pending LDS-DMA

  for (int kt = 0; kt < nk; ++kt) {
    const int cur = kt & 1;
    if (kt + 1 < nk) stage(cur ^ 1, kt + 1);  // next tile's DMA flies under this tile's MFMAs
    const half_t* sa = smem[cur][0];
    const half_t* sw = smem[cur][1];
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      half8_t fa[2], fb[2];
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        int row = wm * 64 + i * 32 + lr;
        fa[i] = *reinterpret_cast<const half8_t*>(&sa[lds_off(row, s * 2 + lg)]);
      }
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        int row = wn * 64 + j * 32 + lr;
        fb[j] = *reinterpret_cast<const half8_t*>(&sw[lds_off(row, s * 2 + lg)]);
      }
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa[i], fb[j], acc[i][j], 0, 0, 0);
    }
    __syncthreads();  // all waves done reading buf[cur]; DMA into buf[cur^1] has landed (vmcnt(0))
  }

  // epilogue. C layout of 32x32 MFMA: col = lane&31, row = (r&3) + 8*(r>>2) + 4*(lane>>5)
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int col = n0 + wn * 64 + j * 32 + lr;
    const float bv = p.bias ? p.bias[col] : 0.f;
    float gv = 1.f;
    if (EPI == EPI_F32) gv = p.gamma ? p.gamma[col] : 1.f;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int m = m0 + wm * 64 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lg;
        if (m < p.M) {
          float v = acc[i][j][r] + bv;
          size_t orow = p.out_seg ? (size_t)(m / p.out_seg) * p.out_seg_stride + p.out_seg_off + (m % p.out_seg)
                                  : (size_t)m;
          if (EPI == EPI_F16) {
            reinterpret_cast<half_t*>(p.out)[orow * p.ldo + col] = (half_t)v;
          } else if (EPI == EPI_GELU_F16) {
            reinterpret_cast<half_t*>(p.out)[orow * p.ldo + col] = (half_t)gelu_erf(v);
          } else {
            v *= gv;
            if (p.resid) {
              size_t rrow = p.resid_mod ? (size_t)(m % p.resid_mod) : (size_t)m;
              v += p.resid[rrow * p.ldr + col];
            }
            reinterpret_cast<float*>(p.out)[orow * p.ldo + col] = v;
          }
        }
      }
    }
  }
}

extern "C" int psam_gemm_f16(const void* A, const void* W, const float* bias, void* out, const float* resid,
                             const float* gamma, int M, int N, int K, int lda, int ldw, int ldo, int ldr,
                             int resid_mod, int out_seg, int out_seg_stride, int out_seg_off, int epilogue,
                             void* stream) {
  if (M <= 0 || N <= 0 || K <= 0 || (N % BN) != 0 || (K % BK) != 0 || (lda % 8) != 0 || (ldw % 8) != 0)
    return PSAM_ERR_ARG;
  if (epilogue < 0 || epilogue > 2) return PSAM_ERR_ARG;
  GemmArgs p;
  p.A = (const half_t*)A;
  p.W = (const half_t*)W;
  p.bias = bias;
  p.out = out;
  p.resid = resid;
  p.gamma = gamma;
  p.M = M;
  p.N = N;
  p.K = K;
  p.lda = lda;
  p.ldw = ldw;
  p.ldo = ldo;
  p.ldr = ldr;
  p.resid_mod = resid_mod;
  p.out_seg = out_seg;
  p.out_seg_stride = out_seg_stride;
  p.out_seg_off = out_seg_off;
  const int ntm = (M + BM - 1) / BM, ntn = N / BN;
  dim3 grid(ntm * ntn), block(256);
  hipStream_t s = (hipStream_t)stream;
  switch (epilogue) {
    case EPI_F16: hipLaunchKernelGGL(gemm_f16_kernel<EPI_F16>, grid, block, 0, s, p); break;
    case EPI_GELU_F16: hipLaunchKernelGGL(gemm_f16_kernel<EPI_GELU_F16>, grid, block, 0, s, p); break;
    default: hipLaunchKernelGGL(gemm_f16_kernel<EPI_F32>, grid, block, 0, s, p); break;
  }
  return psam_launch_status();
}
