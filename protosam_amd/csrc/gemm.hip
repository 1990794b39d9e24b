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
#include "gemm_asm_meta.h"   // generated: PSAM_ASM2_E_* (gemm_asm_gen.py --meta)
#include <stdlib.h>
#include <type_traits>
#include <map>
#include <string>
#include <vector>

enum { EPI_F16 = 0, EPI_GELU_F16 = 1, EPI_F32 = 2, EPI_RELU_F16 = 3 };

struct GemmArgs {
  const half_t* A;
  const half_t* W;
  const float* bias;   // [N] or null
  void* out;           // half or float [*, ldo]
  const float* resid;  // EPI_F32: float [*, ldr] or null; EPI_RELU_F16: HALF [*, ldr] or null (out = relu(acc + bias + resid))
  const float* gamma;  // [N] or null (EPI_F32 only): out = resid + gamma * (acc + bias)
  int M, N, K;
  int lda, ldw, ldo, ldr;
  int resid_mod;       // resid row = resid_mod ? m % resid_mod : m
  int out_seg;         // out row = out_seg ? (m / out_seg) * out_seg_stride + out_seg_off + m % out_seg : m
  int out_seg_stride;
  int out_seg_off;
  int map_mode;  // 0: XCD-region tile map (default); 1: identity; 2: contiguous chunk per XCD
  int head_hd;   // > 0: head-major output, out[(n / head_hd), m, n % head_hd] (planes of [M, head_hd]); EPI_F16 only
  unsigned long long* trace;  // PSAM_GEMM_TRACE: per-workgroup timestamps (tile 10, debugging)
  int wide16;    // fp16 output rows may be stored with 16-byte instructions (ldo % 8 == 0, 16-byte aligned base)
  int stagger;   // start-time spread of the first round of workgroups, in units of s_sleep(8) (tile 10)
  int dbg;       // ablation switches for tools/gemm_ablate.py (PSAM_GEMM_DBG; only the DBG instantiation reads it)
  // ---- LayerNorm folded into the GEMMs either side of it (psam_gemm_f16_ln; tiles 1 / 11 / 14 only) ----
  // producer (EPI_F32, the residual-stream update x = resid + gamma * (acc + bias)): besides x it writes
  //   out16[m, n]                 = fp16(x)   - the NEXT GEMM's A operand (no LayerNorm pass, no cast pass)
  //   stats[(m * parts + n/64)*2] = (sum, sum of squares) of x over the 64 columns [n/64*64, +64) of row m, parts = N / 64
  // consumer (EPI_F16 / EPI_GELU_F16, W already multiplied by the LayerNorm weight): with (mean, rstd) = ln_mr[m] and
  //   s[n] = sum_k W'[n,k]:   out = act( rstd * (acc - mean * s[n]) + bias'[n] ),  bias' = bias + W . ln_bias   (the caller's `bias`)
  half_t* out16;
  int ld16;
  float* stats;
  const float* ln_mr;
  const float* ln_s;
  // ---- split-K (tile 11, EPI_F32): `ksplit` workgroups share one 256x256 tile, each over a contiguous range of K-tiles, and
  // store their raw partial sums to plane `range` of `ks_ws` (fp32 [ksplit][M][N]); splitk_reduce_kernel finishes the epilogue
  int ksplit;
  float* ks_ws;
};


// ---- device geometry, read once from the runtime (a full MI355X reports 256 CUs in 8 XCDs; a CPX / NPS partition fewer) ----
// The XCD-region tile maps below are written for 8 XCDs (block b on XCD b % 8, observed placement, speed only): on any other
// geometry the launchers fall back to the identity map, and the persistent grids / fill tests use the real CU count.
static int g_num_cus = 0, g_num_xcds = 0;
static void device_geometry() {
  if (g_num_cus > 0) return;
  int dev = 0, v = 0;
  (void)hipGetDevice(&dev);
  g_num_cus = (hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && v > 0) ? v : 256;
  g_num_xcds = (hipDeviceGetAttribute(&v, hipDeviceAttributeNumberOfXccs, dev) == hipSuccess && v > 0) ? v : 8;
  (void)hipGetLastError();
}
static inline int num_cus() { device_geometry(); return g_num_cus; }
static inline bool xcd_maps_apply() { device_geometry(); return g_num_xcds == 8 && g_num_cus % 8 == 0; }

// ---- tile -> workgroup mapping ------------------------------------------------------------------------------------
// Block b runs on XCD b % 8 (observed placement; used for speed only, never for correctness) and every XCD has a
// private 4 MiB L2. The tiles an XCD works on CONCURRENTLY (32 CUs x blocks/CU) should therefore share as many A row
// panels and W column panels as possible: the 8 XCDs are laid out as an xm x xn grid over the tile space (the split
// that minimises rows + cols per region) and each XCD walks its region in 4-row strips, column by column, so any 32
// consecutive tiles of an XCD form a ~4 x 8 patch (12 operand panels instead of 33 for a row-major chunk).
// Measured on 8192^3 (256x256 tiles): DMA-only time 1213 us (contiguous chunk per XCD) -> 500 us (2-D spread).
__host__ __device__ __forceinline__ bool tile_map(int bid, int ntm, int ntn, int mode, int& tm, int& tn) {
  if (mode == 1) {
    if (bid >= ntm * ntn) return false;
    tm = bid / ntn;
    tn = bid % ntn;
    return true;
  }
  if (mode == 2) {
    if (bid >= ntm * ntn) return false;
    const int t = xcd_remap(bid, ntm * ntn);
    tm = t / ntn;
    tn = t % ntn;
    return true;
  }
  const int xcd = bid & 7, idx = bid >> 3;
  int xm = 8, xn = 1, best = (ntm + 7) / 8 + ntn;
  {
    int c = (ntm + 3) / 4 + (ntn + 1) / 2;
    if (c < best) { best = c; xm = 4; xn = 2; }
    c = (ntm + 1) / 2 + (ntn + 3) / 4;
    if (c < best) { best = c; xm = 2; xn = 4; }
    c = ntm + (ntn + 7) / 8;
    if (c < best) { best = c; xm = 1; xn = 8; }
  }
  // proportional (balanced) split: region rx owns rows [rx*ntm/xm, (rx+1)*ntm/xm)
  const int rx = xcd / xn, cx = xcd % xn;
  const int r0 = (rx * ntm) / xm, c0 = (cx * ntn) / xn;
  const int nr = ((rx + 1) * ntm) / xm - r0, nc = ((cx + 1) * ntn) / xn - c0;
  if (nr <= 0 || nc <= 0 || idx >= nr * nc) return false;
  const int strip = idx / (4 * nc), rem = idx - strip * 4 * nc;
  const int rows = (nr - strip * 4) < 4 ? (nr - strip * 4) : 4;
  tm = r0 + strip * 4 + rem % rows;
  tn = c0 + rem / rows;
  return true;
}
// Largest number of tiles any XCD region of the 2-D map (mode 0) holds (host side). One workgroup per CU and 32 CUs per
// XCD: a region of 33 tiles costs two rounds where a balanced split costs one (85 x 3 tiles: five regions of 11 x 3).
static int tile_map_max_region(int ntm, int ntn) {
  int xm = 8, xn = 1, best = (ntm + 7) / 8 + ntn;
  int c = (ntm + 3) / 4 + (ntn + 1) / 2;
  if (c < best) { best = c; xm = 4; xn = 2; }
  c = (ntm + 1) / 2 + (ntn + 3) / 4;
  if (c < best) { best = c; xm = 2; xn = 4; }
  c = ntm + (ntn + 7) / 8;
  if (c < best) { best = c; xm = 1; xn = 8; }
  int mx = 0;
  for (int x = 0; x < 8; ++x) {
    const int rx = x / xn, cx = x % xn;
    const int nr = ((rx + 1) * ntm) / xm - (rx * ntm) / xm, nc = ((cx + 1) * ntn) / xn - (cx * ntn) / xn;
    mx = nr * nc > mx ? nr * nc : mx;
  }
  return mx;
}
// map for the one-workgroup-per-CU kernels: the 2-D region map unless its fullest region needs more rounds than a balanced
// split of the tiles (then contiguous, balanced chunks per XCD: mode 2). PSAM_GEMM_MAP overrides.
static int pick_map_mode(int ntm, int ntn) {
  static int forced = -2;
  if (forced == -2) { const char* e = getenv("PSAM_GEMM_MAP"); forced = e ? atoi(e) : -1; }
  if (forced >= 0) return forced;
  if (!xcd_maps_apply()) return 1;   // not the 8-XCD geometry the region maps are laid out for: identity
  const int total = ntm * ntn, ncu = num_cus(), per_xcd = ncu / 8;
  const int rounds_region = (tile_map_max_region(ntm, ntn) + per_xcd - 1) / per_xcd, rounds_even = (total + ncu - 1) / ncu;
  return rounds_region > rounds_even ? 2 : 0;
}
// grid size that covers every region of tile_map (host side)
static int tile_map_grid(int ntm, int ntn, int mode) {
  if (mode != 0) return ntm * ntn;
  int xm = 8, xn = 1, best = (ntm + 7) / 8 + ntn;
  int c = (ntm + 3) / 4 + (ntn + 1) / 2;
  if (c < best) { best = c; xm = 4; xn = 2; }
  c = (ntm + 1) / 2 + (ntn + 3) / 4;
  if (c < best) { best = c; xm = 2; xn = 4; }
  c = ntm + (ntn + 7) / 8;
  if (c < best) { best = c; xm = 1; xn = 8; }
  const int R = (ntm + xm - 1) / xm, Cc = (ntn + xn - 1) / xn;
  return 8 * R * Cc;
}

// Epilogue for one 32x32 accumulator computed as D^T = W_frag . X_frag^T (weights as the MFMA A operand): lane holds
// output row m = mbase + (lane & 31) and, per register group q, four CONSECUTIVE columns n = nbase + 8q + 4*(lane>>5)
// + 0..3, so bias / LayerScale / residual are float4 loads and the result is one 8-byte (fp16) or 16-byte (fp32) store.
template <int EPI>
__device__ __forceinline__ void store_acc32(const f32x16& acc, int m, int nb, const GemmArgs& p) {
  if (m >= p.M) return;
  const size_t orow = p.out_seg ? (size_t)(m / p.out_seg) * p.out_seg_stride + p.out_seg_off + (m % p.out_seg) : (size_t)m;
  const size_t rrow = p.resid_mod ? (size_t)(m % p.resid_mod) : (size_t)m;
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const int n = nb + 8 * q;
    float4 v = make_float4(acc[4 * q], acc[4 * q + 1], acc[4 * q + 2], acc[4 * q + 3]);
    if (p.bias) {
      const float4 b = *reinterpret_cast<const float4*>(p.bias + n);
      v.x += b.x; v.y += b.y; v.z += b.z; v.w += b.w;
    }
    if (EPI == EPI_F16) {
      half4_t h = {(half_t)v.x, (half_t)v.y, (half_t)v.z, (half_t)v.w};
      *reinterpret_cast<half4_t*>(reinterpret_cast<half_t*>(p.out) + orow * p.ldo + n) = h;
    } else if (EPI == EPI_GELU_F16) {
      half4_t h = {(half_t)gelu_erf(v.x), (half_t)gelu_erf(v.y), (half_t)gelu_erf(v.z), (half_t)gelu_erf(v.w)};
      *reinterpret_cast<half4_t*>(reinterpret_cast<half_t*>(p.out) + orow * p.ldo + n) = h;
    } else {
      if (p.gamma) {
        const float4 g = *reinterpret_cast<const float4*>(p.gamma + n);
        v.x *= g.x; v.y *= g.y; v.z *= g.z; v.w *= g.w;
      }
      if (p.resid) {
        const float4 r = *reinterpret_cast<const float4*>(p.resid + rrow * p.ldr + n);
        v.x += r.x; v.y += r.y; v.z += r.z; v.w += r.w;
      }
      *reinterpret_cast<float4*>(reinterpret_cast<float*>(p.out) + orow * p.ldo + n) = v;
    }
  }
}

// LDS-staged epilogue for a 64-row x 64-column slab held by one wave as 2 x 2 transposed 32x32 accumulators: the wave
// parks the slab in its own 16 KiB of (now idle) stage memory, then re-reads it so that 16 lanes cover one full 64-column
// row: every global access instruction then touches 4 rows x 256 contiguous bytes (fp32) / 128 bytes (fp16) instead of
// 32 rows x 32 bytes. Measured before: the fp32 residual epilogue ran at ~1.5 TB/s (proj GEMM 230 us for 107 GFLOP).
// 16-byte chunks are XOR-swizzled by (row & 15) so the ds_write_b128 of 8 consecutive rows hit distinct banks.
// residual rows of a 64x64 slab in the staged epilogue's lane mapping, fetched EARLY (before the k-loop) so the HBM
// latency of `x += ...` hides under the MFMAs; in-place updates are safe because the thread that reads resid[m][n] is
// the thread that later writes out[m][n].
__device__ __forceinline__ void prefetch_resid(const GemmArgs& p, int mbase, int nbase, int lane, float4 (&r)[16]) {
  const int n = nbase + (lane & 15) * 4;
#pragma unroll
  for (int it = 0; it < 16; ++it) {
    int m = mbase + it * 4 + (lane >> 4);
    m = m < p.M ? m : p.M - 1;
    const size_t rrow = p.resid_mod ? (size_t)(m % p.resid_mod) : (size_t)m;
    r[it] = *reinterpret_cast<const float4*>(p.resid + rrow * p.ldr + n);
  }
}

__device__ __forceinline__ void slab_park(const f32x16 (&a00), const f32x16 (&a01), const f32x16 (&a10),
                                          const f32x16 (&a11), float* __restrict__ slab, int lane) {
  const int lr = lane & 31, lg = lane >> 5;
  const f32x16* accs[2][2] = {{&a00, &a01}, {&a10, &a11}};
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int row = i * 32 + lr;
        const int chunk = (j * 32 + 8 * q + 4 * lg) >> 2;
        const f32x16& a = *accs[i][j];
        *reinterpret_cast<float4*>(&slab[row * 64 + ((chunk ^ (row & 15)) << 2)]) =
            make_float4(a[4 * q], a[4 * q + 1], a[4 * q + 2], a[4 * q + 3]);
      }
}

template <int EPI, bool PRE = false, bool LNF = false>
__device__ __forceinline__ void slab_emit(const float* __restrict__ slab, int mbase, int nbase, int lane,
                                          const GemmArgs& p, const float4* pre = nullptr) {
  // same-wave LDS traffic is ordered; the compiler inserts the lgkmcnt wait for the reads below
  const int c4 = lane & 15;
  const int n = nbase + c4 * 4;
  float4 bv = make_float4(0.f, 0.f, 0.f, 0.f), gv = make_float4(1.f, 1.f, 1.f, 1.f);
  if (p.bias) bv = *reinterpret_cast<const float4*>(p.bias + n);
  if (EPI == EPI_F32 && p.gamma) gv = *reinterpret_cast<const float4*>(p.gamma + n);
#pragma unroll
  for (int it = 0; it < 16; ++it) {
    const int row = it * 4 + (lane >> 4);
    const int m = mbase + row;
    float4 v = *reinterpret_cast<const float4*>(&slab[row * 64 + ((c4 ^ (row & 15)) << 2)]);
    if (m >= p.M && !(LNF && EPI == EPI_F32 && p.stats)) continue;
    v.x += bv.x; v.y += bv.y; v.z += bv.z; v.w += bv.w;
    const size_t orow = p.out_seg ? (size_t)(m / p.out_seg) * p.out_seg_stride + p.out_seg_off + (m % p.out_seg) : (size_t)m;
    if (m >= p.M) {
      // (out-of-range row kept only for the wave-wide shuffles of the row statistics below)
    } else if (EPI == EPI_F16) {
      half4_t h = {(half_t)v.x, (half_t)v.y, (half_t)v.z, (half_t)v.w};
      if (p.head_hd) {  // packed qkv written head-major: plane (which*H + h) of [M, hd], 4 columns never straddle a head
        const int pl = n / p.head_hd;
        *reinterpret_cast<half4_t*>(reinterpret_cast<half_t*>(p.out) + ((size_t)pl * p.M + m) * p.head_hd + (n - pl * p.head_hd)) = h;
      } else {
        *reinterpret_cast<half4_t*>(reinterpret_cast<half_t*>(p.out) + orow * p.ldo + n) = h;
      }
    } else if (EPI == EPI_GELU_F16) {
      half4_t h = {(half_t)gelu_erf(v.x), (half_t)gelu_erf(v.y), (half_t)gelu_erf(v.z), (half_t)gelu_erf(v.w)};
      *reinterpret_cast<half4_t*>(reinterpret_cast<half_t*>(p.out) + orow * p.ldo + n) = h;
    } else if (EPI == EPI_RELU_F16) {  // conv + folded BatchNorm (+ identity) + ReLU of a ResNet bottleneck
      if (p.resid) {
        const half4_t r = *reinterpret_cast<const half4_t*>(reinterpret_cast<const half_t*>(p.resid) + (size_t)m * p.ldr + n);
        v.x += (float)r[0]; v.y += (float)r[1]; v.z += (float)r[2]; v.w += (float)r[3];
      }
      half4_t h = {(half_t)fmaxf(v.x, 0.f), (half_t)fmaxf(v.y, 0.f), (half_t)fmaxf(v.z, 0.f), (half_t)fmaxf(v.w, 0.f)};
      *reinterpret_cast<half4_t*>(reinterpret_cast<half_t*>(p.out) + orow * p.ldo + n) = h;
    } else {
      v.x *= gv.x; v.y *= gv.y; v.z *= gv.z; v.w *= gv.w;
      if (PRE) {
        const float4 r = pre[it];
        v.x += r.x; v.y += r.y; v.z += r.z; v.w += r.w;
      } else if (p.resid) {
        const size_t rrow = p.resid_mod ? (size_t)(m % p.resid_mod) : (size_t)m;
        const float4 r = *reinterpret_cast<const float4*>(p.resid + rrow * p.ldr + n);
        v.x += r.x; v.y += r.y; v.z += r.z; v.w += r.w;
      }
      *reinterpret_cast<float4*>(reinterpret_cast<float*>(p.out) + orow * p.ldo + n) = v;
      if (LNF && p.out16) {   // folded LayerNorm, producer side (16 lanes hold this row's 64 columns)
        half4_t h = {(half_t)v.x, (half_t)v.y, (half_t)v.z, (half_t)v.w};
        *reinterpret_cast<half4_t*>(p.out16 + orow * p.ld16 + n) = h;
      }
    }
    if (LNF && EPI == EPI_F32 && p.stats) {   // (every lane of the wave takes part in the shuffles: rows beyond M contribute nothing)
      float s1 = m < p.M ? (v.x + v.y) + (v.z + v.w) : 0.f;
      float s2 = m < p.M ? (v.x * v.x + v.y * v.y) + (v.z * v.z + v.w * v.w) : 0.f;
#pragma unroll
      for (int o = 1; o < 16; o <<= 1) { s1 += __shfl_xor(s1, o, 64); s2 += __shfl_xor(s2, o, 64); }
      if (c4 == 0 && m < p.M)
        *reinterpret_cast<float2*>(p.stats + (orow * (p.N >> 6) + (nbase >> 6)) * 2) = make_float2(s1, s2);
    }
  }
}

template <int EPI, bool PRE = false, bool LNF = false>
__device__ __forceinline__ void store_slab_staged(const f32x16 (&a00), const f32x16 (&a01), const f32x16 (&a10),
                                                  const f32x16 (&a11), float* __restrict__ slab, int mbase, int nbase,
                                                  int lane, const GemmArgs& p, const float4* pre = nullptr) {
  slab_park(a00, a01, a10, a11, slab, lane);
  slab_emit<EPI, PRE, LNF>(slab, mbase, nbase, lane, p, pre);
}

// fp16-output slabs (EPI_F16 / EPI_GELU_F16): bias / GELU are applied in fp32 in the accumulator layout and the slab is parked
// as fp16 (half the LDS bytes of the fp32 parking), then re-read so that 8 lanes cover one 64-column row with 16 bytes each:
// every global store is a dwordx4 over 8 rows x 128 contiguous bytes. The CU's store path is ISSUE-bound (~40 cycles per
// store instruction whatever its width: 256 dwordx2 stores per 256x256 tile took 5.6 us of a 45 us tile), so halving the
// instruction count is what counts. 16-byte chunks XOR-swizzled by (row >> 1) & 7: the 8-byte parking writes and the
// 16-byte reads are both conflict-free.
template <int EPI, bool LNF = false>
__device__ __forceinline__ void slab_park16(const f32x16 (&a00), const f32x16 (&a01), const f32x16 (&a10),
                                            const f32x16 (&a11), half_t* __restrict__ slab, int nbase, int lane,
                                            const GemmArgs& p, int mbase = 0) {
  const int lr = lane & 31, lg = lane >> 5;
  const f32x16* accs[2][2] = {{&a00, &a01}, {&a10, &a11}};
  float2 mr[2] = {make_float2(0.f, 1.f), make_float2(0.f, 1.f)};   // folded LayerNorm: (mean, rstd) of this lane's two rows
  if (LNF && p.ln_mr) {
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      int m = mbase + i * 32 + lr;
      m = m < p.M ? m : p.M - 1;
      mr[i] = *reinterpret_cast<const float2*>(p.ln_mr + (size_t)m * 2);
    }
  }
#pragma unroll
  for (int j = 0; j < 2; ++j)
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      float4 bv = make_float4(0.f, 0.f, 0.f, 0.f), sv = make_float4(0.f, 0.f, 0.f, 0.f);
      if (p.bias) bv = *reinterpret_cast<const float4*>(p.bias + nbase + j * 32 + 8 * q + 4 * lg);
      if (LNF && p.ln_mr) sv = *reinterpret_cast<const float4*>(p.ln_s + nbase + j * 32 + 8 * q + 4 * lg);
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        const int row = i * 32 + lr;
        const int chunk = j * 4 + q;
        const f32x16& a = *accs[i][j];
        float4 v;
        if (LNF && p.ln_mr) {
          const float mu = mr[i].x, rs = mr[i].y;
          v = make_float4(fmaf(fmaf(-mu, sv.x, a[4 * q]), rs, bv.x), fmaf(fmaf(-mu, sv.y, a[4 * q + 1]), rs, bv.y),
                          fmaf(fmaf(-mu, sv.z, a[4 * q + 2]), rs, bv.z), fmaf(fmaf(-mu, sv.w, a[4 * q + 3]), rs, bv.w));
        } else {
          v = make_float4(a[4 * q] + bv.x, a[4 * q + 1] + bv.y, a[4 * q + 2] + bv.z, a[4 * q + 3] + bv.w);
        }
        if (EPI == EPI_GELU_F16) { v.x = gelu_erf(v.x); v.y = gelu_erf(v.y); v.z = gelu_erf(v.z); v.w = gelu_erf(v.w); }
        half4_t h = {(half_t)v.x, (half_t)v.y, (half_t)v.z, (half_t)v.w};
        *reinterpret_cast<half4_t*>(&slab[row * 64 + ((chunk ^ ((row >> 1) & 7)) << 3) + 4 * lg]) = h;
      }
    }
}

__device__ __forceinline__ void slab_emit16(const half_t* __restrict__ slab, int mbase, int nbase, int lane,
                                            const GemmArgs& p) {
  const int c8 = lane & 7;
  const int n = nbase + c8 * 8;
  int pl = 0, ncol = n;
  if (p.head_hd) { pl = n / p.head_hd; ncol = n - pl * p.head_hd; }   // 8 columns never straddle a head (hd % 8 == 0)
#pragma unroll
  for (int it = 0; it < 8; ++it) {
    const int row = it * 8 + (lane >> 3);
    const int m = mbase + row;
    const half8_t v = *reinterpret_cast<const half8_t*>(&slab[row * 64 + ((c8 ^ ((row >> 1) & 7)) << 3)]);
    if (m >= p.M) continue;
    half_t* dst;
    if (p.head_hd) {
      dst = reinterpret_cast<half_t*>(p.out) + ((size_t)pl * p.M + m) * p.head_hd + ncol;
    } else {
      const size_t orow = p.out_seg ? (size_t)(m / p.out_seg) * p.out_seg_stride + p.out_seg_off + (m % p.out_seg) : (size_t)m;
      dst = reinterpret_cast<half_t*>(p.out) + orow * p.ldo + n;
    }
    *reinterpret_cast<half8_t*>(dst) = v;
  }
}

// 32-row x 64-column fp32 slabs for the `x += gamma * (acc + bias)` epilogue of the 128x64 wave tile
__device__ __forceinline__ void prefetch_resid32(const GemmArgs& p, int mbase, int nbase, int lane, float4 (&r)[8]) {
  const int n = nbase + (lane & 15) * 4;
#pragma unroll
  for (int it = 0; it < 8; ++it) {
    int m = mbase + it * 4 + (lane >> 4);
    m = m < p.M ? m : p.M - 1;
    const size_t rrow = p.resid_mod ? (size_t)(m % p.resid_mod) : (size_t)m;
    r[it] = *reinterpret_cast<const float4*>(p.resid + rrow * p.ldr + n);
  }
}

__device__ __forceinline__ void slab_park32(const f32x16 (&a0), const f32x16 (&a1), float* __restrict__ slab, int lane) {
  const int lr = lane & 31, lg = lane >> 5;
  const f32x16* accs[2] = {&a0, &a1};
#pragma unroll
  for (int j = 0; j < 2; ++j)
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int chunk = (j * 32 + 8 * q + 4 * lg) >> 2;
      const f32x16& a = *accs[j];
      *reinterpret_cast<float4*>(&slab[lr * 64 + ((chunk ^ (lr & 15)) << 2)]) =
          make_float4(a[4 * q], a[4 * q + 1], a[4 * q + 2], a[4 * q + 3]);
    }
}

__device__ __forceinline__ void slab_emit32(const float* __restrict__ slab, int mbase, int nbase, int lane,
                                            const GemmArgs& p, const float4 (&pre)[8]) {
  const int c4 = lane & 15;
  const int n = nbase + c4 * 4;
  float4 bv = make_float4(0.f, 0.f, 0.f, 0.f), gv = make_float4(1.f, 1.f, 1.f, 1.f);
  if (p.bias) bv = *reinterpret_cast<const float4*>(p.bias + n);
  if (p.gamma) gv = *reinterpret_cast<const float4*>(p.gamma + n);
#pragma unroll
  for (int it = 0; it < 8; ++it) {
    const int row = it * 4 + (lane >> 4);
    const int m = mbase + row;
    float4 v = *reinterpret_cast<const float4*>(&slab[row * 64 + ((c4 ^ (row & 15)) << 2)]);
    if (m >= p.M) continue;
    const size_t orow = p.out_seg ? (size_t)(m / p.out_seg) * p.out_seg_stride + p.out_seg_off + (m % p.out_seg) : (size_t)m;
    const float4 r = pre[it];
    v.x = (v.x + bv.x) * gv.x + r.x; v.y = (v.y + bv.y) * gv.y + r.y;
    v.z = (v.z + bv.z) * gv.z + r.z; v.w = (v.w + bv.w) * gv.w + r.w;
    *reinterpret_cast<float4*>(reinterpret_cast<float*>(p.out) + orow * p.ldo + n) = v;
  }
}

// 128-row x 64-column wave tile (two slabs) with the residual rows of BOTH slabs requested before anything waits on
// them (16 float4 loads in flight per lane, issued ahead of the LDS parking). Without this the `x += ...` epilogues (proj,
// lin2) ran 30-40 % below the fp16-output GEMMs of the same shape (profiles/r01_f: 528 / 712 vs 830 / 1010 TFLOP/s).
template <int EPI>
__device__ __forceinline__ void store_wave_tile_128x64(const f32x16 (&acc)[2][2][2], float* __restrict__ slab, int mbase,
                                                       int nbase, int lane, const GemmArgs& p) {
  if (EPI == EPI_F32 && p.resid != nullptr) {
    // four 32-row slabs; the residual rows of slab s+1 are requested before slab s is emitted (two 8-float4 sets in
    // flight), so only the first request's latency is exposed and the register peak stays below the spill line
    float4 ra[8], rb[8];
    float* slab2[2] = {slab, slab + 2048};
    prefetch_resid32(p, mbase, nbase, lane, ra);
#pragma unroll
    for (int sidx = 0; sidx < 4; ++sidx) {
      const int a = sidx >> 1, i = sidx & 1;
      float* sl = slab2[sidx & 1];
      slab_park32(acc[a][i][0], acc[a][i][1], sl, lane);
      __builtin_amdgcn_sched_barrier(0);
      if (sidx < 3) prefetch_resid32(p, mbase + (sidx + 1) * 32, nbase, lane, (sidx & 1) ? ra : rb);
      __builtin_amdgcn_sched_barrier(0);
      slab_emit32(sl, mbase + sidx * 32, nbase, lane, p, (sidx & 1) ? rb : ra);
      __builtin_amdgcn_sched_barrier(0);
    }
  } else if (EPI == EPI_F16 || EPI == EPI_GELU_F16) {
    half_t* slab16 = reinterpret_cast<half_t*>(slab);   // two 8 KiB fp16 slabs inside the wave's 16 KiB
#pragma unroll
    for (int a = 0; a < 2; ++a) {
      slab_park16<EPI>(acc[a][0][0], acc[a][0][1], acc[a][1][0], acc[a][1][1], slab16 + a * 4096, nbase, lane, p);
      slab_emit16(slab16 + a * 4096, mbase + a * 64, nbase, lane, p);
    }
  } else {
#pragma unroll
    for (int a = 0; a < 2; ++a)
      store_slab_staged<EPI, false>(acc[a][0][0], acc[a][0][1], acc[a][1][0], acc[a][1][1], slab, mbase + a * 64, nbase,
                                    lane, p);
  }
}

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

template <int EPI, bool LNF = false>
__global__ __launch_bounds__(256) void gemm_f16_kernel(GemmArgs p) {
  __shared__ __attribute__((aligned(16))) half_t smem[2][2][BM * BK];  // [buf][A|W] 64 KiB

  const int ntn = p.N / BN;
  const int ntm = (p.M + BM - 1) / BM;
  int tm, tn;
  if (!tile_map(blockIdx.x, ntm, ntn, p.map_mode, tm, tn)) return;
  const int m0 = tm * BM;
  const int n0 = tn * BN;

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
  // residual prefetch (EPI_F32): 16 float4 per lane, landed long before the epilogue
  float4 rpre[16];
  const bool has_res = (EPI == EPI_F32) && p.resid != nullptr;
  if (EPI == EPI_F32 && has_res) prefetch_resid(p, m0 + wm * 64, n0 + wn * 64, lane, rpre);
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
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fb[j], fa[i], acc[i][j], 0, 0, 0);  // D^T
    }
    __syncthreads();  // all waves done reading buf[cur]; DMA into buf[cur^1] has landed (vmcnt(0))
  }

  // epilogue: each wave stages its 64x64 slab through its own quarter of the (idle) 64 KiB stage memory; the loop's
  // closing __syncthreads() guarantees nobody still reads the k-tile buffers
  float* slab = reinterpret_cast<float*>(&smem[0][0][0]) + wv * 4096;
  if (EPI == EPI_F32 && has_res)
    store_slab_staged<EPI, true, LNF>(acc[0][0], acc[0][1], acc[1][0], acc[1][1], slab, m0 + wm * 64, n0 + wn * 64, lane, p,
                                      rpre);
  else if ((EPI == EPI_F16 || EPI == EPI_GELU_F16) && p.wide16) {
    half_t* slab16 = reinterpret_cast<half_t*>(slab);
    slab_park16<EPI, LNF>(acc[0][0], acc[0][1], acc[1][0], acc[1][1], slab16, n0 + wn * 64, lane, p, m0 + wm * 64);
    slab_emit16(slab16, m0 + wm * 64, n0 + wn * 64, lane, p);
  } else
    store_slab_staged<EPI, false, LNF>(acc[0][0], acc[0][1], acc[1][0], acc[1][1], slab, m0 + wm * 64, n0 + wn * 64, lane, p);
}

// =====================================================================================================
// Large-tile variant: 256 x BN x 32 k-steps, 512 threads = 8 waves (2 x 4), wave tile 128 x BN/4, 4-deep LDS ring
// (4 x 32 KiB for BN = 256) filled by direct-to-LDS DMA three k-steps ahead. The load path (L2 -> LDS, ~20 B/clk/CU)
// is what bounds a 128x128 tile (64 FLOP/B); the 256x256 tile halves the bytes per FLOP. One raw s_barrier per
// k-step; the DMA queue is never drained inside the loop: `s_waitcnt vmcnt(2*LPT)` only retires the k-step about to
// be read and leaves the next two in flight across the barrier.
//   iteration t:  wait(tile t landed, own pieces) -> barrier (everyone's pieces landed; everyone finished reading
//                 tile t-1) -> issue DMA of tile t+3 into the slot of tile t-1 -> MFMAs on tile t.
template <int BN_>
struct G256 {
  static constexpr int BM_ = 256, BK_ = 32, NST = 4;
  static constexpr int WN = BN_ / 4;        // wave tile width
  static constexpr int NJ = WN / 32;        // 32-wide MFMA tiles per wave along N
  static constexpr int A_EL = BM_ * BK_;    // halfs per stage (A)
  static constexpr int B_EL = BN_ * BK_;
  static constexpr int ST_EL = A_EL + B_EL;
  static constexpr int PA = 2;              // 1-KiB A pieces per wave per stage (16 pieces / 8 waves)
  static constexpr int PB = BN_ / 128;      // 1-KiB B pieces per wave per stage
  static constexpr int LPT = PA + PB;       // DMA instructions per thread per stage
  static constexpr int LDS_BYTES = NST * ST_EL * 2;
};

__device__ __forceinline__ int lds_off32(int row, int chunk) {
  // [rows][32] fp16 tile (64-byte rows, 4 rows per 256-byte bank row): slot ^= (row>>2)&3 is conflict-free for the
  // 16-lane groups of ds_read_b128
  return row * 32 + ((chunk ^ ((row >> 2) & 3)) << 3);
}

template <int N>
__device__ __forceinline__ void wait_vmcnt() {
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

template <int EPI, int BN_, int STAG>
__global__ __launch_bounds__(512) void gemm256_f16_kernel(GemmArgs p) {
  using C = G256<BN_>;
  extern __shared__ __attribute__((aligned(16))) half_t ring[];  // the ONLY LDS object of this kernel

  const int ntn = p.N / BN_;
  const int ntm = (p.M + C::BM_ - 1) / C::BM_;
  int tm, tn;
  if (!tile_map(blockIdx.x, ntm, ntn, p.map_mode, tm, tn)) return;
  const int m0 = tm * C::BM_;
  const int n0 = tn * BN_;

  const int t = threadIdx.x;
  const int lane = t & 63;
  const int wv = __builtin_amdgcn_readfirstlane(t >> 6);
  const int wm = wv >> 2, wn = wv & 3;
  const int lr = lane & 31, lg = lane >> 5;

  // DMA source pointers: piece q = 16 tile rows; lane -> (row = q*16 + lane/4, slot = lane%4), chunk = slot ^ ((row>>2)&3)
  const half_t* ag[C::PA];
  const half_t* wg[C::PB];
#pragma unroll
  for (int j = 0; j < C::PA; ++j) {
    const int row = (wv * C::PA + j) * 16 + (lane >> 2);
    const int chunk = (lane & 3) ^ ((row >> 2) & 3);
    int am = m0 + row;
    am = am < p.M ? am : p.M - 1;
    ag[j] = p.A + (size_t)am * p.lda + chunk * 8;
  }
#pragma unroll
  for (int j = 0; j < C::PB; ++j) {
    const int row = (wv * C::PB + j) * 16 + (lane >> 2);
    const int chunk = (lane & 3) ^ ((row >> 2) & 3);
    wg[j] = p.W + (size_t)(n0 + row) * p.ldw + chunk * 8;
  }

  auto stage = [&](int kt) {
    half_t* base = ring + (kt & (C::NST - 1)) * C::ST_EL;
#pragma unroll
    for (int j = 0; j < C::PA; ++j) glds16(ag[j] + kt * C::BK_, base + (wv * C::PA + j) * 512);
#pragma unroll
    for (int j = 0; j < C::PB; ++j) glds16(wg[j] + kt * C::BK_, base + C::A_EL + (wv * C::PB + j) * 512);
  };

  f32x16 acc[4][C::NJ];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < C::NJ; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  // residual rows of the first epilogue pass, fetched before any DMA is queued (they are the OLDEST entries of the
  // in-order vmcnt queue, so the counted waits below are unchanged)
  float4 rpre[16];
  const bool has_res = (EPI == EPI_F32) && (C::NJ == 2) && p.resid != nullptr;
  if (EPI == EPI_F32 && C::NJ == 2 && has_res) prefetch_resid(p, m0 + wm * 128, n0 + wn * 64, lane, rpre);
  const int nk = p.K / C::BK_;
  if (STAG == 0) {
    // plain ring: one barrier per k-step, DMA of k-step t+3 issued right after it, fragments read and consumed in place
    stage(0);
    if (nk > 1) stage(1);
    if (nk > 2) stage(2);
    for (int kt = 0; kt < nk; ++kt) {
      if (kt + 2 < nk)
        wait_vmcnt<2 * C::LPT>();
      else if (kt + 1 < nk)
        wait_vmcnt<C::LPT>();
      else
        wait_vmcnt<0>();
      __builtin_amdgcn_s_barrier();
      asm volatile("" ::: "memory");
      if (kt + 3 < nk) stage(kt + 3);
      const half_t* sa = ring + (kt & (C::NST - 1)) * C::ST_EL;
      const half_t* sw = sa + C::A_EL;
#pragma unroll
      for (int s = 0; s < 2; ++s) {
        half8_t fa[4], fb[C::NJ];
#pragma unroll
        for (int i = 0; i < 4; ++i)
          fa[i] = *reinterpret_cast<const half8_t*>(&sa[lds_off32(wm * 128 + i * 32 + lr, s * 2 + lg)]);
#pragma unroll
        for (int j = 0; j < C::NJ; ++j)
          fb[j] = *reinterpret_cast<const half8_t*>(&sw[lds_off32(wn * C::WN + j * 32 + lr, s * 2 + lg)]);
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
          for (int j = 0; j < C::NJ; ++j)
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fb[j], fa[i], acc[i][j], 0, 0, 0);  // D^T
      }
    }
  } else {
  // Two wave groups (waves 0-3 / 4-7: one wave of each per SIMD) run the same phase sequence one barrier apart, so
  // while one group is in its MFMA phase the other issues DMA and reads fragments:
  //   L_s: issue DMA of k-step s+3 (into the slot of k-step s-1), read the 12 fragments of k-step s into registers,
  //        wait until this wave's pieces of k-step s+1 have landed (two k-steps stay in flight), lgkmcnt(0), barrier
  //   M_s: 16 MFMAs, barrier
  // Slot reuse is safe: k-step s-1 was last read in L_{s-1} (lagging group: one phase before the leading group's L_s)
  // and those reads were retired by the lgkmcnt(0) before that phase's closing barrier. A k-step is read only after a
  // barrier that every wave passed after waiting for its own pieces of it.
  stage(0);
  if (nk > 1) stage(1);
  if (nk > 2) stage(2);
  if (nk > 2)
    wait_vmcnt<2 * C::LPT>();
  else if (nk > 1)
    wait_vmcnt<C::LPT>();
  else
    wait_vmcnt<0>();
  __builtin_amdgcn_s_barrier();
  if (wm == 1) __builtin_amdgcn_s_barrier();  // stagger the second group by one phase
  asm volatile("" ::: "memory");

  for (int kt = 0; kt < nk; ++kt) {
    // ---- L phase -------------------------------------------------------------------------------------------------
    if (kt + 3 < nk) stage(kt + 3);
    const half_t* sa = ring + (kt & (C::NST - 1)) * C::ST_EL;
    const half_t* sw = sa + C::A_EL;
    half8_t fa[2][4], fb[2][C::NJ];
#pragma unroll
    for (int s = 0; s < 2; ++s) {
#pragma unroll
      for (int i = 0; i < 4; ++i)
        fa[s][i] = *reinterpret_cast<const half8_t*>(&sa[lds_off32(wm * 128 + i * 32 + lr, s * 2 + lg)]);
#pragma unroll
      for (int j = 0; j < C::NJ; ++j)
        fb[s][j] = *reinterpret_cast<const half8_t*>(&sw[lds_off32(wn * C::WN + j * 32 + lr, s * 2 + lg)]);
    }
    if (kt + 3 < nk)
      wait_vmcnt<2 * C::LPT>();
    else if (kt + 2 < nk)
      wait_vmcnt<C::LPT>();
    else
      wait_vmcnt<0>();
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
    // ---- M phase -------------------------------------------------------------------------------------------------
    __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int s = 0; s < 2; ++s)
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < C::NJ; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fb[s][j], fa[s][i], acc[i][j], 0, 0, 0);  // D^T
    __builtin_amdgcn_s_setprio(0);
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
  }
  if (wm == 0) __builtin_amdgcn_s_barrier();  // balance the stagger

  }

#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < C::NJ; ++j)
      store_acc32<EPI>(acc[i][j], m0 + wm * 128 + i * 32 + lr, n0 + wn * C::WN + j * 32 + 4 * lg, p);
}

// =====================================================================================================
// Wave-specialised variant: 384 x 128 x 32 k-steps, 512 threads = 6 consumer waves (3 x 2, wave tile 128 x 64) + 2 loader
// waves that issue ALL direct-to-LDS DMA (16 one-KiB pieces each per k-step) and never touch the matrix pipe. A wave
// stalled at VMEM issue cannot issue its MFMAs (in-order issue); with every wave both loading and computing, DMA time
// and MFMA time add up (measured: 195 + 164 us ~ 365 us on 32768x3840x1280), with dedicated loaders they overlap.
// Same 4-deep ring and barrier protocol as the plain ring kernel: loader waits for its pieces of k-step t, everyone
// meets at the barrier, loader then refills the slot of k-step t-1 with k-step t+3 while the consumers work on t.
template <int EPI>
__global__ __launch_bounds__(512) void gemm_ws_f16_kernel(GemmArgs p) {
  constexpr int BMW = 384, BNW = 128, BKW = 32, NST = 4;
  constexpr int A_EL = BMW * BKW, B_EL = BNW * BKW, ST_EL = A_EL + B_EL;  // 16384 halfs = 32 KiB per stage
  constexpr int NPIECE = (BMW + BNW) / 16;                              // 32 one-KiB pieces per stage
  constexpr int LP = NPIECE / 2;                                        // per loader wave
  extern __shared__ __attribute__((aligned(16))) half_t ring[];

  const int ntn = p.N / BNW;
  const int ntm = (p.M + BMW - 1) / BMW;
  int tm, tn;
  if (!tile_map(blockIdx.x, ntm, ntn, p.map_mode, tm, tn)) return;
  const int m0 = tm * BMW, n0 = tn * BNW;
  const int t = threadIdx.x, lane = t & 63;
  const int wv = __builtin_amdgcn_readfirstlane(t >> 6);
  const int nk = p.K / BKW;

  if (wv >= 6) {
    // ---------------- loader wave ----------------
    const int ld = wv - 6;
    const half_t* src[LP];
#pragma unroll
    for (int j = 0; j < LP; ++j) {
      const int piece = ld * LP + j;                 // 0..23: A rows, 24..31: W rows
      const int row = piece * 16 + (lane >> 2);      // row inside the stacked [A | W] stage image
      const int chunk = (lane & 3) ^ ((row >> 2) & 3);
      if (piece < BMW / 16) {
        int am = m0 + row;
        am = am < p.M ? am : p.M - 1;
        src[j] = p.A + (size_t)am * p.lda + chunk * 8;
      } else {
        src[j] = p.W + (size_t)(n0 + row - BMW) * p.ldw + chunk * 8;
      }
    }
    auto stage = [&](int kt) {
      half_t* base = ring + (kt & (NST - 1)) * ST_EL + (ld * LP) * 512;
#pragma unroll
      for (int j = 0; j < LP; ++j) glds16(src[j] + kt * BKW, base + j * 512);
    };
    stage(0);
    if (nk > 1) stage(1);
    if (nk > 2) stage(2);
    for (int kt = 0; kt < nk; ++kt) {
      if (kt + 2 < nk)
        wait_vmcnt<2 * LP>();
      else if (kt + 1 < nk)
        wait_vmcnt<LP>();
      else
        wait_vmcnt<0>();
      __builtin_amdgcn_s_barrier();
      asm volatile("" ::: "memory");
      if (kt + 3 < nk) stage(kt + 3);
    }
    return;
  }

  // ---------------- consumer wave ----------------
  const int wm = wv >> 1, wn = wv & 1;
  const int lr = lane & 31, lg = lane >> 5;
  f32x16 acc[4][2];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
  for (int kt = 0; kt < nk; ++kt) {
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    const half_t* sa = ring + (kt & (NST - 1)) * ST_EL;
    const half_t* sw = sa + A_EL;
#pragma unroll
    for (int s = 0; s < 2; ++s) {
      half8_t fa[4], fb[2];
#pragma unroll
      for (int i = 0; i < 4; ++i)
        fa[i] = *reinterpret_cast<const half8_t*>(&sa[lds_off32(wm * 128 + i * 32 + lr, s * 2 + lg)]);
#pragma unroll
      for (int j = 0; j < 2; ++j)
        fb[j] = *reinterpret_cast<const half8_t*>(&sw[lds_off32(wn * 64 + j * 32 + lr, s * 2 + lg)]);
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fb[j], fa[i], acc[i][j], 0, 0, 0);  // D^T
    }
  }
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
      store_acc32<EPI>(acc[i][j], m0 + wm * 128 + i * 32 + lr, n0 + wn * 64 + j * 32 + 4 * lg, p);
}


// =====================================================================================================
// 256 x 256 x 64, 8 waves (2 x 4, wave tile 128 x 64), "8-phase" schedule (tile 7).
// A K-tile is consumed in four phases, one 64 x 32 quadrant of the wave tile (8 MFMA 32x32x16) each, in the order
// (A0,B0) (A0,B1) (A1,B1) (A1,B0) so only one operand sub-tile changes per phase (12 / 4 / 8 / 0 ds_read_b128).
// LDS = 2 buffers x 4 half-tiles (A0 A1 B0 B1) of [128][64] fp16 = 128 KiB. Half-tile Ah holds, for BOTH wave rows,
// sub-tile h of their 128 rows (local row wr*64 + i <-> block row wr*128 + h*64 + i); Bh likewise for the four wave
// columns (local row wc*32 + i <-> block column wc*64 + h*32 + i), so every wave finishes reading A0/B0 in phase 1, B1 in
// phase 2 and A1 in phase 3 and the half-tile can be refilled two phases later. Each phase issues the DMA of ONE
// half-tile (2 x global_load_lds_dwordx4 per lane), in consumption order, ~6 phases ahead of its first read:
//     phase 1(t): B1(t+1)   phase 2(t): A1(t+1)   phase 3(t): A0(t+2)   phase 4(t): B0(t+2)
// followed by s_waitcnt vmcnt(8): the four newest half-tiles stay in flight, everything a wave issued before them has
// landed - which is exactly what the NEXT phase reads. The queue is never drained inside the loop.
// The two wave rows (one wave of each per SIMD) run the same phase sequence one barrier apart, so on every SIMD one
// wave is in its MFMA segment while the other reads fragments and issues DMA.
// Ordering: a half-tile is read only after a barrier that every wave passed after its own counted wait for it (phase
// p waits, phase p+1 reads; the lagging group's wait is one barrier later, its read too). A half-tile is refilled >= 2
// phases after its last read, whose lgkmcnt(0) every wave passed before the intervening barriers.
__device__ __forceinline__ int lds_off64(int row, int chunk) {
  return row * 64 + ((chunk ^ ((row >> 1) & 7)) << 3);
}

template <int EPI, bool DBG = false>
__global__ __launch_bounds__(512) void gemm8p_f16_kernel(GemmArgs p) {
  extern __shared__ __attribute__((aligned(16))) half_t ring[];  // [buf 2][A0 A1 B0 B1][128][64]
  constexpr int HT = 128 * 64;                                     // halfs per half-tile
  const int ntn = p.N / 256;
  const int ntm = (p.M + 255) / 256;
  int tm, tn;
  if (!tile_map(blockIdx.x, ntm, ntn, p.map_mode, tm, tn)) return;
  const int m0 = tm * 256, n0 = tn * 256;
  const int t = threadIdx.x, lane = t & 63;
  const int wv = __builtin_amdgcn_readfirstlane(t >> 6);
  const int wr = wv >> 2, wc = wv & 3;
  const int lr = lane & 31, lg = lane >> 5;

  // DMA sources. Instruction j of a half-tile covers local rows j*64 + t/8 (8 lanes = one 128-byte row), slot t%8.
  const half_t* asrc[2];  // half 0; half 1 = + 64 rows
  const half_t* bsrc[2];  // half 0; half 1 = + 32 rows
  int a_clamp[2][2];
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int lrow = j * 64 + (t >> 3);
    const int chunk = (t & 7) ^ ((lrow >> 1) & 7);
    const int arow = (lrow >> 6) * 128 + (lrow & 63);       // + h*64
    const int brow = (lrow >> 5) * 64 + (lrow & 31);        // + h*32
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      int am = m0 + arow + h * 64;
      a_clamp[j][h] = (am < p.M ? am : p.M - 1) - (m0 + arow);   // row delta actually fetched (ragged last tile)
    }
    asrc[j] = p.A + (size_t)(m0 + arow) * p.lda + chunk * 8;
    bsrc[j] = p.W + (size_t)(n0 + brow) * p.ldw + chunk * 8;
  }
  // stage half-tile `which` (0 A0, 1 A1, 2 B0, 3 B1) of K-tile kt
  const int dbg = DBG ? p.dbg : 0;  // 1: no DMA in the loop, 2: no fragment reads, 4: no MFMA, 8: no epilogue, 16: no barriers
  auto stage = [&](int which, int kt) {
    if (DBG && (dbg & 1) && kt > 1) return;
    half_t* dst = ring + ((kt & 1) * 4 + which) * HT + wv * 512;
    const int h = which & 1;
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const half_t* g = which < 2 ? asrc[j] + (ptrdiff_t)a_clamp[j][h] * p.lda : bsrc[j] + (size_t)h * 32 * p.ldw;
      glds16(g + kt * 64, dst + j * 4096);
    }
  };

  f32x16 acc[2][2][2];  // [a][i][b]: rows m0 + wr*128 + a*64 + i*32 + lr, cols n0 + wc*64 + b*32 + ...
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int b = 0; b < 2; ++b)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[a][i][b][r] = 0.f;

  const int nk = p.K / 64;
  // prologue: A0 B0 B1 A1 of K-tile 0, A0 B0 of K-tile 1 (the steady schedule's phases 3/4 of "K-tile -1")
  stage(0, 0); stage(2, 0); stage(3, 0); stage(1, 0);
  if (nk > 1) { stage(0, 1); stage(2, 1); wait_vmcnt<8>(); } else { wait_vmcnt<4>(); }
  __builtin_amdgcn_s_barrier();
  if (wr == 1) __builtin_amdgcn_s_barrier();  // stagger the second wave row by one barrier
  asm volatile("" ::: "memory");

  half8_t fa[2][4], fb0[4], fb1[4];
  const int arow0 = wr * 64 + lr, brow0 = wc * 32 + lr;

#define PHASE_SYNC_IN()                                   \
  if (!(DBG && (dbg & 16))) __builtin_amdgcn_s_barrier(); \
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");      \
  __builtin_amdgcn_sched_barrier(0);                      \
  __builtin_amdgcn_s_setprio(1);
#define PHASE_SYNC_OUT()                                  \
  __builtin_amdgcn_s_setprio(0);                          \
  __builtin_amdgcn_sched_barrier(0);                      \
  if (!(DBG && (dbg & 16))) __builtin_amdgcn_s_barrier(); \
  asm volatile("" ::: "memory");
#define RD(dst, off) if (!(DBG && (dbg & 2))) dst = *reinterpret_cast<const half8_t*>(&buf[off])
#define MMA(c, a_, b_) if (!(DBG && (dbg & 4))) c = __builtin_amdgcn_mfma_f32_32x32x16_f16(a_, b_, c, 0, 0, 0)

  if (DBG) {
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int s2 = 0; s2 < 4; ++s2) fa[i][s2] = half8_t{0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
    for (int s2 = 0; s2 < 4; ++s2) fb0[s2] = fb1[s2] = half8_t{0, 0, 0, 0, 0, 0, 0, 0};
  }
  for (int kt = 0; kt < nk; ++kt) {
    const half_t* buf = ring + (kt & 1) * 4 * HT;
    const bool more1 = kt + 1 < nk, more2 = kt + 2 < nk;
    // ---- phase 1: read B0, A0; stage B1(kt+1); quadrant (A0, B0) -------------------------------------------------
#pragma unroll
    for (int s2 = 0; s2 < 4; ++s2)
      RD(fb0[s2], 2 * HT + lds_off64(brow0, s2 * 2 + lg));
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int s2 = 0; s2 < 4; ++s2)
        RD(fa[i][s2], 0 * HT + lds_off64(arow0 + i * 32, s2 * 2 + lg));
    if (more1) stage(3, kt + 1);
    if (more2) wait_vmcnt<8>(); else wait_vmcnt<0>();
    PHASE_SYNC_IN();
#pragma unroll
    for (int s2 = 0; s2 < 4; ++s2)
#pragma unroll
      for (int i = 0; i < 2; ++i)
        MMA(acc[0][i][0], fb0[s2], fa[i][s2]);
    PHASE_SYNC_OUT();
    // ---- phase 2: read B1; stage A1(kt+1); quadrant (A0, B1) -----------------------------------------------------
#pragma unroll
    for (int s2 = 0; s2 < 4; ++s2)
      RD(fb1[s2], 3 * HT + lds_off64(brow0, s2 * 2 + lg));
    if (more1) stage(1, kt + 1);
    if (more2) wait_vmcnt<8>(); else wait_vmcnt<0>();
    PHASE_SYNC_IN();
#pragma unroll
    for (int s2 = 0; s2 < 4; ++s2)
#pragma unroll
      for (int i = 0; i < 2; ++i)
        MMA(acc[0][i][1], fb1[s2], fa[i][s2]);
    PHASE_SYNC_OUT();
    // ---- phase 3: read A1; stage A0(kt+2); quadrant (A1, B1) -----------------------------------------------------
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int s2 = 0; s2 < 4; ++s2)
        RD(fa[i][s2], 1 * HT + lds_off64(arow0 + i * 32, s2 * 2 + lg));
    if (more2) { stage(0, kt + 2); wait_vmcnt<8>(); } else { wait_vmcnt<0>(); }
    PHASE_SYNC_IN();
#pragma unroll
    for (int s2 = 0; s2 < 4; ++s2)
#pragma unroll
      for (int i = 0; i < 2; ++i)
        MMA(acc[1][i][1], fb1[s2], fa[i][s2]);
    PHASE_SYNC_OUT();
    // ---- phase 4: no reads; stage B0(kt+2); quadrant (A1, B0) ----------------------------------------------------
    if (more2) { stage(2, kt + 2); wait_vmcnt<8>(); } else { wait_vmcnt<0>(); }
    PHASE_SYNC_IN();
#pragma unroll
    for (int s2 = 0; s2 < 4; ++s2)
#pragma unroll
      for (int i = 0; i < 2; ++i)
        MMA(acc[1][i][0], fb0[s2], fa[i][s2]);
    PHASE_SYNC_OUT();
  }
#undef PHASE_SYNC_IN
#undef PHASE_SYNC_OUT
#undef RD
#undef MMA
  if (wr == 0) __builtin_amdgcn_s_barrier();  // balance the stagger
  if (DBG && (dbg & 8)) return;

  // epilogue: every wave parks its two 64x64 slabs, one after the other, in its own 16 KiB of the (idle, fully landed,
  // no longer read) ring and re-reads them row-contiguously: each store instruction covers 4 rows x 128/256 bytes
  float* slab = reinterpret_cast<float*>(ring) + wv * 4096;
  store_wave_tile_128x64<EPI>(acc, slab, m0 + wr * 128, n0 + wc * 64, lane, p);
}

template <int EPI, bool DBG = false>
static void launch8p(const GemmArgs& p, hipStream_t s) {
  constexpr int LDS = 2 * 4 * 128 * 64 * 2;
  static bool attr = false;
  if (!attr) {
    (void)hipFuncSetAttribute((const void*)gemm8p_f16_kernel<EPI, DBG>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS);
    attr = true;
  }
  const int ntm = (p.M + 255) / 256, ntn = p.N / 256;
  hipLaunchKernelGGL((gemm8p_f16_kernel<EPI, DBG>), dim3(tile_map_grid(ntm, ntn, p.map_mode)), dim3(512), LDS, s, p);
}


// =====================================================================================================
// Tile 8: the 8-phase kernel's geometry, half-tile DMA schedule and LDS image, with ONE barrier per phase instead of
// two. The two wave rows run the same phase in the same barrier interval but in opposite order: row 0 opens the
// interval with the phase's MFMAs (its fragments were read in the previous interval) and then reads the NEXT phase's
// fragments; row 1 reads the phase's fragments first and then issues its MFMAs. On every SIMD one wave therefore starts
// on the matrix pipe while the other starts on LDS, and they swap without a rendezvous. An empty barrier interval costs
// ~270 cycles on this chip (measured: tile 7 with everything but the barriers removed), more than the 256 cycles of a
// phase's MFMAs, so halving the barrier count matters more than any instruction placement inside a phase.
// Row 0 reads a half-tile one interval before row 1, so the counted wait moves one phase earlier: vmcnt(6) (three
// half-tiles in flight). Refill distance (>= 2 phases after the last read of either row) is unchanged.
template <int EPI>
__global__ __launch_bounds__(512) void gemm8h_f16_kernel(GemmArgs p) {
  extern __shared__ __attribute__((aligned(16))) half_t ring[];  // [buf 2][A0 A1 B0 B1][128][64]
  constexpr int HT = 128 * 64;
  const int ntn = p.N / 256;
  const int ntm = (p.M + 255) / 256;
  int tm, tn;
  if (!tile_map(blockIdx.x, ntm, ntn, p.map_mode, tm, tn)) return;
  const int m0 = tm * 256, n0 = tn * 256;
  const int t = threadIdx.x, lane = t & 63;
  const int wv = __builtin_amdgcn_readfirstlane(t >> 6);
  const int wr = wv >> 2, wc = wv & 3;
  const int lr = lane & 31, lg = lane >> 5;

  // DMA sources as 32-bit element offsets from the (wave-uniform) operand bases: instruction j of a half-tile covers
  // local rows j*64 + t/8 (8 lanes = one 128-byte row), LDS slot t%8 <- source chunk slot ^ ((row>>1)&7)
  unsigned aoff[2][2], boff[2];
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int lrow = j * 64 + (t >> 3);
    const int chunk = (t & 7) ^ ((lrow >> 1) & 7);
    const int arow = (lrow >> 6) * 128 + (lrow & 63);       // + h*64
    const int brow = (lrow >> 5) * 64 + (lrow & 31);        // + h*32
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      int am = m0 + arow + h * 64;
      am = am < p.M ? am : p.M - 1;                          // ragged last row tile: re-read the last row
      aoff[j][h] = (unsigned)am * (unsigned)p.lda + chunk * 8;
    }
    boff[j] = (unsigned)(n0 + brow) * (unsigned)p.ldw + chunk * 8;
  }
  const unsigned bh = 32u * (unsigned)p.ldw;
  auto stage = [&](int which, int kt) {  // 0 A0, 1 A1, 2 B0, 3 B1
    half_t* dst = ring + ((kt & 1) * 4 + which) * HT + wv * 512;
    const int h = which & 1;
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const half_t* g = which < 2 ? p.A + (aoff[j][h] + (unsigned)kt * 64u) : p.W + (boff[j] + h * bh + (unsigned)kt * 64u);
      glds16(g, dst + j * 4096);
    }
  };

  f32x16 acc[2][2][2];
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int b = 0; b < 2; ++b)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[a][i][b][r] = 0.f;

  half8_t fa[2][4], fb0[4], fb1[4];
  const int arow0 = wr * 64 + lr, brow0 = wc * 32 + lr;
  const int nk = p.K / 64;

#define RD_A(bufp, h)                                                                                             \
  _Pragma("unroll") for (int i = 0; i < 2; ++i) _Pragma("unroll") for (int s2 = 0; s2 < 4; ++s2)                   \
      fa[i][s2] = *reinterpret_cast<const half8_t*>(&(bufp)[(h) * HT + lds_off64(arow0 + i * 32, s2 * 2 + lg)]);
#define RD_B(dst, bufp, h)                                                                                        \
  _Pragma("unroll") for (int s2 = 0; s2 < 4; ++s2)                                                                 \
      dst[s2] = *reinterpret_cast<const half8_t*>(&(bufp)[(2 + (h)) * HT + lds_off64(brow0, s2 * 2 + lg)]);
#define MMA_Q(a, b, fb)                                                                                           \
  __builtin_amdgcn_s_setprio(1);                                                                                  \
  _Pragma("unroll") for (int s2 = 0; s2 < 4; ++s2) _Pragma("unroll") for (int i = 0; i < 2; ++i)                   \
      acc[a][i][b] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fb[s2], fa[i][s2], acc[a][i][b], 0, 0, 0);           \
  __builtin_amdgcn_s_setprio(0);
#define LGKM0() asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); __builtin_amdgcn_sched_barrier(0)
#define END_INTERVAL(more)                                                                                        \
  if (more) wait_vmcnt<6>(); else wait_vmcnt<0>();                                                                \
  __builtin_amdgcn_sched_barrier(0);                                                                              \
  __builtin_amdgcn_s_barrier();                                                                                   \
  asm volatile("" ::: "memory");                                                                                  \
  __builtin_amdgcn_sched_barrier(0)

  stage(0, 0); stage(2, 0); stage(3, 0); stage(1, 0);
  if (nk > 1) { stage(0, 1); stage(2, 1); wait_vmcnt<8>(); } else { wait_vmcnt<4>(); }
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");
  if (wr == 0) {
    RD_B(fb0, ring, 0)
    RD_A(ring, 0)
  }
  if (nk > 1) wait_vmcnt<6>(); else wait_vmcnt<2>();
  __builtin_amdgcn_sched_barrier(0);
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");
  __builtin_amdgcn_sched_barrier(0);

  auto main_loop = [&](auto role_c) {
  constexpr int ROLE = decltype(role_c)::value;
  for (int kt = 0; kt < nk; ++kt) {
    const half_t* buf = ring + (kt & 1) * 4 * HT;
    const half_t* nbuf = ring + ((kt + 1) & 1) * 4 * HT;
    const bool more1 = kt + 1 < nk, more2 = kt + 2 < nk;
    // ---- phase 1: quadrant (A0, B0); DMA B1(kt+1) ---------------------------------------------------------------
    if (ROLE == 0) {
      LGKM0();
      MMA_Q(0, 0, fb0)
      __builtin_amdgcn_sched_barrier(0);
      RD_B(fb1, buf, 1)
      if (more1) stage(3, kt + 1);
    } else {
      RD_B(fb0, buf, 0)
      RD_A(buf, 0)
      if (more1) stage(3, kt + 1);
      LGKM0();
      MMA_Q(0, 0, fb0)
    }
    END_INTERVAL(more2);
    // ---- phase 2: quadrant (A0, B1); DMA A1(kt+1) ---------------------------------------------------------------
    if (ROLE == 0) {
      LGKM0();
      MMA_Q(0, 1, fb1)
      __builtin_amdgcn_sched_barrier(0);
      RD_A(buf, 1)
      if (more1) stage(1, kt + 1);
    } else {
      RD_B(fb1, buf, 1)
      if (more1) stage(1, kt + 1);
      LGKM0();
      MMA_Q(0, 1, fb1)
    }
    END_INTERVAL(more2);
    // ---- phase 3: quadrant (A1, B1); DMA A0(kt+2) ---------------------------------------------------------------
    if (ROLE == 0) {
      LGKM0();
      MMA_Q(1, 1, fb1)
      __builtin_amdgcn_sched_barrier(0);
      if (more2) stage(0, kt + 2);
    } else {
      RD_A(buf, 1)
      if (more2) stage(0, kt + 2);
      LGKM0();
      MMA_Q(1, 1, fb1)
    }
    END_INTERVAL(more2);
    // ---- phase 4: quadrant (A1, B0); DMA B0(kt+2); row 0 reads (A0, B0) of K-tile kt+1 ---------------------------
    if (ROLE == 0) {
      MMA_Q(1, 0, fb0)
      __builtin_amdgcn_sched_barrier(0);
      if (more1) {
        RD_B(fb0, nbuf, 0)
        RD_A(nbuf, 0)
      }
      if (more2) stage(2, kt + 2);
    } else {
      if (more2) stage(2, kt + 2);
      MMA_Q(1, 0, fb0)
    }
    END_INTERVAL(more2);
  }
  };
  if (wr == 0) main_loop(std::integral_constant<int, 0>{}); else main_loop(std::integral_constant<int, 1>{});
#undef RD_A
#undef RD_B
#undef MMA_Q
#undef LGKM0
#undef END_INTERVAL

  float* slab = reinterpret_cast<float*>(ring) + wv * 4096;
  store_wave_tile_128x64<EPI>(acc, slab, m0 + wr * 128, n0 + wc * 64, lane, p);
}

template <int EPI>
static void launch8h(const GemmArgs& p, hipStream_t s) {
  constexpr int LDS = 2 * 4 * 128 * 64 * 2;
  static bool attr = false;
  if (!attr) {
    (void)hipFuncSetAttribute((const void*)gemm8h_f16_kernel<EPI>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS);
    attr = true;
  }
  const int ntm = (p.M + 255) / 256, ntn = p.N / 256;
  hipLaunchKernelGGL((gemm8h_f16_kernel<EPI>), dim3(tile_map_grid(ntm, ntn, p.map_mode)), dim3(512), LDS, s, p);
}


// =====================================================================================================
// Tile 9 (experimental): 256 x 256 x 64, FOUR waves (2 x 2), wave tile 128 x 128 = 4 x 4 MFMA 32x32x16 tiles (256
// accumulator registers, one wave per SIMD), operands staged through REGISTERS (global_load_dwordx4 -> ds_write_b128) into
// two 64 KiB LDS buffers. Rationale: with one wave per SIMD nothing else can cover an issue stall, so the cheap-to-issue
// plain loads (the data lands asynchronously) replace the direct-to-LDS DMA (tens of cycles of issue each), and the
// 128 x 128 wave tile needs a third less LDS read traffic per FLOP than 128 x 64. One barrier per K-tile.
template <int EPI>
__global__ __launch_bounds__(256) void gemm4w_f16_kernel(GemmArgs p) {
  extern __shared__ __attribute__((aligned(16))) half_t ring[];  // [buf 2][A | B][256][64]
  constexpr int TE = 256 * 64;                                     // halfs per operand tile
  const int ntn = p.N / 256;
  const int ntm = (p.M + 255) / 256;
  int tm, tn;
  if (!tile_map(blockIdx.x, ntm, ntn, p.map_mode, tm, tn)) return;
  const int m0 = tm * 256, n0 = tn * 256;
  const int t = threadIdx.x, lane = t & 63;
  const int wv = __builtin_amdgcn_readfirstlane(t >> 6);
  const int wr = wv >> 1, wc = wv & 1;
  const int lr = lane & 31, lg = lane >> 5;

  // staging: piece j (0..7) of an operand tile = rows j*32 + t/8, 16-byte slot t%8 of the 128-byte row; the LDS image is
  // XOR-swizzled on the write (slot ^ ((row>>1)&7)) exactly as the fragment reads expect
  const int srow = t >> 3, sslot = t & 7;
  unsigned aoff[8], boff[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    int am = m0 + j * 32 + srow;
    am = am < p.M ? am : p.M - 1;
    aoff[j] = (unsigned)am * (unsigned)p.lda + sslot * 8;
    boff[j] = (unsigned)(n0 + j * 32 + srow) * (unsigned)p.ldw + sslot * 8;
  }
  int soff[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) soff[j] = lds_off64(j * 32 + srow, sslot);

  f32x16 acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  const int nk = p.K / 64;
  // The staging loads are inline asm: written as plain C++ loads hipcc either sinks them next to their ds_writes (latency
  // fully exposed) or, when pinned at the top with a sched_barrier, parks them in scratch. As asm the 16 requests are issued
  // at the top of the iteration, fly under the 64 MFMAs, and one explicit s_waitcnt ahead of the LDS writes orders them.
  typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
  u32x4 ra[8], rb[8];
  unsigned abyte[8], bbyte[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) { abyte[j] = aoff[j] * 2u; bbyte[j] = boff[j] * 2u; }
  // The staging loads are inline asm: written as plain C++ loads hipcc either sinks them next to their ds_writes (latency
  // fully exposed) or, when pinned with a sched_barrier, parks them in scratch. Each request produces a fresh value that is
  // consumed (s_waitcnt, ds_write) in the SAME iteration: a value in flight across the loop back-edge can be "copied" by a
  // compiler-inserted v_mov before it has landed (tried: requests re-issued right after the LDS writes = one full iteration
  // of cover and 3-4 % faster, but wrong results).
#define G4W_LD(j, kb)                                                                                              \
  asm volatile("global_load_dwordx4 %0, %1, %2" : "=&v"(ra[j]) : "v"(abyte[j] + (kb)), "s"(p.A) : "memory");       \
  asm volatile("global_load_dwordx4 %0, %1, %2" : "=&v"(rb[j]) : "v"(bbyte[j] + (kb)), "s"(p.W) : "memory");
#define G4W_WAIT()                                                                                                 \
  asm volatile("s_waitcnt vmcnt(0)"                                                                                \
               : "+v"(ra[0]), "+v"(ra[1]), "+v"(ra[2]), "+v"(ra[3]), "+v"(ra[4]), "+v"(ra[5]), "+v"(ra[6]), "+v"(ra[7]),  \
                 "+v"(rb[0]), "+v"(rb[1]), "+v"(rb[2]), "+v"(rb[3]), "+v"(rb[4]), "+v"(rb[5]), "+v"(rb[6]), "+v"(rb[7])   \
               :: "memory")
#pragma unroll
  for (int j = 0; j < 8; ++j) { G4W_LD(j, 0u) }
  G4W_WAIT();
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    *reinterpret_cast<u32x4*>(ring + soff[j]) = ra[j];
    *reinterpret_cast<u32x4*>(ring + TE + soff[j]) = rb[j];
  }
  __syncthreads();
  const int arow = wr * 128 + lr, brow = wc * 128 + lr;
  for (int kt = 0; kt < nk; ++kt) {
    const half_t* sa = ring + (kt & 1) * 2 * TE;
    const half_t* sb = sa + TE;
    half_t* da = ring + ((kt + 1) & 1) * 2 * TE;
    // next K-tile -> registers (the last iteration re-reads its own tile: no branch around the loads). The 16 requests are
    // issued in the shadows of k-substep 0's MFMAs, the 16 LDS writes in those of k-substep 3 (by then the data has had
    // ~1500 cycles to land): with one wave per SIMD an instruction only costs matrix-pipe time if no MFMA follows it soon
    // enough. Fragments of k-substep s2+1 are requested before the 16 MFMAs of s2 (two register sets).
    const unsigned kb = (unsigned)(kt + 1 < nk ? kt + 1 : kt) * 128u;
    half8_t fa[2][4], fb[2][4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      fa[0][i] = *reinterpret_cast<const half8_t*>(&sa[lds_off64(arow + i * 32, lg)]);
      fb[0][i] = *reinterpret_cast<const half8_t*>(&sb[lds_off64(brow + i * 32, lg)]);
    }
#pragma unroll
    for (int s2 = 0; s2 < 4; ++s2) {
      if (s2 < 3) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          fa[(s2 + 1) & 1][i] = *reinterpret_cast<const half8_t*>(&sa[lds_off64(arow + i * 32, (s2 + 1) * 2 + lg)]);
          fb[(s2 + 1) & 1][i] = *reinterpret_cast<const half8_t*>(&sb[lds_off64(brow + i * 32, (s2 + 1) * 2 + lg)]);
        }
      }
      __builtin_amdgcn_sched_barrier(0);
      if (s2 == 3) G4W_WAIT();
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        if (s2 == 0) {
#pragma unroll
          for (int j = 2 * i; j < 2 * i + 2; ++j) { G4W_LD(j, kb) }
        }
        if (s2 == 3) {
#pragma unroll
          for (int j = 2 * i; j < 2 * i + 2; ++j) {
            *reinterpret_cast<u32x4*>(da + soff[j]) = ra[j];
            *reinterpret_cast<u32x4*>(da + TE + soff[j]) = rb[j];
          }
        }
#pragma unroll
        for (int j = 0; j < 4; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fb[s2 & 1][j], fa[s2 & 1][i], acc[i][j], 0, 0, 0);  // D^T
        __builtin_amdgcn_sched_barrier(0);
      }
    }
    __syncthreads();
  }
#undef G4W_LD
#undef G4W_WAIT

  float* slab = reinterpret_cast<float*>(ring) + wv * 4096;
#pragma unroll
  for (int i = 0; i < 4; i += 2)
#pragma unroll
    for (int j = 0; j < 4; j += 2)
      store_slab_staged<EPI, false>(acc[i][j], acc[i][j + 1], acc[i + 1][j], acc[i + 1][j + 1], slab,
                                    m0 + wr * 128 + i * 32, n0 + wc * 128 + j * 32, lane, p);
}

template <int EPI>
static void launch4w(const GemmArgs& p, hipStream_t s) {
  constexpr int LDS = 2 * 2 * 256 * 64 * 2;
  static bool attr = false;
  if (!attr) {
    (void)hipFuncSetAttribute((const void*)gemm4w_f16_kernel<EPI>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS);
    attr = true;
  }
  const int ntm = (p.M + 255) / 256, ntn = p.N / 256;
  hipLaunchKernelGGL((gemm4w_f16_kernel<EPI>), dim3(tile_map_grid(ntm, ntn, p.map_mode)), dim3(256), LDS, s, p);
}


// =====================================================================================================
// Tile 10: the 8-phase kernel (tile 7: geometry, LDS image, staggered wave rows, two barriers per phase) with the K-tile cut
// the other way: a phase is one 64-row x 64-column half of the wave tile over HALF the K-tile (2 k-substeps), i.e. 8 MFMAs on
// FOUR independent accumulators (tile 7: two accumulators x 4 k-substeps, every MFMA depending on the one two slots earlier),
// and the fragment reads are 8 / 8 / 4 / 4 per phase instead of 12 / 4 / 8 / 0:
//   phase 1: (A0 k01) x (B0 B1 k01)   phase 2: (A0 k23) x (B0 B1 k23)   phase 3: (A1 k01) x kept B k01   phase 4: (A1 k23) x kept B k23
// Half-tiles A0, B0, B1 are last read in phase 2, A1 in phase 4; DMA (one half-tile per phase, >= 2 phases after its last
// read, consumption order):  phase 1(t): B0(t+1)   phase 2(t): B1(t+1)   phase 3(t): A1(t+1)   phase 4(t): A0(t+2)
// Waits: phase 4 leaves the two newest half-tiles in flight (vmcnt 4: A0 B0 B1 of t+1 have landed for phase 1),
// phases 1-3 leave three (vmcnt 6: by phase 2 this retires A1(t) for phase 3).
template <int EPI>
__global__ __launch_bounds__(512) void gemm8k_f16_kernel(GemmArgs p) {
  extern __shared__ __attribute__((aligned(16))) half_t ring[];  // [buf 2][A0 A1 B0 B1][128][64]
  constexpr int HT = 128 * 64;
  const int ntn = p.N / 256;
  const int ntm = (p.M + 255) / 256;
  int tm, tn;
  if (!tile_map(blockIdx.x, ntm, ntn, p.map_mode, tm, tn)) return;
  const int m0 = tm * 256, n0 = tn * 256;
  const int t = threadIdx.x, lane = t & 63;
  const int wv = __builtin_amdgcn_readfirstlane(t >> 6);
  const int wr = wv >> 2, wc = wv & 3;
  const int lr = lane & 31, lg = lane >> 5;
  if (p.dbg & 32) return;
  unsigned long long tr0 = 0, tr1 = 0, tr2 = 0;
  unsigned long long cy0 = 0;
  if (p.trace) { tr0 = wall_clock64(); cy0 = __builtin_amdgcn_s_memtime(); }
  if (p.stagger > 0 && blockIdx.x < 256) {
    const int n = (int)(((blockIdx.x * 167u) & 255u) * (unsigned)p.stagger) >> 8;
    for (int i = 0; i < n; ++i) __builtin_amdgcn_s_sleep(8);
  }

  unsigned aoff[2][2], boff[2];
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int lrow = j * 64 + (t >> 3);
    const int chunk = (t & 7) ^ ((lrow >> 1) & 7);
    const int arow = (lrow >> 6) * 128 + (lrow & 63);
    const int brow = (lrow >> 5) * 64 + (lrow & 31);
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      int am = m0 + arow + h * 64;
      am = am < p.M ? am : p.M - 1;
      aoff[j][h] = (unsigned)am * (unsigned)p.lda + chunk * 8;
    }
    boff[j] = (unsigned)(n0 + brow) * (unsigned)p.ldw + chunk * 8;
  }
  const unsigned bh = 32u * (unsigned)p.ldw;
  auto stage = [&](int which, int kt) {  // 0 A0, 1 A1, 2 B0, 3 B1
    half_t* dst = ring + ((kt & 1) * 4 + which) * HT + wv * 512;
    const int h = which & 1;
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const half_t* g = which < 2 ? p.A + (aoff[j][h] + (unsigned)kt * 64u) : p.W + (boff[j] + h * bh + (unsigned)kt * 64u);
      glds16(g, dst + j * 4096);
    }
  };

  f32x16 acc[2][2][2];  // [a][i][b]
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int b = 0; b < 2; ++b)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[a][i][b][r] = 0.f;

  half8_t fa[2][2];      // [i][k within the half]
  half8_t fb[2][2][2];   // [k half][b][k within the half]: both halves stay resident for phases 3 / 4
  const int arow0 = wr * 64 + lr, brow0 = wc * 32 + lr;
  const int nk = p.K / 64;

#define RD_A(bufp, h, kh)                                                                                         \
  _Pragma("unroll") for (int i = 0; i < 2; ++i) _Pragma("unroll") for (int k2 = 0; k2 < 2; ++k2)                   \
      fa[i][k2] = *reinterpret_cast<const half8_t*>(&(bufp)[(h) * HT + lds_off64(arow0 + i * 32, ((kh) * 2 + k2) * 2 + lg)]);
#define RD_B(bufp, kh)                                                                                            \
  _Pragma("unroll") for (int b = 0; b < 2; ++b) _Pragma("unroll") for (int k2 = 0; k2 < 2; ++k2)                   \
      fb[kh][b][k2] = *reinterpret_cast<const half8_t*>(&(bufp)[(2 + b) * HT + lds_off64(brow0, ((kh) * 2 + k2) * 2 + lg)]);
#define MMA_H(a, kh)                                                                                              \
  _Pragma("unroll") for (int k2 = 0; k2 < 2; ++k2) _Pragma("unroll") for (int i = 0; i < 2; ++i)                   \
      _Pragma("unroll") for (int b = 0; b < 2; ++b)                                                               \
          acc[a][i][b] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fb[kh][b][k2], fa[i][k2], acc[a][i][b], 0, 0, 0);
#define PHASE_SYNC_IN()                                   \
  __builtin_amdgcn_s_barrier();                           \
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");      \
  __builtin_amdgcn_sched_barrier(0);                      \
  __builtin_amdgcn_s_setprio(1);
#define PHASE_SYNC_OUT()                                  \
  __builtin_amdgcn_s_setprio(0);                          \
  __builtin_amdgcn_sched_barrier(0);                      \
  __builtin_amdgcn_s_barrier();                           \
  asm volatile("" ::: "memory");

  stage(0, 0); stage(2, 0); stage(3, 0); stage(1, 0);
  if (nk > 1) { stage(0, 1); wait_vmcnt<4>(); } else { wait_vmcnt<2>(); }
  __builtin_amdgcn_s_barrier();
  if (wr == 1) __builtin_amdgcn_s_barrier();  // stagger the second wave row by one barrier
  asm volatile("" ::: "memory");
  if (p.trace) tr1 = wall_clock64();

  for (int kt = 0; kt < nk; ++kt) {
    const half_t* buf = ring + (kt & 1) * 4 * HT;
    const bool more1 = kt + 1 < nk, more2 = kt + 2 < nk;
    // ---- phase 1 ---------------------------------------------------------------------------------------------------
    RD_B(buf, 0)
    __builtin_amdgcn_sched_barrier(0);
    RD_A(buf, 0, 0)
    if (more1) stage(2, kt + 1);
    if (more1) wait_vmcnt<6>(); else wait_vmcnt<0>();
    PHASE_SYNC_IN();
    MMA_H(0, 0)
    PHASE_SYNC_OUT();
    // ---- phase 2 ---------------------------------------------------------------------------------------------------
    RD_B(buf, 1)
    __builtin_amdgcn_sched_barrier(0);
    RD_A(buf, 0, 1)
    if (more1) stage(3, kt + 1);
    if (more1) wait_vmcnt<6>(); else wait_vmcnt<0>();
    PHASE_SYNC_IN();
    MMA_H(0, 1)
    PHASE_SYNC_OUT();
    // ---- phase 3 ---------------------------------------------------------------------------------------------------
    RD_A(buf, 1, 0)
    if (more1) stage(1, kt + 1);
    if (more1) wait_vmcnt<6>(); else wait_vmcnt<0>();
    PHASE_SYNC_IN();
    MMA_H(1, 0)
    PHASE_SYNC_OUT();
    // ---- phase 4 ---------------------------------------------------------------------------------------------------
    RD_A(buf, 1, 1)
    if (more2) { stage(0, kt + 2); wait_vmcnt<4>(); } else if (more1) { wait_vmcnt<2>(); } else { wait_vmcnt<0>(); }
    PHASE_SYNC_IN();
    MMA_H(1, 1)
    PHASE_SYNC_OUT();
  }
#undef RD_A
#undef RD_B
#undef MMA_H
#undef PHASE_SYNC_IN
#undef PHASE_SYNC_OUT
  if (wr == 0) __builtin_amdgcn_s_barrier();  // balance the stagger
  if (p.trace) tr2 = wall_clock64();
  if (p.dbg & 8) { if (acc[0][0][0][0] == 123.456f) reinterpret_cast<half_t*>(p.out)[0] = (half_t)acc[1][1][1][3]; return; }

  float* slab = reinterpret_cast<float*>(ring) + wv * 4096;
  if (p.dbg & 64) {   // LDS part of the epilogue only
    float sum = 0.f;
#pragma unroll
    for (int a = 0; a < 2; ++a) {
      slab_park(acc[a][0][0], acc[a][0][1], acc[a][1][0], acc[a][1][1], slab, lane);
#pragma unroll
      for (int it = 0; it < 16; ++it) {
        const int row = it * 4 + (lane >> 4);
        float4 v = *reinterpret_cast<const float4*>(&slab[row * 64 + (((lane & 15) ^ (row & 15)) << 2)]);
        sum += v.x + v.y + v.z + v.w;
      }
    }
    if (sum == 123.456f) reinterpret_cast<half_t*>(p.out)[0] = (half_t)sum;
    return;
  }
  if (p.dbg & 128) {  // direct stores from the accumulator layout
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int b = 0; b < 2; ++b)
          store_acc32<EPI>(acc[a][i][b], m0 + wr * 128 + a * 64 + i * 32 + lr, n0 + wc * 64 + b * 32 + 4 * lg, p);
    return;
  }
  store_wave_tile_128x64<EPI>(acc, slab, m0 + wr * 128, n0 + wc * 64, lane, p);
  if (p.trace && lane == 0) {
    const unsigned long long tr3 = wall_clock64();
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    const unsigned long long tr4 = wall_clock64();
    unsigned hw;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
    unsigned xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    unsigned long long* o = p.trace + ((size_t)blockIdx.x * 8 + wv) * 8;
    o[0] = tr0; o[1] = tr1; o[2] = tr2; o[3] = tr3; o[4] = tr4; o[5] = hw; o[6] = xcc; o[7] = __builtin_amdgcn_s_memtime() - cy0;
  }
}

template <int EPI>
static void launch8k(const GemmArgs& p, hipStream_t s) {
  constexpr int LDS = 2 * 4 * 128 * 64 * 2;
  static bool attr = false;
  if (!attr) {
    (void)hipFuncSetAttribute((const void*)gemm8k_f16_kernel<EPI>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS);
    attr = true;
  }
  const int ntm = (p.M + 255) / 256, ntn = p.N / 256;
  hipLaunchKernelGGL((gemm8k_f16_kernel<EPI>), dim3(tile_map_grid(ntm, ntn, p.map_mode)), dim3(512), LDS, s, p);
}

// ---------------------------------------------------------------------------------------------------------------------
// Tile 11: the 8-phase kernel above made PERSISTENT for the fp16-output epilogues. One workgroup per CU walks tiles
// bid, bid + grid, ... of the same XCD-aware map. What it buys (per-workgroup timeline, tools/gemm_trace.py, 65536x3840x1280:
// prologue 2.3 us + k-loop 35.8 us + epilogue 4.2 us + 0.9 us until the CU's next workgroup starts): the next tile's first
// five half-tile DMAs are issued BEFORE the epilogue of the finished tile and land while its stores drain, and there is no
// workgroup turn-around. The epilogue therefore cannot park its slabs in the ring: each wave owns 4 KiB of the 32 KiB that
// the 128 KiB ring leaves free and emits its 128x64 tile as four 32-row fp16 slabs (4 dwordx4 stores each).
template <bool LN>
__device__ __forceinline__ void slab_park16h(const f32x16 (&a0), const f32x16 (&a1), half_t* __restrict__ slab,
                                             const float4 (&bv)[2][4], int lane, bool gelu, const float4 (&sv)[2][4],
                                             float2 mr) {
  const int lr = lane & 31, lg = lane >> 5;
  const f32x16* accs[2] = {&a0, &a1};
#pragma unroll
  for (int j = 0; j < 2; ++j)
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const f32x16& a = *accs[j];
      const float4 b = bv[j][q];
      float4 v;
      if (LN) {   // folded LayerNorm, consumer side: rstd * (acc - mean * s[n]) + bias'[n]
        const float4 sn = sv[j][q];
        v = make_float4(fmaf(fmaf(-mr.x, sn.x, a[4 * q]), mr.y, b.x), fmaf(fmaf(-mr.x, sn.y, a[4 * q + 1]), mr.y, b.y),
                        fmaf(fmaf(-mr.x, sn.z, a[4 * q + 2]), mr.y, b.z), fmaf(fmaf(-mr.x, sn.w, a[4 * q + 3]), mr.y, b.w));
      } else {
        v = make_float4(a[4 * q] + b.x, a[4 * q + 1] + b.y, a[4 * q + 2] + b.z, a[4 * q + 3] + b.w);
      }
      if (gelu) { v.x = gelu_erf(v.x); v.y = gelu_erf(v.y); v.z = gelu_erf(v.z); v.w = gelu_erf(v.w); }
      half4_t h = {(half_t)v.x, (half_t)v.y, (half_t)v.z, (half_t)v.w};
      *reinterpret_cast<half4_t*>(&slab[lr * 64 + (((j * 4 + q) ^ ((lr >> 1) & 7)) << 3) + 4 * lg]) = h;
    }
}

__device__ __forceinline__ void slab_emit16h(const half_t* __restrict__ slab, int mbase, int nbase, int lane,
                                             const GemmArgs& p) {
  const int c8 = lane & 7;
  const int n = nbase + c8 * 8;
  int pl = 0, ncol = n;
  if (p.head_hd) { pl = n / p.head_hd; ncol = n - pl * p.head_hd; }
#pragma unroll
  for (int it = 0; it < 4; ++it) {
    const int row = it * 8 + (lane >> 3);
    const int m = mbase + row;
    const half8_t v = *reinterpret_cast<const half8_t*>(&slab[row * 64 + ((c8 ^ ((row >> 1) & 7)) << 3)]);
    if (m >= p.M) continue;
    half_t* dst;
    if (p.head_hd) {
      dst = reinterpret_cast<half_t*>(p.out) + ((size_t)pl * p.M + m) * p.head_hd + ncol;
    } else {
      const size_t orow = p.out_seg ? (size_t)(m / p.out_seg) * p.out_seg_stride + p.out_seg_off + (m % p.out_seg) : (size_t)m;
      dst = reinterpret_cast<half_t*>(p.out) + orow * p.ldo + n;
    }
    *reinterpret_cast<half8_t*>(dst) = v;
  }
}

// Epilogue of one 128x64 wave tile of the persistent kernels, through the wave's private 4 KiB slab.
struct NoNext { __device__ __forceinline__ void operator()() const {} };
// `issue_next` (tile 11): requests the next tile's first half-tiles. It is called AFTER the epilogue's own first global loads
// (bias / LayerScale / first residual rows) have been issued: loads complete in order, so requested behind 80 KiB of DMA they
// would hold the whole epilogue until those tiles have landed.
template <int EPI, bool LNF = false, bool SPLITK = false, class NEXT = NoNext>
__device__ __forceinline__ void persist_epilogue(const f32x16 (&acc)[2][2][2], half_t* __restrict__ slab, int mbase, int nbase,
                                                 int lane, const GemmArgs& p, int krange = 0, NEXT issue_next = NEXT()) {
  const int lr = lane & 31, lg = lane >> 5;
  if constexpr (EPI == EPI_F32) {
    // x (+)= gamma * (acc + bias) in fp32: eight 32x32 blocks through the 4 KiB slab, 8 lanes per 128-byte row; the
    // residual rows of block s+1 are requested before block s is emitted
    float* slabf = reinterpret_cast<float*>(slab);
    const int c = lane & 7, r8 = lane >> 3;
    auto pre = [&](int blk, float4 (&r)[4]) {
      const int mb = mbase + (blk >> 1) * 32, n = nbase + (blk & 1) * 32 + c * 4;
#pragma unroll
      for (int it = 0; it < 4; ++it) {
        int m = mb + it * 8 + r8;
        m = m < p.M ? m : p.M - 1;
        const size_t rrow = p.resid_mod ? (size_t)(m % p.resid_mod) : (size_t)m;
        r[it] = (p.resid && !SPLITK) ? *reinterpret_cast<const float4*>(p.resid + rrow * p.ldr + n) : make_float4(0.f, 0.f, 0.f, 0.f);
      }
    };
    float4 bvv[2], gvv[2];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      bvv[j] = (p.bias && !SPLITK) ? *reinterpret_cast<const float4*>(p.bias + nbase + j * 32 + c * 4) : make_float4(0.f, 0.f, 0.f, 0.f);
      gvv[j] = p.gamma ? *reinterpret_cast<const float4*>(p.gamma + nbase + j * 32 + c * 4) : make_float4(1.f, 1.f, 1.f, 1.f);
    }
    float4 ra[4], rb[4];
    float ps1[4] = {0.f, 0.f, 0.f, 0.f}, ps2[4] = {0.f, 0.f, 0.f, 0.f};   // folded LayerNorm: row sums over the wave's 64 columns
    half4_t hx[2][4];                                                      // and half(x) of the current 32-row strip
    pre(0, ra);
#pragma unroll
    for (int blk = 0; blk < 8; ++blk) {
      const f32x16& a = acc[blk >> 2][(blk >> 1) & 1][blk & 1];
#pragma unroll
      for (int q = 0; q < 4; ++q)
        *reinterpret_cast<float4*>(&slabf[lr * 32 + (((2 * q + lg) ^ ((lr >> 1) & 7)) << 2)]) =
            make_float4(a[4 * q], a[4 * q + 1], a[4 * q + 2], a[4 * q + 3]);
      __builtin_amdgcn_sched_barrier(0);
      if (blk < 7) pre(blk + 1, (blk & 1) ? ra : rb);
      __builtin_amdgcn_sched_barrier(0);
#ifndef PSAM_EPI32_NEXT_AT
#define PSAM_EPI32_NEXT_AT 0   // block after whose residual request the next tile's first DMAs go out. Loads complete in order, so
#endif                         // residual rows requested behind 80 KiB of DMA wait for it; issuing the DMA only after the LAST request
                               // (6) measured slightly slower than right away (997 vs 1001 TFLOP/s in the pipeline): the DMA's own
                               // latency is the longer pole
      if (blk == PSAM_EPI32_NEXT_AT) {
        issue_next();
        __builtin_amdgcn_sched_barrier(0);
      }
      const float4 (&r)[4] = (blk & 1) ? rb : ra;
      const float4 bv = bvv[blk & 1], gv = gvv[blk & 1];
      const int mb = mbase + (blk >> 1) * 32, n = nbase + (blk & 1) * 32 + c * 4;
#pragma unroll
      for (int it = 0; it < 4; ++it) {
        const int row = it * 8 + r8;
        const int m = mb + row;
        float4 v = *reinterpret_cast<const float4*>(&slabf[row * 32 + ((c ^ ((row >> 1) & 7)) << 2)]);
        if (m >= p.M) continue;
        const size_t orow = p.out_seg ? (size_t)(m / p.out_seg) * p.out_seg_stride + p.out_seg_off + (m % p.out_seg) : (size_t)m;
        if constexpr (!SPLITK) {
          v.x = (v.x + bv.x) * gv.x + r[it].x; v.y = (v.y + bv.y) * gv.y + r[it].y;
          v.z = (v.z + bv.z) * gv.z + r[it].z; v.w = (v.w + bv.w) * gv.w + r[it].w;
        }
        if constexpr (SPLITK) {   // raw partial sums of this K range (bias, gamma and the residual are the reduce kernel's)
          *reinterpret_cast<float4*>(p.ks_ws + ((size_t)krange * p.M + m) * p.N + n) = v;
        } else {
          *reinterpret_cast<float4*>(reinterpret_cast<float*>(p.out) + orow * p.ldo + n) = v;
        }
        if (LNF && p.out16) hx[blk & 1][it] = half4_t{(half_t)v.x, (half_t)v.y, (half_t)v.z, (half_t)v.w};
        if (LNF && p.stats) {
          ps1[it] += (v.x + v.y) + (v.z + v.w);
          ps2[it] += (v.x * v.x + v.y * v.y) + (v.z * v.z + v.w * v.w);
        }
      }
      if (LNF && p.out16 && (blk & 1)) {
        // half(x) of this 32-row x 64-column strip: through the (consumed) slab so that 8 lanes cover a row with 16 bytes each -
        // 4 dwordx4 stores instead of 8 dwordx2 (the epilogue is store-ISSUE-bound)
        half_t* slabh = reinterpret_cast<half_t*>(slabf);
#pragma unroll
        for (int jb = 0; jb < 2; ++jb)
#pragma unroll
          for (int it = 0; it < 4; ++it) {
            const int row = it * 8 + r8;
            *reinterpret_cast<half4_t*>(&slabh[row * 64 + (((jb * 4 + (c >> 1)) ^ ((row >> 1) & 7)) << 3) + (c & 1) * 4]) =
                hx[jb][it];
          }
#pragma unroll
        for (int it = 0; it < 4; ++it) {
          const int row = it * 8 + r8;
          const int m = mbase + (blk >> 1) * 32 + row;
          const half8_t hv = *reinterpret_cast<const half8_t*>(&slabh[row * 64 + ((c ^ ((row >> 1) & 7)) << 3)]);
          if (m >= p.M) continue;
          const size_t orow = p.out_seg ? (size_t)(m / p.out_seg) * p.out_seg_stride + p.out_seg_off + (m % p.out_seg) : (size_t)m;
          *reinterpret_cast<half8_t*>(p.out16 + orow * p.ld16 + nbase + c * 8) = hv;
        }
      }
      if (LNF && p.stats && (blk & 1)) {   // both 32-column blocks of this 32-row strip are in: 8 lanes hold one row's 64 columns
#pragma unroll
        for (int it = 0; it < 4; ++it) {
          float s1 = ps1[it], s2 = ps2[it];
#pragma unroll
          for (int o = 1; o < 8; o <<= 1) { s1 += __shfl_xor(s1, o, 64); s2 += __shfl_xor(s2, o, 64); }
          const int m = mbase + (blk >> 1) * 32 + it * 8 + r8;
          const size_t orow = p.out_seg ? (size_t)(m / p.out_seg) * p.out_seg_stride + p.out_seg_off + (m % p.out_seg) : (size_t)m;
          if (c == 0 && m < p.M)
            *reinterpret_cast<float2*>(p.stats + (orow * (p.N >> 6) + (nbase >> 6)) * 2) = make_float2(s1, s2);
          ps1[it] = 0.f; ps2[it] = 0.f;
        }
      }
      __builtin_amdgcn_sched_barrier(0);
    }
  } else {  // epilogue of the finished tile from the wave's private slab
    float4 bv[2][4], sv[2][4];
    float2 mrv[4];
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        bv[j][q] = p.bias ? *reinterpret_cast<const float4*>(p.bias + nbase + j * 32 + 8 * q + 4 * lg)
                          : make_float4(0.f, 0.f, 0.f, 0.f);
        if (LNF) sv[j][q] = *reinterpret_cast<const float4*>(p.ln_s + nbase + j * 32 + 8 * q + 4 * lg);
      }
    if (LNF) {   // every request of the epilogue in flight at once (one exposed latency, as for the bias)
#pragma unroll
      for (int sidx = 0; sidx < 4; ++sidx) {
        int m = mbase + sidx * 32 + lr;
        m = m < p.M ? m : p.M - 1;
        mrv[sidx] = *reinterpret_cast<const float2*>(p.ln_mr + (size_t)m * 2);
      }
    }
    __builtin_amdgcn_sched_barrier(0);
    issue_next();
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int sidx = 0; sidx < 4; ++sidx) {
      if (LNF) {
        slab_park16h<true>(acc[sidx >> 1][sidx & 1][0], acc[sidx >> 1][sidx & 1][1], slab, bv, lane, EPI == EPI_GELU_F16, sv,
                           mrv[sidx]);
      } else {
        slab_park16h<false>(acc[sidx >> 1][sidx & 1][0], acc[sidx >> 1][sidx & 1][1], slab, bv, lane, EPI == EPI_GELU_F16, bv,
                            make_float2(0.f, 1.f));
      }
      slab_emit16h(slab, mbase + sidx * 32, nbase, lane, p);
    }
  }
}

template <int EPI, bool LNF = false, bool SPLITK = false>
__global__ __launch_bounds__(512) void gemm8kp_f16_kernel(GemmArgs p, int total) {
  extern __shared__ __attribute__((aligned(16))) half_t ring[];  // [buf 2][A0 A1 B0 B1][128][64] + 8 x 4 KiB slabs
  constexpr int HT = 128 * 64;
  const int ntn = p.N / 256;
  const int ntm = (p.M + 255) / 256;
  const int t = threadIdx.x, lane = t & 63;
  const int wv = __builtin_amdgcn_readfirstlane(t >> 6);
  const int wr = wv >> 2, wc = wv & 3;
  const int lr = lane & 31, lg = lane >> 5;
  half_t* slab = ring + 8 * HT + wv * 2048;
  const int nk = p.K / 64;
  const unsigned bh = 32u * (unsigned)p.ldw;
  const int arow0 = wr * 64 + lr, brow0 = wc * 32 + lr;

  // SPLITK: work item = (tile, K range): item j is range j % ksplit of tile j / ksplit; `total` counts items
  const int KSP = SPLITK ? p.ksplit : 1;
  int idx = blockIdx.x, tm = 0, tn = 0;
  while (idx < total && !tile_map(idx / KSP, ntm, ntn, p.map_mode, tm, tn)) idx += gridDim.x;
  if (idx >= total) return;
  int k_lo = SPLITK ? (idx % KSP) * nk / KSP : 0, k_hi = SPLITK ? (idx % KSP + 1) * nk / KSP : nk;

  unsigned aoff[2][2], boff[2];
  auto offsets = [&](int m0, int n0) {
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int lrow = j * 64 + (t >> 3);
      const int chunk = (t & 7) ^ ((lrow >> 1) & 7);
      const int arow = (lrow >> 6) * 128 + (lrow & 63);
      const int brow = (lrow >> 5) * 64 + (lrow & 31);
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        int am = m0 + arow + h * 64;
        am = am < p.M ? am : p.M - 1;
        aoff[j][h] = (unsigned)am * (unsigned)p.lda + chunk * 8;
      }
      boff[j] = (unsigned)(n0 + brow) * (unsigned)p.ldw + chunk * 8;
    }
  };
  auto stage = [&](int which, int kt) {  // 0 A0, 1 A1, 2 B0, 3 B1
    half_t* dst = ring + ((kt & 1) * 4 + which) * HT + wv * 512;
    const int h = which & 1;
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const half_t* g = which < 2 ? p.A + (aoff[j][h] + (unsigned)kt * 64u) : p.W + (boff[j] + h * bh + (unsigned)kt * 64u);
      glds16(g, dst + j * 4096);
    }
  };
  auto prologue = [&](int k0, int k1) {
    stage(0, k0); stage(2, k0); stage(3, k0); stage(1, k0);
    if (k1 - k0 > 1) stage(0, k0 + 1);
  };

  if (p.stagger > 0) {
    const int n = (int)(((blockIdx.x * 167u) & 255u) * (unsigned)p.stagger) >> 8;
    for (int i = 0; i < n; ++i) __builtin_amdgcn_s_sleep(8);
  }
  offsets(tm * 256, tn * 256);
  prologue(k_lo, k_hi);

  half8_t fa[2][2];      // [i][k within the half]
  half8_t fb[2][2][2];   // [k half][b][k within the half]

#define RD_A(bufp, h, kh)                                                                                         \
  _Pragma("unroll") for (int i = 0; i < 2; ++i) _Pragma("unroll") for (int k2 = 0; k2 < 2; ++k2)                   \
      fa[i][k2] = *reinterpret_cast<const half8_t*>(&(bufp)[(h) * HT + lds_off64(arow0 + i * 32, ((kh) * 2 + k2) * 2 + lg)]);
#define RD_B(bufp, kh)                                                                                            \
  _Pragma("unroll") for (int b = 0; b < 2; ++b) _Pragma("unroll") for (int k2 = 0; k2 < 2; ++k2)                   \
      fb[kh][b][k2] = *reinterpret_cast<const half8_t*>(&(bufp)[(2 + b) * HT + lds_off64(brow0, ((kh) * 2 + k2) * 2 + lg)]);
#define MMA_H(a, kh)                                                                                              \
  _Pragma("unroll") for (int k2 = 0; k2 < 2; ++k2) _Pragma("unroll") for (int i = 0; i < 2; ++i)                   \
      _Pragma("unroll") for (int b = 0; b < 2; ++b)                                                               \
          acc[a][i][b] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fb[kh][b][k2], fa[i][k2], acc[a][i][b], 0, 0, 0);
#define PHASE_SYNC_IN()                                   \
  __builtin_amdgcn_s_barrier();                           \
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");      \
  __builtin_amdgcn_sched_barrier(0);                      \
  __builtin_amdgcn_s_setprio(1);
#define PHASE_SYNC_OUT()                                  \
  __builtin_amdgcn_s_setprio(0);                          \
  __builtin_amdgcn_sched_barrier(0);                      \
  __builtin_amdgcn_s_barrier();                           \
  asm volatile("" ::: "memory");

  bool first = true;
  for (;;) {
    const int m0 = tm * 256, n0 = tn * 256;
    f32x16 acc[2][2][2];  // [a][i][b]
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
          for (int r = 0; r < 16; ++r) acc[a][i][b][r] = 0.f;

    // first tile: the counted wait of the 8-phase prologue. Later tiles: the previous tile's stores were issued AFTER this
    // tile's first DMAs, and a count says nothing about which of loads and stores are still in flight, so wait for all
    // (the DMAs landed during the epilogue; stores are acknowledged ~0.1 us after issue)
    if (first) { if (k_hi - k_lo > 1) wait_vmcnt<4>(); else wait_vmcnt<2>(); } else { wait_vmcnt<0>(); }
    first = false;
    __builtin_amdgcn_s_barrier();
    if (wr == 1) __builtin_amdgcn_s_barrier();  // stagger the second wave row by one barrier
    asm volatile("" ::: "memory");

    for (int kt = k_lo; kt < k_hi; ++kt) {
      const half_t* buf = ring + (kt & 1) * 4 * HT;
      const bool more1 = kt + 1 < k_hi, more2 = kt + 2 < k_hi;
#define LOAD_PHASE(RDS, ST) RDS __builtin_amdgcn_sched_barrier(0); ST
      LOAD_PHASE(RD_B(buf, 0) __builtin_amdgcn_sched_barrier(0); RD_A(buf, 0, 0), if (more1) stage(2, kt + 1);)
      if (more1) wait_vmcnt<6>(); else wait_vmcnt<0>();
      PHASE_SYNC_IN();
      MMA_H(0, 0)
      PHASE_SYNC_OUT();
      LOAD_PHASE(RD_B(buf, 1) __builtin_amdgcn_sched_barrier(0); RD_A(buf, 0, 1), if (more1) stage(3, kt + 1);)
      if (more1) wait_vmcnt<6>(); else wait_vmcnt<0>();
      PHASE_SYNC_IN();
      MMA_H(0, 1)
      PHASE_SYNC_OUT();
      LOAD_PHASE(RD_A(buf, 1, 0), if (more1) stage(1, kt + 1);)
      if (more1) wait_vmcnt<6>(); else wait_vmcnt<0>();
      PHASE_SYNC_IN();
      MMA_H(1, 0)
      PHASE_SYNC_OUT();
      LOAD_PHASE(RD_A(buf, 1, 1), if (more2) stage(0, kt + 2);)
      if (more2) { wait_vmcnt<4>(); } else if (more1) { wait_vmcnt<2>(); } else { wait_vmcnt<0>(); }
      PHASE_SYNC_IN();
      MMA_H(1, 1)
      PHASE_SYNC_OUT();
#undef LOAD_PHASE
    }
    if (wr == 0) __builtin_amdgcn_s_barrier();  // balance the stagger: every wave is done with the ring here

    // next tile of this workgroup: request its first half-tiles now
    int nidx = idx + gridDim.x, ntm_ = 0, ntn_ = 0;
    while (nidx < total && !tile_map(nidx / KSP, ntm, ntn, p.map_mode, ntm_, ntn_)) nidx += gridDim.x;
    const bool have = nidx < total;
    const int nk_lo = SPLITK && have ? (nidx % KSP) * nk / KSP : 0, nk_hi = SPLITK && have ? (nidx % KSP + 1) * nk / KSP : nk;
    // (every wave is past the ring here: whichever slots the next item's first K-tiles map to are free)
    auto issue_next = [&]() {
      if (have) {
        offsets(ntm_ * 256, ntn_ * 256);
        prologue(nk_lo, nk_hi);
      }
    };
    __builtin_amdgcn_sched_barrier(0);
#ifndef PSAM_EPI_EARLY_NEXT
#define PSAM_EPI_EARLY_NEXT 1   // 0: the previous order (next tile's DMA before the epilogue's loads), for A/B builds
#endif
#if PSAM_EPI_EARLY_NEXT
    persist_epilogue<EPI, LNF, SPLITK>(acc, slab, m0 + wr * 128, n0 + wc * 64, lane, p, SPLITK ? idx % KSP : 0, issue_next);
#else
    issue_next();
    __builtin_amdgcn_sched_barrier(0);
    persist_epilogue<EPI, LNF, SPLITK>(acc, slab, m0 + wr * 128, n0 + wc * 64, lane, p, SPLITK ? idx % KSP : 0);
#endif
    if (!have) break;
    idx = nidx; tm = ntm_; tn = ntn_; k_lo = nk_lo; k_hi = nk_hi;
  }
#undef RD_A
#undef RD_B
#undef MMA_H
#undef PHASE_SYNC_IN
#undef PHASE_SYNC_OUT
}

template <int EPI, bool LNF = false>
static void launch8kp(const GemmArgs& p, hipStream_t s) {
  constexpr int LDS = 2 * 4 * 128 * 64 * 2 + 8 * 4096;
  static bool attr = false;
  if (!attr) {
    (void)hipFuncSetAttribute((const void*)gemm8kp_f16_kernel<EPI, LNF>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS);
    attr = true;
  }
  const int ntm = (p.M + 255) / 256, ntn = p.N / 256;
  GemmArgs q = p;
  q.map_mode = pick_map_mode(ntm, ntn);
  const int total = tile_map_grid(ntm, ntn, q.map_mode);
  hipLaunchKernelGGL((gemm8kp_f16_kernel<EPI, LNF>), dim3(total < num_cus() ? total : num_cus()), dim3(512), LDS, s, q, total);
}

// Split-K form of tile 11 for the residual update of a FEW tiles with a LONG K (one slice through fc2: 4096x1280x5120 = 80
// tiles of 256x256 for 256 CUs): `ksplit` workgroups per tile, each over nk / ksplit K-tiles, storing raw partial sums to a
// caller-registered workspace (psam_gemm_set_workspace); splitk_reduce_kernel then applies out = resid + gamma * (sum + bias)
// in a fixed order (deterministic; an fp32-atomic epilogue was measured 2.5x SLOWER than no split at all - device-scope float
// atomics are resolved beyond the per-XCD L2s).
__global__ __launch_bounds__(256) void splitk_reduce_kernel(const float* __restrict__ ws, int ks, const float* __restrict__ bias,
                                                            const float* __restrict__ gamma, const float* resid, int ldr,
                                                            int resid_mod, float* out, int ldo, int M, int N) {
  const int n4 = N >> 2;
  const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= (size_t)M * n4) return;
  const int m = (int)(i / n4), n = (int)(i - (size_t)m * n4) * 4;
  float4 v = bias ? *reinterpret_cast<const float4*>(bias + n) : make_float4(0.f, 0.f, 0.f, 0.f);
  for (int r = 0; r < ks; ++r) {
    const float4 q = *reinterpret_cast<const float4*>(ws + ((size_t)r * M + m) * N + n);
    v.x += q.x; v.y += q.y; v.z += q.z; v.w += q.w;
  }
  if (gamma) {
    const float4 g = *reinterpret_cast<const float4*>(gamma + n);
    v.x *= g.x; v.y *= g.y; v.z *= g.z; v.w *= g.w;
  }
  if (resid) {
    const size_t rrow = resid_mod ? (size_t)(m % resid_mod) : (size_t)m;
    const float4 r = *reinterpret_cast<const float4*>(resid + rrow * ldr + n);
    v.x += r.x; v.y += r.y; v.z += r.z; v.w += r.w;
  }
  *reinterpret_cast<float4*>(out + (size_t)m * ldo + n) = v;
}

static float* g_ks_ws = nullptr;
static size_t g_ks_ws_bytes = 0;
extern "C" int psam_gemm_set_workspace(void* ptr, size_t bytes) {   // device scratch for the split-K partial sums (null: no split-K)
  if (ptr && (reinterpret_cast<uintptr_t>(ptr) & 15)) return PSAM_ERR_ARG;
  g_ks_ws = (float*)ptr;
  g_ks_ws_bytes = ptr ? bytes : 0;
  return PSAM_OK;
}

static void launch8kp_splitk(const GemmArgs& p, hipStream_t s) {
  constexpr int LDS = 2 * 4 * 128 * 64 * 2 + 8 * 4096;
  static bool attr = false;
  if (!attr) {
    (void)hipFuncSetAttribute((const void*)gemm8kp_f16_kernel<EPI_F32, false, true>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS);
    attr = true;
  }
  const int ntm = (p.M + 255) / 256, ntn = p.N / 256;
  GemmArgs q = p;
  q.map_mode = 1;   // identity map: the items of one tile are neighbours
  q.ks_ws = g_ks_ws;
  const int total = ntm * ntn * q.ksplit;
  hipLaunchKernelGGL((gemm8kp_f16_kernel<EPI_F32, false, true>), dim3(total < num_cus() ? total : num_cus()), dim3(512), LDS, s, q, total);
  const size_t n4 = (size_t)p.M * (p.N / 4);
  hipLaunchKernelGGL(splitk_reduce_kernel, dim3((unsigned)((n4 + 255) / 256)), dim3(256), 0, s, g_ks_ws, q.ksplit, p.bias, p.gamma,
                     p.resid, p.ldr, p.resid_mod, reinterpret_cast<float*>(p.out), p.ldo, p.M, p.N);
}

// ---------------------------------------------------------------------------------------------------------------------
// Tile 14: tile 11 with TWO phases of 16 MFMAs per K-tile instead of four of 8 (same 256x256x64 macro tile, 8 waves as 2 x 4,
// wave tile 128x64, same ring, persistent walk and epilogues; results bit-identical to tiles 10 / 11: every accumulator sees
// its k16 steps in the same order).
// Why: per barrier interval one wave row issues MFMAs (8 x 32 = 256 cycles in tile 11) while the other reads fragments and
// issues DMA; the measured interval is ~370 cycles, and the part above 256 is per-interval latency (the load row's LDS read
// latency + DMA issue + the barrier hand-off), not throughput - an interval with nothing but its barriers costs ~250 cycles
// (tools/abl_matrix.sh). Twice the work per interval amortises it: 512 MFMA cycles against 16 / 8 ds_read_b128 + 4 DMA.
//   phase A(t): MFMA rows 0..63 of the wave tile x 64 columns x K = 64   (16 MFMA; A0, B0, B1 fragments: 16 ds_read_b128)
//   phase B(t): MFMA rows 64..127                                       (16 MFMA; A1 fragments: 8 reads, B fragments kept)
// DMA issue (two half-tiles per load part, in consumption order, 4 - 6 intervals ahead of the first read):
//   load A(t): B0(t+1) B1(t+1)      load B(t): A1(t+1) A0(t+2)          prologue: A0(0) B0(0) B1(0) A1(0) A0(1)
// Counted waits at the END of a load part, before its barrier (what the NEXT load part reads has landed for every wave once
// the barrier is passed): after load A vmcnt(6) - only A0(t+1) and this part's four stay in flight, A1(t) is in; after load B
// vmcnt(4). lgkmcnt(0) also sits BEFORE the barrier: the lagging wave row's reads of A0(t) retire before the leading row,
// one interval later, refills that half-tile with A0(t+2).
template <int EPI, bool LNF = false>
__global__ __launch_bounds__(512) void gemm8q_f16_kernel(GemmArgs p, int total) {
  extern __shared__ __attribute__((aligned(16))) half_t ring[];  // [buf 2][A0 A1 B0 B1][128][64] + 8 x 4 KiB slabs
  constexpr int HT = 128 * 64;
  const int ntn = p.N / 256;
  const int ntm = (p.M + 255) / 256;
  const int t = threadIdx.x, lane = t & 63;
  const int wv = __builtin_amdgcn_readfirstlane(t >> 6);
  const int wr = wv >> 2, wc = wv & 3;
  const int lr = lane & 31, lg = lane >> 5;
  half_t* slab = ring + 8 * HT + wv * 2048;
  const int nk = p.K / 64;
  const unsigned bh = 32u * (unsigned)p.ldw;
  const int arow0 = wr * 64 + lr, brow0 = wc * 32 + lr;

  int idx = blockIdx.x, tm = 0, tn = 0;
  while (idx < total && !tile_map(idx, ntm, ntn, p.map_mode, tm, tn)) idx += gridDim.x;
  if (idx >= total) return;

  unsigned aoff[2][2], boff[2];
  auto offsets = [&](int m0, int n0) {
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int lrow = j * 64 + (t >> 3);
      const int chunk = (t & 7) ^ ((lrow >> 1) & 7);
      const int arow = (lrow >> 6) * 128 + (lrow & 63);
      const int brow = (lrow >> 5) * 64 + (lrow & 31);
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        int am = m0 + arow + h * 64;
        am = am < p.M ? am : p.M - 1;
        aoff[j][h] = (unsigned)am * (unsigned)p.lda + chunk * 8;
      }
      boff[j] = (unsigned)(n0 + brow) * (unsigned)p.ldw + chunk * 8;
    }
  };
  auto stage = [&](int which, int kt) {  // 0 A0, 1 A1, 2 B0, 3 B1
    half_t* dst = ring + ((kt & 1) * 4 + which) * HT + wv * 512;
    const int h = which & 1;
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const half_t* g = which < 2 ? p.A + (aoff[j][h] + (unsigned)kt * 64u) : p.W + (boff[j] + h * bh + (unsigned)kt * 64u);
      glds16(g, dst + j * 4096);
    }
  };
  auto prologue = [&]() {
    stage(0, 0); stage(2, 0); stage(3, 0); stage(1, 0);
    if (nk > 1) stage(0, 1);
  };

  offsets(tm * 256, tn * 256);
  prologue();

  half8_t fa[2][4];   // [i][k16]
  half8_t fb[2][4];   // [b][k16]

#define RDQ_A(bufp, h)                                                                                            \
  _Pragma("unroll") for (int i = 0; i < 2; ++i) _Pragma("unroll") for (int k4 = 0; k4 < 4; ++k4)                   \
      fa[i][k4] = *reinterpret_cast<const half8_t*>(&(bufp)[(h) * HT + lds_off64(arow0 + i * 32, k4 * 2 + lg)]);
#define RDQ_B(bufp)                                                                                               \
  _Pragma("unroll") for (int b = 0; b < 2; ++b) _Pragma("unroll") for (int k4 = 0; k4 < 4; ++k4)                   \
      fb[b][k4] = *reinterpret_cast<const half8_t*>(&(bufp)[(2 + b) * HT + lds_off64(brow0, k4 * 2 + lg)]);
#define MMAQ(a)                                                                                                   \
  _Pragma("unroll") for (int k4 = 0; k4 < 4; ++k4) _Pragma("unroll") for (int i = 0; i < 2; ++i)                   \
      _Pragma("unroll") for (int b = 0; b < 2; ++b)                                                               \
          acc[a][i][b] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fb[b][k4], fa[i][k4], acc[a][i][b], 0, 0, 0);
#define Q_SYNC_IN()                                       \
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");      \
  __builtin_amdgcn_sched_barrier(0);                      \
  __builtin_amdgcn_s_barrier();                           \
  __builtin_amdgcn_sched_barrier(0);                      \
  __builtin_amdgcn_s_setprio(1);
#define Q_SYNC_OUT()                                      \
  __builtin_amdgcn_s_setprio(0);                          \
  __builtin_amdgcn_sched_barrier(0);                      \
  __builtin_amdgcn_s_barrier();                           \
  asm volatile("" ::: "memory");

  bool first = true;
  for (;;) {
    const int m0 = tm * 256, n0 = tn * 256;
    f32x16 acc[2][2][2];  // [a][i][b]
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
          for (int r = 0; r < 16; ++r) acc[a][i][b][r] = 0.f;

    // first tile: A0(0) B0(0) B1(0) have landed, A1(0) A0(1) may still fly. Later tiles: see tile 11 (stores and loads share
    // the counter, so wait for all; the DMAs landed during the epilogue)
    if (first) { if (nk > 1) wait_vmcnt<4>(); else wait_vmcnt<2>(); } else { wait_vmcnt<0>(); }
    first = false;
    __builtin_amdgcn_s_barrier();
    if (wr == 1) __builtin_amdgcn_s_barrier();  // stagger the second wave row by one barrier
    asm volatile("" ::: "memory");

    for (int kt = 0; kt < nk; ++kt) {
      const half_t* buf = ring + (kt & 1) * 4 * HT;
      const bool more1 = kt + 1 < nk, more2 = kt + 2 < nk;
      // ---- load A: B0 B1 A0 fragments of K-tile kt; DMA B0 B1 of kt + 1
      RDQ_B(buf) __builtin_amdgcn_sched_barrier(0); RDQ_A(buf, 0) __builtin_amdgcn_sched_barrier(0);
      if (more1) { stage(2, kt + 1); stage(3, kt + 1); }
      if (more1) wait_vmcnt<6>(); else wait_vmcnt<0>();
      Q_SYNC_IN();
      MMAQ(0)
      Q_SYNC_OUT();
      // ---- load B: A1 fragments; DMA A1 of kt + 1, A0 of kt + 2
      RDQ_A(buf, 1) __builtin_amdgcn_sched_barrier(0);
      if (more1) stage(1, kt + 1);
      if (more2) stage(0, kt + 2);
      if (more2) { wait_vmcnt<4>(); } else if (more1) { wait_vmcnt<2>(); } else { wait_vmcnt<0>(); }
      Q_SYNC_IN();
      MMAQ(1)
      Q_SYNC_OUT();
    }
    if (wr == 0) __builtin_amdgcn_s_barrier();  // balance the stagger: every wave is done with the ring here

    // next tile of this workgroup: request its first half-tiles now
    int nidx = idx + gridDim.x, ntm_ = 0, ntn_ = 0;
    while (nidx < total && !tile_map(nidx, ntm, ntn, p.map_mode, ntm_, ntn_)) nidx += gridDim.x;
    const bool have = nidx < total;
    if (have) {
      offsets(ntm_ * 256, ntn_ * 256);
      prologue();
    }
    __builtin_amdgcn_sched_barrier(0);

    persist_epilogue<EPI, LNF>(acc, slab, m0 + wr * 128, n0 + wc * 64, lane, p);
    if (!have) break;
    idx = nidx; tm = ntm_; tn = ntn_;
  }
#undef RDQ_A
#undef RDQ_B
#undef MMAQ
#undef Q_SYNC_IN
#undef Q_SYNC_OUT
}

template <int EPI, bool LNF = false>
static void launch8q(const GemmArgs& p, hipStream_t s) {
  constexpr int LDS = 2 * 4 * 128 * 64 * 2 + 8 * 4096;
  static bool attr = false;
  if (!attr) {
    (void)hipFuncSetAttribute((const void*)gemm8q_f16_kernel<EPI, LNF>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS);
    attr = true;
  }
  const int ntm = (p.M + 255) / 256, ntn = p.N / 256;
  GemmArgs q = p;
  q.map_mode = pick_map_mode(ntm, ntn);
  const int total = tile_map_grid(ntm, ntn, q.map_mode);
  hipLaunchKernelGGL((gemm8q_f16_kernel<EPI, LNF>), dim3(total < num_cus() ? total : num_cus()), dim3(512), LDS, s, q, total);
}

// ---------------------------------------------------------------------------------------------------------------------
// Tile 13: FOUR waves (one per SIMD), 128x128 wave tiles, same 256x256x64 macro tile, DMA ring and persistent walk as tile
// 11. Why: tools/micro/mfma_shadow_bench.hip - one wave per SIMD, back-to-back 32x32x16 MFMAs with this kernel's load mix
// in their shadows (2 ds_read_b128 + 1 LDS-DMA per 4 MFMAs, two fragment sets alternating without copies) runs 35.3
// cycles per MFMA = 2260 cycles per K-tile, against 2970 measured for the 8-wave kernels: a 128x128 wave tile needs 32
// fragment reads per 64 MFMAs instead of 48, and a single in-order stream has no barrier-coupled intervals.
// Schedule per K-tile kt (four k16 sub-steps of 16 MFMAs; fragment sets alternate, so the loop has no register rotation):
//   S0: MFMA set 0 | read (kt, s1) -> set 1 | DMA pieces 8..15 of K-tile kt+1
//   S1: MFMA set 1 | read (kt, s2) -> set 0
//   S2: MFMA set 0 | read (kt, s3) -> set 1 ; lgkmcnt(0) ; vmcnt(0) ; ONE barrier
//   S3: MFMA set 1 | read (kt+1, s0) -> set 0 | DMA pieces 0..7 of K-tile kt+2 (into the buffer the barrier just freed)
// so a DMA piece is waited for 2 to 4 sub-steps (1100-2300 cycles) after it was issued.
template <int EPI>
__global__ __launch_bounds__(256, 1) void gemm4p_f16_kernel(GemmArgs p, int total) {
  extern __shared__ __attribute__((aligned(16))) half_t ring[];  // [buf 2][A0 A1 B0 B1][128][64] + 4 x 8 KiB slabs
  constexpr int HT = 128 * 64;
  const int ntn = p.N / 256;
  const int ntm = (p.M + 255) / 256;
  const int t = threadIdx.x, lane = t & 63;
  const int wv = __builtin_amdgcn_readfirstlane(t >> 6);
  const int wr = wv >> 1, wc = wv & 1;
  const int lr = lane & 31, lg = lane >> 5;
  half_t* slab = ring + 8 * HT + wv * 4096;
  const int nk = p.K / 64;

  int idx = blockIdx.x, tm = 0, tn = 0;
  while (idx < total && !tile_map(idx, ntm, ntn, p.map_mode, tm, tn)) idx += gridDim.x;
  if (idx >= total) return;

  // DMA piece (which, j): rows j*32 + t/8 of half-tile `which` (8 lanes per 128-byte row), LDS slot t%8 <- source chunk
  unsigned aoff[2][4], boff[2][4];
  auto offsets = [&](int m0, int n0) {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int row = j * 32 + (t >> 3);
      const int chunk = (t & 7) ^ ((row >> 1) & 7);
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        int am = m0 + h * 128 + row;
        am = am < p.M ? am : p.M - 1;
        aoff[h][j] = (unsigned)am * (unsigned)p.lda + chunk * 8;
        boff[h][j] = (unsigned)(n0 + h * 128 + row) * (unsigned)p.ldw + chunk * 8;
      }
    }
  };
  auto piece = [&](int q, int kt) {   // q = 0..15: which = q >> 2 (A0 A1 B0 B1), j = q & 3
    const int which = q >> 2, j = q & 3;
    half_t* dst = ring + ((kt & 1) * 4 + which) * HT + (j * 32 + wv * 8) * 64;
    const half_t* g = which < 2 ? p.A + (aoff[which][j] + (unsigned)kt * 64u) : p.W + (boff[which - 2][j] + (unsigned)kt * 64u);
    if (!(kt > 1 && (p.dbg & 1))) glds16(g, dst);
  };

  offsets(tm * 256, tn * 256);
#pragma unroll
  for (int q = 0; q < 16; ++q) piece(q, 0);
  if (nk > 1) {
#pragma unroll
    for (int q = 0; q < 16; ++q) piece(q, 1);
  }

  half8_t fa[2][4], fb[2][4];   // [set][row block / column block]
  const int arow = lr, brow = lr;
#define SB() __builtin_amdgcn_sched_barrier(0)
#define RDA(set, bufp, s, r) \
  fa[set][r] = *reinterpret_cast<const half8_t*>(&(bufp)[wr * HT + lds_off64(arow + (r) * 32, (s) * 2 + lg)])
#define RDB(set, bufp, s, r) \
  fb[set][r] = *reinterpret_cast<const half8_t*>(&(bufp)[(2 + wc) * HT + lds_off64(brow + (r) * 32, (s) * 2 + lg)])
#define MF(set, rb, cb) \
  acc[(cb) >> 1][(rb) >> 1][(rb) & 1][(cb) & 1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fb[set][cb], fa[set][rb], acc[(cb) >> 1][(rb) >> 1][(rb) & 1][(cb) & 1], 0, 0, 0)
// one k16 sub-step: 16 MFMAs on fragment set `set`; X0..X7 go behind MFMAs 0..7, Y0..Y7 behind MFMAs 8..15
#define SUBSTEP(set, X0, X1, X2, X3, X4, X5, X6, X7, Y0, Y1, Y2, Y3, Y4, Y5, Y6, Y7)                                  \
  MF(set, 0, 0); X0; SB(); MF(set, 0, 1); X1; SB(); MF(set, 0, 2); X2; SB(); MF(set, 0, 3); X3; SB();                 \
  MF(set, 1, 0); X4; SB(); MF(set, 1, 1); X5; SB(); MF(set, 1, 2); X6; SB(); MF(set, 1, 3); X7; SB();                 \
  MF(set, 2, 0); Y0; SB(); MF(set, 2, 1); Y1; SB(); MF(set, 2, 2); Y2; SB(); MF(set, 2, 3); Y3; SB();                 \
  MF(set, 3, 0); Y4; SB(); MF(set, 3, 1); Y5; SB(); MF(set, 3, 2); Y6; SB(); MF(set, 3, 3); Y7; SB();

  bool first = true;
  unsigned long long tcyc = 0, twall = 0, tks = 0;
  for (;;) {
    const int m0 = tm * 256, n0 = tn * 256;
    f32x16 acc[2][2][2][2];  // [column half][a][i][b]: rows (2a+i)*32, columns (2*half+b)*32 of the 128x128 wave tile
#pragma unroll
    for (int c = 0; c < 2; ++c)
#pragma unroll
      for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
          for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[c][a][i][b][r] = 0.f;

    // K-tile 0 landed (the 8 pieces of K-tile 1 may fly); later tiles: the previous tile's stores are in the count
    if (first) { if (nk > 1) wait_vmcnt<16>(); else wait_vmcnt<0>(); } else { wait_vmcnt<0>(); }
    first = false;
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    RDA(0, ring, 0, 0); RDA(0, ring, 0, 1); RDA(0, ring, 0, 2); RDA(0, ring, 0, 3);
    RDB(0, ring, 0, 0); RDB(0, ring, 0, 1); RDB(0, ring, 0, 2); RDB(0, ring, 0, 3);
    SB();

    unsigned long long c0 = 0, w0 = 0;
    if (p.trace) { c0 = __builtin_amdgcn_s_memtime(); w0 = wall_clock64(); }
    for (int kt = 0; kt < nk; ++kt) {
      const half_t* buf = ring + (kt & 1) * 4 * HT;
      const half_t* nbuf = ring + ((kt + 1) & 1) * 4 * HT;
      const bool more1 = kt + 1 < nk, more2 = kt + 2 < nk;
      SUBSTEP(0, RDA(1, buf, 1, 0), RDA(1, buf, 1, 1), RDA(1, buf, 1, 2), RDA(1, buf, 1, 3), RDB(1, buf, 1, 0), RDB(1, buf, 1, 1), RDB(1, buf, 1, 2), RDB(1, buf, 1, 3),
 , , , , , , , )
      SUBSTEP(1, RDA(0, buf, 2, 0), RDA(0, buf, 2, 1), RDA(0, buf, 2, 2), RDA(0, buf, 2, 3), RDB(0, buf, 2, 0), RDB(0, buf, 2, 1), RDB(0, buf, 2, 2), RDB(0, buf, 2, 3), , , , , , , , )
      SUBSTEP(0, RDA(1, buf, 3, 0), RDA(1, buf, 3, 1), RDA(1, buf, 3, 2), RDA(1, buf, 3, 3), RDB(1, buf, 3, 0), RDB(1, buf, 3, 1), RDB(1, buf, 3, 2), RDB(1, buf, 3, 3), , , , , , , , )
      // every wave is done reading this K-tile's buffer and has its pieces of the next one: one barrier per K-tile
      if (!(p.dbg & 2)) {
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
      }
      asm volatile("" ::: "memory");
      // (after the last K-tile these reads fetch stale ring contents that nobody uses: no second copy of the MFMA stream)
      SUBSTEP(1, RDA(0, nbuf, 0, 0); if (more2) piece(0, kt + 2), RDA(0, nbuf, 0, 1); if (more2) piece(1, kt + 2), RDA(0, nbuf, 0, 2); if (more2) piece(2, kt + 2), RDA(0, nbuf, 0, 3); if (more2) piece(3, kt + 2), RDB(0, nbuf, 0, 0); if (more2) piece(4, kt + 2), RDB(0, nbuf, 0, 1); if (more2) piece(5, kt + 2), RDB(0, nbuf, 0, 2); if (more2) piece(6, kt + 2), RDB(0, nbuf, 0, 3); if (more2) piece(7, kt + 2),
                 if (more2) piece(8, kt + 2), if (more2) piece(9, kt + 2), if (more2) piece(10, kt + 2), if (more2) piece(11, kt + 2), if (more2) piece(12, kt + 2), if (more2) piece(13, kt + 2), if (more2) piece(14, kt + 2), if (more2) piece(15, kt + 2))
    }

    if (p.trace) { tcyc += __builtin_amdgcn_s_memtime() - c0; twall += wall_clock64() - w0; tks += nk; }
    // next tile of this workgroup: request its first K-tile and half of the second now (the ring is idle)
    int nidx = idx + gridDim.x, ntm_ = 0, ntn_ = 0;
    while (nidx < total && !tile_map(nidx, ntm, ntn, p.map_mode, ntm_, ntn_)) nidx += gridDim.x;
    const bool have = nidx < total;
    if (have) {
      offsets(ntm_ * 256, ntn_ * 256);
#pragma unroll
      for (int q = 0; q < 16; ++q) piece(q, 0);
      if (nk > 1) {
#pragma unroll
        for (int q = 0; q < 16; ++q) piece(q, 1);
      }
    }
    __builtin_amdgcn_sched_barrier(0);
    persist_epilogue<EPI>(acc[0], slab, m0 + wr * 128, n0 + wc * 128, lane, p);
    persist_epilogue<EPI>(acc[1], slab, m0 + wr * 128, n0 + wc * 128 + 64, lane, p);
    if (!have) break;
    idx = nidx; tm = ntm_; tn = ntn_;
  }
  if (p.trace && t == 0) { p.trace[blockIdx.x * 4 + 0] = tcyc; p.trace[blockIdx.x * 4 + 1] = twall; p.trace[blockIdx.x * 4 + 2] = tks; }
#undef SB
#undef RDA
#undef RDB
#undef MF
#undef SUBSTEP
}

template <int EPI>
static void launch4p(const GemmArgs& p, hipStream_t s) {
  constexpr int LDS = 2 * 4 * 128 * 64 * 2 + 4 * 8192;
  static bool attr = false;
  if (!attr) {
    (void)hipFuncSetAttribute((const void*)gemm4p_f16_kernel<EPI>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS);
    attr = true;
  }
  const int ntm = (p.M + 255) / 256, ntn = p.N / 256;
  GemmArgs q = p;
  q.map_mode = pick_map_mode(ntm, ntn);
  const int total = tile_map_grid(ntm, ntn, q.map_mode);
  static const char* tr = getenv("PSAM_GEMM_TRACE");
  if (tr) { (void)hipMalloc((void**)&q.trace, 256 * 4 * sizeof(unsigned long long)); (void)hipMemsetAsync(q.trace, 0, 256 * 32, s); }
  hipLaunchKernelGGL((gemm4p_f16_kernel<EPI>), dim3(total < num_cus() ? total : num_cus()), dim3(256), LDS, s, q, total);
  if (tr) {   // debugging aid: cycles and wall time inside the k-loops (s_memtime / s_memrealtime at 100 MHz)
    std::vector<unsigned long long> h(256 * 4);
    (void)hipStreamSynchronize(s);
    (void)hipMemcpy(h.data(), q.trace, h.size() * 8, hipMemcpyDeviceToHost);
    (void)hipFree(q.trace);
    double c = 0, w = 0, n = 0;
    for (int b = 0; b < 256; ++b) { c += (double)h[b * 4]; w += (double)h[b * 4 + 1]; n += (double)h[b * 4 + 2]; }
    if (n > 0) fprintf(stderr, "tile13 k-loop: %.0f cycles, %.3f us per K-tile -> %.3f GHz\n", c / n, w / n * 0.01, c / (w * 10.0));
  }
}

template <int EPI>
static void launch_ws(const GemmArgs& p, hipStream_t s) {
  constexpr int LDS = 4 * (384 + 128) * 32 * 2;
  static bool attr = false;
  if (!attr) {
    (void)hipFuncSetAttribute((const void*)gemm_ws_f16_kernel<EPI>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS);
    attr = true;
  }
  const int ntm = (p.M + 383) / 384, ntn = p.N / 128;
  hipLaunchKernelGGL((gemm_ws_f16_kernel<EPI>), dim3(tile_map_grid(ntm, ntn, p.map_mode)), dim3(512), LDS, s, p);
}

template <int EPI, int BN_, int STAG>
static void launch256(const GemmArgs& p, hipStream_t s) {
  using C = G256<BN_>;
  static bool attr = false;
  if (!attr) {
    (void)hipFuncSetAttribute((const void*)gemm256_f16_kernel<EPI, BN_, STAG>,
                              hipFuncAttributeMaxDynamicSharedMemorySize, C::LDS_BYTES);
    attr = true;
  }
  const int ntm = (p.M + 255) / 256, ntn = p.N / BN_;
  hipLaunchKernelGGL((gemm256_f16_kernel<EPI, BN_, STAG>), dim3(tile_map_grid(ntm, ntn, p.map_mode)), dim3(512),
                     C::LDS_BYTES, s, p);
}


// ---- tile 15: the hand-scheduled assembly kernels of gemm_asm_gen.py (code object embedded by gemm_asm_blob.S) ----------------
// Four waves / 128x128 wave tiles / whole K-tile of fragments in registers / LDS-DMA two K-tiles ahead through buffer descriptors;
// one persistent workgroup per CU walks a host-built list of 256x256 tiles (the same XCD-aware order as tile 11). Same MFMA and the
// same k order as tiles 10 / 11 / 13, same epilogue arithmetic: bit-identical results (tests/test_kernels_core_gpu.py).
extern "C" const unsigned char psam_gemm_asm_co[];
extern "C" const unsigned char psam_gemm_asm_co_end[];
struct AsmGemmArgs {
  const void* A; const void* W; const void* bias; void* out; const void* resid; const void* gamma; const void* tab;
  int M, N, K, lda, ldw, ldo, ldr, G, flags, pad;
  void* trace;   // experiment variants only (gemm_asm_gen.py --experiments): uint4 per workgroup {k-loop cycles, epilogue cycles, K-tiles, 0}
};
static_assert(sizeof(AsmGemmArgs) == 104, "kernarg layout of gemm_asm_gen.py");
static hipModule_t g_asm_mod = nullptr;
static std::map<int, std::vector<hipFunction_t>> g_asm_fns;   // family * 1000 + variant -> {f16, gelu, f32}
static int g_asm_variant = 0;   // 0 = the shipped schedule; > 0: experiment builds (kernel names carry the suffix _v<n>)
static int g_asm_state = 0;     // 0 not tried, 1 loaded, -1 failed
extern "C" int psam_gemm_asm_variant(int v) {
  g_asm_variant = v;
  return PSAM_OK;
}
static const hipFunction_t* asm_load(int family = 1) {
  if (g_asm_state == 0) {
    g_asm_state = -1;
    const char* path = getenv("PSAM_GEMM_ASM_CO");          // (experiments: a code object built from another schedule)
    hipError_t st = path ? hipModuleLoad(&g_asm_mod, path) : hipModuleLoadData(&g_asm_mod, psam_gemm_asm_co);
    if (st != hipSuccess) { (void)hipGetLastError(); return nullptr; }
    g_asm_state = 1;
  }
  if (g_asm_state < 0) return nullptr;
  const int key = family * 1000 + g_asm_variant;
  auto it = g_asm_fns.find(key);
  if (it != g_asm_fns.end()) return it->second.data();
  const char* names[3] = {"f16", "gelu", "f32"};
  std::vector<hipFunction_t> f(3, nullptr);
  for (int i = 0; i < 3; ++i) {
    std::string n = std::string(family == 2 ? "psam_gemm_asm2_" : "psam_gemm_asm_") + names[i] +
                    (g_asm_variant > 0 ? "_v" + std::to_string(g_asm_variant) : std::string());
    if (hipModuleGetFunction(&f[i], g_asm_mod, n.c_str()) != hipSuccess) { (void)hipGetLastError(); return nullptr; }
  }
  return (g_asm_fns[key] = f).data();
}
struct AsmTable { int grid; int* dev; };
static std::map<unsigned long long, AsmTable> g_asm_tabs;
// work list of workgroup b: entry i at tab[i * G + b] = tm | tn << 16, terminated (and padded two rows deep) by -1
// halves > 0 (tile 16): every 256x256 tile of the map becomes its 256x128 halves (tm, 2 tn), (tm, 2 tn + 1) back to back - they share
// the A panel - and `halves` is the number of 128-column blocks of the matrix (the last tile column may have one)
static const AsmTable* asm_table(int ntm, int ntn, int mode, int halves = 0) {
  int dev = 0;
  (void)hipGetDevice(&dev);
  const unsigned long long key = ((unsigned long long)ntm << 40) | ((unsigned long long)ntn << 20) | ((unsigned long long)(halves ? 1 : 0) << 19) |
                                 ((unsigned long long)(halves & 1) << 18) | ((unsigned long long)mode << 8) | (unsigned)dev;
  auto it = g_asm_tabs.find(key);
  if (it != g_asm_tabs.end()) return &it->second;
  const int total = tile_map_grid(ntm, ntn, mode);
  const int G = total < num_cus() ? total : num_cus();
  std::vector<std::vector<int>> lists(G);
  for (int b = 0; b < G; ++b)
    for (int idx = b; idx < total; idx += G) {
      int tm = 0, tn = 0;
      if (!tile_map(idx, ntm, ntn, mode, tm, tn)) continue;
      if (!halves) { lists[b].push_back(tm | (tn << 16)); continue; }
      lists[b].push_back(tm | ((2 * tn) << 16));
      if (2 * tn + 1 < halves) lists[b].push_back(tm | ((2 * tn + 1) << 16));
    }
  size_t rows = 0;
  for (auto& l : lists) rows = l.size() > rows ? l.size() : rows;
  rows += 3;
  std::vector<int> h(rows * G, -1);
  for (int b = 0; b < G; ++b)
    for (size_t i = 0; i < lists[b].size(); ++i) h[i * G + b] = lists[b][i];
  AsmTable t;
  t.grid = G;
  t.dev = nullptr;
  if (hipMalloc((void**)&t.dev, h.size() * sizeof(int)) != hipSuccess) return nullptr;
  if (hipMemcpy(t.dev, h.data(), h.size() * sizeof(int), hipMemcpyHostToDevice) != hipSuccess) return nullptr;
  return &(g_asm_tabs[key] = t);
}
static bool asm_eligible(const GemmArgs& p, int epilogue, bool lnf) {
  if (lnf || p.head_hd || p.out_seg || p.resid_mod || epilogue > EPI_F32) return false;
  if (p.N % 256 || p.K % 64 || p.K < 128 || p.M < 1) return false;
  if ((p.lda % 8) || (p.ldw % 8) || (reinterpret_cast<uintptr_t>(p.A) & 15) || (reinterpret_cast<uintptr_t>(p.W) & 15)) return false;
  const unsigned long long lim = 0xffffffffull;
  if ((unsigned long long)p.M * p.lda * 2 > lim || (unsigned long long)p.N * p.ldw * 2 > lim) return false;
  if (p.bias && (reinterpret_cast<uintptr_t>(p.bias) & 15)) return false;
  if (epilogue == EPI_F32) {
    if ((p.ldo % 4) || (reinterpret_cast<uintptr_t>(p.out) & 15) || (unsigned long long)p.M * p.ldo * 4 > lim) return false;
    if (p.resid && ((p.ldr % 4) || (reinterpret_cast<uintptr_t>(p.resid) & 15) || (unsigned long long)p.M * p.ldr * 4 > lim)) return false;
    if (p.gamma && (reinterpret_cast<uintptr_t>(p.gamma) & 15)) return false;
  } else {
    if (!p.wide16 || (unsigned long long)p.M * p.ldo * 2 > lim) return false;
  }
  return true;
}
// tile 16 (half-tile ping-pong, epilogue hidden under the next half-tile): same operand rules, N in blocks of 128, and the K loop
// must be long enough to carry the previous half-tile's epilogue (E + 1 K-tiles)
static bool asm2_eligible(const GemmArgs& p, int epilogue, bool lnf) {
  if (epilogue > EPI_F32 || (p.N % 128)) return false;
  GemmArgs q = p;
  q.N = 256;   // (the 256-column rule of the first family does not apply)
  if (!asm_eligible(q, epilogue, lnf)) return false;
  if ((unsigned long long)p.N * p.ldw * 2 > 0xffffffffull) return false;
  const int e = epilogue == EPI_F16 ? PSAM_ASM2_E_F16 : epilogue == EPI_GELU_F16 ? PSAM_ASM2_E_GELU : PSAM_ASM2_E_F32;
  return p.K / 64 >= e + 1 && p.K / 64 >= 3;
}
static int launch_asm(const GemmArgs& p, int epilogue, hipStream_t s, int family = 1) {
  const hipFunction_t* fns = asm_load(family);
  if (!fns) return PSAM_ERR_LAUNCH;
  // family 2 walks 256x128 half-tiles in the same XCD-aware order: the workgroups of an XCD that run side by side then share A panels
  // (one workgroup doing both halves of a 256x256 tile back to back re-read its A panel from beyond the L2: measured 20 % slower on fc2)
  const int ntm = (p.M + 255) / 256, ntn = family == 2 ? p.N / 128 : p.N / 256;
  const AsmTable* t = asm_table(ntm, ntn, pick_map_mode(ntm, ntn));
  if (!t) return PSAM_ERR_LAUNCH;
  AsmGemmArgs a;
  a.A = p.A; a.W = p.W; a.bias = p.bias; a.out = p.out; a.resid = p.resid; a.gamma = p.gamma; a.tab = t->dev;
  a.M = p.M; a.N = p.N; a.K = p.K; a.lda = p.lda; a.ldw = p.ldw; a.ldo = p.ldo; a.ldr = p.resid ? p.ldr : 0; a.G = t->grid;
  a.flags = p.gamma ? 1 : 0;
  a.pad = 0;
  a.trace = nullptr;
  static const char* tr = getenv("PSAM_GEMM_ASM_TRACE");
  const bool trace = tr && g_asm_variant > 0;
  if (trace) { (void)hipMalloc(&a.trace, (size_t)t->grid * 16); (void)hipMemsetAsync(a.trace, 0, (size_t)t->grid * 16, s); }
  size_t sz = sizeof(a);
  void* extra[] = {HIP_LAUNCH_PARAM_BUFFER_POINTER, &a, HIP_LAUNCH_PARAM_BUFFER_SIZE, &sz, HIP_LAUNCH_PARAM_END};
  if (hipModuleLaunchKernel(fns[epilogue], t->grid, 1, 1, 256, 1, 1, 0, s, nullptr, extra) != hipSuccess) {
    (void)hipGetLastError();
    return PSAM_ERR_LAUNCH;
  }
  if (trace) {   // debugging aid (synchronous): shader cycles inside the k-loops and the epilogues, wave 0 of every workgroup
    std::vector<unsigned> h((size_t)t->grid * 4);
    (void)hipStreamSynchronize(s);
    (void)hipMemcpy(h.data(), a.trace, h.size() * 4, hipMemcpyDeviceToHost);
    (void)hipFree(a.trace);
    double loop = 0, epi = 0, kt = 0;
    for (int b = 0; b < t->grid; ++b) { loop += h[b * 4]; epi += h[b * 4 + 1]; kt += h[b * 4 + 2]; }
    const double tiles = kt / (p.K / 64);
    if (kt > 0) fprintf(stderr, "asm v%d %dx%dx%d epi%d: %.0f cycles per K-tile, %.0f cycles per epilogue (%.1f tiles per workgroup)\n", g_asm_variant, p.M, p.N,
                        p.K, epilogue, loop / kt, epi / tiles, tiles / t->grid);
  }
  return PSAM_OK;
}

// tile choice: 0 = auto, 1 = 128x128x64 double buffer, 2 = 256x128 / 3 = 256x256 staggered ring, 5 = 256x256 plain ring,
// 6 = wave-specialised 384x128, 7 = 256x256x64 8-phase, 8 = 7 with one barrier per phase, 9 = four waves with 128x128
// wave tiles and register staging (experimental), 10 = 8-phase with K-split phases, 11 = its persistent form (the default
// large tile), 13 = four waves / 128x128 wave tiles / LDS-DMA / persistent (experimental, same speed as 11)
// (PSAM_GEMM_TILE env var or psam_gemm_set_tile)
static int g_tile_override = -1;
extern "C" int psam_gemm_set_tile(int t) {
  g_tile_override = t;
  return PSAM_OK;
}
static int pick_tile(int M, int N, int K, int epilogue) {
  if (g_tile_override < 0) {
    const char* e = getenv("PSAM_GEMM_TILE");
    g_tile_override = e ? atoi(e) : 0;
  }
  if (g_tile_override > 0) return g_tile_override;
  // measured on MI355X (tools/gemm_tiles.py, within-run A/B): the 256x256 8-phase kernel (7) beats the 128x128
  // double-buffered kernel (1) by 5-20 % whenever its 256-tiles fill the 256 CUs to >= 80 % in their last round (one
  // workgroup per CU, so a part-filled round is lost time); otherwise the 128x128 kernel with two workgroups per CU
  // wins. 5 = plain 4-deep ring, 3 = its two-phase staggered variant, 6 = wave-specialised loaders (kept for A/B).
  if (N % 256 == 0) {
    const long t256 = (long)((M + 255) / 256) * (N / 256);
    const long ncu = num_cus();
    const long rounds = (t256 + ncu - 1) / ncu;
    // the fp32 residual epilogue with a short K (proj: 20 K-tiles) is better served by two workgroups per CU unless the
    // 256-tiles fill their rounds completely (65536x1280x1280: 603 vs 525 TFLOP/s; 32768x1280x1280, 2.5 rounds: 653 vs 689)
    const bool short_f32 = epilogue == EPI_F32 && K < 2048;
    // 10 = 8-phase with K-split phases (7 = its quadrant-phase predecessor); 11 = its persistent form (fp16 outputs)
    // K >= 768: DINOv2-B's shapes at 16 slices (M = 20752) measured 5-45 % faster on the persistent 256-tile kernel than on the
    // 128-tile one (tools/gemm_tiles.py 1,11: qkv 790-820 vs 750, proj 730-760 vs 500-620, fc1 870 vs 740 TFLOP/s; fp32 epilogue
    // 557 vs 497); per-slice calls (M = 1297) fail the fill test and stay on the 128-tile kernel (350 vs 200)
    if (K >= 768 && t256 * 100 >= rounds * ncu * (short_f32 ? 95 : 80)) {
      static int f32p = -1;
      if (f32p < 0) { const char* e = getenv("PSAM_GEMM_F32_PERSIST"); f32p = e ? atoi(e) : 1; }
      static int asm_on = -1;   // the assembly kernels (tile 15) take every shape the persistent HIP kernel took, when eligible (gemm_dispatch)
      if (asm_on < 0) { const char* e = getenv("PSAM_GEMM_ASM"); asm_on = e ? atoi(e) : 1; }
      if (asm_on) return 15;
      return epilogue == EPI_F32 && !f32p ? 10 : 11;
    }
  }
  return 1;
}

struct LnFold {   // see GemmArgs: LayerNorm folded into the GEMMs either side of it
  void* out16 = nullptr;
  int ld16 = 0;
  float* stats = nullptr;
  const float* ln_mr = nullptr;
  const float* ln_s = nullptr;
};

static int gemm_dispatch(const void* A, const void* W, const float* bias, void* out, const float* resid,
                         const float* gamma, int M, int N, int K, int lda, int ldw, int ldo, int ldr,
                         int resid_mod, int out_seg, int out_seg_stride, int out_seg_off, int epilogue, int head_hd,
                         void* stream, const LnFold& ln = LnFold()) {
  if (M <= 0 || N <= 0 || K <= 0 || (N % BN) != 0 || (K % BK) != 0 || (lda % 8) != 0 || (ldw % 8) != 0 ||
      (ldo % 4) != 0 || (resid && (ldr % 4) != 0))
    return PSAM_ERR_ARG;
  if (epilogue < 0 || epilogue > 3) return PSAM_ERR_ARG;
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
  { static int mm = -2; if (mm == -2) { const char* e = getenv("PSAM_GEMM_MAP"); mm = e ? atoi(e) : (xcd_maps_apply() ? 0 : 1); } p.map_mode = mm; }
  { const char* e = getenv("PSAM_GEMM_DBG"); p.dbg = e ? atoi(e) : 0; }
  p.trace = nullptr;
  { static int st = -1; if (st < 0) { const char* e = getenv("PSAM_GEMM_STAGGER"); st = e ? atoi(e) : 0; } p.stagger = st; }
  p.head_hd = head_hd;
  p.out16 = (half_t*)ln.out16;
  p.ld16 = ln.ld16;
  p.stats = ln.stats;
  p.ln_mr = ln.ln_mr;
  p.ln_s = ln.ln_s;
  const bool ln_prod = ln.out16 || ln.stats, ln_cons = ln.ln_mr || ln.ln_s;
  if (ln_prod && (epilogue != EPI_F32 || (ln.out16 && ((ln.ld16 % 8) != 0 || (reinterpret_cast<uintptr_t>(ln.out16) & 15) != 0)))) return PSAM_ERR_ARG;
  if (ln_cons && (!(epilogue == EPI_F16 || epilogue == EPI_GELU_F16) || !ln.ln_mr || !ln.ln_s)) return PSAM_ERR_ARG;
  const int ntm = (M + BM - 1) / BM, ntn = N / BN;
  dim3 grid(tile_map_grid(ntm, ntn, p.map_mode)), block(256);
  hipStream_t s = (hipStream_t)stream;
  int tsel = epilogue == EPI_RELU_F16 ? 1 : pick_tile(M, N, K, epilogue);
  // the slab epilogues store fp16 rows with 16-byte instructions when the layout allows (tiles 7 / 8 / 10 / 11 / 15 require it)
  p.wide16 = (ldo % 8) == 0 && (reinterpret_cast<uintptr_t>(out) & 15) == 0 && (!head_hd || head_hd % 8 == 0);
  // the assembly kernels (tile 15) take plain row-major operands; everything else they were picked for goes to the persistent HIP kernel
  if (tsel == 16 && !asm2_eligible(p, epilogue, ln_prod || ln_cons)) tsel = 15;
  if (tsel == 15 && !asm_eligible(p, epilogue, ln_prod || ln_cons)) tsel = (N % 256 == 0) ? 11 : 1;
  if (head_hd && tsel != 1 && tsel != 7 && tsel != 8 && tsel != 10 && tsel != 11 && tsel != 13 && tsel != 14) tsel = 1;   // the head-major store lives in the staged epilogue  // the ReLU epilogue lives in the 128x128 kernel
  if ((tsel == 7 || tsel == 8 || tsel == 10 || tsel == 11 || tsel == 13 || tsel == 14) && epilogue != EPI_F32 && !p.wide16) tsel = 1;
  if (ln_prod || ln_cons) {   // the folded-LayerNorm epilogues live in the 128x128 kernel and the persistent 256x256 ones
    if (tsel != 11 && tsel != 14) tsel = 1;
    if (ln_cons && !p.wide16) return PSAM_ERR_ARG;
  }
  if (tsel == 15) return launch_asm(p, epilogue, s);
  if (tsel == 16) return launch_asm(p, epilogue, s, 2);
  if ((tsel == 3 || tsel == 5) && N % 256 == 0) {
    if (tsel == 3) {
      if (epilogue == EPI_F16) launch256<EPI_F16, 256, 1>(p, s);
      else if (epilogue == EPI_GELU_F16) launch256<EPI_GELU_F16, 256, 1>(p, s);
      else launch256<EPI_F32, 256, 1>(p, s);
    } else {
      if (epilogue == EPI_F16) launch256<EPI_F16, 256, 0>(p, s);
      else if (epilogue == EPI_GELU_F16) launch256<EPI_GELU_F16, 256, 0>(p, s);
      else launch256<EPI_F32, 256, 0>(p, s);
    }
    return psam_launch_status();
  }
  if ((tsel == 11 || tsel == 13 || tsel == 14) && N % 256 == 0 && epilogue != EPI_F32 && !p.wide16) tsel = 10;
  if (tsel == 13 && N % 256 == 0) {
    if (epilogue == EPI_F16) launch4p<EPI_F16>(p, s);
    else if (epilogue == EPI_GELU_F16) launch4p<EPI_GELU_F16>(p, s);
    else launch4p<EPI_F32>(p, s);
    return psam_launch_status();
  }
  p.ksplit = 1;
  p.ks_ws = nullptr;
  // split-K (see launch8kp_splitk): only where the automatic choice fell back to the 128x128 kernel because too few 256-tiles
  // exist, every K range keeps >= 16 K-tiles (the partial stores + the reduce pass cost about as much as 12) and the
  // registered workspace holds the partial sums
  {
    static int ks_on = -1;
    if (ks_on < 0) { const char* e = getenv("PSAM_GEMM_SPLITK"); ks_on = e ? atoi(e) : 1; }
    const bool auto_sel = g_tile_override <= 0;
    if (ks_on && auto_sel && g_ks_ws && tsel == 1 && epilogue == EPI_F32 && !ln_prod && !ln_cons && !head_hd && N % 256 == 0 &&
        out_seg == 0 && (ldo % 4) == 0) {
      const int t256 = ((M + 255) / 256) * (N / 256), nkt = K / 64;
      // measured (tools/gemm_splitk_bench.py, us split / 128-tile): 4096x1280x5120 (3 ranges of 26-27 K-tiles, 240 items) 80 / 108;
      // 4096x1024x4096 (4 x 16, 256 items) 58 / 52; 4096x768x3072 (3 x 16, 144 items) 44 / 40; 1297x768x3072 (3 x 16, 54) 34 / 37:
      // it pays only with >= 24 K-tiles per range and the CUs at least three quarters busy
      int ks = num_cus() / t256;
      if (ks > nkt / 24) ks = nkt / 24;
      if (ks > 8) ks = 8;
      while (ks >= 2 && (size_t)ks * M * N * sizeof(float) > g_ks_ws_bytes) --ks;
      if (ks >= 2 && t256 * ks * 4 >= num_cus() * 3) {
        p.ksplit = ks;
        launch8kp_splitk(p, s);
        return psam_launch_status();
      }
    }
  }
  const bool lnf = ln_prod || ln_cons;
  if (lnf && (tsel == 14 || tsel == 11) && N % 256 == 0) {   // separate instantiations: the plain kernels stay as they were
    if (tsel == 14) {
      if (epilogue == EPI_F16) launch8q<EPI_F16, true>(p, s);
      else if (epilogue == EPI_GELU_F16) launch8q<EPI_GELU_F16, true>(p, s);
      else launch8q<EPI_F32, true>(p, s);
    } else {
      if (epilogue == EPI_F16) launch8kp<EPI_F16, true>(p, s);
      else if (epilogue == EPI_GELU_F16) launch8kp<EPI_GELU_F16, true>(p, s);
      else launch8kp<EPI_F32, true>(p, s);
    }
    return psam_launch_status();
  }
  if (lnf) {
    switch (epilogue) {
      case EPI_F16: hipLaunchKernelGGL((gemm_f16_kernel<EPI_F16, true>), grid, block, 0, s, p); break;
      case EPI_GELU_F16: hipLaunchKernelGGL((gemm_f16_kernel<EPI_GELU_F16, true>), grid, block, 0, s, p); break;
      default: hipLaunchKernelGGL((gemm_f16_kernel<EPI_F32, true>), grid, block, 0, s, p); break;
    }
    return psam_launch_status();
  }
  if (tsel == 14 && N % 256 == 0) {
    if (epilogue == EPI_F16) launch8q<EPI_F16>(p, s);
    else if (epilogue == EPI_GELU_F16) launch8q<EPI_GELU_F16>(p, s);
    else launch8q<EPI_F32>(p, s);
    return psam_launch_status();
  }
  if (tsel == 11 && N % 256 == 0) {
    if (epilogue == EPI_F16) launch8kp<EPI_F16>(p, s);
    else if (epilogue == EPI_GELU_F16) launch8kp<EPI_GELU_F16>(p, s);
    else launch8kp<EPI_F32>(p, s);
    return psam_launch_status();
  }
  if (tsel == 10 && N % 256 == 0) {
    static const char* trace_path = getenv("PSAM_GEMM_TRACE");
    const size_t nblk = (size_t)tile_map_grid((M + 255) / 256, N / 256, p.map_mode);
    if (trace_path) {   // debugging aid: dump per-wave timestamps of this launch (synchronous)
      (void)hipMalloc((void**)&p.trace, nblk * 64 * sizeof(unsigned long long));
      (void)hipMemsetAsync(p.trace, 0, nblk * 64 * sizeof(unsigned long long), s);
    }
    if (epilogue == EPI_F16) launch8k<EPI_F16>(p, s);
    else if (epilogue == EPI_GELU_F16) launch8k<EPI_GELU_F16>(p, s);
    else launch8k<EPI_F32>(p, s);
    if (trace_path) {
      std::vector<unsigned long long> h(nblk * 64);
      (void)hipStreamSynchronize(s);
      (void)hipMemcpy(h.data(), p.trace, h.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost);
      (void)hipFree(p.trace);
      FILE* f = fopen(trace_path, "wb");
      if (f) { fwrite(h.data(), sizeof(unsigned long long), h.size(), f); fclose(f); }
    }
    return psam_launch_status();
  }
  if (tsel == 9 && N % 256 == 0) {
    if (epilogue == EPI_F16) launch4w<EPI_F16>(p, s);
    else if (epilogue == EPI_GELU_F16) launch4w<EPI_GELU_F16>(p, s);
    else launch4w<EPI_F32>(p, s);
    return psam_launch_status();
  }
  if (tsel == 8 && N % 256 == 0) {
    if (epilogue == EPI_F16) launch8h<EPI_F16>(p, s);
    else if (epilogue == EPI_GELU_F16) launch8h<EPI_GELU_F16>(p, s);
    else launch8h<EPI_F32>(p, s);
    return psam_launch_status();
  }
  if (tsel == 7 && N % 256 == 0) {
    if (epilogue == EPI_F16 && p.dbg) launch8p<EPI_F16, true>(p, s);
    else if (epilogue == EPI_F16) launch8p<EPI_F16>(p, s);
    else if (epilogue == EPI_GELU_F16) launch8p<EPI_GELU_F16>(p, s);
    else launch8p<EPI_F32>(p, s);
    return psam_launch_status();
  }
  if (tsel == 6) {
    if (epilogue == EPI_F16) launch_ws<EPI_F16>(p, s);
    else if (epilogue == EPI_GELU_F16) launch_ws<EPI_GELU_F16>(p, s);
    else launch_ws<EPI_F32>(p, s);
    return psam_launch_status();
  }
  if (tsel == 2) {
    if (epilogue == EPI_F16) launch256<EPI_F16, 128, 1>(p, s);
    else if (epilogue == EPI_GELU_F16) launch256<EPI_GELU_F16, 128, 1>(p, s);
    else launch256<EPI_F32, 128, 1>(p, s);
    return psam_launch_status();
  }
  switch (epilogue) {
    case EPI_F16: hipLaunchKernelGGL(gemm_f16_kernel<EPI_F16>, grid, block, 0, s, p); break;
    case EPI_GELU_F16: hipLaunchKernelGGL(gemm_f16_kernel<EPI_GELU_F16>, grid, block, 0, s, p); break;
    case EPI_RELU_F16: hipLaunchKernelGGL(gemm_f16_kernel<EPI_RELU_F16>, grid, block, 0, s, p); break;
    default: hipLaunchKernelGGL(gemm_f16_kernel<EPI_F32>, grid, block, 0, s, p); break;
  }
  return psam_launch_status();
}

extern "C" int psam_gemm_f16(const void* A, const void* W, const float* bias, void* out, const float* resid,
                             const float* gamma, int M, int N, int K, int lda, int ldw, int ldo, int ldr,
                             int resid_mod, int out_seg, int out_seg_stride, int out_seg_off, int epilogue,
                             void* stream) {
  // Column split for a tile count just above one round of the persistent kernel (one slice through fc1: 4096x5120x1280 is
  // 16 x 20 = 320 tiles of 256x256 for 256 CUs - two rounds for 1.25 rounds of work, or 2.5 rounds of the 128-tile kernel):
  // the columns that fill the CUs exactly once go to the persistent kernel, the rest to whatever the picker chooses for
  // them (every kernel accumulates an element's K-tiles in the same order, so the results do not depend on the split).
  {
    static int ns_on = -1;
    if (ns_on < 0) { const char* e = getenv("PSAM_GEMM_NSPLIT"); ns_on = e ? atoi(e) : 1; }
    const int ncu = num_cus();
    const int ntm = (M + 255) / 256;
    if (ns_on && g_tile_override <= 0 && (epilogue == EPI_F16 || epilogue == EPI_GELU_F16) && out_seg == 0 && N % 256 == 0 &&
        K >= 768 && ntm > 0 && ntm <= ncu) {
      const int ntn = N / 256, c1 = ncu / ntm;                       // whole tile columns in one full round
      if (c1 >= 1 && c1 < ntn && ntm * c1 * 100 >= ncu * 95 && (ntn - c1) * 2 <= c1) {   // remainder at most half a round
        const int n1 = c1 * 256;
        const half_t* Wh = (const half_t*)W;
        int st = gemm_dispatch(A, W, bias, out, resid, gamma, M, n1, K, lda, ldw, ldo, ldr, resid_mod, 0, 0, 0, epilogue, 0, stream);
        if (st != PSAM_OK) return st;
        return gemm_dispatch(A, Wh + (size_t)n1 * ldw, bias ? bias + n1 : nullptr, (half_t*)out + n1, resid, gamma ? gamma + n1 : nullptr,
                             M, N - n1, K, lda, ldw, ldo, ldr, resid_mod, 0, 0, 0, epilogue, 0, stream);
      }
    }
  }
  return gemm_dispatch(A, W, bias, out, resid, gamma, M, N, K, lda, ldw, ldo, ldr, resid_mod, out_seg, out_seg_stride,
                       out_seg_off, epilogue, 0, stream);
}

// psam_gemm_f16 with a LayerNorm folded into the GEMMs either side of it (see GemmArgs): the residual-stream GEMM (epilogue 2)
// also emits fp16(x) and per-row partial sums, the consuming GEMM (epilogue 0 / 1) runs on fp16(x) with W pre-multiplied by
// the LayerNorm weight and corrects by (mean, rstd) per row. No LayerNorm pass, no cast pass, same arithmetic up to rounding
// (modeling/image_encoder.py:174-193: norm1 -> attn.qkv, norm2 -> mlp.lin1; DINOv2 Block likewise).
extern "C" int psam_gemm_f16_ln(const void* A, const void* W, const float* bias, void* out, const float* resid,
                                const float* gamma, int M, int N, int K, int lda, int ldw, int ldo, int ldr, int resid_mod,
                                int out_seg, int out_seg_stride, int out_seg_off, int epilogue, void* out16, int ld16,
                                float* stats, const float* ln_mr, const float* ln_s, void* stream) {
  LnFold ln;
  ln.out16 = out16; ln.ld16 = ld16; ln.stats = stats; ln.ln_mr = ln_mr; ln.ln_s = ln_s;
  if (stats && (N % 64) != 0) return PSAM_ERR_ARG;
  return gemm_dispatch(A, W, bias, out, resid, gamma, M, N, K, lda, ldw, ldo, ldr, resid_mod, out_seg, out_seg_stride,
                       out_seg_off, epilogue, 0, stream, ln);
}

// The packed qkv projection with a HEAD-MAJOR result: out half [N / hd planes][M][hd] (plane = which*H + h), so that the
// attention kernels read each head's Q / K / V rows as contiguous hd-vectors (a window row is 14 x 160 contiguous bytes)
// instead of 160-byte slices 7680 bytes apart. Same arithmetic as psam_gemm_f16 with epilogue 0.
extern "C" int psam_gemm_f16_heads(const void* A, const void* W, const float* bias, void* out, int M, int N, int K, int lda,
                                   int ldw, int hd, void* stream) {
  if (hd <= 0 || (hd % 4) != 0 || (N % hd) != 0) return PSAM_ERR_ARG;
  return gemm_dispatch(A, W, bias, out, nullptr, nullptr, M, N, K, lda, ldw, /*ldo (checked only)*/ N, 0, 0, 0, 0, 0, EPI_F16,
                       hd, stream);
}
