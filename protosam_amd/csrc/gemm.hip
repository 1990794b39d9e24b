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
#include <string.h>
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
  int wide16;    // fp16 output rows may be stored with 16-byte instructions (ldo % 8 == 0, 16-byte aligned base)
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
// (per DEVICE: one process may drive several GPUs; every cache below that holds device-side state is keyed by hipGetDevice())
struct DevGeom { int cus = 0, xcds = 0; };
static DevGeom g_geom[64];
static inline int cur_device() { int dev = 0; (void)hipGetDevice(&dev); return dev; }
static const DevGeom& device_geometry() {
  const int dev = cur_device();
  DevGeom& g = g_geom[dev & 63];
  if (g.cus > 0) return g;
  int v = 0;
  g.cus = (hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && v > 0) ? v : 256;
  g.xcds = (hipDeviceGetAttribute(&v, hipDeviceAttributeNumberOfXccs, dev) == hipSuccess && v > 0) ? v : 8;
  (void)hipGetLastError();
  return g;
}
static inline int num_cus() { return device_geometry().cus; }
// Cap on the persistent grids (psam_gemm_set_option("max_wgs", n) / PSAM_GEMM_MAX_WGS; 0 = none): with half the CUs per launch, two
// streams run their GEMMs side by side on disjoint CUs (one's epilogue traffic beside the other's k-loops)
static int g_max_wgs = -1;
static inline int eff_cus() {
  if (g_max_wgs < 0) { const char* e = getenv("PSAM_GEMM_MAX_WGS"); g_max_wgs = e ? atoi(e) : 0; }
  const int n = num_cus();
  return (g_max_wgs > 0 && g_max_wgs < n) ? g_max_wgs : n;
}
static inline bool xcd_maps_apply() { const DevGeom& g = device_geometry(); return g.xcds == 8 && g.cus % 8 == 0; }

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

template <int N>
__device__ __forceinline__ void wait_vmcnt() {
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

// ---------------------------------------------------------------------------------------------------------------------
// Tile 12 (round 5): the 128x128x64 kernel above for launches with AT MOST ONE workgroup per CU (one DINOv2 slice: 1297 x 768
// is 66 tiles; one SAM ViT-B image: 4096 x 768 is 192): a ring of four K-tile buffers (128 KiB of LDS), the DMA three K-tiles
// ahead, counted waits. With two buffers a lone workgroup pays one L2 / HBM round trip per K-tile (the next tile is requested
// when the current one is consumed, 16 MFMAs = ~0.25 us later it is awaited: 1297x768x3072 ran 48 K-tiles in 37 us = 0.75 us
// each; two co-resident workgroups hide that for each other, a lone one cannot). Same MFMAs, same k order, same epilogue code:
// results are bit-identical to tile 1 (tests/test_kernels_core_gpu.py test_gemm_deep_ring).
template <int EPI>
__global__ __launch_bounds__(256) void gemm_f16_deep_kernel(GemmArgs p) {
  extern __shared__ __attribute__((aligned(16))) half_t ring[];      // [4 buffers][A | W][128 * 64]
  constexpr int NB = 4, TILE = BM * BK;
  const int ntn = p.N / BN;
  const int ntm = (p.M + BM - 1) / BM;
  int tm, tn;
  if (!tile_map(blockIdx.x, ntm, ntn, p.map_mode, tm, tn)) return;
  const int m0 = tm * BM, n0 = tn * BN;
  const int t = threadIdx.x;
  const int lane = t & 63;
  const int wv = __builtin_amdgcn_readfirstlane(t >> 6);
  const int wm = wv >> 1, wn = wv & 1;
  const int lr = lane & 31, lg = lane >> 5;
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
  auto stage = [&](int kt) {      // eight 1-KiB pieces per wave
    half_t* b = ring + (size_t)(kt & (NB - 1)) * 2 * TILE;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      glds16(ag[j] + kt * BK, b + (wv * 4 + j) * 8 * BK);
      glds16(wg[j] + kt * BK, b + TILE + (wv * 4 + j) * 8 * BK);
    }
  };
  const int nk = p.K / BK;
  // residual rows first (EPI_F32): older than every DMA, so every counted wait below covers them
  float4 rpre[16];
  const bool has_res = (EPI == EPI_F32) && p.resid != nullptr;
  if (EPI == EPI_F32 && has_res) prefetch_resid(p, m0 + wm * 64, n0 + wn * 64, lane, rpre);
  stage(0);
  if (nk > 1) stage(1);
  if (nk > 2) stage(2);
  for (int kt = 0; kt < nk; ++kt) {
    // tile kt has landed for this wave (the stages of tiles kt + 1, kt + 2 may stay in flight), then for all of them; the same
    // barrier says every wave is done reading tile kt - 1, whose buffer the next stage overwrites
    const int rem = nk - 1 - kt;
    if (rem >= 2) wait_vmcnt<16>(); else if (rem == 1) wait_vmcnt<8>(); else wait_vmcnt<0>();
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    if (kt + 3 < nk) stage(kt + 3);
    const half_t* sa = ring + (size_t)(kt & (NB - 1)) * 2 * TILE;
    const half_t* sw = sa + TILE;
    // one wave per SIMD: nothing else hides the LDS latency, so the fragments of k-step s + 1 are requested before the MFMAs of
    // k-step s (two register sets; the compiler's own order was read - wait - multiply, 4 x (LDS latency + 128 cycles) per K-tile)
    half8_t fa[2][2], fb[2][2];
#define PSAM_DEEP_RD(set, s_)                                                                                                        \
    _Pragma("unroll") for (int i = 0; i < 2; ++i)                                                                                    \
        fa[set][i] = *reinterpret_cast<const half8_t*>(&sa[lds_off(wm * 64 + i * 32 + lr, (s_) * 2 + lg)]);                          \
    _Pragma("unroll") for (int j = 0; j < 2; ++j)                                                                                    \
        fb[set][j] = *reinterpret_cast<const half8_t*>(&sw[lds_off(wn * 64 + j * 32 + lr, (s_) * 2 + lg)]);
    PSAM_DEEP_RD(0, 0)
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      __builtin_amdgcn_sched_barrier(0);
      if (s < 3) { PSAM_DEEP_RD((s + 1) & 1, s + 1) }
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fb[s & 1][j], fa[s & 1][i], acc[i][j], 0, 0, 0);  // D^T
    }
#undef PSAM_DEEP_RD
  }
  __syncthreads();   // every wave is done with the ring: the epilogue parks its slabs in it
  float* slab = reinterpret_cast<float*>(ring) + wv * 4096;
  if (EPI == EPI_F32 && has_res)
    store_slab_staged<EPI, true, false>(acc[0][0], acc[0][1], acc[1][0], acc[1][1], slab, m0 + wm * 64, n0 + wn * 64, lane, p, rpre);
  else if ((EPI == EPI_F16 || EPI == EPI_GELU_F16) && p.wide16) {
    half_t* slab16 = reinterpret_cast<half_t*>(slab);
    slab_park16<EPI, false>(acc[0][0], acc[0][1], acc[1][0], acc[1][1], slab16, n0 + wn * 64, lane, p, m0 + wm * 64);
    slab_emit16(slab16, m0 + wm * 64, n0 + wn * 64, lane, p);
  } else
    store_slab_staged<EPI, false, false>(acc[0][0], acc[0][1], acc[1][0], acc[1][1], slab, m0 + wm * 64, n0 + wn * 64, lane, p);
}
// ---------------------------------------------------------------------------------------------------------------------
// Tile 13 (round 5): 64x64x64 tiles for launches whose 128x128 tiles would leave most CUs idle (one DINOv2 slice through proj /
// fc2: 1297 x 768 is 66 tiles of 128x128 but 252 of 64x64). What bounds such a launch is not the MFMA rate but what ONE CU can keep
// in flight towards the L2 (~60 GB/s per CU measured on 1297x768x3072: 48 K-tiles of 32 KiB took 0.54 us each on the deep-ring
// 128-tile kernel whatever the prefetch depth): spreading the same bytes over four times as many CUs is what the library does
// there (19 us against our 37). Four waves (2 x 2), one 32x32x16 accumulator each, a ring of four 16-KiB K-tile buffers (two
// workgroups per CU), DMA three K-tiles ahead with counted waits; bias / LayerScale / residual operands requested before the
// k-loop; direct 8-/16-byte stores from the accumulator layout (rows of 32 bytes: fine for outputs this small).
template <int EPI>
__global__ __launch_bounds__(256, 2) void gemm_f16_s64_kernel(GemmArgs p) {
  extern __shared__ __attribute__((aligned(16))) half_t ring[];      // [4 buffers][A | W][64 * 64]
  constexpr int NB = 4, TILE = 64 * BK;
  const int ntn = p.N / 64;
  const int ntm = (p.M + 63) / 64;
  int tm, tn;
  if (!tile_map(blockIdx.x, ntm, ntn, p.map_mode, tm, tn)) return;
  const int m0 = tm * 64, n0 = tn * 64;
  const int t = threadIdx.x;
  const int lane = t & 63;
  const int wv = __builtin_amdgcn_readfirstlane(t >> 6);
  const int wm = wv >> 1, wn = wv & 1;
  const int lr = lane & 31, lg = lane >> 5;
  const half_t* ag[2];
  const half_t* wg[2];
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int row = (wv * 2 + j) * 8 + (lane >> 3);
    const int chunk = (lane & 7) ^ ((row >> 1) & 7);
    int am = m0 + row;
    am = am < p.M ? am : p.M - 1;
    ag[j] = p.A + (size_t)am * p.lda + chunk * 8;
    wg[j] = p.W + (size_t)(n0 + row) * p.ldw + chunk * 8;
  }
  auto stage = [&](int kt) {      // four 1-KiB pieces per wave
    half_t* b = ring + (size_t)(kt & (NB - 1)) * 2 * TILE;
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      glds16(ag[j] + kt * BK, b + (wv * 2 + j) * 8 * BK);
      glds16(wg[j] + kt * BK, b + TILE + (wv * 2 + j) * 8 * BK);
    }
  };
  // epilogue operands of this lane (row m, columns nb + 8 q + 0..3), requested before the first DMA: older than every counted wait
  const int m = m0 + wm * 32 + lr, nb = n0 + wn * 32 + 4 * lg;
  const bool mval = m < p.M;
  const size_t rrow = p.resid_mod ? (size_t)((mval ? m : 0) % p.resid_mod) : (size_t)(mval ? m : 0);
  float4 bv[4], gv[4], rv[4];
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    bv[q] = p.bias ? *reinterpret_cast<const float4*>(p.bias + nb + 8 * q) : make_float4(0.f, 0.f, 0.f, 0.f);
    if (EPI == EPI_F32) {
      gv[q] = p.gamma ? *reinterpret_cast<const float4*>(p.gamma + nb + 8 * q) : make_float4(1.f, 1.f, 1.f, 1.f);
      rv[q] = p.resid ? *reinterpret_cast<const float4*>(p.resid + rrow * p.ldr + nb + 8 * q) : make_float4(0.f, 0.f, 0.f, 0.f);
    }
  }
  asm volatile("" ::: "memory");
  f32x16 acc;
#pragma unroll
  for (int r = 0; r < 16; ++r) acc[r] = 0.f;
  const int nk = p.K / BK;
  stage(0);
  if (nk > 1) stage(1);
  if (nk > 2) stage(2);
  for (int kt = 0; kt < nk; ++kt) {
    const int rem = nk - 1 - kt;
    if (rem >= 2) wait_vmcnt<8>(); else if (rem == 1) wait_vmcnt<4>(); else wait_vmcnt<0>();
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    if (kt + 3 < nk) stage(kt + 3);
    const half_t* sa = ring + (size_t)(kt & (NB - 1)) * 2 * TILE;
    const half_t* sw = sa + TILE;
    half8_t fa[4], fb[4];
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      fa[s] = *reinterpret_cast<const half8_t*>(&sa[lds_off(wm * 32 + lr, s * 2 + lg)]);
      fb[s] = *reinterpret_cast<const half8_t*>(&sw[lds_off(wn * 32 + lr, s * 2 + lg)]);
    }
#pragma unroll
    for (int s = 0; s < 4; ++s) acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(fb[s], fa[s], acc, 0, 0, 0);  // D^T
  }
  if (!mval) return;
  const size_t orow = p.out_seg ? (size_t)(m / p.out_seg) * p.out_seg_stride + p.out_seg_off + (m % p.out_seg) : (size_t)m;
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const int n = nb + 8 * q;
    float4 v = make_float4(acc[4 * q] + bv[q].x, acc[4 * q + 1] + bv[q].y, acc[4 * q + 2] + bv[q].z, acc[4 * q + 3] + bv[q].w);
    if (EPI == EPI_F16) {
      half4_t h = {(half_t)v.x, (half_t)v.y, (half_t)v.z, (half_t)v.w};
      *reinterpret_cast<half4_t*>(reinterpret_cast<half_t*>(p.out) + orow * p.ldo + n) = h;
    } else if (EPI == EPI_GELU_F16) {
      half4_t h = {(half_t)gelu_erf(v.x), (half_t)gelu_erf(v.y), (half_t)gelu_erf(v.z), (half_t)gelu_erf(v.w)};
      *reinterpret_cast<half4_t*>(reinterpret_cast<half_t*>(p.out) + orow * p.ldo + n) = h;
    } else {
      if (p.gamma) { v.x *= gv[q].x; v.y *= gv[q].y; v.z *= gv[q].z; v.w *= gv[q].w; }
      if (p.resid) { v.x += rv[q].x; v.y += rv[q].y; v.z += rv[q].z; v.w += rv[q].w; }
      *reinterpret_cast<float4*>(reinterpret_cast<float*>(p.out) + orow * p.ldo + n) = v;
    }
  }
}
template <int EPI>
static void launch_s64(const GemmArgs& p, hipStream_t s) {
  constexpr int LDS = 4 * 2 * 64 * BK * 2;
  static bool attr = false;
  if (!attr) {
    (void)hipFuncSetAttribute((const void*)gemm_f16_s64_kernel<EPI>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS);
    attr = true;
  }
  const int ntm = (p.M + 63) / 64, ntn = p.N / 64;
  hipLaunchKernelGGL((gemm_f16_s64_kernel<EPI>), dim3(tile_map_grid(ntm, ntn, p.map_mode)), dim3(256), LDS, s, p);
}

template <int EPI>
static void launch_deep(const GemmArgs& p, dim3 grid, hipStream_t s) {
  constexpr int LDS = 4 * 2 * BM * BK * 2;
  static bool attr = false;
  if (!attr) {
    (void)hipFuncSetAttribute((const void*)gemm_f16_deep_kernel<EPI>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS);
    attr = true;
  }
  hipLaunchKernelGGL((gemm_f16_deep_kernel<EPI>), grid, dim3(256), LDS, s, p);
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




// ---------------------------------------------------------------------------------------------------------------------
// Tile 11: the 8-phase kernel above made PERSISTENT for the fp16-output epilogues. One workgroup per CU walks tiles
// bid, bid + grid, ... of the same XCD-aware map. What it buys (per-workgroup timeline of a round-2 trace build, 65536x3840x1280:
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

// The workspace is registered per DEVICE (the device current at registration), and users on different streams are ordered through an
// event: a split-K GEMM on stream B first waits for the previous user's reduce pass on stream A (ProtoSAM.overlap_streams runs the two
// encoders on two streams; both may take this path).
struct KsWorkspace { float* ptr = nullptr; size_t bytes = 0; hipEvent_t done = nullptr; hipStream_t last = nullptr; bool used = false; };
static std::map<int, KsWorkspace> g_ks;
static KsWorkspace* ks_workspace() {
  int dev = 0;
  (void)hipGetDevice(&dev);
  auto it = g_ks.find(dev);
  return it == g_ks.end() || !it->second.ptr ? nullptr : &it->second;
}
extern "C" int psam_gemm_set_workspace(void* ptr, size_t bytes) {   // device scratch for the split-K partial sums (null: no split-K)
  if (ptr && (reinterpret_cast<uintptr_t>(ptr) & 15)) return PSAM_ERR_ARG;
  int dev = 0;
  (void)hipGetDevice(&dev);
  KsWorkspace& w = g_ks[dev];
  w.ptr = (float*)ptr;
  w.bytes = ptr ? bytes : 0;
  w.used = false;
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
  KsWorkspace* w = ks_workspace();
  q.ks_ws = w->ptr;
  if (w->used && w->last != s) (void)hipStreamWaitEvent(s, w->done, 0);   // another stream's partial sums may still be in the planes
  const int total = ntm * ntn * q.ksplit;
  hipLaunchKernelGGL((gemm8kp_f16_kernel<EPI_F32, false, true>), dim3(total < num_cus() ? total : num_cus()), dim3(512), LDS, s, q, total);
  const size_t n4 = (size_t)p.M * (p.N / 4);
  hipLaunchKernelGGL(splitk_reduce_kernel, dim3((unsigned)((n4 + 255) / 256)), dim3(256), 0, s, w->ptr, q.ksplit, p.bias, p.gamma,
                     p.resid, p.ldr, p.resid_mod, reinterpret_cast<float*>(p.out), p.ldo, p.M, p.N);
  if (!w->done) (void)hipEventCreateWithFlags(&w->done, hipEventDisableTiming);
  (void)hipEventRecord(w->done, s);
  w->last = s;
  w->used = true;
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
  // folded-LayerNorm producer (psam_gemm_asm_f32_ln only; the other kernels' kernarg segment ends at 104)
  void* out16; float* stats; int ld16, pad2;
};
static_assert(sizeof(AsmGemmArgs) == 128, "kernarg layout of gemm_asm_gen.py");
// The code object is loaded once per DEVICE (a hipModule_t / hipFunction_t belongs to the device that was current at load time).
struct AsmModule { hipModule_t mod = nullptr; int state = 0; std::map<int, std::vector<hipFunction_t>> fns; std::map<std::string, hipFunction_t> named; };
static std::map<int, AsmModule> g_asm;   // device -> module; fns: family * 1000 + variant -> {f16, gelu, f32, ...}; state 0 not tried, 1 loaded, -1 failed
static int g_asm_variant = 0;   // 0 = the shipped schedule; > 0: experiment builds (kernel names carry the suffix _v<n>)
extern "C" int psam_gemm_asm_variant(int v) {
  g_asm_variant = v;
  return PSAM_OK;
}
static AsmModule* asm_module() {
  AsmModule& m = g_asm[cur_device()];
  if (m.state == 0) {
    m.state = -1;
    const char* path = getenv("PSAM_GEMM_ASM_CO");          // (experiments: a code object built from another schedule)
    hipError_t st = path ? hipModuleLoad(&m.mod, path) : hipModuleLoadData(&m.mod, psam_gemm_asm_co);
    if (st != hipSuccess) { (void)hipGetLastError(); return nullptr; }
    m.state = 1;
  }
  return m.state < 0 ? nullptr : &m;
}
static const hipFunction_t* asm_load(int family = 1) {
  AsmModule* m = asm_module();
  if (!m) return nullptr;
  const int key = family * 1000 + g_asm_variant;
  auto it = m->fns.find(key);
  if (it != m->fns.end()) return it->second.data();
  // [0..2]: f16 / gelu / f32; [3..5] (family 1, shipped schedule): the same with the default cache policy in the epilogue (_l2);
  // [6..8]: the folded-LayerNorm forms (_ln: consumers f16 / gelu, producer f32)
  const char* names[3] = {"f16", "gelu", "f32"};
  const char* suffix[3] = {"", "_l2", "_ln"};
  const int nf = (family == 1 && g_asm_variant == 0) ? 9 : 3;
  std::vector<hipFunction_t> f(9, nullptr);
  for (int i = 0; i < nf; ++i) {
    std::string n = std::string(family == 2 ? "psam_gemm_asm2_" : "psam_gemm_asm_") + names[i % 3] + suffix[i / 3] +
                    (g_asm_variant > 0 ? "_v" + std::to_string(g_asm_variant) : std::string());
    if (hipModuleGetFunction(&f[i], m->mod, n.c_str()) != hipSuccess) {
      (void)hipGetLastError();
      if (i < 6) return nullptr;
      f[i] = nullptr;              // (a _ln form that is not built: the dispatcher keeps those launches on the HIP kernels)
    }
  }
  for (int i = nf; i < 6; ++i) f[i] = f[i - 3];
  return (m->fns[key] = f).data();
}
// other kernels of the same code object (csrc/gattn_asm_gen.py): looked up by name from attention.hip
hipFunction_t psam_asm_function(const char* name) {
  AsmModule* m = asm_module();
  if (!m) return nullptr;
  auto it = m->named.find(name);
  if (it != m->named.end()) return it->second;
  hipFunction_t f = nullptr;
  if (hipModuleGetFunction(&f, m->mod, name) != hipSuccess) { (void)hipGetLastError(); f = nullptr; }
  return m->named[name] = f;
}
struct AsmTable { int grid; int* dev; };
static std::map<unsigned long long, AsmTable> g_asm_tabs;
// work list of workgroup b: entry i at tab[i * G + b] = tm | tn << 16, terminated (and padded two rows deep) by -1
// halves > 0 (tile 16): every 256x256 tile of the map becomes its 256x128 halves (tm, 2 tn), (tm, 2 tn + 1) back to back - they share
// the A panel - and `halves` is the number of 128-column blocks of the matrix (the last tile column may have one)
static const AsmTable* asm_table(int ntm, int ntn, int mode, int halves = 0, int wg_per_cu = 1) {
  int dev = 0;
  (void)hipGetDevice(&dev);
  if (ntm >= 4096 || ntn >= (1 << 20)) return nullptr;   // (the key's fields: 12 bits of ntm below the CU count)
  const unsigned long long key = ((unsigned long long)eff_cus() << 52) | ((unsigned long long)ntm << 40) | ((unsigned long long)ntn << 20) | ((unsigned long long)(halves ? 1 : 0) << 19) |
                                 ((unsigned long long)(halves & 1) << 18) | ((unsigned long long)(wg_per_cu - 1) << 16) | ((unsigned long long)mode << 8) | (unsigned)dev;
  auto it = g_asm_tabs.find(key);
  if (it != g_asm_tabs.end()) return &it->second;
  const int total = tile_map_grid(ntm, ntn, mode);
  const int G = total < eff_cus() * wg_per_cu ? total : eff_cus() * wg_per_cu;
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
// lnf: 0 plain, 1 a folded-LayerNorm launch (tile 15 has _ln forms of its kernels, tile 16 none)
static bool asm_eligible(const GemmArgs& p, int epilogue, int lnf) {
  if (lnf) {
    const hipFunction_t* f = asm_load(1);
    const bool cons = p.ln_mr || p.ln_s, prod = p.out16 || p.stats;
    if (!f || g_asm_variant != 0 || (cons && prod)) return false;
    if (cons && (!f[6 + epilogue] || !p.ln_mr || !p.ln_s || !p.bias || (reinterpret_cast<uintptr_t>(p.ln_mr) & 15) || (reinterpret_cast<uintptr_t>(p.ln_s) & 15))) return false;
    if (prod && (!f[8] || !p.out16 || !p.stats)) return false;
  }
  if (p.head_hd || p.out_seg || epilogue > EPI_F32) return false;
  // resid_mod (residual row = output row % resid_mod: the position table of a patch embedding, image_encoder.py:107-108): the fp32
  // epilogue of the 256-tile kernels masks the row of a tile's origin - a power of two that whole tiles divide
  if (p.resid_mod && (epilogue != EPI_F32 || !p.resid || p.resid_mod < 256 || (p.resid_mod & (p.resid_mod - 1)) != 0 || g_asm_variant != 0)) return false;
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
static bool asm2_eligible(const GemmArgs& p, int epilogue, int lnf) {
  if (epilogue > EPI_F32 || (p.N % 128)) return false;
  GemmArgs q = p;
  q.N = 256;   // (the 256-column rule of the first family does not apply)
  if (lnf || p.resid_mod || !asm_eligible(q, epilogue, 0)) return false;
  if ((unsigned long long)p.N * p.ldw * 2 > 0xffffffffull) return false;
  const int e = epilogue == EPI_F16 ? PSAM_ASM2_E_F16 : epilogue == EPI_GELU_F16 ? PSAM_ASM2_E_GELU : PSAM_ASM2_E_F32;
  return p.K / 64 >= e + 1 && p.K / 64 >= 3;
}
static int launch_asm(const GemmArgs& p, int epilogue, hipStream_t s, int family = 1) {
  const hipFunction_t* fns = asm_load(family);
  if (!fns) return PSAM_ERR_LAUNCH;
  // family 2 walks 256x128 half-tiles in the same XCD-aware order: the workgroups of an XCD that run side by side then share A panels
  // (one workgroup doing both halves of a 256x256 tile back to back re-read its A panel from beyond the L2: measured 20 % slower on fc2)
  const int ntm = (p.M + 255) / 256, ntn = family >= 2 ? p.N / 128 : p.N / 256;
  const AsmTable* t = asm_table(ntm, ntn, pick_map_mode(ntm, ntn));
  if (!t) return PSAM_ERR_LAUNCH;
  AsmGemmArgs a;
  a.A = p.A; a.W = p.W; a.bias = p.bias; a.out = p.out; a.resid = p.resid; a.gamma = p.gamma; a.tab = t->dev;
  a.M = p.M; a.N = p.N; a.K = p.K; a.lda = p.lda; a.ldw = p.ldw; a.ldo = p.ldo; a.ldr = p.resid ? p.ldr : 0; a.G = t->grid;
  a.flags = p.gamma ? 1 : 0;
  a.pad = (family == 1 && p.resid_mod > 0) ? p.resid_mod - 1 : -1;      // (S_RMASK of gemm_asm_gen.py; the half-tile kernels ignore it)
  a.trace = nullptr;
  const bool ln_cons = family == 1 && p.ln_mr && p.ln_s, ln_prod = family == 1 && (p.out16 || p.stats);
  if (ln_cons) {   // folded LayerNorm, consumer: (mean, rstd) rows + -mean fragments in `resid`, the s fragments (behind the N floats) in `gamma`
    a.resid = p.ln_mr;
    a.gamma = p.ln_s + p.N;
  }
  a.out16 = p.out16; a.stats = p.stats; a.ld16 = p.ld16; a.pad2 = 0;
  static const char* tr = getenv("PSAM_GEMM_ASM_TRACE");
  const bool trace = tr && g_asm_variant > 0;
  if (trace) { (void)hipMalloc(&a.trace, (size_t)t->grid * 32); (void)hipMemsetAsync(a.trace, 0, (size_t)t->grid * 32, s); }
  size_t sz = ln_prod ? sizeof(a) : 104;     // (the kernarg segment each kernel declares)
  void* extra[] = {HIP_LAUNCH_PARAM_BUFFER_POINTER, &a, HIP_LAUNCH_PARAM_BUFFER_SIZE, &sz, HIP_LAUNCH_PARAM_END};
  // streaming epilogue (nt loads / stores) when the output is far beyond the 32 MB of L2 and only the next launch reads it; the
  // default policy when it may still be there (gemm_asm_gen.py variants())
  const size_t out_bytes = (size_t)p.M * p.N * (epilogue == EPI_F32 ? 4 : 2);
  const int fsel = (ln_cons || ln_prod) ? 6 + epilogue : epilogue + ((family == 1 && out_bytes <= ((size_t)48 << 20)) ? 3 : 0);
  if (hipModuleLaunchKernel(fns[fsel], t->grid, 1, 1, 256, 1, 1, 0, s, nullptr, extra) != hipSuccess) {
    (void)hipGetLastError();
    return PSAM_ERR_LAUNCH;
  }
  if (trace) {   // debugging aid (synchronous): shader cycles inside the k-loops and the epilogues, wave 0 of every workgroup
    std::vector<unsigned> h((size_t)t->grid * 8);
    (void)hipStreamSynchronize(s);
    (void)hipMemcpy(h.data(), a.trace, h.size() * 4, hipMemcpyDeviceToHost);
    (void)hipFree(a.trace);
    double loop = 0, epi = 0, kt = 0, tot = 0, rt = 0;
    unsigned tmax = 0;
    for (int b = 0; b < t->grid; ++b) rt += h[(size_t)t->grid * 4 + b * 4];
    for (int b = 0; b < t->grid; ++b) { loop += h[b * 4]; epi += h[b * 4 + 1]; kt += h[b * 4 + 2]; tot += h[b * 4 + 3]; tmax = h[b * 4 + 3] > tmax ? h[b * 4 + 3] : tmax; }
    const double tiles = kt / (p.K / 64);
    if (kt > 0) fprintf(stderr, "asm v%d %dx%dx%d epi%d: %.0f cycles per K-tile, %.0f cycles per epilogue (%.1f tiles per workgroup) kernel entry to exit: mean %.0f max %u cycles = %.1f us (shader clock %.0f MHz)\n", g_asm_variant, p.M, p.N,
                        p.K, epilogue, loop / kt, epi / tiles, tiles / t->grid, tot / t->grid, tmax, rt / t->grid * 0.01, rt > 0 ? tot / rt * 100.0 : 0.0);
  }
  return PSAM_OK;
}

// ---- split-K on the assembly tile for ONE slice through a long-K residual GEMM (round 6) ---------------------------------------
// One SAM ViT-H image through mlp.lin2 is 4096 x 1280 x 5120: 80 tiles of 256 x 256 for 256 CUs - the half-tile kernels reach 160
// items (67 us, 0.6 of the chip). Here an item is (tile, K range): 80 x 3 = 240 workgroups side by side, each over 26-27 K-tiles of
// psam_gemm_asm_f32_sk (the tile-15 k-loop; plain fp32 stores of the partial sums to plane r of a CALLER-OWNED workspace, so the path
// is safe inside a captured graph: nothing is shared between streams or graphs), then ONE pass that finishes the residual update
// AND the LayerNorm that follows it in the block stack: x += bias + sum_r plane_r (fixed order: deterministic), out16 = LN(x) - the
// separate LayerNorm pass of the one-slice path disappears. modeling/common.py:13-26 + image_encoder.py:174-193.
static const AsmTable* asm_table_splitk(int ntm, int ntn, int ks) {
  int dev = 0;
  (void)hipGetDevice(&dev);
  if (ntm >= 4096 || ntn >= (1 << 16) || ks < 2 || ks > 8) return nullptr;
  const unsigned long long key = (1ull << 63) | ((unsigned long long)eff_cus() << 52) | ((unsigned long long)ntm << 40) | ((unsigned long long)ntn << 20) |
                                 ((unsigned long long)ks << 8) | (unsigned)dev;
  auto it = g_asm_tabs.find(key);
  if (it != g_asm_tabs.end()) return &it->second;
  const int mode = pick_map_mode(ntm, ntn);
  const int ntiles = tile_map_grid(ntm, ntn, mode);
  std::vector<int> items;                                  // the ranges of a tile are neighbours: they share both operand panels' rows
  for (int idx = 0; idx < ntiles; ++idx) {
    int tm = 0, tn = 0;
    if (!tile_map(idx, ntm, ntn, mode, tm, tn)) continue;
    for (int r = 0; r < ks; ++r) items.push_back(tm | (r << 12) | (tn << 16));
  }
  const int total = (int)items.size();
  const int G = total < eff_cus() ? total : eff_cus();
  const size_t rows = (size_t)(total + G - 1) / G + 3;
  std::vector<int> h(rows * G, -1);
  for (int i = 0; i < total; ++i) h[(size_t)(i / G) * G + (i % G)] = items[i];
  AsmTable t;
  t.grid = G;
  t.dev = nullptr;
  if (hipMalloc((void**)&t.dev, h.size() * sizeof(int)) != hipSuccess) return nullptr;
  if (hipMemcpy(t.dev, h.data(), h.size() * sizeof(int), hipMemcpyHostToDevice) != hipSuccess) return nullptr;
  return &(g_asm_tabs[key] = t);
}

// x[M,N] (fp32, ldx) += bias + sum over the ks planes of ws ([ks][Mp][N] fp32, Mp = M rounded up to whole 256-row tiles) in the order r = 0, 1, ...; then, per row,
// out16 = fp16(LayerNorm(x) * ln_w + ln_b) (two-pass mean / variance as layernorm_kernel) or fp16(x) when ln_w is null. One wave per
// row, the row in registers. HBM: (ks + 1) * 4 N read, 4 N + 2 N written per row.
#define SKR_MAXV 8      // N <= 64 * 4 * 8 = 2048
__global__ __launch_bounds__(256) void splitk_reduce_ln_kernel(const float* __restrict__ ws, int ks, const float* __restrict__ bias,
                                                               float* __restrict__ x, int ldx, const float* __restrict__ ln_w,
                                                               const float* __restrict__ ln_b, float eps, half_t* __restrict__ out16, int ld16,
                                                               int M, int N) {
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (row >= M) return;
  const int nv = N >> 2;
  const size_t Mp = (size_t)((M + 255) & ~255);
  float4 v[SKR_MAXV];
  float* xr = x + (size_t)row * ldx;
#pragma unroll
  for (int k = 0; k < SKR_MAXV; ++k) {
    const int i = lane + 64 * k;
    if (i < nv) v[k] = bias ? reinterpret_cast<const float4*>(bias)[i] : make_float4(0.f, 0.f, 0.f, 0.f);
  }
  for (int r = 0; r < ks; ++r) {
    const float4* pr = reinterpret_cast<const float4*>(ws + ((size_t)r * Mp + row) * N);
#pragma unroll
    for (int k = 0; k < SKR_MAXV; ++k) {
      const int i = lane + 64 * k;
      if (i < nv) { const float4 q = pr[i]; v[k].x += q.x; v[k].y += q.y; v[k].z += q.z; v[k].w += q.w; }
    }
  }
  float s = 0.f;
#pragma unroll
  for (int k = 0; k < SKR_MAXV; ++k) {
    const int i = lane + 64 * k;
    if (i < nv) {
      const float4 q = reinterpret_cast<const float4*>(xr)[i];
      v[k].x += q.x; v[k].y += q.y; v[k].z += q.z; v[k].w += q.w;
      reinterpret_cast<float4*>(xr)[i] = v[k];
      s += (v[k].x + v[k].y) + (v[k].z + v[k].w);
    }
  }
  if (!out16) return;
  float mean = 0.f, rstd = 1.f;
  if (ln_w) {
    mean = wave_sum(s) / (float)N;
    float q = 0.f;
#pragma unroll
    for (int k = 0; k < SKR_MAXV; ++k) {
      const int i = lane + 64 * k;
      if (i < nv) {
        const float a = v[k].x - mean, c = v[k].y - mean, d = v[k].z - mean, e = v[k].w - mean;
        q += (a * a + c * c) + (d * d + e * e);
      }
    }
    rstd = 1.0f / sqrtf(wave_sum(q) / (float)N + eps);
  }
#pragma unroll
  for (int k = 0; k < SKR_MAXV; ++k) {
    const int i = lane + 64 * k;
    if (i < nv) {
      float o0 = v[k].x, o1 = v[k].y, o2 = v[k].z, o3 = v[k].w;
      if (ln_w) {
        const float4 ww = reinterpret_cast<const float4*>(ln_w)[i], bb = reinterpret_cast<const float4*>(ln_b)[i];
        o0 = (o0 - mean) * rstd * ww.x + bb.x; o1 = (o1 - mean) * rstd * ww.y + bb.y;
        o2 = (o2 - mean) * rstd * ww.z + bb.z; o3 = (o3 - mean) * rstd * ww.w + bb.w;
      }
      half4_t h = {(half_t)o0, (half_t)o1, (half_t)o2, (half_t)o3};
      *reinterpret_cast<half4_t*>(out16 + (size_t)row * ld16 + i * 4) = h;
    }
  }
}

// How many K ranges psam_gemm_f16_splitk_ln would use for this shape on this device (0: the shape does not pay - few tiles AND a long K
// are needed: every range keeps >= 16 K-tiles, the items fill >= 3/4 of the CUs in one round; or the assembly kernel is not loaded).
extern "C" int psam_gemm_splitk_ranges(int M, int N, int K) {
  if (M <= 0 || N <= 0 || K <= 0 || (N % 256) || (K % 64) || N > 64 * 4 * SKR_MAXV) return 0;
  const int tiles = ((M + 255) / 256) * (N / 256), nkt = K / 64, ncu = eff_cus();
  if (tiles * 2 > ncu) return 0;
  int ks = ncu / tiles;
  if (ks > 8) ks = 8;
  while (ks >= 2 && nkt / ks < 16) --ks;
  if (ks < 2 || tiles * ks * 4 < ncu * 3) return 0;
  return psam_asm_function("psam_gemm_asm_f32_sk") ? ks : 0;
}

// x[M,N] (fp32, in place) += A[M,K] . W[N,K]^T + bias; out16 (optional, fp16 [M, ld16]) = LayerNorm(x; ln_w, ln_b, eps), or fp16(x) when
// ln_w is null. ws: caller-owned fp32 scratch of at least ks * Mp * N elements, Mp = M rounded up to a multiple of 256 (ks =
// psam_gemm_splitk_ranges(M, N, K), which must be >= 2).
// Deterministic; touches no library-owned state, so it may be captured into a graph.
extern "C" int psam_gemm_f16_splitk_ln(const void* A, const void* W, const float* bias, float* x, int M, int N, int K, int lda, int ldw,
                                       int ldx, int ks, float* ws, const float* ln_w, const float* ln_b, float eps, void* out16, int ld16,
                                       void* stream) {
  if (ks < 2 || ks > 8 || ks != psam_gemm_splitk_ranges(M, N, K) || !ws || !x || (ldx % 4) || (ln_w && !ln_b) || (out16 && (ld16 % 4)))
    return PSAM_ERR_ARG;
  if ((reinterpret_cast<uintptr_t>(ws) | reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(bias) | reinterpret_cast<uintptr_t>(ln_w) |
       reinterpret_cast<uintptr_t>(ln_b)) & 15 || (reinterpret_cast<uintptr_t>(out16) & 7))
    return PSAM_ERR_ARG;
  GemmArgs p;
  p.A = (const half_t*)A; p.W = (const half_t*)W; p.bias = nullptr; p.out = ws; p.resid = nullptr; p.gamma = nullptr;
  p.M = M; p.N = N; p.K = K; p.lda = lda; p.ldw = ldw; p.ldo = N; p.ldr = 0;
  p.resid_mod = 0; p.out_seg = 0; p.out_seg_stride = 0; p.out_seg_off = 0; p.map_mode = 0; p.head_hd = 0;
  p.out16 = nullptr; p.ld16 = 0; p.stats = nullptr; p.ln_mr = nullptr; p.ln_s = nullptr; p.wide16 = false; p.ksplit = 1; p.ks_ws = nullptr;
  if (!asm_eligible(p, EPI_F32, 0) || (unsigned long long)ks * ((M + 255) & ~255) * N * 4 > 0xffffffffull) return PSAM_ERR_ARG;
  hipFunction_t fn = psam_asm_function("psam_gemm_asm_f32_sk");
  const AsmTable* t = asm_table_splitk((M + 255) / 256, N / 256, ks);
  if (!fn || !t) return PSAM_ERR_LAUNCH;
  AsmGemmArgs a;
  a.A = A; a.W = W; a.bias = nullptr; a.out = ws; a.resid = nullptr; a.gamma = nullptr; a.tab = t->dev;
  a.M = M; a.N = N; a.K = K; a.lda = lda; a.ldw = ldw; a.ldo = N; a.ldr = 0; a.G = t->grid; a.flags = 0; a.pad = ks;
  a.trace = nullptr; a.out16 = nullptr; a.stats = nullptr; a.ld16 = 0; a.pad2 = 0;
  size_t sz = 104;
  void* extra[] = {HIP_LAUNCH_PARAM_BUFFER_POINTER, &a, HIP_LAUNCH_PARAM_BUFFER_SIZE, &sz, HIP_LAUNCH_PARAM_END};
  hipStream_t s = (hipStream_t)stream;
  if (hipModuleLaunchKernel(fn, t->grid, 1, 1, 256, 1, 1, 0, s, nullptr, extra) != hipSuccess) {
    (void)hipGetLastError();
    return PSAM_ERR_LAUNCH;
  }
  hipLaunchKernelGGL(splitk_reduce_ln_kernel, dim3((M + 3) / 4), dim3(256), 0, s, ws, ks, bias, x, ldx, ln_w, ln_b, eps, (half_t*)out16, ld16, M, N);
  return psam_launch_status();
}

// tile choice: 0 = auto; 1 = 128x128x64, four waves, two workgroups per CU (HIP); 11 = 256x256x64 persistent 8-wave kernel (HIP; the
// folded-LayerNorm, head-major and split-K forms live here); 15 = the assembly kernels of gemm_asm_gen.py (256x256 tiles, four
// waves, the default large tile); 16 = the half-tile ping-pong assembly kernels of gemm_asm2_gen.py (experimental). The other
// HIP schedules of rounds 1 / 2 (tiles 2 ... 10, 13, 14) lost to these and are gone (history: DESIGN.md).
// (PSAM_GEMM_TILE env var or psam_gemm_set_tile)
// dispatch switches (A/B and tests): initial values from the environment, psam_gemm_set_option overrides at run time
enum { OPT_ASM = 0, OPT_HALF, OPT_SPLITK, OPT_NSPLIT, OPT_DEEP, OPT_SMALL, OPT_COUNT };
static const char* const g_opt_names[OPT_COUNT] = {"asm", "half_tiles", "splitk", "nsplit", "deep", "small"};
static const char* const g_opt_env[OPT_COUNT] = {"PSAM_GEMM_ASM", "PSAM_GEMM_HALF", "PSAM_GEMM_SPLITK", "PSAM_GEMM_NSPLIT", "PSAM_GEMM_DEEP", "PSAM_GEMM_SMALL"};
static int g_opt[OPT_COUNT] = {-1, -1, -1, -1, -1, -1};
static int gemm_option(int i) {
  if (g_opt[i] < 0) { const char* e = getenv(g_opt_env[i]); g_opt[i] = e ? (atoi(e) != 0) : 1; }
  return g_opt[i];
}
extern "C" int psam_gemm_set_option(const char* name, int value) {
  if (name && strcmp(name, "max_wgs") == 0) { g_max_wgs = value > 0 ? value : 0; return PSAM_OK; }
  for (int i = 0; i < OPT_COUNT; ++i)
    if (name && strcmp(name, g_opt_names[i]) == 0) { g_opt[i] = value != 0; return PSAM_OK; }
  return PSAM_ERR_ARG;
}
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
  // a 256x256-tile kernel (one persistent workgroup per CU) wins whenever its tiles fill the CUs to >= 80 % in their last round (a
  // part-filled round is lost time); otherwise the 128x128 kernel with two workgroups per CU (measured, tools/gemm_asm_check.py bench)
  if (N % 256 == 0) {
    const long t256 = (long)((M + 255) / 256) * (N / 256);
    const long ncu = eff_cus();
    const long rounds = (t256 + ncu - 1) / ncu;
    // the fp32 residual epilogue with a short K (proj: 20 K-tiles) is better served by two workgroups per CU unless the
    // 256-tiles fill their rounds completely (65536x1280x1280: 603 vs 525 TFLOP/s; 32768x1280x1280, 2.5 rounds: 653 vs 689)
    const bool short_f32 = epilogue == EPI_F32 && K < 2048;
    // K >= 768: DINOv2-B's shapes at 16 slices (M = 20752) measured 5-45 % faster on the persistent 256-tile kernel than on the
    // 128-tile one (tools/gemm_tiles.py 1,11: qkv 790-820 vs 750, proj 730-760 vs 500-620, fc1 870 vs 740 TFLOP/s; fp32 epilogue
    // 557 vs 497); per-slice calls (M = 1297) fail the fill test and stay on the 128-tile kernel (350 vs 200)
    // (round 5, tools/r05/gemm_small_sweep.py: with an fp16 / GELU epilogue the 256-tile assembly kernel still wins at half-filled rounds -
    // 4096x2304x768, 144 tiles: 20.9 us against 23.8 on the half tiles; 4096x3072x768, 192 tiles: 29.5 against 32-34 on the 128-tile
    // kernel; 2594x3072x768, 132 tiles: 26.1 against 31.5 - a launch that leaves CUs idle runs at a higher clock under the power cap;
    // at 99 tiles (2594x2304x768) it loses, 19.9 against 15.4)
    // (one partial round only: 4096x5120x1280 - 320 tiles, 1.25 rounds - stays on the half tiles, 64 against 71.5 us)
    const int fill = short_f32 ? 95 : (epilogue == EPI_F32 || rounds > 1) ? 80 : 50;
    if (K >= 768 && t256 * 100 >= rounds * ncu * fill) {
      // the assembly kernels (tile 15) take every shape the persistent HIP kernel took, when eligible (gemm_dispatch)
      return gemm_option(OPT_ASM) ? 15 : 11;
    }
  }
  // too few 256x256 tiles for the CUs (one slice through proj / fc1 / fc2: 80 ... 320 tiles): the half-tile assembly kernels have
  // twice the work items (4096x1280x5120: 852 vs 576 TFLOP/s on tile 15, 671 split-K; 4096x5120x1280: 832 vs 734; gemm_dispatch
  // falls back to the 128x128 kernel when their layout rules or minimum K are not met)
  if (N % 128 == 0 && K >= 768) {
    const int half_on = gemm_option(OPT_HALF);
    const long th = (long)((M + 255) / 256) * (N / 128), ncu = eff_cus();
    // (>= 0.9 of the CUs half-filled: 5330x768x3072 - one 1022^2 DINOv2 slice through fc2, 126 half-tiles - 592 vs 465 TFLOP/s on the 128-tile kernel)
    if (half_on && th * 20 >= ncu * 9) return 16;
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
  if (tsel == 16 && !asm2_eligible(p, epilogue, ln_prod || ln_cons)) tsel = (g_tile_override > 0 || ln_prod || ln_cons) ? 15 : 1;
  if (tsel == 15 && !asm_eligible(p, epilogue, ln_prod || ln_cons)) tsel = (N % 256 == 0) ? 11 : 1;
  if (tsel != 1 && tsel != 11 && tsel != 12 && tsel != 13 && tsel != 15 && tsel != 16) tsel = (N % 256 == 0) ? 11 : 1;   // (tiles 2 ... 14 of rounds 1 / 2 are gone)
  if (tsel == 11 && (N % 256) != 0) tsel = 1;
  if (tsel == 11 && epilogue != EPI_F32 && !p.wide16) tsel = 1;   // the persistent kernel stores fp16 rows with 16-byte instructions
  if (ln_cons && !p.wide16) return PSAM_ERR_ARG;                  // (the folded-LayerNorm epilogues live in tiles 1 and 11)
  {   // PSAM_GEMM_LOG=1: which kernel family every distinct launch shape goes to (stderr, once per shape)
    static const bool logging = getenv("PSAM_GEMM_LOG") != nullptr;
    if (logging) {
      static std::map<std::string, int> seen;
      char key[160];
      snprintf(key, sizeof(key), "%d x %d x %d epilogue %d ln %d%d head_hd %d out_seg %d resid_mod %d -> tile %d", M, N, K, epilogue, (int)ln_prod,
               (int)ln_cons, head_hd, out_seg, resid_mod, tsel);
      if (seen[key]++ == 0) fprintf(stderr, "psam_gemm_f16: %s\n", key);
    }
  }
  if (tsel == 15) return launch_asm(p, epilogue, s);
  if (tsel == 16) return launch_asm(p, epilogue, s, 2);
  p.ksplit = 1;
  p.ks_ws = nullptr;
  // split-K (see launch8kp_splitk): only where the automatic choice fell back to the 128x128 kernel because too few 256-tiles
  // exist, every K range keeps >= 16 K-tiles (the partial stores + the reduce pass cost about as much as 12) and the
  // registered workspace holds the partial sums
  {
    const int ks_on = gemm_option(OPT_SPLITK);
    const bool auto_sel = g_tile_override <= 0;
    const KsWorkspace* ksw = ks_workspace();
    // never inside a stream capture (ops.GraphCache): the one workspace per device is ordered between streams by a host-tracked
    // event, which a replayed graph neither waits for nor records - two graphs replayed on two streams would share the planes
    hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
    if (ksw && hipStreamIsCapturing(s, &cap) != hipSuccess) { (void)hipGetLastError(); cap = hipStreamCaptureStatusNone; }
    if (ks_on && auto_sel && ksw && cap == hipStreamCaptureStatusNone && tsel == 1 && epilogue == EPI_F32 && !ln_prod && !ln_cons && !head_hd && N % 256 == 0 &&
        out_seg == 0 && (ldo % 4) == 0) {
      const int t256 = ((M + 255) / 256) * (N / 256), nkt = K / 64;
      // measured (tools/gemm_splitk_bench.py, us split / 128-tile): 4096x1280x5120 (3 ranges of 26-27 K-tiles, 240 items) 80 / 108;
      // 4096x1024x4096 (4 x 16, 256 items) 58 / 52; 4096x768x3072 (3 x 16, 144 items) 44 / 40; 1297x768x3072 (3 x 16, 54) 34 / 37:
      // it pays only with >= 24 K-tiles per range and the CUs at least three quarters busy
      int ks = num_cus() / t256;
      if (ks > nkt / 24) ks = nkt / 24;
      if (ks > 8) ks = 8;
      while (ks >= 2 && (size_t)ks * M * N * sizeof(float) > ksw->bytes) --ks;
      if (ks >= 2 && t256 * ks * 4 >= num_cus() * 3) {
        p.ksplit = ks;
        launch8kp_splitk(p, s);
        return psam_launch_status();
      }
    }
  }
  const bool lnf = ln_prod || ln_cons;
  // tile 13: 64x64 tiles where the 128x128 tiles would leave more than half of the CUs idle (gemm_f16_s64_kernel); psam_gemm_set_option("small", 0)
  // / PSAM_GEMM_SMALL=0 keeps the larger tiles; psam_gemm_set_tile(13) forces it
  // (its epilogue has no head-major plane addressing: psam_gemm_f16_heads stays on the 128-tile kernel)
  if ((tsel == 13 || (tsel == 1 && g_tile_override <= 0 && gemm_option(OPT_SMALL) && (int)grid.x * 2 <= num_cus())) && !lnf && !head_hd && K >= 64 &&
      epilogue != EPI_RELU_F16 && (epilogue == EPI_F32 || (ldo % 4) == 0)) {
    switch (epilogue) {
      case EPI_F16: launch_s64<EPI_F16>(p, s); break;
      case EPI_GELU_F16: launch_s64<EPI_GELU_F16>(p, s); break;
      default: launch_s64<EPI_F32>(p, s); break;
    }
    return psam_launch_status();
  }
  if (tsel == 13) tsel = 1;
  // tile 12: the 128x128 kernel with a four-deep K-tile ring where a launch leaves at most one workgroup per CU (gemm_f16_deep_kernel;
  // bit-identical to tile 1). PSAM_GEMM_DEEP=0 / psam_gemm_set_option("deep", 0) keeps tile 1; psam_gemm_set_tile(12) forces it.
  if ((tsel == 12 || (tsel == 1 && g_tile_override <= 0 && gemm_option(OPT_DEEP) && (int)grid.x <= num_cus())) && !lnf && K >= 256 &&
      epilogue != EPI_RELU_F16) {
    switch (epilogue) {
      case EPI_F16: launch_deep<EPI_F16>(p, grid, s); break;
      case EPI_GELU_F16: launch_deep<EPI_GELU_F16>(p, grid, s); break;
      default: launch_deep<EPI_F32>(p, grid, s); break;
    }
    return psam_launch_status();
  }
  if (tsel == 12) tsel = 1;
  if (tsel == 11) {
    if (lnf) {   // separate instantiations: the plain kernels stay as they were
      if (epilogue == EPI_F16) launch8kp<EPI_F16, true>(p, s);
      else if (epilogue == EPI_GELU_F16) launch8kp<EPI_GELU_F16, true>(p, s);
      else launch8kp<EPI_F32, true>(p, s);
    } else {
      if (epilogue == EPI_F16) launch8kp<EPI_F16>(p, s);
      else if (epilogue == EPI_GELU_F16) launch8kp<EPI_GELU_F16>(p, s);
      else launch8kp<EPI_F32>(p, s);
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
  // them. Every kernel accumulates an element's K-tiles in the same order; tiles 1 / 11 / 15 add the bias AFTER the products and are
  // bit-identical to each other, the half-tile assembly kernels (tile 16) start the accumulators FROM the bias: when the remainder
  // goes to tile 16 its columns differ from the others by that rounding order (a few fp32 ulps before the fp16 rounding:
  // tests/test_kernels_core_gpu.py test_gemm_half_tile_pingpong bounds it by one fp16 ulp).
  {
    const int ns_on = gemm_option(OPT_NSPLIT);
    const int ncu = num_cus();
    const int ntm = (M + 255) / 256;
    // (round 4: with its GELU arithmetic hidden under the MFMAs the half-tile kernel takes the whole fc1 shape faster than the split:
    // 4096x5120x1280 64-66 us vs 49 + 26 us; one ProtoSAM.forward per slice 98.4-100.4 vs 96.3-96.7 slices/s)
    const bool whole16 = epilogue == EPI_GELU_F16 && pick_tile(M, N, K, epilogue) == 16 && K / 64 >= PSAM_ASM2_E_GELU + 1 &&
                         (lda % 8) == 0 && (ldw % 8) == 0 && (ldo % 8) == 0 && ((reinterpret_cast<uintptr_t>(A) | reinterpret_cast<uintptr_t>(W) |
                           reinterpret_cast<uintptr_t>(out) | reinterpret_cast<uintptr_t>(bias)) & 15) == 0;
    if (ns_on && !whole16 && g_tile_override <= 0 && (epilogue == EPI_F16 || epilogue == EPI_GELU_F16) && out_seg == 0 && N % 256 == 0 &&
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
