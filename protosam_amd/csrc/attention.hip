// Fused multi-head self-attention (flash style, online softmax in fp32) on MFMA 16x16x32 f16.
//
// Replaces, per head, softmax((q*scale) k^T [+ decomposed rel-pos bias]) v of
//   * DINOv2 `Attention` / `MemEffAttention` (global, N = 1 + (S/14)^2 tokens, no bias)  [external hub
//     model; call site /root/reference/models/grid_proto_fewshot.py:90-91]
//   * SAM `Attention.forward` (models/segment_anything/modeling/image_encoder.py:235-251) in both its
//     global form (blocks in global_attn_indexes, 64x64 tokens) and its 14x14 windowed form
//     (`window_partition` / `window_unpartition`, image_encoder.py:254-300) with
//     `add_decomposed_rel_pos` (image_encoder.py:337-372).
//
// Input is the packed projection `qkv` fp16 [B, N, 3, H, hd] exactly as `self.qkv(x).reshape(B, N, 3, H, -1)`
// lays it out, so no permute/contiguous copies exist. Output is fp16 [B, N, H*hd] (heads recombined).
//
// Work decomposition: one workgroup = 32*NW query rows of one (batch, head[, window]); each wave owns
// 32 query rows. The product is computed "swapped": S^T = K Q^T and O^T = V^T P^T, so that every lane
// owns one query column of the score tile: row max / row sum are in-lane plus two cross-lane shuffles,
// the P registers are already the B operand of the second MFMA, and the online rescale of O^T is lane
// local. K tiles (64 keys) are staged row-major in LDS, V tiles transposed ([hd][keys]) so both MFMA A
// operands are 16-byte ds_read_b128; the next tile's global loads are issued before the current
// tile's MFMAs.
//
// Window mode folds `window_partition` into index math: key j of window (wy,wx) is token
// (wy*14 + j/14, wx*14 + j%14); positions outside the 64x64 map are the zero-padded tokens of the
// reference (image_encoder.py:267-271) whose q/k/v equal the qkv bias -> read from `pad_row`.
// Rel-pos bias uses the UNSCALED q (image_encoder.py:242-245): rel_h/rel_w are produced by
// psam_relpos from the same fp16 q and added in fp32 after the scale.
#include "common.h"
#include <stdlib.h>
#include <map>
#include <type_traits>
#include <utility>
#include <vector>

struct AttnArgs {
  const half_t* qkv;      // [B, N, 3, H, HD]
  half_t* out;            // [B, N, H*HD]
  const float* rel_h;     // mode1: [B,H,N,gw(64)]  mode2: [B,H,N,16]
  const float* rel_w;
  const half_t* pad_row;  // mode2: [3, H, HD] = fp16(qkv bias)
  const half_t* relq;     // mode2: [B,H,N,2(hi,lo),32] = (rel_h[0:ws] | rel_w[0:ws] | 0) / scale, split in two halfs; or null:
  const half_t* rpack;    // mode2: psam_relpos' windowed table pack [2][2][32][HDP]: the rel-pos terms are computed in-kernel
  int B, N, H;
  float scale;
  int gh, gw, ws, nwx, nwin;  // token grid, window size, windows per row, windows per image
  int nqb;                    // query blocks per (batch, head) (global modes)
  int dbg;                    // ablation switches (PSAM_ATTN_DBG): 1 no K/V global loads, 2 no LDS staging, 4 no tile compute
  long long ts, hs, ws_;      // qkv strides in halfs: token, head, which (q/k/v). token-major [B,N,3,H,hd]: 3*H*hd, hd, H*hd;
                              // head-major [3,H,B*N,hd] (psam_gemm_f16_heads): hd, B*N*hd, H*B*N*hd
};

#define KT 64  // keys per tile

// FULL (global modes only): the host guarantees N % 64 == 0, so no key of any tile is masked and every staging load is in
// range - the zero-fills, null checks and per-element mask selects (a quarter of the loop's VALU issue in a kernel whose
// SIMDs are 87 % issue-busy) drop out at compile time.
// native 16-byte vector for the staging registers: HIP's `uint4` is a struct, and its copies reach the optimiser as
// cross-address-space memcpys that keep the staging array in memory (promoted to LDS / scratch)
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

// V2: softmax section restructured for instruction-level parallelism (see compute_tile). Both forms are kept selectable
// (PSAM_ATTN_V2=0/1) for within-process A/B; they agree to rounding (the row sum of V2 is taken over the fp16-rounded
// probabilities that enter the PV product, by the matrix pipe).
template <int HD, int MODE, int NW, bool FULL = false, bool V2 = false>
// HIP's second launch-bound argument is the minimum number of WAVES PER SIMD (not CUDA's blocks per multiprocessor): the
// 7-wave window kernel needs 4 per SIMD (<= 128 VGPRs) for two workgroups to be co-resident on a CU
__global__ __launch_bounds__(NW * 64, (MODE == 2 ? 4 : 2)) void attn_kernel(AttnArgs p) {
  constexpr int NT = NW * 64;
  constexpr int QB = NW * 32;
  constexpr int HDP = (HD + 31) / 32 * 32;
  constexpr int KS = HDP / 32;
  constexpr int DT = HD / 16;
  constexpr int CH = HD / 8;       // 16-byte chunks per row
  // LDS images are XOR-swizzled in 16-byte chunks instead of padded (bank model of MI355X_MICROARCH.md, checked with
  // SQ_LDS_BANK_CONFLICT: the padded [64][104] / [80][72] images cost 2x on every fragment read and ~6x on the transposing
  // V writes - 64 % of all LDS cycles of the global kernel were conflict cycles):
  //   K  [row][HDP]:  chunk ^= (row >> 2) & 3            (hd = 80: 12 chunks per row)
  //                   chunk ^= row&3 | ((row>>3)&1)<<2   (hd = 64: 8 chunks per row)
  //   V^T[d][VLD]:    chunk ^= (d >> VSH) & VMSK         (64-key tile: 10 chunks per row, (d>>4)&7 for hd 80, (d>>2)&7 for
  //                                                       hd 64; the window kernel keeps its padded images)
  constexpr bool KSWZ = HDP != HD && MODE != 2;          // hd = 80, global modes (the window kernel sits at its 128-VGPR
                                                         // budget: with the swizzle arithmetic it spills 92 bytes and runs 12 %
                                                         // slower, so its K image stays padded and its V^T padding is 30 chunks)
  constexpr bool KSWZ64 = HDP == HD && MODE != 2;        // hd = 64, global modes: 8 chunks per row, chunk ^= r&3 | (r>>3&1)<<2
  constexpr int KLD = (KSWZ || KSWZ64) ? HDP : HDP + 8;  // Ks row stride (halfs)
  // window mode keeps the WHOLE window (196 keys in 3 x 64 + 1 x 16 MFMA key tiles; K rows zero-padded to 200, V^T columns to 224; 78.7 KB -> two
  // workgroups per CU) resident in LDS:
  // one load phase with every request in flight at once and one barrier, instead of a load/store/2-barrier round per
  // 64-key tile (measured before: 40 us per window-head for ~2 us of MFMA work - latency and rendezvous bound)
  constexpr int KRES = MODE == 2 ? 200 : KT;          // key rows resident in Ks (masked rows of the last key tile clamp to 199)
  constexpr int VCOL = MODE == 2 ? 224 : KT;          // key columns resident in Vt
  constexpr int VLD = VCOL + 16;   // Vt row stride (halfs): window 30 chunks, padded only (conflict-free fragment reads by the
                                   // bank model; 29 chunks cost 2x); 64-key tile: 10 chunks, swizzled
  constexpr int VSH = (MODE == 2 || HD == 80) ? 4 : 2;
  constexpr int VMSK = MODE == 2 ? 0 : 7;
  constexpr int NKL = (KRES * CH + NT - 1) / NT;        // K chunk loads per thread (per load phase)
  constexpr int NVL = ((VCOL / 2) * CH + NT - 1) / NT;  // V chunk-pair loads per thread
  const float LOG2E = 1.4426950408889634f;
  const float RESCALE_THR = 8.0f;  // log2 units

  __shared__ __attribute__((aligned(16))) half_t Ks[KRES * KLD];
  __shared__ __attribute__((aligned(16))) half_t Vt[HD * VLD];
  auto koff_of = [](int row, int chunk) {
    const int c = KSWZ ? (chunk ^ ((row >> 2) & 3)) : (KSWZ64 ? (chunk ^ ((row & 3) | (((row >> 3) & 1) << 2))) : chunk);
    return row * KLD + (c << 3);
  };
  auto voff_of = [](int d, int chunk) { return d * VLD + ((chunk ^ ((d >> VSH) & VMSK)) << 3); };
  __shared__ unsigned short klut[MODE == 2 ? 256 : 1];  // window key index -> kh | (kw << 8)
  constexpr int RWLD = 64;                               // rel_w stage row (floats); 16-byte chunks XOR-swizzled by q & 15
  __shared__ __attribute__((aligned(16))) float relw_s[MODE == 1 ? QB * RWLD : 4];  // rel_w[q][kw] * log2(e)

  const int t = threadIdx.x;
  const int lane = t & 63, wv = t >> 6;
  const int li = lane & 15, g = lane >> 4;
  // Workgroup -> (query block, head, batch/window) with XCD locality (block i runs on XCD i % 8, each XCD has its own L2):
  // the workgroups that touch the same cache lines are given ids that differ by a multiple of 8.
  //   window mode: the 16 heads of one window read interleaved 160-byte slices of the same qkv rows (2.25 lines fetched
  //   per slice when alone) -> all heads of a window on one XCD;  global mode: the N/QB query blocks of one (batch, head)
  //   stream the same K/V -> all query blocks of a head on one XCD (one K/V fetch per head instead of one per XCD).
  int h, b, qblk = 0, win = 0, wy = 0, wx = 0;
  {
    const int per = MODE == 2 ? p.H : p.nqb;                     // workgroups that share lines
    const int nshare = MODE == 2 ? p.B * p.nwin : p.B * p.H;     // independent groups
    const int g = blockIdx.x / (8 * per), r = blockIdx.x % (8 * per);
    const int grp = g * 8 + (r & 7), idx = r >> 3;
    if (grp >= nshare) return;
    if (MODE == 2) {
      h = idx;
      b = grp / p.nwin;
      win = grp % p.nwin;
      wy = win / p.nwx;
      wx = win % p.nwx;
    } else {
      qblk = idx;
      h = grp % p.H;
      b = grp / p.H;
    }
  }
  const int N = p.N, H = p.H;
  const size_t rs = (size_t)p.ts;  // qkv token stride (halfs)
  const half_t* qkv_b = p.qkv + (size_t)b * N * rs;
  constexpr int WS = 14;   // the only window size the resident-window schedule is laid out for (checked by the host entry): a
                           // compile-time divisor turns the ~20 runtime integer divisions per thread into multiply-shifts
  const int nkeys = MODE == 2 ? WS * WS : N;
  const int ntiles = (nkeys + KT - 1) / KT;

  // ---- index helpers --------------------------------------------------------------------
  // window-local index j -> token index, or -1 for a zero-padded position
  auto win_token = [&](int j) -> int {
    int y = wy * WS + j / WS, x = wx * WS + j % WS;
    return (y < p.gh && x < p.gw) ? y * p.gw + x : -1;
  };

  // ---- query fragments (B operand of S^T = K Q^T), kept in registers -----------------------
  const int qrow_blk = qblk * QB + wv * 32;  // first query row (global or window-local) of this wave
  int qtok[2];
  bool qvalid[2];
  half8_t qf[2][KS];
#pragma unroll
  for (int qt = 0; qt < 2; ++qt) {
    int q = qrow_blk + qt * 16 + li;
    int tok;
    if (MODE == 2) {
      tok = q < nkeys ? win_token(q) : -1;
    } else {
      tok = q < N ? q : -1;
    }
    qvalid[qt] = tok >= 0;
    qtok[qt] = tok >= 0 ? tok : 0;
    const half_t* qp = qkv_b + (size_t)qtok[qt] * rs + (size_t)h * p.hs;
#pragma unroll
    for (int s = 0; s < KS; ++s) {
      int c0 = s * 32 + g * 8;
      if (c0 < HD) {
        qf[qt][s] = *reinterpret_cast<const half8_t*>(qp + c0);
      } else {
#pragma unroll
        for (int e = 0; e < 8; ++e) qf[qt][s][e] = (half_t)0.f;
      }
    }
  }

  // ---- rel-pos bias state ------------------------------------------------------------------
  // MODE 1: rel_w[q][0..63] * log2(e) is identical for every key tile (a 64-key tile is one key row): staged once in LDS
  const float* relh_q[2] = {nullptr, nullptr};
  if (MODE == 1) {
#pragma unroll
    for (int qt = 0; qt < 2; ++qt)
      relh_q[qt] = p.rel_h + (((size_t)b * H + h) * N + qtok[qt]) * (size_t)p.gw;
    for (int idx = t; idx < QB * 16; idx += NT) {
      const int qr = idx >> 4, c4 = idx & 15;
      int q = qblk * QB + qr;
      q = q < N ? q : N - 1;
      float4 v = *reinterpret_cast<const float4*>(p.rel_w + (((size_t)b * H + h) * N + q) * (size_t)p.gw + c4 * 4);
      v.x *= LOG2E; v.y *= LOG2E; v.z *= LOG2E; v.w *= LOG2E;
      *reinterpret_cast<float4*>(&relw_s[qr * RWLD + ((c4 ^ (qr & 15)) << 2)]) = v;
    }
  }
  // MODE 2: the decomposed rel-pos bias is folded into the S^T MFMA as 32 extra k-slots:
  //   B operand (query side) = [rel_h(q, 0..ws) | rel_w(q, 0..ws) | 0] / scale  (fp16 hi + lo parts, from psam_relpos)
  //   A operand (key side)   = one-hot(kh) | one-hot(kw)  built in registers per key tile
  half8_t qaug[2][2];
  if (MODE == 2) {
    for (int idx = t; idx < 256; idx += NT) {
      int kh = idx / WS, kw = idx - kh * WS;
      klut[idx] = (unsigned short)(kh | (kw << 8));
    }
    if (p.rpack == nullptr) {
#pragma unroll
      for (int qt = 0; qt < 2; ++qt) {
        const half_t* rq = p.relq + (((size_t)b * H + h) * N + qtok[qt]) * 64 + g * 8;
        qaug[qt][0] = *reinterpret_cast<const half8_t*>(rq);
        qaug[qt][1] = *reinterpret_cast<const half8_t*>(rq + 32);
      }
    } else {
      // Fused `add_decomposed_rel_pos` query side (image_encoder.py:337-372; what psam_relpos writes to `relq` otherwise):
      // T[r][q] = q . R[r] for all 2*14-1 rows of both tables as MFMAs with the (hi + lo split) table as the A operand and
      // the query fragments already in registers as B; the lane then gathers, through a wave-private corner of the (still
      // empty) K stage, the 28 values its k-slots need: slot j < 14: T_h[qy + 13 - j], slot 14 + j: T_w[qx + 13 - j].
      f32x4 D[2][2][2];
#pragma unroll
      for (int tab = 0; tab < 2; ++tab)
#pragma unroll
        for (int tile = 0; tile < 2; ++tile) {
#pragma unroll
          for (int qt = 0; qt < 2; ++qt) D[tab][tile][qt] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
          for (int part = 0; part < 2; ++part)
#pragma unroll
            for (int s = 0; s < KS; ++s) {
              const half8_t rf = *reinterpret_cast<const half8_t*>(
                  p.rpack + ((size_t)((tab * 2 + part) * 32 + tile * 16 + li)) * HDP + s * 32 + g * 8);
#pragma unroll
              for (int qt = 0; qt < 2; ++qt)
                D[tab][tile][qt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(rf, qf[qt][s], D[tab][tile][qt], 0, 0, 0);
            }
        }
      float* ts = reinterpret_cast<float*>(Ks) + wv * 1024;   // [tab 2][r 32][q 16] fp32 = 4 KiB per wave, one q-tile at a time
      const float inv_scale = 1.0f / p.scale;
#pragma unroll
      for (int qt = 0; qt < 2; ++qt) {
#pragma unroll
        for (int tab = 0; tab < 2; ++tab)
#pragma unroll
          for (int tile = 0; tile < 2; ++tile)
#pragma unroll
            for (int i = 0; i < 4; ++i) ts[(tab * 32 + tile * 16 + g * 4 + i) * 16 + li] = D[tab][tile][qt][i];
        const int q = qrow_blk + qt * 16 + li;           // window-local query index (< 224)
        const int qy = (q * 4682) >> 16, qx = q - 14 * qy;   // q / 14, q % 14
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          const int j = g * 8 + e;
          float v = 0.f;
          if (j < 28) {
            const int tab = j >= 14;
            const int r = (tab ? qx : qy) + 13 - (tab ? j - 14 : j);
            v = ts[(tab * 32 + r) * 16 + li] * inv_scale;
          }
          const half_t hi = (half_t)v;
          qaug[qt][0][e] = hi;
          qaug[qt][1][e] = (half_t)(v - (float)hi);
        }
      }
    }
  }

  if (MODE == 2 && p.rpack != nullptr) __syncthreads();   // every wave is done with its rel-pos scratch inside Ks
  // zero the K pad columns once (they are never overwritten)
  if (HDP > HD) {
    constexpr int PC = (HDP - HD) / 8;
    for (int idx = t; idx < KRES * PC; idx += NT) {
      int key = idx / PC, c = idx % PC;
      *reinterpret_cast<uint4*>(&Ks[koff_of(key, HD / 8 + c)]) = make_uint4(0, 0, 0, 0);
    }
  }

  // ---- K/V tile prefetch registers -----------------------------------------------------------
  u32x4 kreg[NKL];
  u32x4 vreg[NVL][2];

  auto key_src = [&](int kidx, int which) -> const half_t* {
    // pointer to the hd-vector of key `kidx` (tile-global index), or nullptr if masked
    if (!FULL && kidx >= nkeys) return nullptr;
    if (MODE == 2) {
      int tok = win_token(kidx);
      if (tok < 0) return p.pad_row + ((size_t)which * H + h) * HD;
      return qkv_b + (size_t)tok * rs + (size_t)which * p.ws_ + (size_t)h * p.hs;
    }
    return qkv_b + (size_t)kidx * rs + (size_t)which * p.ws_ + (size_t)h * p.hs;
  };

  auto load_tile = [&](int tile) {
#pragma unroll
    for (int i = 0; i < NKL; ++i) {
      int idx = t + i * NT;
      if constexpr (FULL) {   // every lane loads (out-of-range lanes re-read the last chunk and never store it): no partially
                    // defined registers, so the compiler keeps all requests in flight
        idx = idx < KRES * CH ? idx : KRES * CH - 1;
        const int key = idx / CH, c = idx % CH;
        kreg[i] = *reinterpret_cast<const u32x4*>(key_src(tile * KT + key, 1) + c * 8);
      } else {
        kreg[i] = u32x4{0u, 0u, 0u, 0u};
        if (idx < KRES * CH) {
          int key = idx / CH, c = idx % CH;
          const half_t* src = key_src(tile * KT + key, 1);
          if (src) kreg[i] = *reinterpret_cast<const u32x4*>(src + c * 8);
        }
      }
    }
#pragma unroll
    for (int i = 0; i < NVL; ++i) {
      int idx = t + i * NT;
      if constexpr (FULL) {
        idx = idx < (VCOL / 2) * CH ? idx : (VCOL / 2) * CH - 1;
        const int kp = idx / CH, c = idx % CH;
        vreg[i][0] = *reinterpret_cast<const u32x4*>(key_src(tile * KT + 2 * kp, 2) + c * 8);
        vreg[i][1] = *reinterpret_cast<const u32x4*>(key_src(tile * KT + 2 * kp + 1, 2) + c * 8);
      } else {
        vreg[i][0] = u32x4{0u, 0u, 0u, 0u};
        vreg[i][1] = u32x4{0u, 0u, 0u, 0u};
        if (idx < (VCOL / 2) * CH) {
          int kp = idx / CH, c = idx % CH;
          const half_t* s0 = key_src(tile * KT + 2 * kp, 2);
          const half_t* s1 = key_src(tile * KT + 2 * kp + 1, 2);
          if (s0) vreg[i][0] = *reinterpret_cast<const u32x4*>(s0 + c * 8);
          if (s1) vreg[i][1] = *reinterpret_cast<const u32x4*>(s1 + c * 8);
        }
      }
    }
  };

  auto store_tile = [&]() {
#pragma unroll
    for (int i = 0; i < NKL; ++i) {
      int idx = t + i * NT;
      if (idx < KRES * CH) {
        int key = idx / CH, c = idx % CH;
        *reinterpret_cast<u32x4*>(&Ks[koff_of(key, c)]) = kreg[i];
      }
    }
#pragma unroll
    for (int i = 0; i < NVL; ++i) {
      int idx = t + i * NT;
      if (idx < (VCOL / 2) * CH) {
        int kp = idx / CH, c = idx % CH;
        const uint32_t a[4] = {vreg[i][0].x, vreg[i][0].y, vreg[i][0].z, vreg[i][0].w};
        const uint32_t bb[4] = {vreg[i][1].x, vreg[i][1].y, vreg[i][1].z, vreg[i][1].w};
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          // elements 2e, 2e+1 of both keys -> Vt[c*8+2e][2kp..2kp+1], Vt[c*8+2e+1][2kp..2kp+1]
          uint32_t lo = (a[e] & 0xffffu) | (bb[e] << 16);
          uint32_t hi = (a[e] >> 16) | (bb[e] & 0xffff0000u);
          *reinterpret_cast<uint32_t*>(&Vt[voff_of(c * 8 + 2 * e, kp >> 2) + 2 * (kp & 3)]) = lo;
          *reinterpret_cast<uint32_t*>(&Vt[voff_of(c * 8 + 2 * e + 1, kp >> 2) + 2 * (kp & 3)]) = hi;
        }
      }
    }
  };

  // ---- online-softmax state -------------------------------------------------------------------
  f32x4 ot[DT][2];
#pragma unroll
  for (int d = 0; d < DT; ++d)
#pragma unroll
    for (int qt = 0; qt < 2; ++qt)
#pragma unroll
      for (int r = 0; r < 4; ++r) ot[d][qt][r] = 0.f;
  const float sl2 = p.scale * LOG2E;
  float mrun[2] = {-INFINITY, -INFINITY};
  float lrun[2] = {0.f, 0.f};
  // V2: row sums on the matrix pipe, l^T += 1 . P^T (every row of the 16 x 16 result holds the complete sum over the 32 keys of
  // the k-step, all four lane groups included), rescaled together with O^T
  f32x4 lt[2] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}};
  half8_t ones;
#pragma unroll
  for (int e = 0; e < 8; ++e) ones[e] = (half_t)1.f;

  if (!(p.dbg & 1)) load_tile(0);
  __syncthreads();  // pad-column zeroing + rel tables visible
  if (!(p.dbg & 2)) store_tile();
  __syncthreads();

  // one 64-key tile (NTT = 4 MFMA key tiles) or, for the window's last 4 keys, a single 16-key tile (NTT = 1);
  // `koff` = first resident key row / V^T column of the tile
  // kbase = global index of the tile's first key (masking, rel-pos), koff = its first resident row in Ks / column in Vt
  auto compute_tile = [&](auto ntt_c, int kbase, int koff, bool last_partial) {
    constexpr int NTT = decltype(ntt_c)::value;
    // S^T = K Q^T
    f32x4 st[NTT][2];
#pragma unroll
    for (int tt = 0; tt < NTT; ++tt)
#pragma unroll
      for (int qt = 0; qt < 2; ++qt)
#pragma unroll
        for (int r = 0; r < 4; ++r) st[tt][qt][r] = 0.f;
#pragma unroll
    for (int tt = 0; tt < NTT; ++tt) {
      const int krow = (tt >> 1) * 32 + (li >> 2) * 8 + (tt & 1) * 4 + (li & 3);
#pragma unroll
      for (int s = 0; s < KS; ++s) {
        const int kr = (MODE == 2 && NTT < 4) ? min(koff + krow, KRES - 1) : koff + krow;  // clamped rows are masked keys
        half8_t kf = *reinterpret_cast<const half8_t*>(&Ks[koff_of(kr, s * 4 + g)]);
#pragma unroll
        for (int qt = 0; qt < 2; ++qt)
          st[tt][qt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(kf, qf[qt][s], st[tt][qt], 0, 0, 0);
      }
      if (MODE == 2) {
        const int kidx = kbase + krow;
        const unsigned lut = klut[kidx & 255];
        const int kh = lut & 0xff, kw = (int)(lut >> 8) + WS;  // slots of the two one-hots
        half8_t oh;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          const int slot = g * 8 + e;
          oh[e] = (kidx < nkeys && (slot == kh || slot == kw)) ? (half_t)1.f : (half_t)0.f;
        }
#pragma unroll
        for (int qt = 0; qt < 2; ++qt) {
          st[tt][qt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(oh, qaug[qt][0], st[tt][qt], 0, 0, 0);
          st[tt][qt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(oh, qaug[qt][1], st[tt][qt], 0, 0, 0);
        }
      }
    }

    // scale + bias + mask, online softmax (base-2 domain: scale and bias are pre-multiplied by log2(e))
    half8_t pf[2][(NTT + 1) / 2];
    if constexpr ((NTT & 1) != 0) {
#pragma unroll
      for (int qt = 0; qt < 2; ++qt)
#pragma unroll
        for (int e = 0; e < 8; ++e) pf[qt][NTT / 2][e] = (half_t)0.f;
    }
    if constexpr (!V2) {
#pragma unroll
    for (int qt = 0; qt < 2; ++qt) {
      // bias that is constant over this lane's keys of the tile is added to the row max instead of to every element
      float bh2 = 0.f;
      if (MODE == 1) bh2 = relh_q[qt][kbase / KT] * LOG2E;
      float mx = -INFINITY;
#pragma unroll
      for (int tt = 0; tt < NTT; ++tt) {
        float4 rw4 = make_float4(0.f, 0.f, 0.f, 0.f);
        if (MODE == 1)
          rw4 = *reinterpret_cast<const float4*>(
              &relw_s[(wv * 32 + qt * 16 + li) * RWLD + ((((tt >> 1) * 8 + g * 2 + (tt & 1)) ^ li) << 2)]);
        const float rwv[4] = {rw4.x, rw4.y, rw4.z, rw4.w};
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          float sv = MODE == 1 ? fmaf(st[tt][qt][r], sl2, rwv[r]) : st[tt][qt][r] * sl2;
          if (!FULL && last_partial) {
            const int kidx = kbase + (tt >> 1) * 32 + g * 8 + (tt & 1) * 4 + r;
            if (kidx >= nkeys) sv = -INFINITY;
          }
          st[tt][qt][r] = sv;
          mx = fmaxf(mx, sv);
        }
      }
      mx = fmaxf(mx, __shfl_xor(mx, 16, 64));
      mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
      mx += bh2;
      // lazy rescale: keep the running max while the tile max exceeds it by < 2^RESCALE_THR (P stays <= 2^THR, fine in
      // fp16/fp32); the branch is wave-uniform and alpha == 1 for lanes whose max did not grow.
      float mref = mrun[qt];
      if (!__all(mx <= mrun[qt] + RESCALE_THR)) {
        const float mnew = fmaxf(mrun[qt], mx);
        const float alpha = __builtin_amdgcn_exp2f(mrun[qt] - mnew);
        mrun[qt] = mnew;
        mref = mnew;
        lrun[qt] *= alpha;
#pragma unroll
        for (int d = 0; d < DT; ++d)
#pragma unroll
          for (int r = 0; r < 4; ++r) ot[d][qt][r] *= alpha;
      }
      const float moff = mref - bh2;  // exp2(sv + bh2 - mref)
      float ps = 0.f;
#pragma unroll
      for (int tt = 0; tt < NTT; ++tt) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const float pv = __builtin_amdgcn_exp2f(st[tt][qt][r] - moff);
          ps += pv;
          pf[qt][tt >> 1][(tt & 1) * 4 + r] = (half_t)pv;
        }
      }
      lrun[qt] += ps;
    }
    } else {
      // One straight-line block for BOTH query tiles: the per-tile decision is a single wave-uniform branch (alpha == 1 for
      // rows that did not grow), so the two tiles' max / exp / convert chains interleave; maxima and sums are reduced as
      // trees (the serial `ps += pv` / `mx = max(mx, sv)` chains were 16 dependent operations per tile with two waves per
      // SIMD to hide them); the row sum itself is left to the matrix pipe (below).
      float bh2[2] = {0.f, 0.f}, mx[2];
#pragma unroll
      for (int qt = 0; qt < 2; ++qt) {
        if (MODE == 1) bh2[qt] = relh_q[qt][kbase / KT] * LOG2E;
        float mt[NTT];
#pragma unroll
        for (int tt = 0; tt < NTT; ++tt) {
          float4 rw4 = make_float4(0.f, 0.f, 0.f, 0.f);
          if (MODE == 1)
            rw4 = *reinterpret_cast<const float4*>(
                &relw_s[(wv * 32 + qt * 16 + li) * RWLD + ((((tt >> 1) * 8 + g * 2 + (tt & 1)) ^ li) << 2)]);
          const float rwv[4] = {rw4.x, rw4.y, rw4.z, rw4.w};
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            float sv = MODE == 1 ? fmaf(st[tt][qt][r], sl2, rwv[r]) : st[tt][qt][r] * sl2;
            if (!FULL && last_partial) {
              const int kidx = kbase + (tt >> 1) * 32 + g * 8 + (tt & 1) * 4 + r;
              if (kidx >= nkeys) sv = -INFINITY;
            }
            st[tt][qt][r] = sv;
          }
          mt[tt] = fmaxf(fmaxf(st[tt][qt][0], st[tt][qt][1]), fmaxf(st[tt][qt][2], st[tt][qt][3]));
        }
        float m = mt[0];
        if constexpr (NTT == 2) m = fmaxf(mt[0], mt[1]);
        if constexpr (NTT == 4) m = fmaxf(fmaxf(mt[0], mt[1]), fmaxf(mt[2], mt[3]));
        mx[qt] = m;
      }
#pragma unroll
      for (int qt = 0; qt < 2; ++qt) {
        mx[qt] = fmaxf(mx[qt], __shfl_xor(mx[qt], 16, 64));
        mx[qt] = fmaxf(mx[qt], __shfl_xor(mx[qt], 32, 64));
        mx[qt] += bh2[qt];
      }
      if (!__all(mx[0] <= mrun[0] + RESCALE_THR && mx[1] <= mrun[1] + RESCALE_THR)) {
#pragma unroll
        for (int qt = 0; qt < 2; ++qt) {
          const float mnew = fmaxf(mrun[qt], mx[qt]);
          const float alpha = __builtin_amdgcn_exp2f(mrun[qt] - mnew);
          mrun[qt] = mnew;
#pragma unroll
          for (int r = 0; r < 4; ++r) lt[qt][r] *= alpha;
#pragma unroll
          for (int d = 0; d < DT; ++d)
#pragma unroll
            for (int r = 0; r < 4; ++r) ot[d][qt][r] *= alpha;
        }
      }
#pragma unroll
      for (int qt = 0; qt < 2; ++qt) {
        const float moff = mrun[qt] - bh2[qt];  // exp2(sv + bh2 - mrun)
#pragma unroll
        for (int tt = 0; tt < NTT; ++tt)
#pragma unroll
          for (int r = 0; r < 4; ++r)
            pf[qt][tt >> 1][(tt & 1) * 4 + r] = (half_t)__builtin_amdgcn_exp2f(st[tt][qt][r] - moff);
      }
    }

    // O^T += V^T P^T
    if constexpr (V2) {
#pragma unroll
      for (int s2 = 0; s2 < (NTT + 1) / 2; ++s2)
#pragma unroll
        for (int qt = 0; qt < 2; ++qt) lt[qt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ones, pf[qt][s2], lt[qt], 0, 0, 0);
    }
#pragma unroll
    for (int d = 0; d < DT; ++d) {
#pragma unroll
      for (int s2 = 0; s2 < (NTT + 1) / 2; ++s2) {
        half8_t vf = *reinterpret_cast<const half8_t*>(&Vt[voff_of(d * 16 + li, (koff >> 3) + s2 * 4 + g)]);
#pragma unroll
        for (int qt = 0; qt < 2; ++qt)
          ot[d][qt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(vf, pf[qt][s2], ot[d][qt], 0, 0, 0);
      }
    }
  };

  if (MODE == 2) {
    // the whole window is resident: no further loads, no barriers; each wave walks the key tiles on its own
    // 32-key chunks (two MFMA key tiles = one PV k-step) keep the score / probability registers at half the size of a
    // 64-key tile, which is what lets the kernel fit 128 VGPRs (4 waves per SIMD = two workgroups per CU)
    if (!(p.dbg & 4)) {
      for (int c = 0; c < 6; ++c) compute_tile(std::integral_constant<int, 2>{}, c * 32, c * 32, false);
      compute_tile(std::integral_constant<int, 1>{}, 192, 192, true);
    }
  } else {
    for (int tile = 0; tile < ntiles; ++tile) {
      if (tile + 1 < ntiles) load_tile(tile + 1);
      compute_tile(std::integral_constant<int, 4>{}, tile * KT, 0, (tile == ntiles - 1) && (nkeys % KT != 0));
      __syncthreads();  // everyone done reading Ks/Vt
      if (tile + 1 < ntiles) {
        store_tile();
        __syncthreads();
      }
    }
  }

  // ---- normalise and store: lane holds O^T[d = dt*16 + g*4 + r][q = li] ---------------------------
#pragma unroll
  for (int qt = 0; qt < 2; ++qt) {
    float l;
    if constexpr (V2) {
      l = lt[qt][0];      // complete row sum of query column li (identical in the four rows and lane groups)
    } else {
      l = lrun[qt];
      l += __shfl_xor(l, 16, 64);
      l += __shfl_xor(l, 32, 64);
    }
    const float inv = 1.0f / l;
    if (qvalid[qt]) {
      half_t* op = p.out + ((size_t)b * N + qtok[qt]) * ((size_t)H * HD) + (size_t)h * HD;
#pragma unroll
      for (int d = 0; d < DT; ++d) {
        half4_t o = {(half_t)(ot[d][qt][0] * inv), (half_t)(ot[d][qt][1] * inv), (half_t)(ot[d][qt][2] * inv),
                     (half_t)(ot[d][qt][3] * inv)};
        *reinterpret_cast<half4_t*>(op + d * 16 + g * 4) = o;
      }
    }
  }
}

// ---------------------------------------------------------------------------------------------
// Window attention, second form (`wattn_kernel`, the default for mode 2 when the caller passes `rpack`): same arithmetic as
// attn_kernel<HD, 2, 7>, restructured around what bounded that kernel (VALU issue and the staging round trip; the matrix pipe
// was 7 % busy):
//   * K and V of the window go global -> LDS by DMA (`global_load_lds_dwordx4`) as plain row-major [208][80] images: no
//     staging registers, no LDS store instructions, no transposing writes. The V^T fragments of O^T += V^T P^T come from the
//     row-major image through `ds_read_b64_tr_b16` (two per 16 x 32 fragment). Rows are 160 bytes for both head sizes (hd = 64
//     leaves two 16-byte slots per row unused): with the LDS row ORDER chosen on the DMA's source side (below), every
//     `ds_read_b128` K-fragment read and every transposing V read is bank-conflict-free by the model of MI355X_MICROARCH.md;
//   * key k = 32c + 8a + 4t + b of chunk c sits in LDS row 32c + 16t + 4a + b of the K image (row li of MFMA key tile 2c + t
//     is LDS row 16(2c + t) + li: the fragment rows of one read are consecutive) and in row 32c + 16(a>>1) + 8t + 4(a&1) + b
//     of the V image (the eight rows two neighbouring 16-lane groups transpose-read are consecutive);
//   * the one-hot key-side operands of the folded rel-pos bias are a constant of the window geometry: read from a table the
//     host appends to `rpack` (one 16-byte load per key tile) instead of being rebuilt from compares per tile (a third of the
//     old loop's VALU instructions);
//   * softmax: max over the raw scores, then p = exp2(fma(s, scale*log2e, -m)); both query tiles in one straight-line block;
//   * the V DMA is issued after the rel-pos prologue (whose scratch aliases the V image) and lands under the first chunk's
//     scores + softmax; two barriers per workgroup in all.
// rpack layout (ops.pack_rel_tables(windowed=True)): [2][2][32][HDP] tables | [13][64][8] one-hot fragments | 128 zeros.
typedef __fp16 fp16x4_t __attribute__((__vector_size__(4 * sizeof(__fp16))));

template <int HD>
__global__ __launch_bounds__(448, 4) void wattn_kernel(AttnArgs p) {
  constexpr int NW = 7, NT = NW * 64;
  constexpr int HDP = (HD + 31) / 32 * 32;
  constexpr int KS = HDP / 32;
  constexpr int DT = HD / 16;
  constexpr int CH = HD / 8;          // 16-byte chunks of a head vector
  constexpr int RC = 10;              // 16-byte slots per LDS row
  constexpr int RLD = RC * 8;         // LDS row stride (halfs)
  constexpr int WS = 14, NKEY = WS * WS;
  constexpr int KROWS = 208;          // 13 MFMA key tiles
  constexpr int NINS = (KROWS * RC + 63) / 64;   // DMA wave-instructions per image (33)
  const float LOG2E = 1.4426950408889634f;
  const float RESCALE_THR = 8.0f;

  __shared__ __attribute__((aligned(16))) half_t Ks[KROWS * RLD + 64];   // + the third k-step's over-read of the last row
  __shared__ __attribute__((aligned(16))) half_t Vs[KROWS * RLD];

  const int t = threadIdx.x;
  const int lane = t & 63, wv = __builtin_amdgcn_readfirstlane(t >> 6);
  const int li = lane & 15, g = lane >> 4;
  int h, b, win;
  {
    const int per = p.H, nshare = p.B * p.nwin;
    const int gq = blockIdx.x / (8 * per), r = blockIdx.x % (8 * per);
    const int grp = gq * 8 + (r & 7);
    if (grp >= nshare) return;
    h = r >> 3;
    b = grp / p.nwin;
    win = grp % p.nwin;
  }
  const int wy = win / p.nwx, wx = win % p.nwx;
  const int N = p.N, H = p.H;
  const size_t rs = (size_t)p.ts;
  const half_t* qkv_b = p.qkv + (size_t)b * N * rs;
  const half_t* oh_tab = p.rpack + 2 * 2 * 32 * HDP;
  const half_t* zero_row = oh_tab + 13 * 64 * 8;
  auto win_token = [&](int j) -> int {
    const int y = wy * WS + j / WS, x = wx * WS + j % WS;
    return (y < p.gh && x < p.gw) ? y * p.gw + x : -1;
  };
  // source of one 16-byte slot of the K (which = 1) or V (which = 2) image
  auto dma_image = [&](int which, half_t* img) {
#pragma unroll
    for (int i = 0; i < (NINS + NW - 1) / NW; ++i) {
      const int j = wv + i * NW;                    // wave-uniform
      if (j < NINS) {
        const int S = j * 64 + lane;
        const int R = S / RC, c = S - R * RC;
        const int rho = R & 31, C = R >> 5;
        int key;
        if (which == 1) key = C * 32 + ((rho >> 2) & 3) * 8 + (rho >> 4) * 4 + (rho & 3);
        else key = C * 32 + ((rho >> 4) * 2 + ((rho >> 2) & 1)) * 8 + ((rho >> 3) & 1) * 4 + (rho & 3);
        const half_t* src = zero_row;
        if (key < NKEY) {
          const int tok = win_token(key);
          src = tok >= 0 ? qkv_b + (size_t)tok * rs + (size_t)which * p.ws_ + (size_t)h * p.hs
                         : p.pad_row + ((size_t)which * H + h) * HD;
        }
        if (R < KROWS && c < CH)
          __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src + c * 8),
                                           (__attribute__((address_space(3))) void*)(img + j * 512), 16, 0, 0);
      }
    }
  };
  if (!(p.dbg & 1)) dma_image(1, Ks);

  // ---- query fragments ---------------------------------------------------------------------------------
  const int qrow_blk = wv * 32;
  int qtok[2];
  bool qvalid[2];
  half8_t qf[2][KS];
#pragma unroll
  for (int qt = 0; qt < 2; ++qt) {
    const int q = qrow_blk + qt * 16 + li;
    const int tok = q < NKEY ? win_token(q) : -1;
    qvalid[qt] = tok >= 0;
    qtok[qt] = tok >= 0 ? tok : 0;
    const half_t* qp = qkv_b + (size_t)qtok[qt] * rs + (size_t)h * p.hs;
#pragma unroll
    for (int s = 0; s < KS; ++s) {
      const int c0 = s * 32 + g * 8;
      if (c0 < HD && !(p.dbg & 64)) {
        qf[qt][s] = *reinterpret_cast<const half8_t*>(qp + c0);
      } else {
#pragma unroll
        for (int e = 0; e < 8; ++e) qf[qt][s][e] = (half_t)0.f;
      }
    }
  }

  // ---- rel-pos query terms (add_decomposed_rel_pos, image_encoder.py:337-372), as in attn_kernel; scratch = the V image -----
  half8_t qaug[2][2];
#pragma unroll
  for (int qt = 0; qt < 2; ++qt)
#pragma unroll
    for (int e = 0; e < 8; ++e) qaug[qt][0][e] = qaug[qt][1][e] = (half_t)0.f;
  if (!(p.dbg & 2)) {
    float* ts = reinterpret_cast<float*>(Vs) + wv * 1024;   // [tab 2][r 32][q 16] fp32 = 4 KiB per wave
    const float inv_scale = 1.0f / p.scale;
#pragma unroll
    for (int qt = 0; qt < 2; ++qt) {
#pragma unroll
      for (int tab = 0; tab < 2; ++tab)
#pragma unroll
        for (int tile = 0; tile < 2; ++tile) {
          f32x4 D = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
          for (int part = 0; part < 2; ++part)
#pragma unroll
            for (int s = 0; s < KS; ++s) {
              const half8_t rf = *reinterpret_cast<const half8_t*>(
                  p.rpack + ((size_t)((tab * 2 + part) * 32 + tile * 16 + li)) * HDP + s * 32 + g * 8);
              D = __builtin_amdgcn_mfma_f32_16x16x32_f16(rf, qf[qt][s], D, 0, 0, 0);
            }
#pragma unroll
          for (int i = 0; i < 4; ++i) ts[(tab * 32 + tile * 16 + g * 4 + i) * 16 + li] = D[i];
        }
      const int q = qrow_blk + qt * 16 + li;
      const int qy = (q * 4682) >> 16, qx = q - 14 * qy;
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const int j = g * 8 + e;
        float v = 0.f;
        if (j < 28) {
          const int tab = j >= 14;
          const int r = (tab ? qx : qy) + 13 - (tab ? j - 14 : j);
          v = ts[(tab * 32 + r) * 16 + li] * inv_scale;
        }
        const half_t hi = (half_t)v;
        qaug[qt][0][e] = hi;
        qaug[qt][1][e] = (half_t)(v - (float)hi);
      }
    }
  }
  // K has landed (loads return in order and the query / table loads above were consumed); every wave is done with the scratch
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (!(p.dbg & 1)) dma_image(2, Vs);

  f32x4 ot[DT][2];
#pragma unroll
  for (int d = 0; d < DT; ++d)
#pragma unroll
    for (int qt = 0; qt < 2; ++qt) ot[d][qt] = f32x4{0.f, 0.f, 0.f, 0.f};
  const float sl2 = p.scale * LOG2E;
  float mrun[2] = {-INFINITY, -INFINITY};
  float lrun[2] = {0.f, 0.f};

  auto chunk = [&](auto ntt_c, int c, bool first) {
    constexpr int NTT = decltype(ntt_c)::value;
    half8_t oh[NTT];
#pragma unroll
    for (int tt = 0; tt < NTT; ++tt) {
      if (!(p.dbg & 32)) oh[tt] = *reinterpret_cast<const half8_t*>(oh_tab + ((size_t)((c * 2 + tt) * 64 + lane)) * 8);
      else oh[tt] = qaug[0][1];
    }
    f32x4 st[NTT][2];
#pragma unroll
    for (int tt = 0; tt < NTT; ++tt) {
#pragma unroll
      for (int qt = 0; qt < 2; ++qt) st[tt][qt] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int s = 0; s < KS; ++s) {
        const half8_t kf = *reinterpret_cast<const half8_t*>(&Ks[(c * 32 + tt * 16 + li) * RLD + (s * 4 + g) * 8]);
#pragma unroll
        for (int qt = 0; qt < 2; ++qt)
          st[tt][qt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(kf, qf[qt][s], st[tt][qt], 0, 0, 0);
      }
#pragma unroll
      for (int qt = 0; qt < 2; ++qt) {
        st[tt][qt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(oh[tt], qaug[qt][0], st[tt][qt], 0, 0, 0);
        st[tt][qt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(oh[tt], qaug[qt][1], st[tt][qt], 0, 0, 0);
      }
    }
    if (NTT == 1 && g != 0) {     // the last key tile holds keys 192..207: only 192..195 (lane group 0) exist
#pragma unroll
      for (int qt = 0; qt < 2; ++qt) st[0][qt] = f32x4{-INFINITY, -INFINITY, -INFINITY, -INFINITY};
    }
    float mx[2];
    if (!(p.dbg & 8)) {
#pragma unroll
    for (int qt = 0; qt < 2; ++qt) {
      float m = fmaxf(fmaxf(st[0][qt][0], st[0][qt][1]), fmaxf(st[0][qt][2], st[0][qt][3]));
      if constexpr (NTT == 2)
        m = fmaxf(m, fmaxf(fmaxf(st[1][qt][0], st[1][qt][1]), fmaxf(st[1][qt][2], st[1][qt][3])));
      mx[qt] = m;
    }
#pragma unroll
    for (int qt = 0; qt < 2; ++qt) {
      mx[qt] = fmaxf(mx[qt], __shfl_xor(mx[qt], 16, 64));
      mx[qt] = fmaxf(mx[qt], __shfl_xor(mx[qt], 32, 64));
      mx[qt] *= sl2;
    }
    if (!__all(mx[0] <= mrun[0] + RESCALE_THR && mx[1] <= mrun[1] + RESCALE_THR)) {
#pragma unroll
      for (int qt = 0; qt < 2; ++qt) {
        const float mnew = fmaxf(mrun[qt], mx[qt]);
        const float alpha = __builtin_amdgcn_exp2f(mrun[qt] - mnew);
        mrun[qt] = mnew;
        lrun[qt] *= alpha;
#pragma unroll
        for (int d = 0; d < DT; ++d)
#pragma unroll
          for (int r = 0; r < 4; ++r) ot[d][qt][r] *= alpha;
      }
    }
    }
    half8_t pf[2];
    if (p.dbg & 8) {
#pragma unroll
      for (int qt = 0; qt < 2; ++qt)
#pragma unroll
        for (int e = 0; e < 8; ++e) pf[qt][e] = (half_t)st[0][qt][e & 3];
    } else
#pragma unroll
    for (int qt = 0; qt < 2; ++qt) {
      const float nm = -mrun[qt];
      float pv[NTT][4];
#pragma unroll
      for (int tt = 0; tt < NTT; ++tt)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          pv[tt][r] = __builtin_amdgcn_exp2f(fmaf(st[tt][qt][r], sl2, nm));
          pf[qt][tt * 4 + r] = (half_t)pv[tt][r];
        }
      float ps = (pv[0][0] + pv[0][1]) + (pv[0][2] + pv[0][3]);
      if constexpr (NTT == 2) ps += (pv[1][0] + pv[1][1]) + (pv[1][2] + pv[1][3]);
      else {
#pragma unroll
        for (int e = 4; e < 8; ++e) pf[qt][e] = (half_t)0.f;
      }
      lrun[qt] += ps;
    }
    if (first) {   // the V image: every wave's DMA complete and visible
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();
    }
    // O^T += V^T P^T: the 16 (d) x 32 (keys) fragment = two transposing reads of [4 keys][16 d] blocks per 16-lane group
    const int vrow = c * 32 + (NTT == 2 ? (g >> 1) * 16 : 0) + (g & 1) * 4 + (li >> 2);
    const half_t* vb = &Vs[vrow * RLD + (li & 3) * 4];
    if (!(p.dbg & 16))
#pragma unroll
    for (int d = 0; d < DT; ++d) {
      const fp16x4_t v0 = __builtin_amdgcn_ds_read_tr16_b64_v4f16((__attribute__((address_space(3))) fp16x4_t*)(vb + d * 16));
      fp16x4_t v1 = {0, 0, 0, 0};
      if constexpr (NTT == 2)
        v1 = __builtin_amdgcn_ds_read_tr16_b64_v4f16((__attribute__((address_space(3))) fp16x4_t*)(vb + d * 16 + 8 * RLD));
      half8_t vf;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        vf[e] = (half_t)v0[e];
        vf[4 + e] = (half_t)v1[e];
      }
#pragma unroll
      for (int qt = 0; qt < 2; ++qt) ot[d][qt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(vf, pf[qt], ot[d][qt], 0, 0, 0);
    }
  };

  if (!(p.dbg & 4)) {
    chunk(std::integral_constant<int, 2>{}, 0, true);
#pragma unroll 1
    for (int c = 1; c < 6; ++c) chunk(std::integral_constant<int, 2>{}, c, false);
    chunk(std::integral_constant<int, 1>{}, 6, false);
  }

#pragma unroll
  for (int qt = 0; qt < 2; ++qt) {
    float l = lrun[qt];
    l += __shfl_xor(l, 16, 64);
    l += __shfl_xor(l, 32, 64);
    const float inv = 1.0f / l;
    if (qvalid[qt]) {
      half_t* op = p.out + ((size_t)b * N + qtok[qt]) * ((size_t)H * HD) + (size_t)h * HD;
#pragma unroll
      for (int d = 0; d < DT; ++d) {
        half4_t o = {(half_t)(ot[d][qt][0] * inv), (half_t)(ot[d][qt][1] * inv), (half_t)(ot[d][qt][2] * inv),
                     (half_t)(ot[d][qt][3] * inv)};
        *reinterpret_cast<half4_t*>(op + d * 16 + g * 4) = o;
      }
    }
  }
}

// ---------------------------------------------------------------------------------------------
// Window attention, persistent pipelined form (`wattn_p_kernel`; PSAM_WATTN=2, the default when the caller passes `rpack`).
// The ablation of wattn_kernel (tools/attn_win_ablate.py, 16 slices: launch + stores 56 us, query loads 47, rel-pos prologue
// ~100, K/V DMA ~60, key chunks ~120, SUM ~ the 344-400 us measured) showed that nothing overlaps: every workgroup of the grid
// runs the same phase at the same time, so the chip alternates between a load burst and a compute burst. Here ONE workgroup
// per CU walks a contiguous list of (window, head) items and overlaps them explicitly:
//   * 14 waves; wave w owns ONE 16-row query tile (13 tiles cover the 196 queries; wave 13 only helps with the DMA);
//   * K is double-buffered: the K image and the query fragments of item i+1 are requested at the top of item i; V is
//     single-buffered: V(i+1) is requested at the end of item i (once every wave is done with V(i)) and lands under the rel-pos
//     prologue and the first chunk's scores of item i+1; counted vmcnt waits (loads complete in order);
//   * the rel-pos tables and the one-hot fragments live in LDS for the lifetime of the workgroup (the per-item loads of the
//     24 KiB table from L2 were most of the old prologue's time);
//   * two barriers per item.
// (Tried: the 14th wave, whose query rows do not exist, as a dedicated loader issuing all 66 DMA pieces of an item - one wave's
// serial address arithmetic + DMA issue takes as long as the whole item, 314 -> 488 us; the same wave merely skipping its
// pointless arithmetic - no measurable change. Both dropped.)
// LDS: K 2 x 33 280 + V 33 280 + tables 20 480 + one-hot 13 312 + 14 x 2 KiB prologue scratch = 162 304 bytes.
#ifndef PSAM_WATTN_NTT
#define PSAM_WATTN_NTT 4   // key tiles of 16 per chunk. 64-key chunks (4) against 32-key ones (2): 322 vs 314 us in round 2 - no gain; 343 vs 359 us
#endif                     // in round 3, once the DMA address arithmetic was out of the way (the kernel is bound by its VALU + MFMA issue)

template <int HD>
__global__ __launch_bounds__(896, 4) void wattn_p_kernel(AttnArgs p, int nitems) {
  constexpr int NW = 14, NT = NW * 64;
  constexpr int HDP = (HD + 31) / 32 * 32;
  constexpr int KS = HDP / 32;
  constexpr int DT = HD / 16;
  constexpr int CH = HD / 8;
  constexpr int RC = 10, RLD = RC * 8;
  constexpr int WS = 14, NKEY = WS * WS;
  constexpr int KROWS = 208;
  constexpr int IMG = KROWS * RLD;                   // halfs per image
  constexpr int NINS = (KROWS * RC + 63) / 64;       // 33 DMA wave-instructions per image
  constexpr int NDMA = (NINS + NW - 1) / NW;         // at most 3 per wave
  const float LOG2E = 1.4426950408889634f;
  const float RESCALE_THR = 8.0f;

  __shared__ __attribute__((aligned(16))) half_t Kb0[IMG];
  __shared__ __attribute__((aligned(16))) half_t Kb1[IMG];
  __shared__ __attribute__((aligned(16))) half_t Vs[IMG];
  __shared__ __attribute__((aligned(16))) half_t Tab[2 * 2 * 32 * RLD];
  __shared__ __attribute__((aligned(16))) half_t Oht[13 * 64 * 8];
  __shared__ __attribute__((aligned(16))) float Scr[NW * 512];

  const int t = threadIdx.x;
  const int lane = t & 63, wv = __builtin_amdgcn_readfirstlane(t >> 6);
  const int li = lane & 15, g = lane >> 4;
  const int N = p.N, H = p.H;
  const size_t rs = (size_t)p.ts;
  const half_t* zero_row = p.rpack + 2 * 2 * 32 * HDP + 13 * 64 * 8;

  // items of this workgroup: the (window, head) pairs whose window index is congruent to this XCD, window-major, a contiguous
  // share per workgroup (consecutive items are heads of one window: neighbouring 160-byte slices of the same token rows)
  const int ngrp = p.B * p.nwin;
  const int x = blockIdx.x & 7, sx = blockIdx.x >> 3;
  const int nwg_x = ((int)gridDim.x + 7 - x) >> 3;                 // workgroups on this XCD
  const int cnt_x = ((ngrp + 7 - x) >> 3) * H;                     // items of this XCD
  const int it0 = (int)((long long)sx * cnt_x / nwg_x), it1 = (int)((long long)(sx + 1) * cnt_x / nwg_x);
  if (it0 >= it1) return;

  // Per DMA piece (one wave instruction = 64 lanes x 16 bytes = 6.4 image rows) the geometry of this lane's chunk is a function of the
  // lane and the piece only: packed ONCE per workgroup into one register per piece - kx | ky << 4 of the K image's key, the same
  // << 8 for the V image's key (the two images order their rows differently), chunk << 16, bit 20 / 21: the K / V slot exists and
  // its key is one of the window's 196 (the rows of the keys 196..207 are zeroed once, below, and never loaded). Round 2 recomputed
  // all of it per piece and item (53 VALU instructions per piece, a third of the kernel's VALU issue - and the kernel is bound by
  // exactly that: ~1000 VALU instructions per wave and item on SIMDs that hold four waves); three registers fit the budget.
  int G[NDMA];
#pragma unroll
  for (int i = 0; i < NDMA; ++i) {
    const int S = (wv + i * NW) * 64 + lane;
    const int R = S / RC, c = S - R * RC;
    const int rho = R & 31, C = R >> 5;
    const int keyK = C * 32 + ((rho >> 2) & 3) * 8 + (rho >> 4) * 4 + (rho & 3);
    const int keyV = C * 32 + ((rho >> 4) * 2 + ((rho >> 2) & 1)) * 8 + ((rho >> 3) & 1) * 4 + (rho & 3);
    const int kK = keyK < NKEY ? keyK : 0, kV = keyV < NKEY ? keyV : 0;
    const int kyK = kK / WS, kxK = kK - kyK * WS, kyV = kV / WS, kxV = kV - kyV * WS;
    const bool slot = R < KROWS && c < CH;
    G[i] = kxK | (kyK << 4) | (kxV << 8) | (kyV << 12) | (c << 16) | ((slot && keyK < NKEY) ? (1 << 20) : 0) | ((slot && keyV < NKEY) ? (1 << 21) : 0);
  }
  // K-image DMA instructions this wave ACTUALLY issues per item: a piece whose 64 lanes all fall on the rows of the keys
  // 196..207 is skipped by dma_image (no lane passes its slot test, the compiler branches around the load: pieces 31 / 32, the
  // third pieces of waves 3 / 4), so it must not be counted by the vmcnt in front of the first V read either - with the
  // nominal 33 = 5 x 3 + 9 x 2 those two waves waited one request short and could pass the barrier with a piece of their own
  // V(it) still in flight. (Pieces that straddle the image edge issue TWO loads - image rows and pad row - which only makes
  // the counted wait stricter.)
  int ndma_w = 0;
#pragma unroll
  for (int i = 0; i < NDMA; ++i)
    if (wv + i * NW < NINS && __any((G[i] >> 20) & 1)) ++ndma_w;
  ndma_w = __builtin_amdgcn_readfirstlane(ndma_w);
  const int qidx = wv * 16 + li;                     // this lane's query (window-local; >= 196: none)
  const int qy = (qidx * 4682) >> 16, qx = qidx - 14 * qy;

  struct Item { int b, h, wy14, wx14, grpq; };
  auto item_at = [&](int it) -> Item {               // with divisions: once per workgroup
    const int grpq = it / H, grp = grpq * 8 + x;
    const int win = grp % p.nwin;
    return Item{grp / p.nwin, it - grpq * H, (win / p.nwx) * WS, (win % p.nwx) * WS, grpq};
  };
  auto item_next = [&](const Item& im) -> Item {     // consecutive items: the next head, or head 0 of this XCD's next window
    Item n = im;
    if (++n.h == H) {
      n.h = 0;
      ++n.grpq;
      const int grp = n.grpq * 8 + x;
      const int win = grp % p.nwin;
      n.b = grp / p.nwin;
      n.wy14 = (win / p.nwx) * WS;
      n.wx14 = (win % p.nwx) * WS;
    }
    return n;
  };
  typedef int i32x4 __attribute__((ext_vector_type(4)));
  const int rs2 = (int)rs * 2;                       // bytes per token row of qkv
  auto dma_image = [&](const Item& im, int which, half_t* img) {
    // wave-uniform buffer descriptors (raw, no stride): the head's slice of q / k / v for this item's image, and the pad row
    const half_t* base = p.qkv + (size_t)im.b * N * rs + (size_t)which * p.ws_ + (size_t)im.h * p.hs;
    const half_t* padw = p.pad_row + ((size_t)which * H + im.h) * HD;
    i32x4 srd, srd_pad;
    srd.x = __builtin_amdgcn_readfirstlane((int)(unsigned)(size_t)base);
    srd.y = __builtin_amdgcn_readfirstlane((int)(unsigned)((size_t)base >> 32));
    srd.z = 0x7fffffff;
    srd.w = 0x00020000;
    srd_pad.x = __builtin_amdgcn_readfirstlane((int)(unsigned)(size_t)padw);
    srd_pad.y = __builtin_amdgcn_readfirstlane((int)(unsigned)((size_t)padw >> 32));
    srd_pad.z = HD * 2;
    srd_pad.w = 0x00020000;
#pragma unroll
    for (int i = 0; i < NDMA; ++i) {
      if (wv + i * NW < NINS) {
        int gq = G[i];
        asm volatile("" : "+v"(gq));   // opaque: the fields are unpacked here, per piece - hoisted out of the item loop they are spilled, and
                                       // every reload of a spill is a VMEM load the compiler awaits with vmcnt(0): the prefetch would drain
        const int kk = which == 1 ? gq : (gq >> 8);
        const int y = im.wy14 + ((kk >> 4) & 15), xx = im.wx14 + (kk & 15);
        const unsigned c16 = (unsigned)((gq >> 16) & 15) * 16u;
        const unsigned voff = (unsigned)((y * p.gw + xx) * rs2) + c16;
        // The DMA is issued as inline asm, not through __builtin_amdgcn_global_load_lds: the compiler models the builtin as a
        // pending LDS write and - across the item loop, where it cannot see the hand-placed counted waits - protects every
        // later read of the same array with s_waitcnt vmcnt(0), which drains the NEXT item's prefetch (seen in the ISA in front
        // of the first V read of every item). Unknown to its model, the DMA can only make its own waits more conservative.
        const unsigned ldsb = (unsigned)(size_t)(__attribute__((address_space(3))) half_t*)(img + (wv + i * NW) * 512);
        const int m0v = __builtin_amdgcn_readfirstlane(ldsb);
        if (gq & (1 << (19 + which))) {
          if (y < p.gh && xx < p.gw) {
            asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tbuffer_load_dwordx4 %0, %1, 0 offen lds"
                         :: "v"(voff), "s"(srd), "s"(m0v) : "memory", "m0");
          } else {   // tokens beyond the image edge (the last window row / column): the padded map's value there, fp16(qkv bias)
            asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tbuffer_load_dwordx4 %0, %1, 0 offen lds"
                         :: "v"(c16), "s"(srd_pad), "s"(m0v) : "memory", "m0");
          }
        }
      }
    }
  };
  auto load_q = [&](const Item& im, half8_t (&q)[KS], int& qtok, bool& qvalid) {
    const int y = im.wy14 + qy, xx = im.wx14 + qx;
    qvalid = qidx < NKEY && y < p.gh && xx < p.gw;
    qtok = qvalid ? y * p.gw + xx : 0;
    const half_t* qp = p.qkv + ((size_t)im.b * N + qtok) * rs + (size_t)im.h * p.hs;
#pragma unroll
    for (int s = 0; s < KS; ++s) {
      const int c0 = s * 32 + g * 8;
      if (c0 < HD) {
        q[s] = *reinterpret_cast<const half8_t*>(qp + c0);
      } else {
#pragma unroll
        for (int e = 0; e < 8; ++e) q[s][e] = (half_t)0.f;
      }
    }
  };

  // ---- once per workgroup: tables into LDS, first item's K / Q / V ---------------------------------------------------
  Item cur = item_at(it0);
  dma_image(cur, 1, Kb0);
  half8_t qf[KS];
  int qtok;
  bool qvalid;
  load_q(cur, qf, qtok, qvalid);
  for (int idx = t; idx < 2 * 2 * 32 * RC; idx += NT) {            // [tab][part][32 rows][80]: the tables without their k padding
    const int row = idx / RC, c = idx - row * RC;
    uint4 v = make_uint4(0, 0, 0, 0);
    if (c < CH) v = *reinterpret_cast<const uint4*>(p.rpack + (size_t)row * HDP + c * 8);
    *reinterpret_cast<uint4*>(&Tab[row * RLD + c * 8]) = v;
  }
  for (int idx = t; idx < 13 * 64; idx += NT)
    *reinterpret_cast<uint4*>(&Oht[idx * 8]) = *reinterpret_cast<const uint4*>(p.rpack + 2 * 2 * 32 * HDP + (size_t)idx * 8);
  for (int idx = t; idx < KROWS * RC; idx += NT) {                 // image rows of the keys 196..207: zero, once (no DMA piece touches them)
    const int R = idx / RC, c = idx - R * RC;
    const int rho = R & 31, C = R >> 5;
    const int keyK = C * 32 + ((rho >> 2) & 3) * 8 + (rho >> 4) * 4 + (rho & 3);
    const int keyV = C * 32 + ((rho >> 4) * 2 + ((rho >> 2) & 1)) * 8 + ((rho >> 3) & 1) * 4 + (rho & 3);
    if (keyK >= NKEY) {
      *reinterpret_cast<uint4*>(&Kb0[R * RLD + c * 8]) = make_uint4(0, 0, 0, 0);
      *reinterpret_cast<uint4*>(&Kb1[R * RLD + c * 8]) = make_uint4(0, 0, 0, 0);
    }
    if (keyV >= NKEY) *reinterpret_cast<uint4*>(&Vs[R * RLD + c * 8]) = make_uint4(0, 0, 0, 0);
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  dma_image(cur, 2, Vs);

  const float sl2 = p.scale * LOG2E;
  const float inv_scale = 1.0f / p.scale;
  float* ts = Scr + wv * 512;                                      // [32 r][16 q] fp32, one table at a time

  // The two K buffers are separate arrays and the item body is instantiated once per buffer parity, so that every K-fragment
  // read names its array at compile time: a read through a run-time selected pointer cannot be told apart from the pending DMA
  // into the OTHER buffer, and the compiler put s_waitcnt vmcnt(0) in front of the first one - draining the next item's
  // prefetch at the start of every item (seen in the ISA).
  auto body = [&](auto par_c, int it) {
    constexpr int PAR = decltype(par_c)::value;
    const half_t* Ks = PAR ? Kb1 : Kb0;
    half_t* Knext = PAR ? Kb0 : Kb1;
    // requests for the next item: K image into the other buffer, query fragments into registers
    const bool has_next = it + 1 < it1;
    Item nxt = cur;
    half8_t qn[KS];
    int qtok_n = 0;
    bool qvalid_n = false;
    if (has_next) {
      nxt = item_next(cur);
      dma_image(nxt, 1, Knext);
      load_q(nxt, qn, qtok_n, qvalid_n);
    } else {
#pragma unroll
      for (int s = 0; s < KS; ++s)
#pragma unroll
        for (int e = 0; e < 8; ++e) qn[s][e] = (half_t)0.f;
    }

    // ---- rel-pos query terms of this wave's query tile (image_encoder.py:337-372) ------------------------------------
    half8_t qaug[2];
    {
      float vals[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) vals[e] = 0.f;
#pragma unroll
      for (int tab = 0; tab < 2; ++tab) {
#pragma unroll
        for (int tile = 0; tile < 2; ++tile) {
          f32x4 D = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
          for (int part = 0; part < 2; ++part)
#pragma unroll
            for (int s = 0; s < KS; ++s) {
              const half8_t rf = *reinterpret_cast<const half8_t*>(
                  &Tab[((tab * 2 + part) * 32 + tile * 16 + li) * RLD + (s * 4 + g) * 8]);
              D = __builtin_amdgcn_mfma_f32_16x16x32_f16(rf, qf[s], D, 0, 0, 0);
            }
          // (scratch writes as inline asm for the same reason as the gather reads below; the s_nops cover the MFMA-result ->
          // LDS-data hazard the compiler would otherwise pad)
          const unsigned waddr = (unsigned)(size_t)(__attribute__((address_space(3))) float*)(ts + (tile * 16 + g * 4) * 16 + li);
          asm volatile("s_nop 7\n\ts_nop 7\n\tds_write2_b32 %0, %1, %2 offset1:16\n\tds_write2_b32 %0, %3, %4 offset0:32 offset1:48"
                       :: "v"(waddr), "v"(D[0]), "v"(D[1]), "v"(D[2]), "v"(D[3]) : "memory");
        }
        // The gather reads are issued as inline asm: for an ordinary LDS load the compiler cannot rule out that it reads what a
        // pending LDS-DMA writes and puts s_waitcnt vmcnt(0) in front of it - which here would wait for the NEXT item's K image
        // and query loads, requested a moment ago (seen in the ISA; it made the prefetch across items void). LDS operations of
        // one wave execute in order, so the reads see the ds_writes above.
        float gv[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          const int j = g * 8 + e;                       // k-slot: 0..13 rel_h, 14..27 rel_w, 28..31 unused
          const int r = ((tab ? qx : qy) + 13 - (tab ? j - 14 : j)) & 31;   // every lane reads; lanes that do not own the slot discard
          const unsigned addr = (unsigned)(size_t)(__attribute__((address_space(3))) float*)(ts + r * 16 + li);
          asm volatile("ds_read_b32 %0, %1" : "=v"(gv[e]) : "v"(addr));
        }
        asm volatile("s_waitcnt lgkmcnt(0)"
                     : "+v"(gv[0]), "+v"(gv[1]), "+v"(gv[2]), "+v"(gv[3]), "+v"(gv[4]), "+v"(gv[5]), "+v"(gv[6]), "+v"(gv[7]));
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          const int j = g * 8 + e;
          const bool mine = tab == 0 ? (j < 14) : (j >= 14 && j < 28);
          vals[e] = mine ? gv[e] * inv_scale : vals[e];
        }
      }
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const half_t hi = (half_t)vals[e];
        qaug[0][e] = hi;
        qaug[1][e] = (half_t)(vals[e] - (float)hi);
      }
    }

    f32x4 ot[DT];
#pragma unroll
    for (int d = 0; d < DT; ++d) ot[d] = f32x4{0.f, 0.f, 0.f, 0.f};
    float mrun = -INFINITY, lrun = 0.f;

    // NTT key tiles of 16 (T0 = first tile): 64-key chunks amortise the per-chunk reductions / rescale decision over twice the
    // scores of a 32-key chunk and give the wave four independent MFMA chains
    auto chunk = [&](auto ntt_c, int T0, bool first) {
      constexpr int NTT = decltype(ntt_c)::value;
      constexpr int NS2 = (NTT + 1) / 2;
      f32x4 st[NTT];
#pragma unroll
      for (int tt = 0; tt < NTT; ++tt) {
        st[tt] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int s = 0; s < KS; ++s) {
          const half8_t kf = *reinterpret_cast<const half8_t*>(&Ks[((T0 + tt) * 16 + li) * RLD + (s * 4 + g) * 8]);
          st[tt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(kf, qf[s], st[tt], 0, 0, 0);
        }
        const half8_t oh = *reinterpret_cast<const half8_t*>(&Oht[((T0 + tt) * 64 + lane) * 8]);
        st[tt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(oh, qaug[0], st[tt], 0, 0, 0);
        st[tt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(oh, qaug[1], st[tt], 0, 0, 0);
      }
      if (NTT == 1 && g != 0) st[0] = f32x4{-INFINITY, -INFINITY, -INFINITY, -INFINITY};   // keys 196..207 do not exist
      float mt[NTT];
#pragma unroll
      for (int tt = 0; tt < NTT; ++tt) mt[tt] = fmaxf(fmaxf(st[tt][0], st[tt][1]), fmaxf(st[tt][2], st[tt][3]));
      float mx = mt[0];
      if constexpr (NTT == 2) mx = fmaxf(mt[0], mt[1]);
      if constexpr (NTT == 4) mx = fmaxf(fmaxf(mt[0], mt[1]), fmaxf(mt[2], mt[3]));
      mx = fmaxf(mx, __shfl_xor(mx, 16, 64));
      mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
      mx *= sl2;
      if (!__all(mx <= mrun + RESCALE_THR)) {
        const float mnew = fmaxf(mrun, mx);
        const float alpha = __builtin_amdgcn_exp2f(mrun - mnew);
        mrun = mnew;
        lrun *= alpha;
#pragma unroll
        for (int d = 0; d < DT; ++d)
#pragma unroll
          for (int r = 0; r < 4; ++r) ot[d][r] *= alpha;
      }
      half8_t pf[NS2];
      {
        const float nm = -mrun;
        float psum[NTT];
#pragma unroll
        for (int tt = 0; tt < NTT; ++tt) {
          float pv[4];
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            pv[r] = __builtin_amdgcn_exp2f(fmaf(st[tt][r], sl2, nm));
            pf[tt >> 1][(tt & 1) * 4 + r] = (half_t)pv[r];
          }
          psum[tt] = (pv[0] + pv[1]) + (pv[2] + pv[3]);
        }
        float ps = psum[0];
        if constexpr (NTT == 2) ps = psum[0] + psum[1];
        if constexpr (NTT == 4) ps = (psum[0] + psum[1]) + (psum[2] + psum[3]);
        if constexpr (NTT == 1) {
#pragma unroll
          for (int e = 4; e < 8; ++e) pf[0][e] = (half_t)0.f;
        }
        lrun += ps;
      }
      if (first) {
        // V(it) complete in every wave: the only newer requests of this wave are the next item's K pieces and query loads
        if (has_next) {
          if (ndma_w == 3) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(3 + KS) : "memory");
          else if (ndma_w == 2) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 + KS) : "memory");
          else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(KS) : "memory");   // (no geometry has fewer than two pieces; conservative)
        } else {
          asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        // (a raw barrier: __syncthreads() carries a fence whose vmcnt(0) would drain the next item's prefetch)
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
      }
#pragma unroll
      for (int s2 = 0; s2 < NS2; ++s2) {
        const int vrow = (T0 / 2 + s2) * 32 + (NTT >= 2 ? (g >> 1) * 16 : 0) + (g & 1) * 4 + (li >> 2);
        const half_t* vb = &Vs[vrow * RLD + (li & 3) * 4];
#pragma unroll
        for (int d = 0; d < DT; ++d) {
          const fp16x4_t v0 = __builtin_amdgcn_ds_read_tr16_b64_v4f16((__attribute__((address_space(3))) fp16x4_t*)(vb + d * 16));
          fp16x4_t v1 = {0, 0, 0, 0};
          if constexpr (NTT >= 2)
            v1 = __builtin_amdgcn_ds_read_tr16_b64_v4f16((__attribute__((address_space(3))) fp16x4_t*)(vb + d * 16 + 8 * RLD));
          half8_t vf;
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            vf[e] = (half_t)v0[e];
            vf[4 + e] = (half_t)v1[e];
          }
          ot[d] = __builtin_amdgcn_mfma_f32_16x16x32_f16(vf, pf[s2], ot[d], 0, 0, 0);
        }
      }
    };
    if (PSAM_WATTN_NTT == 4) {
      chunk(std::integral_constant<int, 4>{}, 0, true);
      chunk(std::integral_constant<int, 4>{}, 4, false);
      chunk(std::integral_constant<int, 4>{}, 8, false);
    } else {
      chunk(std::integral_constant<int, 2>{}, 0, true);
#pragma unroll 1
      for (int c = 1; c < 6; ++c) chunk(std::integral_constant<int, 2>{}, c * 2, false);
    }
    chunk(std::integral_constant<int, 1>{}, 12, false);

    // next item's K image and query fragments have landed (they had the whole item). The asm "use" tells the compiler the
    // prefetched fragments are complete HERE: otherwise their first use - the next item's first MFMA, just after that item's
    // own prefetch was issued - gets its s_waitcnt vmcnt(0)
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
    for (int s = 0; s < KS; ++s) asm volatile("" : "+v"(qn[s]));
    {
      float l = lrun;
      l += __shfl_xor(l, 16, 64);
      l += __shfl_xor(l, 32, 64);
      const float inv = 1.0f / l;
      if (qvalid) {
        half_t* op = p.out + ((size_t)cur.b * N + qtok) * ((size_t)H * HD) + (size_t)cur.h * HD;
#pragma unroll
        for (int d = 0; d < DT; ++d) {
          half4_t o = {(half_t)(ot[d][0] * inv), (half_t)(ot[d][1] * inv), (half_t)(ot[d][2] * inv), (half_t)(ot[d][3] * inv)};
          *reinterpret_cast<half4_t*>(op + d * 16 + g * 4) = o;
        }
      }
    }
    // every wave is done with K(it) and V(it); the output stores stay in flight across the barrier (they are older than
    // anything the next item waits for with a count)
    __builtin_amdgcn_s_barrier();
    if (has_next) dma_image(nxt, 2, Vs);
    cur = nxt;
#pragma unroll
    for (int s = 0; s < KS; ++s) qf[s] = qn[s];
    qtok = qtok_n;
    qvalid = qvalid_n;
  };
#pragma unroll 1
  for (int it = it0; it < it1; it += 2) {
    body(std::integral_constant<int, 0>{}, it);
    if (it + 1 < it1) body(std::integral_constant<int, 1>{}, it + 1);
  }
}

// ---------------------------------------------------------------------------------------------
// Global attention, DMA-fed form (`gattn_kernel`; PSAM_GATTN=1): attn_kernel<HD, 0 | 1, 4>'s arithmetic (V2 softmax) with the
// K/V staging of the window kernels: each 64-key tile goes global -> LDS by `global_load_lds_dwordx4` as row-major [64][80]
// images (K rows in MFMA-row order, V rows in the order the transposing reads want, see wattn_kernel), double-buffered, one
// barrier per tile; V^T fragments through `ds_read_b64_tr_b16`. No staging registers (28 VGPRs), no LDS store instructions, no
// transposing writes, no second barrier: the ablation of attn_kernel had the staging at 830 of 1798 us, un-overlapped with the
// 1180 us of scores + softmax + PV.
// (Tried on top: rel_w folded into the score MFMA as 64 one-hot k-slots - 16 more MFMAs per tile for one fma + one LDS read +
// one subtract less per score, no rel_w stage in LDS: 2083 -> 2183 us, dropped. The score accumulators started at
// (rel_h - running max) / (scale*log2e), so that the per-score subtraction only happens on tiles that raise the maximum: 2066 us,
// within the noise - dropped as well.)
template <int HD, int MODE, bool FULL>
__global__ __launch_bounds__(256, 2) void gattn_kernel(AttnArgs p) {
  constexpr int NW = 4, NT = 256, QB = 128;
  constexpr int HDP = (HD + 31) / 32 * 32;
  constexpr int KS = HDP / 32;
  constexpr int DT = HD / 16;
  constexpr int CH = HD / 8;
  constexpr int RC = 10, RLD = RC * 8;
  constexpr int IMG = KT * RLD;                        // halfs per 64-row image
  constexpr int NINS = (KT * RC) / 64;                 // 10 DMA wave-instructions per image
  constexpr int NDMA = (NINS + NW - 1) / NW;           // at most 3 per wave
  const float LOG2E = 1.4426950408889634f;
  const float RESCALE_THR = 8.0f;

  __shared__ __attribute__((aligned(16))) half_t Kb[2 * IMG];
  __shared__ __attribute__((aligned(16))) half_t Vb[2 * IMG];
  constexpr int RWLD = 64;
  __shared__ __attribute__((aligned(16))) float relw_s[MODE == 1 ? QB * RWLD : 4];

  const int t = threadIdx.x;
  const int lane = t & 63, wv = __builtin_amdgcn_readfirstlane(t >> 6);
  const int li = lane & 15, g = lane >> 4;
  int h, b, qblk;
  {
    const int per = p.nqb, nshare = p.B * p.H;
    const int gq = blockIdx.x / (8 * per), r = blockIdx.x % (8 * per);
    const int grp = gq * 8 + (r & 7);
    if (grp >= nshare) return;
    qblk = r >> 3;
    h = grp % p.H;
    b = grp / p.H;
  }
  const int N = p.N, H = p.H;
  const size_t rs = (size_t)p.ts;
  const half_t* qkv_b = p.qkv + (size_t)b * N * rs;
  const int ntiles = (N + KT - 1) / KT;

  // per-lane constants of the DMA pieces: key offset inside the tile and 16-byte chunk, for the K and the V image
  int dkey[2][NDMA], dchunk[NDMA];
#pragma unroll
  for (int i = 0; i < NDMA; ++i) {
    const int S = (wv + i * NW) * 64 + lane;
    const int R = S / RC, c = S - R * RC;
    const int rho = R & 31, C = R >> 5;
    dkey[0][i] = C * 32 + ((rho >> 2) & 3) * 8 + (rho >> 4) * 4 + (rho & 3);
    dkey[1][i] = C * 32 + ((rho >> 4) * 2 + ((rho >> 2) & 1)) * 8 + ((rho >> 3) & 1) * 4 + (rho & 3);
    dchunk[i] = c;
  }
  const half_t* kbase = qkv_b + (size_t)p.ws_ + (size_t)h * p.hs;        // which = 1
  const half_t* vbase = qkv_b + 2 * (size_t)p.ws_ + (size_t)h * p.hs;    // which = 2
  auto dma_tile = [&](int tile, int buf) {
#pragma unroll
    for (int i = 0; i < NDMA; ++i) {
      if (wv + i * NW < NINS) {
        int kk = tile * KT + dkey[0][i], kv = tile * KT + dkey[1][i];
        if (!FULL) { kk = kk < N ? kk : N - 1; kv = kv < N ? kv : N - 1; }   // masked keys: finite data, probabilities are 0
        if (dchunk[i] < CH) {
          __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(kbase + (size_t)kk * rs + dchunk[i] * 8),
                                           (__attribute__((address_space(3))) void*)(Kb + buf * IMG + (wv + i * NW) * 512), 16, 0, 0);
          __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(vbase + (size_t)kv * rs + dchunk[i] * 8),
                                           (__attribute__((address_space(3))) void*)(Vb + buf * IMG + (wv + i * NW) * 512), 16, 0, 0);
        }
      }
    }
  };
  dma_tile(0, 0);

  // ---- query fragments ---------------------------------------------------------------------------------
  const int qrow_blk = qblk * QB + wv * 32;
  int qtok[2];
  bool qvalid[2];
  half8_t qf[2][KS];
#pragma unroll
  for (int qt = 0; qt < 2; ++qt) {
    const int q = qrow_blk + qt * 16 + li;
    qvalid[qt] = q < N;
    qtok[qt] = q < N ? q : 0;
    const half_t* qp = qkv_b + (size_t)qtok[qt] * rs + (size_t)h * p.hs;
#pragma unroll
    for (int s = 0; s < KS; ++s) {
      const int c0 = s * 32 + g * 8;
      if (c0 < HD) {
        qf[qt][s] = *reinterpret_cast<const half8_t*>(qp + c0);
      } else {
#pragma unroll
        for (int e = 0; e < 8; ++e) qf[qt][s][e] = (half_t)0.f;
      }
    }
  }
  const float* relh_q[2] = {nullptr, nullptr};
  if (MODE == 1) {
#pragma unroll
    for (int qt = 0; qt < 2; ++qt) relh_q[qt] = p.rel_h + (((size_t)b * H + h) * N + qtok[qt]) * (size_t)p.gw;
    for (int idx = t; idx < QB * 16; idx += NT) {
      const int qr = idx >> 4, c4 = idx & 15;
      int q = qblk * QB + qr;
      q = q < N ? q : N - 1;
      float4 v = *reinterpret_cast<const float4*>(p.rel_w + (((size_t)b * H + h) * N + q) * (size_t)p.gw + c4 * 4);
      v.x *= LOG2E; v.y *= LOG2E; v.z *= LOG2E; v.w *= LOG2E;
      *reinterpret_cast<float4*>(&relw_s[qr * RWLD + ((c4 ^ (qr & 15)) << 2)]) = v;
    }
  }

  f32x4 ot[DT][2];
#pragma unroll
  for (int d = 0; d < DT; ++d)
#pragma unroll
    for (int qt = 0; qt < 2; ++qt) ot[d][qt] = f32x4{0.f, 0.f, 0.f, 0.f};
  const float sl2 = p.scale * LOG2E;
  float mrun[2] = {-INFINITY, -INFINITY};
  f32x4 lt[2] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}};
  half8_t ones;
#pragma unroll
  for (int e = 0; e < 8; ++e) ones[e] = (half_t)1.f;

  // rel_h of (query, key row) is one scalar per 64-key tile: requested one tile ahead and BEFORE that tile's DMA pieces - loads
  // complete in order, so a request issued after the DMA would wait for the whole next tile to land
  float bh2n[2] = {0.f, 0.f};
  if (MODE == 1) {
#pragma unroll
    for (int qt = 0; qt < 2; ++qt) bh2n[qt] = relh_q[qt][0];
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();   // tile 0 and the rel_w stage are in

#pragma unroll 1
  for (int tile = 0; tile < ntiles; ++tile) {
    const int buf = tile & 1;
    float bh2[2] = {bh2n[0] * LOG2E, bh2n[1] * LOG2E};
    if (MODE == 1 && tile + 1 < ntiles) {
#pragma unroll
      for (int qt = 0; qt < 2; ++qt) bh2n[qt] = relh_q[qt][tile + 1];
    }
    if (tile + 1 < ntiles) dma_tile(tile + 1, buf ^ 1);
    const half_t* Ks = Kb + buf * IMG;
    const half_t* Vs = Vb + buf * IMG;
    const int kbase_i = tile * KT;
    const bool last_partial = !FULL && (tile == ntiles - 1) && (N % KT != 0);
    // S^T = K Q^T (four 16-key MFMA tiles)
    f32x4 st[4][2];
#pragma unroll
    for (int tt = 0; tt < 4; ++tt) {
#pragma unroll
      for (int qt = 0; qt < 2; ++qt) st[tt][qt] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int s = 0; s < KS; ++s) {
        // (hd = 80: the third k-step's lane groups 2 / 3 face zero query columns; they re-read chunk 9 instead of running into
        // the next row - finite data either way, but the row after the LAST one may be LDS nobody wrote)
        const int kc = (s * 4 + 3 < CH) ? s * 4 + g : min(s * 4 + g, CH - 1);
        const half8_t kf = *reinterpret_cast<const half8_t*>(&Ks[(tt * 16 + li) * RLD + kc * 8]);
#pragma unroll
        for (int qt = 0; qt < 2; ++qt) st[tt][qt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(kf, qf[qt][s], st[tt][qt], 0, 0, 0);
      }
    }
    half8_t pf[2][2];
    {
      float mx[2];
#pragma unroll
      for (int qt = 0; qt < 2; ++qt) {
        float mt[4];
#pragma unroll
        for (int tt = 0; tt < 4; ++tt) {
          float4 rw4 = make_float4(0.f, 0.f, 0.f, 0.f);
          if (MODE == 1)
            rw4 = *reinterpret_cast<const float4*>(
                &relw_s[(wv * 32 + qt * 16 + li) * RWLD + ((((tt >> 1) * 8 + g * 2 + (tt & 1)) ^ li) << 2)]);
          const float rwv[4] = {rw4.x, rw4.y, rw4.z, rw4.w};
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            // MODE 0: the raw score stays (scale > 0: the max commutes with it; folded into the exp2 argument below)
            float sv = MODE == 1 ? fmaf(st[tt][qt][r], sl2, rwv[r]) : st[tt][qt][r];
            if (!FULL && last_partial) {
              const int kidx = kbase_i + (tt >> 1) * 32 + g * 8 + (tt & 1) * 4 + r;
              if (kidx >= N) sv = -INFINITY;
            }
            st[tt][qt][r] = sv;
          }
          mt[tt] = fmaxf(fmaxf(st[tt][qt][0], st[tt][qt][1]), fmaxf(st[tt][qt][2], st[tt][qt][3]));
        }
        mx[qt] = fmaxf(fmaxf(mt[0], mt[1]), fmaxf(mt[2], mt[3]));
      }
#pragma unroll
      for (int qt = 0; qt < 2; ++qt) {
        mx[qt] = fmaxf(mx[qt], __shfl_xor(mx[qt], 16, 64));
        mx[qt] = fmaxf(mx[qt], __shfl_xor(mx[qt], 32, 64));
        if (MODE == 1) mx[qt] += bh2[qt];
        else mx[qt] *= sl2;
      }
      if (!__all(mx[0] <= mrun[0] + RESCALE_THR && mx[1] <= mrun[1] + RESCALE_THR)) {
#pragma unroll
        for (int qt = 0; qt < 2; ++qt) {
          const float mnew = fmaxf(mrun[qt], mx[qt]);
          const float alpha = __builtin_amdgcn_exp2f(mrun[qt] - mnew);
          mrun[qt] = mnew;
#pragma unroll
          for (int r = 0; r < 4; ++r) lt[qt][r] *= alpha;
#pragma unroll
          for (int d = 0; d < DT; ++d)
#pragma unroll
            for (int r = 0; r < 4; ++r) ot[d][qt][r] *= alpha;
        }
      }
#pragma unroll
      for (int qt = 0; qt < 2; ++qt) {
        const float moff = mrun[qt] - bh2[qt];
#pragma unroll
        for (int tt = 0; tt < 4; ++tt)
#pragma unroll
          for (int r = 0; r < 4; ++r)
            pf[qt][tt >> 1][(tt & 1) * 4 + r] =
                (half_t)__builtin_amdgcn_exp2f(MODE == 1 ? st[tt][qt][r] - moff : fmaf(st[tt][qt][r], sl2, -moff));
      }
    }
    // O^T += V^T P^T; row sums on the matrix pipe
#pragma unroll
    for (int s2 = 0; s2 < 2; ++s2)
#pragma unroll
      for (int qt = 0; qt < 2; ++qt) lt[qt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ones, pf[qt][s2], lt[qt], 0, 0, 0);
#pragma unroll
    for (int s2 = 0; s2 < 2; ++s2) {
      const int vrow = s2 * 32 + (g >> 1) * 16 + (g & 1) * 4 + (li >> 2);
      const half_t* vb = &Vs[vrow * RLD + (li & 3) * 4];
#pragma unroll
      for (int d = 0; d < DT; ++d) {
        const fp16x4_t v0 = __builtin_amdgcn_ds_read_tr16_b64_v4f16((__attribute__((address_space(3))) fp16x4_t*)(vb + d * 16));
        const fp16x4_t v1 = __builtin_amdgcn_ds_read_tr16_b64_v4f16((__attribute__((address_space(3))) fp16x4_t*)(vb + d * 16 + 8 * RLD));
        half8_t vf;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          vf[e] = (half_t)v0[e];
          vf[4 + e] = (half_t)v1[e];
        }
#pragma unroll
        for (int qt = 0; qt < 2; ++qt) ot[d][qt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(vf, pf[qt][s2], ot[d][qt], 0, 0, 0);
      }
    }
    // the next tile has landed (it had this tile's compute time); every wave is done with this tile's buffers
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
  }

#pragma unroll
  for (int qt = 0; qt < 2; ++qt) {
    const float inv = 1.0f / lt[qt][0];
    if (qvalid[qt]) {
      half_t* op = p.out + ((size_t)b * N + qtok[qt]) * ((size_t)H * HD) + (size_t)h * HD;
#pragma unroll
      for (int d = 0; d < DT; ++d) {
        half4_t o = {(half_t)(ot[d][qt][0] * inv), (half_t)(ot[d][qt][1] * inv), (half_t)(ot[d][qt][2] * inv),
                     (half_t)(ot[d][qt][3] * inv)};
        *reinterpret_cast<half4_t*>(op + d * 16 + g * 4) = o;
      }
    }
  }
}

static int g_attn_v2 = -1;
static int g_gattn = -1;   // 1: gattn_kernel for modes 0 / 1
static int g_wattn = -1;   // 1: wattn_kernel for mode 2 (needs rpack with the appended tables), 0: attn_kernel<HD, 2, 7>
extern "C" int psam_attention_set_variant(int v) {   // bit 0: V2 softmax in the global kernels (0 = round 1's serial chains); bit 1: wattn_kernel
  g_attn_v2 = (v & 1) ? 1 : 0;                        // for the windows. Default 5; A/B and tests
  g_gattn = (v & 8) ? 0 : (v & 16) ? 1 : 3;            // bit 3: the register-staged global kernel (attn_kernel); bit 4: gattn_kernel (HIP, DMA-fed)
                                                      // everywhere; neither: the assembly kernel (gattn_asm_gen.py) where it applies
  g_wattn = (v >> 1) & 3;                             // bits 1-2: 0 attn_kernel<HD, 2, 7>, 1 wattn_kernel, 2 the assembly kernel where it
                                                      // applies (else wattn_p_kernel), 3 wattn_p_kernel everywhere
  return PSAM_OK;
}

// ---- the hand-scheduled global kernels of csrc/gattn_asm_gen.py: PSAM_GATTN=3 -------------------------------------------------
//   psam_gattn_asm_{80,64}_rel    mode 1 with rel_h / rel_w given (psam_relpos wrote them), N a multiple of 256
//   psam_gattn_asm_{80,64}_fused  mode 1 with the packed tables given instead (rpack): the rel-pos terms are computed in the kernel
//   psam_gattn_asm_64_norel       mode 0, hd = 64, any N >= 128 (DINOv2: 1297 / 5330 tokens), last key tile masked
hipFunction_t psam_asm_function(const char* name);     // csrc/gemm.hip: the assembly code object
struct GattnAsmArgs {
  const void* qkv; void* out; const void* rel_h; const void* rel_w;
  int N, H; unsigned spare0; int lg_H; float sl2; int rs2, hs2, ws2, NT, orow; float rwmul; int BH;
  int spare1, nvalid; const int* tab;
};
static_assert(sizeof(GattnAsmArgs) == 96, "kernarg layout of gattn_asm_gen.py");
// Work table of the global kernels, per device and (B * H, query blocks): entry wg = (b * H + h) << 10 | query block, -1 = none.
// Workgroup wg runs on XCD wg % 8 (observed placement, speed only): the (b, h)-major item list is cut into eight equal contiguous
// runs, XCD x walks run x - the same number of workgroups on every XCD, and the ones of an XCD that run side by side read the same
// K / V. (12 heads x 21 query blocks of one 1022^2 DINOv2 image: 252 items = one round of 256 CUs; the arithmetic map of round 4
// - eight (b, h) pairs per row of XCDs - ran it in two.)
struct GattnTable { int* dev; int grid; };
static std::map<unsigned long long, GattnTable> g_gattn_tabs;
static const GattnTable* gattn_worklist(int BH, int nqb) {
  int dev = 0;
  (void)hipGetDevice(&dev);
  int cus, xcds;
  psam_device_geometry(&cus, &xcds);
  if (xcds < 1 || xcds > 64) xcds = 8;
  const unsigned long long key = ((unsigned long long)BH << 32) | ((unsigned long long)nqb << 16) | ((unsigned long long)xcds << 8) | (unsigned)dev;
  auto it = g_gattn_tabs.find(key);
  if (it != g_gattn_tabs.end()) return &it->second;
  const long long items = (long long)BH * nqb;
  const int per = (int)((items + xcds - 1) / xcds);
  std::vector<int> h((size_t)per * xcds, -1);
  for (int x = 0; x < xcds; ++x) {
    const long long i0 = items * x / xcds, i1 = items * (x + 1) / xcds;
    for (long long i = i0; i < i1; ++i) h[(size_t)(i - i0) * xcds + x] = (int)(((i / nqb) << 10) | (i % nqb));
  }
  GattnTable t;
  t.grid = per * xcds;
  t.dev = nullptr;
  if (hipMalloc((void**)&t.dev, h.size() * sizeof(int)) != hipSuccess) { (void)hipGetLastError(); return nullptr; }
  if (hipMemcpy(t.dev, h.data(), h.size() * sizeof(int), hipMemcpyHostToDevice) != hipSuccess) { (void)hipGetLastError(); (void)hipFree(t.dev); return nullptr; }
  return &(g_gattn_tabs[key] = t);
}
// kind: 0 rel (tables from HBM), 1 fused (tables computed in the kernel), 2 norel
static bool gattn_asm_eligible(const AttnArgs& p, int mode, int hd, int kind) {
  if (kind == 2) { if (mode != 0 || hd != 64 || p.N < 128) return false; }
  else if (mode != 1 || (hd != 80 && hd != 64) || (p.N % 256) != 0 || p.N < 256 || p.gw != 64) return false;
  if (kind == 0 && (!p.rel_h || !p.rel_w)) return false;
  if (kind == 1 && (!p.rpack || p.N != 4096 || p.ts != 3LL * p.H * hd)) return false;   // (64 x 64 map: the 127-row tables; token-major qkv)
  const long long nqb = (p.N + 255) / 256;
  if (p.H < 1 || p.H > 64 || (long long)p.B * p.H >= 65536 / p.H || nqb > 1023 || (long long)p.B * p.H * nqb >= (1 << 21)) return false;
  const long long lim = 0x7fffffffLL;
  return (long long)p.N * p.ts * 2 < lim && p.hs * 2 < lim && p.ws_ * 2 < lim && (long long)p.N * p.H * hd * 2 < lim;
}
static int launch_gattn_asm(const AttnArgs& p, hipStream_t s, int hd, int kind) {
  hipFunction_t f = psam_asm_function(kind == 2 ? "psam_gattn_asm_64_norel"
                                      : kind == 1 ? (hd == 64 ? "psam_gattn_asm_64_fused" : "psam_gattn_asm_80_fused")
                                                  : (hd == 64 ? "psam_gattn_asm_64_rel" : "psam_gattn_asm_80_rel"));
  if (!f) return PSAM_ERR_LAUNCH;
  GattnAsmArgs a;
  const int nqb = (p.N + 255) / 256;
  a.qkv = p.qkv; a.out = p.out;
  a.rel_h = kind == 1 ? (const void*)p.rpack : (const void*)p.rel_h; a.rel_w = p.rel_w;
  const GattnTable* tab = gattn_worklist(p.B * p.H, nqb);
  if (!tab) return PSAM_ERR_LAUNCH;
  a.N = p.N; a.H = p.H;
  a.spare0 = 0;
  a.lg_H = (65536 + p.H - 1) / p.H;                                               // (the slot carries ceil(2^16 / H))
  a.sl2 = p.scale * 1.4426950408889634f;
  a.rs2 = (int)(p.ts * 2); a.hs2 = (int)(p.hs * 2); a.ws2 = (int)(p.ws_ * 2);
  a.NT = (p.N + 63) / 64; a.orow = p.H * hd * 2;
  a.rwmul = 1.0f / p.scale;      // rel_w is staged as rel_w / scale: it enters the score MFMAs as their accumulator input
  a.BH = p.B * p.H;              // workgroups of the groups beyond it (the grid is rounded up to eight (b, h) per row of XCDs) leave at once
  a.spare1 = 0; a.nvalid = p.N - (a.NT - 1) * 64; a.tab = tab->dev;
  size_t sz = sizeof(a);
  void* extra[] = {HIP_LAUNCH_PARAM_BUFFER_POINTER, &a, HIP_LAUNCH_PARAM_BUFFER_SIZE, &sz, HIP_LAUNCH_PARAM_END};
  const int grid = tab->grid;
  if (hipModuleLaunchKernel(f, grid, 1, 1, 256, 1, 1, 0, s, nullptr, extra) != hipSuccess) {
    (void)hipGetLastError();
    return PSAM_ERR_LAUNCH;
  }
  return PSAM_OK;
}

// ---- the hand-written window kernel of csrc/wattn_asm_gen.py (hd = 80, 14 x 14 windows of a 64 x 64 token map): PSAM_WATTN=3 ----------
struct WattnAsmArgs {
  const void* qkv; void* out; const void* rpack; const void* pad_row; const int* work; const unsigned* geom;
  int N, H, rs2, hs2, ws2, orow, G; float sl2, isc; int pad0, pad1, pad2;
};
static_assert(sizeof(WattnAsmArgs) == 96, "kernarg layout of wattn_asm_gen.py");
static bool wattn_asm_eligible(const AttnArgs& p, int hd) {
  if (hd != 80 || !p.rpack || !p.pad_row || p.gh != 64 || p.gw != 64 || p.ws != 14 || p.N != 4096) return false;
  if (p.hs != hd || p.ws_ != (long long)p.H * hd || p.ts != 3LL * p.H * hd) return false;      // token-major packed qkv
  return p.B <= 255 && p.H <= 255 && (long long)p.N * p.ts * 2 < 0x20000000LL;
}
// Host tables of the kernel, built once per device: (a) DMA geometry - per (image kind, piece of 64 lanes, lane) the byte offset of
// the lane's 16-byte chunk relative to the window's first token, its offset inside a pad row, and per edge class (partial last
// window row / column) the exec masks "token row" / "pad row" of every piece; (b) the work list - per workgroup its (image, head,
// window) items in the XCD-aware order of wattn_p_kernel (windows of one XCD together, the heads of a window back to back).
// (one geometry table per row stride, never rewritten once published: launches already enqueued and captured graphs keep its address)
struct WattnTables { std::map<long long, unsigned*> geoms; std::map<unsigned long long, std::pair<int*, int>> work; };
static std::map<int, WattnTables> g_wattn_tabs;
static const unsigned* wattn_geometry(WattnTables& t, long long rs2) {
  { auto it = t.geoms.find(rs2); if (it != t.geoms.end()) return it->second; }
  constexpr int P = 42, NP = 35, WS = 14, GWID = 64;
  std::vector<unsigned> h((size_t)3 * P * 64 + (size_t)4 * 2 * P * 4, 0u);
  for (int img = 0; img < 2; ++img)
    for (int pc = 0; pc < NP; ++pc)
      for (int l = 0; l < 64; ++l) {
        const int S = pc * 64 + l, R = S / 10, c = S - R * 10;
        int ky, kx;
        if (img == 0) { ky = R >> 4; kx = R & 15; }
        else { const int C = R >> 5, rho = R & 31; ky = 2 * C + ((rho >> 3) & 1); kx = 4 * (2 * (rho >> 4) + ((rho >> 2) & 1)) + (rho & 3); }
        const bool valid = kx < WS && ky < WS;
        h[((size_t)img * P + pc) * 64 + l] = valid ? (unsigned)((long long)(ky * GWID + kx) * rs2 + c * 16) : 0u;
        h[((size_t)2 * P + pc) * 64 + l] = (unsigned)(c * 16);
        for (int cls = 0; cls < 4; ++cls) {
          const bool ey = cls & 2, ex = cls & 1;
          const bool in = valid && !(ey && ky >= 8) && !(ex && kx >= 8), pad = valid && !in;
          unsigned* m = &h[(size_t)3 * P * 64 + (((size_t)cls * 2 + img) * P + pc) * 4];
          if (in) m[l >> 5] |= 1u << (l & 31);
          if (pad) m[2 + (l >> 5)] |= 1u << (l & 31);
        }
      }
  unsigned* g = nullptr;
  if (hipMalloc((void**)&g, h.size() * 4) != hipSuccess) { (void)hipGetLastError(); return nullptr; }
  if (hipMemcpy(g, h.data(), h.size() * 4, hipMemcpyHostToDevice) != hipSuccess) { (void)hipGetLastError(); (void)hipFree(g); return nullptr; }
  return t.geoms[rs2] = g;
}
static const int* wattn_worklist(WattnTables& t, int B, int H, int grid) {
  const unsigned long long key = ((unsigned long long)B << 40) | ((unsigned long long)H << 20) | (unsigned)grid;
  auto it = t.work.find(key);
  if (it != t.work.end()) return it->second.first;
  const int nwin = 25, ngrp = B * nwin;
  std::vector<std::vector<int>> lists(grid);
  size_t rows = 0;
  for (int wg = 0; wg < grid; ++wg) {
    const int x = wg & 7, sx = wg >> 3;
    const int nwg_x = (grid + 7 - x) >> 3;
    const int nwin_x = (ngrp + 7 - x) >> 3;        // windows (b, wy, wx) of XCD x: grp = q * 8 + x
    auto push = [&](int grpq, int hh) {
      const int grp = grpq * 8 + x, win = grp % nwin, b = grp / nwin;
      lists[wg].push_back(b | (hh << 8) | ((win / 5) << 16) | ((win % 5) << 24));
    };
    if (nwg_x % H == 0) {
      // the workgroups of an XCD that run side by side take the H heads of the SAME window(s): a token's q / k / v rows of adjacent
      // heads are 160-byte neighbours, i.e. they share 128-byte lines - fetched once into the XCD's L2 instead of once per head
      // (measured, 16 slices: FETCH_SIZE 834 -> see DESIGN.md; a workgroup keeps ONE head and walks the windows)
      const int lanes = nwg_x / H, hh = sx % H, lane = sx / H;
      for (int q = lane; q < nwin_x; q += lanes) push(q, hh);
    } else {
      const long long cnt_x = (long long)nwin_x * H;
      const int it0 = (int)((long long)sx * cnt_x / nwg_x), it1 = (int)((long long)(sx + 1) * cnt_x / nwg_x);
      for (int i = it0; i < it1; ++i) push(i / H, i % H);
    }
    rows = lists[wg].size() > rows ? lists[wg].size() : rows;
  }
  rows += 2;
  std::vector<int> h(rows * grid, -1);
  for (int wg = 0; wg < grid; ++wg)
    for (size_t i = 0; i < lists[wg].size(); ++i) h[i * grid + wg] = lists[wg][i];
  int* dev = nullptr;
  if (hipMalloc((void**)&dev, h.size() * 4) != hipSuccess) { (void)hipGetLastError(); return nullptr; }
  if (hipMemcpy(dev, h.data(), h.size() * 4, hipMemcpyHostToDevice) != hipSuccess) { (void)hipGetLastError(); return nullptr; }
  t.work[key] = std::make_pair(dev, (int)rows);
  return dev;
}
static int launch_wattn_asm(const AttnArgs& p, hipStream_t s) {
  hipFunction_t f = psam_asm_function("psam_wattn_asm_80");
  if (!f) return PSAM_ERR_LAUNCH;
  int cus, xcds, dev = 0;
  psam_device_geometry(&cus, &xcds);
  (void)hipGetDevice(&dev);
  WattnTables& t = g_wattn_tabs[dev];
  const int nitems = p.B * p.nwin * p.H;
  const int grid = nitems < cus ? nitems : cus;
  WattnAsmArgs a;
  a.geom = wattn_geometry(t, p.ts * 2);
  a.work = wattn_worklist(t, p.B, p.H, grid);
  if (!a.geom || !a.work) return PSAM_ERR_LAUNCH;
  a.qkv = p.qkv; a.out = p.out; a.rpack = p.rpack; a.pad_row = p.pad_row;
  a.N = p.N; a.H = p.H; a.rs2 = (int)(p.ts * 2); a.hs2 = (int)(p.hs * 2); a.ws2 = (int)(p.ws_ * 2); a.orow = p.H * 80 * 2; a.G = grid;
  a.sl2 = p.scale * 1.4426950408889634f; a.isc = 1.0f / p.scale; a.pad0 = a.pad1 = a.pad2 = 0;
  size_t sz = sizeof(a);
  void* extra[] = {HIP_LAUNCH_PARAM_BUFFER_POINTER, &a, HIP_LAUNCH_PARAM_BUFFER_SIZE, &sz, HIP_LAUNCH_PARAM_END};
  if (hipModuleLaunchKernel(f, grid, 1, 1, 14 * 64, 1, 1, 0, s, nullptr, extra) != hipSuccess) {
    (void)hipGetLastError();
    return PSAM_ERR_LAUNCH;
  }
  return PSAM_OK;
}

template <int HD, bool V2>
static int launch_attn(AttnArgs p, int mode, hipStream_t s) {
  if (mode == 2) {
    constexpr int NW = 7;
    p.nqb = 1;
    const int groups8 = (p.B * p.nwin + 7) / 8;
    // 2 (default): the assembly kernel of wattn_asm_gen.py where it applies, else wattn_p_kernel; 3: wattn_p_kernel everywhere
    if (g_wattn < 0) { const char* e = getenv("PSAM_WATTN"); g_wattn = e ? atoi(e) : 2; }
    if (g_wattn == 2 && wattn_asm_eligible(p, HD)) return launch_wattn_asm(p, s);
    if (g_wattn >= 2 && p.rpack != nullptr) {
      int cus, xcds;
      psam_device_geometry(&cus, &xcds);
      const int nitems = p.B * p.nwin * p.H;
      hipLaunchKernelGGL((wattn_p_kernel<HD>), dim3(nitems < cus ? nitems : cus), dim3(14 * 64), 0, s, p, nitems);
      return psam_launch_status();
    }
    if (g_wattn == 1 && p.rpack != nullptr) {
      hipLaunchKernelGGL((wattn_kernel<HD>), dim3(groups8 * 8 * p.H), dim3(NW * 64), 0, s, p);
      return psam_launch_status();
    }
    // the window kernel sits at its 128-VGPR budget (two workgroups per CU): V2's extra live state spills there, so it keeps V1
    hipLaunchKernelGGL((attn_kernel<HD, 2, NW, false, false>), dim3(groups8 * 8 * p.H), dim3(NW * 64), 0, s, p);
  } else {
    constexpr int NW = 4;
    p.nqb = (p.N + NW * 32 - 1) / (NW * 32);
    const int groups8 = (p.B * p.H + 7) / 8;
    dim3 grid(groups8 * 8 * p.nqb), block(NW * 64);
    const bool full = (p.N % 64) == 0;
    if (g_gattn < 0) { const char* e = getenv("PSAM_GATTN"); g_gattn = e ? atoi(e) : 3; }   // 3: the assembly kernel where it applies, else gattn_kernel
    if (mode == 1 && (!p.rel_h || !p.rel_w)) {      // tables only: the fused kernel or nothing (psam_attention_fused_relpos tells the caller)
      if (p.rpack && gattn_asm_eligible(p, mode, HD, 1)) return launch_gattn_asm(p, s, HD, 1);
      return PSAM_ERR_ARG;
    }
    if (g_gattn == 3 && mode == 1 && gattn_asm_eligible(p, mode, HD, 0)) return launch_gattn_asm(p, s, HD, 0);
    if (g_gattn == 3 && mode == 0 && gattn_asm_eligible(p, mode, HD, 2)) return launch_gattn_asm(p, s, HD, 2);
    if (g_gattn && V2) {
      if (mode == 1) {
        if (full) hipLaunchKernelGGL((gattn_kernel<HD, 1, true>), grid, block, 0, s, p);
        else hipLaunchKernelGGL((gattn_kernel<HD, 1, false>), grid, block, 0, s, p);
      } else {
        if (full) hipLaunchKernelGGL((gattn_kernel<HD, 0, true>), grid, block, 0, s, p);
        else hipLaunchKernelGGL((gattn_kernel<HD, 0, false>), grid, block, 0, s, p);
      }
      return psam_launch_status();
    }
    if (mode == 1) {
      if (full) hipLaunchKernelGGL((attn_kernel<HD, 1, NW, true, V2>), grid, block, 0, s, p);
      else hipLaunchKernelGGL((attn_kernel<HD, 1, NW, false, V2>), grid, block, 0, s, p);
    } else {
      if (full) hipLaunchKernelGGL((attn_kernel<HD, 0, NW, true, V2>), grid, block, 0, s, p);
      else hipLaunchKernelGGL((attn_kernel<HD, 0, NW, false, V2>), grid, block, 0, s, p);
    }
  }
  return psam_launch_status();
}

// mode 0: global, no bias.  mode 1: global + decomposed rel-pos (requires gw == 64, N == gh*gw).
// mode 2: ws x ws windows over the gh x gw token map (zero-padded as the reference) + rel-pos folded into the MFMA
//         (relq = psam_relpos' windowed output).
extern "C" int psam_attention_f16(const void* qkv, void* out, const float* rel_h, const float* rel_w,
                                  const void* relq, const void* rpack, const void* pad_row, int B, int N, int H, int hd,
                                  float scale, int mode, int gh, int gw, int ws, int head_major, void* stream) {
  if (B <= 0 || N <= 0 || H <= 0 || mode < 0 || mode > 2) return PSAM_ERR_ARG;
  AttnArgs p;
  if (head_major) {
    p.ts = hd; p.hs = (long long)B * N * hd; p.ws_ = (long long)H * B * N * hd;
  } else {
    p.ts = 3LL * H * hd; p.hs = hd; p.ws_ = (long long)H * hd;
  }
  p.qkv = (const half_t*)qkv;
  p.out = (half_t*)out;
  p.rel_h = rel_h;
  p.rel_w = rel_w;
  p.pad_row = (const half_t*)pad_row;
  p.relq = (const half_t*)relq;
  p.rpack = (const half_t*)rpack;
  p.B = B;
  p.N = N;
  p.H = H;
  p.scale = scale;
  p.gh = gh;
  p.gw = gw;
  p.ws = ws;
  p.nwx = p.nwin = 0;
  { const char* e = getenv("PSAM_ATTN_DBG"); p.dbg = e ? atoi(e) : 0; }
  if (mode == 1) {   // rel_h / rel_w from psam_relpos, or the packed tables alone (the rel-pos terms are then computed in the kernel)
    if (gw != KT || gh * gw != N || ((!rel_h || !rel_w) && !rpack)) return PSAM_ERR_ARG;
  }
  if (mode == 2) {
    // the resident-window schedule is laid out for 14 x 14 = 3 x 64 + 4 keys (SAM's window_size, build_sam.py:73)
    if (ws != 14 || gh * gw != N || (!relq && !rpack) || !pad_row) return PSAM_ERR_ARG;
    p.nwx = (gw + ws - 1) / ws;
    p.nwin = p.nwx * ((gh + ws - 1) / ws);
  }
  hipStream_t s = (hipStream_t)stream;
  if (g_attn_v2 < 0) { const char* e = getenv("PSAM_ATTN_V2"); g_attn_v2 = e ? (atoi(e) != 0) : 1; }
  if (g_attn_v2) {
    if (hd == 64) return launch_attn<64, true>(p, mode, s);
    if (hd == 80) return launch_attn<80, true>(p, mode, s);
  } else {
    if (hd == 64) return launch_attn<64, false>(p, mode, s);
    if (hd == 80) return launch_attn<80, false>(p, mode, s);
  }
  return PSAM_ERR_ARG;
}

// 1 when psam_attention_f16(mode 1) takes the packed rel-pos tables (`rpack` of the global form, ops.pack_rel_tables) in place of
// rel_h / rel_w for this shape: the assembly kernels psam_gattn_asm_{80,64}_fused compute the terms themselves (no psam_relpos)
extern "C" int psam_attention_fused_relpos(int B, int N, int H, int hd, int gh, int gw) {
  if (B <= 0 || N <= 0 || H <= 0 || gw != KT || gh * gw != N) return 0;
  AttnArgs p;
  p.ts = 3LL * H * hd; p.hs = hd; p.ws_ = (long long)H * hd;
  p.B = B; p.N = N; p.H = H; p.gh = gh; p.gw = gw;
  p.rel_h = p.rel_w = nullptr;
  p.rpack = reinterpret_cast<const half_t*>(&p);      // (only tested for presence)
  if (g_gattn < 0) { const char* e = getenv("PSAM_GATTN"); g_gattn = e ? atoi(e) : 3; }
  if (g_gattn != 3 || !gattn_asm_eligible(p, 1, hd, 1)) return 0;
  return psam_asm_function(hd == 64 ? "psam_gattn_asm_64_fused" : "psam_gattn_asm_80_fused") != nullptr;
}

// ---------------------------------------------------------------------------------------------
// Decomposed relative-position terms (models/segment_anything/modeling/image_encoder.py:303-372):
//   rel_h[b,h,n,kh] = q[b,n,h,:] . Rh[qy - kh + (K-1), :],   rel_w[b,h,n,kw] = q . Rw[qx - kw + (K-1), :]
// with (qy,qx) the query's position inside its attention region (whole 64x64 map, or its 14x14 window) and K the
// region side. Rh/Rw are the (2K-1, hd) tables `rel_pos_h/w` (get_rel_pos is the identity gather when the table
// length already equals 2K-1, the only case SAM's 1024 input hits). q is the UNSCALED fp16 query of the packed qkv.
//
// MFMA formulation: T[n, r] = q_n . R[r] for ALL table rows r (a [64 tokens] x [RP rows] x [hd] GEMM per block on
// v_mfma_f32_16x16x32_f16, tables pre-split into fp16 hi + lo parts so the result is fp32-accurate), then each
// (n, r) is scattered to its key index k = pos(n) - r + K-1 when 0 <= k < K.
//   global  (windowed = 0): rel_h / rel_w fp32 [B,H,N,64]
//   windowed             : relq fp16 [B,H,N,2,32] = hi/lo halves of (rel_h[0:K] | rel_w[0:K] | 0) / scale, the extra
//                          k-slots of the window attention's S^T MFMA (caller zero-fills the buffer once).
// Rpack: fp16 [2 (h,w)][2 (hi,lo)][RP][HDP], zero padded; RP = 128 (global) or 32 (windowed).
template <int HD, int QT>
__global__ __launch_bounds__(256) void relpos_mfma_kernel(const half_t* __restrict__ qkv, const half_t* __restrict__ Rpack,
                                                          float* __restrict__ rel_h, float* __restrict__ rel_w,
                                                          half_t* __restrict__ relq, int N, int H, int gw, int K, int RP,
                                                          int windowed, float inv_scale, long long ts, long long hs) {
  // QT query tiles of 16 tokens per wave (64 * QT tokens per workgroup): every table fragment a wave fetches (98 KB of table per
  // wave for the global form) serves QT MFMAs instead of one - the kernel was bound by exactly those L2 fetches
  constexpr int HDP = (HD + 31) / 32 * 32;
  constexpr int KS = HDP / 32;
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int li = lane & 15, g = lane >> 4;
  const int h = blockIdx.y, b = blockIdx.z;
  const int n0 = (blockIdx.x * 4 + wv) * 16 * QT;
  half8_t qf[QT][KS];
#pragma unroll
  for (int t = 0; t < QT; ++t) {
    const half_t* qp = qkv + ((size_t)b * N + n0 + t * 16 + li) * (size_t)ts + (size_t)h * (size_t)hs;   // q = which 0
#pragma unroll
    for (int s = 0; s < KS; ++s) {
      const int c0 = s * 32 + g * 8;
      if (c0 < HD) {
        qf[t][s] = *reinterpret_cast<const half8_t*>(qp + c0);
      } else {
#pragma unroll
        for (int e = 0; e < 8; ++e) qf[t][s][e] = (half_t)0.f;
      }
    }
  }
  const size_t bh = (size_t)b * H + h;
#pragma unroll 1
  for (int tab = 0; tab < 2; ++tab) {
    const half_t* Rhi = Rpack + (size_t)(tab * 2 + 0) * RP * HDP;
    const half_t* Rlo = Rpack + (size_t)(tab * 2 + 1) * RP * HDP;
#pragma unroll 2
    for (int rt = 0; rt < RP / 16; ++rt) {
      half8_t rh[KS], rl[KS];
#pragma unroll
      for (int s = 0; s < KS; ++s) {
        const size_t off = (size_t)(rt * 16 + li) * HDP + s * 32 + g * 8;
        rh[s] = *reinterpret_cast<const half8_t*>(Rhi + off);
        rl[s] = *reinterpret_cast<const half8_t*>(Rlo + off);
      }
#pragma unroll
      for (int t = 0; t < QT; ++t) {
        // table fragment as the A operand, query fragment as B: lane (li, g) then holds token n0 + 16 t + li against FOUR
        // consecutive table rows r0 .. r0 + 3 = four consecutive key indices k0 .. k0 - 3 of one output row: one 16-byte store per
        // lane instead of four scattered 4-byte ones (the kernel was bound by the issue of those)
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int s = 0; s < KS; ++s) {
          acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(rh[s], qf[t][s], acc, 0, 0, 0);
          acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(rl[s], qf[t][s], acc, 0, 0, 0);
        }
        const int n = n0 + t * 16 + li;
        const int r0 = rt * 16 + g * 4;
        int pos = tab == 0 ? n / gw : n % gw;
        if (windowed) pos %= K;
        const int k0 = pos - r0 + K - 1;               // key index of acc[0]; acc[e] belongs to k0 - e
        if (windowed) {
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const int kk = k0 - e;
            if (kk >= 0 && kk < K && r0 + e < 2 * K - 1) {
              const float v = acc[e] * inv_scale;
              const half_t hi = (half_t)v;
              const half_t lo = (half_t)(v - (float)hi);
              half_t* o = relq + (bh * N + n) * 64 + tab * K + kk;
              o[0] = hi;
              o[32] = lo;
            }
          }
        } else {
          float* o = (tab == 0 ? rel_h : rel_w) + (bh * N + n) * 64;
          if (k0 - 3 >= 0 && k0 < K && r0 + 3 < 2 * K - 1) {
            *reinterpret_cast<float4*>(o + k0 - 3) = make_float4(acc[3], acc[2], acc[1], acc[0]);   // 4-byte aligned is enough
          } else {
#pragma unroll
            for (int e = 0; e < 4; ++e) {
              const int kk = k0 - e;
              if (kk >= 0 && kk < K && r0 + e < 2 * K - 1) o[kk] = acc[e];
            }
          }
        }
      }
    }
  }
}

extern "C" int psam_relpos(const void* qkv, const void* Rpack, float* rel_h, float* rel_w, void* relq, int B, int N,
                           int H, int hd, int gw, int K, int windowed, float scale, int head_major, void* stream) {
  if (B <= 0 || N <= 0 || (N % 64) != 0 || K <= 0 || K > 64) return PSAM_ERR_ARG;
  if (windowed ? (K > 16 || !relq) : (!rel_h || !rel_w)) return PSAM_ERR_ARG;
  const int RP = windowed ? 32 : 128;
  dim3 block(256);
  hipStream_t s = (hipStream_t)stream;
  const float inv = 1.0f / scale;
  const long long ts = head_major ? hd : 3LL * H * hd, hs = head_major ? (long long)B * N * hd : hd;
  const bool q4 = (N % 256) == 0;      // four query tiles per wave when the token count allows
  dim3 grid(q4 ? N / 256 : N / 64, H, B);
#define PSAM_RELPOS_LAUNCH(HD_, QT_)                                                                                        \
  hipLaunchKernelGGL((relpos_mfma_kernel<HD_, QT_>), grid, block, 0, s, (const half_t*)qkv, (const half_t*)Rpack, rel_h, rel_w, \
                     (half_t*)relq, N, H, gw, K, RP, windowed, inv, ts, hs)
  if (hd == 64) { if (q4) PSAM_RELPOS_LAUNCH(64, 4); else PSAM_RELPOS_LAUNCH(64, 1); }
  else if (hd == 80) { if (q4) PSAM_RELPOS_LAUNCH(80, 4); else PSAM_RELPOS_LAUNCH(80, 1); }
  else return PSAM_ERR_ARG;
#undef PSAM_RELPOS_LAUNCH
  return psam_launch_status();
}
