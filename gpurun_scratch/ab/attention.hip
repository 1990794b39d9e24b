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
#include <type_traits>

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

static int g_attn_v2 = -1;
extern "C" int psam_attention_set_variant(int v) {   // 0: serial softmax chains (round 1), 1: V2 (default); A/B and tests
  g_attn_v2 = v ? 1 : 0;
  return PSAM_OK;
}

template <int HD, bool V2>
static int launch_attn(AttnArgs p, int mode, hipStream_t s) {
  if (mode == 2) {
    constexpr int NW = 7;
    p.nqb = 1;
    const int groups8 = (p.B * p.nwin + 7) / 8;
    // the window kernel sits at its 128-VGPR budget (two workgroups per CU): V2's extra live state spills there, so it keeps V1
    hipLaunchKernelGGL((attn_kernel<HD, 2, NW, false, false>), dim3(groups8 * 8 * p.H), dim3(NW * 64), 0, s, p);
  } else {
    constexpr int NW = 4;
    p.nqb = (p.N + NW * 32 - 1) / (NW * 32);
    const int groups8 = (p.B * p.H + 7) / 8;
    dim3 grid(groups8 * 8 * p.nqb), block(NW * 64);
    const bool full = (p.N % 64) == 0;
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
  if (mode == 1) {
    if (gw != KT || gh * gw != N || !rel_h || !rel_w) return PSAM_ERR_ARG;
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
template <int HD>
__global__ __launch_bounds__(256) void relpos_mfma_kernel(const half_t* __restrict__ qkv, const half_t* __restrict__ Rpack,
                                                          float* __restrict__ rel_h, float* __restrict__ rel_w,
                                                          half_t* __restrict__ relq, int N, int H, int gw, int K, int RP,
                                                          int windowed, float inv_scale, long long ts, long long hs) {
  constexpr int HDP = (HD + 31) / 32 * 32;
  constexpr int KS = HDP / 32;
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int li = lane & 15, g = lane >> 4;
  const int h = blockIdx.y, b = blockIdx.z;
  const int n0 = blockIdx.x * 64 + wv * 16;
  // A operand: 16 query rows
  half8_t qf[KS];
  {
    const half_t* qp = qkv + ((size_t)b * N + n0 + li) * (size_t)ts + (size_t)h * (size_t)hs;   // q = which 0
#pragma unroll
    for (int s = 0; s < KS; ++s) {
      const int c0 = s * 32 + g * 8;
      if (c0 < HD) {
        qf[s] = *reinterpret_cast<const half8_t*>(qp + c0);
      } else {
#pragma unroll
        for (int e = 0; e < 8; ++e) qf[s][e] = (half_t)0.f;
      }
    }
  }
  const size_t bh = (size_t)b * H + h;
#pragma unroll 1
  for (int tab = 0; tab < 2; ++tab) {
    const half_t* Rhi = Rpack + (size_t)(tab * 2 + 0) * RP * HDP;
    const half_t* Rlo = Rpack + (size_t)(tab * 2 + 1) * RP * HDP;
#pragma unroll 1
    for (int rt = 0; rt < RP / 16; ++rt) {
      f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int s = 0; s < KS; ++s) {
        const size_t off = (size_t)(rt * 16 + li) * HDP + s * 32 + g * 8;
        half8_t rh = *reinterpret_cast<const half8_t*>(Rhi + off);
        half8_t rl = *reinterpret_cast<const half8_t*>(Rlo + off);
        acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(qf[s], rh, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(qf[s], rl, acc, 0, 0, 0);
      }
      // acc[e]: token n0 + g*4 + e, table row r = rt*16 + li
      const int r = rt * 16 + li;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int n = n0 + g * 4 + e;
        int pos = tab == 0 ? n / gw : n % gw;
        if (windowed) pos %= K;
        const int k = pos - r + K - 1;
        if (k >= 0 && k < K && r < 2 * K - 1) {
          if (windowed) {
            const float v = acc[e] * inv_scale;
            const half_t hi = (half_t)v;
            const half_t lo = (half_t)(v - (float)hi);
            half_t* o = relq + (bh * N + n) * 64 + tab * K + k;
            o[0] = hi;
            o[32] = lo;
          } else {
            (tab == 0 ? rel_h : rel_w)[(bh * N + n) * 64 + k] = acc[e];
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
  dim3 grid(N / 64, H, B), block(256);
  hipStream_t s = (hipStream_t)stream;
  const float inv = 1.0f / scale;
  const long long ts = head_major ? hd : 3LL * H * hd, hs = head_major ? (long long)B * N * hd : hd;
  if (hd == 64)
    hipLaunchKernelGGL(relpos_mfma_kernel<64>, grid, block, 0, s, (const half_t*)qkv, (const half_t*)Rpack, rel_h, rel_w,
                       (half_t*)relq, N, H, gw, K, RP, windowed, inv, ts, hs);
  else if (hd == 80)
    hipLaunchKernelGGL(relpos_mfma_kernel<80>, grid, block, 0, s, (const half_t*)qkv, (const half_t*)Rpack, rel_h, rel_w,
                       (half_t*)relq, N, H, gw, K, RP, windowed, inv, ts, hs);
  else
    return PSAM_ERR_ARG;
  return psam_launch_status();
}
