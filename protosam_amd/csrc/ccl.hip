// 8-connected component labelling + per-component statistics on the device.
//
// Replaces the host round trip of models/ProtoSAM.py:602-635: `cv2.connectedComponentsWithStats(pred, connectivity=8)`
// inside util/utils.py:474-494 `get_connected_components` (labels, area, centroids, per-label confidence
// sum(p_fg * [label == j]) / (sum(pred) + 1e-6)), `get_bbox_per_cc` (ProtoSAM.py:242-264: XYXY min/max per label) and
// `get_most_conf_points` with k = 1 (ProtoSAM.py:266-289: arg-max of p_fg inside the component).
//
// Algorithm: every pixel first points at the start of its horizontal run (ballot, no atomics); lock-free union-find
// (link larger root -> smaller with atomicMin, so a component's root is its first pixel in raster order) is then only
// needed at run boundaries; flatten, collect the roots, rank them (labels are numbered by raster order of the first
// pixel; cv2's numbering is an implementation detail and every downstream use is order-invariant, ProtoSAM.py:669),
// then one pass that accumulates the statistics with wave-level pre-aggregation (a 64-pixel row segment almost always
// holds a single label) before the atomics. Integer work throughout, except the confidence sum (fp64 atomics).
//
// Output table (fp64, one D2H): tab[0] = n components found, tab[1] = n kept (<= cap), tab[2] = sum(pred),
// tab[3] = index (0-based) of the most confident component, then per component k at tab[8 + 12*k ..]:
//   {area, sum_x, sum_y, min_x, min_y, max_x, max_y, conf, best_x, best_y, best_p, 0}
#include "common.h"

#define CC_STRIDE 12
#define CC_HDR 8

__device__ __forceinline__ int uf_find(const int* parent, int a) {
  int p = parent[a];
  while (p != a) {
    a = p;
    p = parent[a];
  }
  return a;
}

__device__ __forceinline__ void uf_union(int* parent, int a, int b) {
  for (;;) {
    a = uf_find(parent, a);
    b = uf_find(parent, b);
    if (a == b) return;
    if (a < b) {
      int t = a;
      a = b;
      b = t;
    }
    // a > b: hang a under b if a is still a root
    int old = atomicMin(&parent[a], b);
    if (old == a) return;
    a = old;  // somebody re-parented a concurrently; retry from its new parent
  }
}

// Batched form (psam_ccl_batch): blockIdx.z = image of the batch, every array moves by its per-image stride (elements); the
// single-image entry launches with grid.z = 1 and zero strides.
struct CclStride { long long pred, pfg, labels, parent, counters, roots, acc_i, acc_u, acc_d, fg_sum, tab; };

// init: every foreground pixel points at the start of its horizontal run inside its 64-pixel wave segment (found with
// one ballot, no atomics), so horizontal connectivity inside a segment costs nothing and find() paths stay short.
__global__ __launch_bounds__(256) void ccl_init_kernel(const uint8_t* __restrict__ pred, int H, int W,
                                                       int* __restrict__ parent, int* counters, CclStride st) {
  pred += blockIdx.z * st.pred; parent += blockIdx.z * st.parent; counters += blockIdx.z * st.counters;
  const int x = blockIdx.x * blockDim.x + threadIdx.x, y = blockIdx.y;
  if (x == 0 && y == 0) {
    counters[0] = 0;
    counters[1] = 0;
  }
  const int lane = threadIdx.x & 63;
  const bool fg = x < W && pred[(size_t)y * W + x] != 0;
  const unsigned long long mask = __ballot(fg);
  if (x >= W) return;
  int par = -1;
  if (fg) {
    const unsigned long long below = lane ? (~mask & ((1ull << lane) - 1ull)) : 0ull;  // background lanes left of me
    const int start = below ? 64 - __clzll((long long)below) : 0;
    par = y * W + (x - lane) + start;
  }
  parent[(size_t)y * W + x] = par;
}

// merge: unions only where connectivity is not already implied by a horizontal run (Playne/Hawick-style reduction):
//   W  : only for the first lane of a segment (runs crossing a 64-pixel boundary)
//   N  : unless W and NW are both foreground (then W already linked to NW, which is in N's run)
//   NW : only if N and W are background
//   NE : only if N is background
__global__ __launch_bounds__(256) void ccl_merge_kernel(const uint8_t* __restrict__ pred, int H, int W,
                                                        int* __restrict__ parent, CclStride st) {
  pred += blockIdx.z * st.pred; parent += blockIdx.z * st.parent;
  const int x = blockIdx.x * blockDim.x + threadIdx.x, y = blockIdx.y;
  if (x >= W) return;
  const int p = y * W + x;
  if (!pred[p]) return;
  const bool w = x > 0 && pred[p - 1];
  if (w && (threadIdx.x & 63) == 0) uf_union(parent, p, p - 1);
  if (y == 0) return;
  const int q = p - W;
  const bool n = pred[q] != 0;
  const bool nw = x > 0 && pred[q - 1];
  const bool ne = x + 1 < W && pred[q + 1];
  if (n) {
    if (!(w && nw)) uf_union(parent, p, q);
  } else {
    if (nw && !w) uf_union(parent, p, q - 1);
    if (ne) uf_union(parent, p, q + 1);
  }
}

// flatten: parent[i] := the root of i, and the roots are collected. One pass (round 5; rounds 1-4 ran a read-only find here and a second
// kernel that wrote the roots back): after the merge kernel the ROOTS are final, and a thread that meets a parent another thread has
// already flattened reads either the old ancestor or the root - both on the path to the same root.
__global__ void ccl_flatten_kernel(int n, int* __restrict__ parent, int* __restrict__ roots, int cap, int* counters, CclStride st) {
  parent += blockIdx.z * st.parent; roots += blockIdx.z * st.roots; counters += blockIdx.z * st.counters;
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const int p0 = parent[i];
  if (p0 < 0) return;
  const int r = uf_find(parent, i);
  if (r == i) {
    int k = atomicAdd(&counters[0], 1);
    if (k < cap) roots[k] = i;
  } else if (r != p0) {
    parent[i] = r;
  }
}

// one block: rank the (<= cap) collected roots ascending, label them 1..n, clear the accumulators
__global__ __launch_bounds__(256) void ccl_rank_kernel(int* __restrict__ roots, int cap, const int* counters,
                                                       int* __restrict__ labels, int* __restrict__ acc_i,
                                                       unsigned long long* __restrict__ acc_u, double* __restrict__ acc_d, CclStride st) {
  roots += blockIdx.z * st.roots; counters += blockIdx.z * st.counters; labels += blockIdx.z * st.labels;
  acc_i += blockIdx.z * st.acc_i; acc_u += blockIdx.z * st.acc_u; acc_d += blockIdx.z * st.acc_d;
  extern __shared__ int sroots[];
  const int nall = counters[0];
  const int n = nall < cap ? nall : cap;
  for (int i = threadIdx.x; i < n; i += blockDim.x) sroots[i] = roots[i];
  __syncthreads();
  for (int i = threadIdx.x; i < n; i += blockDim.x) {
    const int r = sroots[i];
    int rank = 0;
    for (int j = 0; j < n; ++j) rank += (sroots[j] < r);
    labels[r] = rank + 1;
    roots[rank] = r;  // sorted in place is unsafe in general; ranks are a permutation and sroots holds the originals
  }
  for (int k = threadIdx.x; k < cap; k += blockDim.x) {
    acc_i[k * 5 + 0] = 0;            // area
    acc_i[k * 5 + 1] = 0x7fffffff;   // min_x
    acc_i[k * 5 + 2] = 0x7fffffff;   // min_y
    acc_i[k * 5 + 3] = -1;           // max_x
    acc_i[k * 5 + 4] = -1;           // max_y
    acc_u[k * 3 + 0] = 0ull;         // sum_x
    acc_u[k * 3 + 1] = 0ull;         // sum_y
    acc_u[k * 3 + 2] = 0ull;         // best (ordered value << 32 | ~index)
    acc_d[k] = 0.0;                  // conf sum
  }
}

__device__ __forceinline__ uint32_t f2ord32(float f) {
  uint32_t u = __float_as_uint(f);
  return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__device__ __forceinline__ float ord2f32(uint32_t u) {
  return __uint_as_float((u & 0x80000000u) ? (u & 0x7fffffffu) : ~u);
}

// One wave walks a 64-pixel-wide strip of STAT_ROWS rows and keeps ONE pending accumulator (label, count, sums, box,
// arg-max key): a row segment's labels are pre-aggregated across the wave as before, and a group that carries the pending
// label is merged into it instead of being sent to memory. For blob-like masks almost every strip sees a single label,
// so the 9 atomics per (wave, row, label) - 16 384 waves hammering the same handful of addresses, 125 us per slice - become
// 9 per (wave, strip).
#define STAT_ROWS 16
__global__ __launch_bounds__(256) void ccl_stats_kernel(const int* __restrict__ parent, int H, int W,
                                                        const float* __restrict__ pfg, int* __restrict__ labels,
                                                        int* __restrict__ acc_i, unsigned long long* __restrict__ acc_u,
                                                        double* __restrict__ acc_d, CclStride st) {
  parent += blockIdx.z * st.parent; pfg += blockIdx.z * st.pfg; labels += blockIdx.z * st.labels;
  acc_i += blockIdx.z * st.acc_i; acc_u += blockIdx.z * st.acc_u; acc_d += blockIdx.z * st.acc_d;
  const int x = blockIdx.x * blockDim.x + threadIdx.x;
  const int lane = threadIdx.x & 63;
  const int y0 = blockIdx.y * STAT_ROWS, y1 = min(y0 + STAT_ROWS, H);
  // pending accumulator (wave-uniform, valid when pl != 0)
  int pl = 0, pcnt = 0, pmnx = 0, pmny = 0, pmxx = 0, pmxy = 0;
  unsigned long long psx = 0, psy = 0, pkey = 0;
  double pps = 0.0;
  auto flush = [&]() {
    if (pl != 0 && lane == 0) {
      const int k = pl - 1;
      atomicAdd(&acc_i[k * 5 + 0], pcnt);
      atomicMin(&acc_i[k * 5 + 1], pmnx);
      atomicMin(&acc_i[k * 5 + 2], pmny);
      atomicMax(&acc_i[k * 5 + 3], pmxx);
      atomicMax(&acc_i[k * 5 + 4], pmxy);
      atomicAdd(&acc_u[k * 3 + 0], psx);
      atomicAdd(&acc_u[k * 3 + 1], psy);
      atomicMax(&acc_u[k * 3 + 2], pkey);
      atomicAdd(&acc_d[k], pps);
    }
    pl = 0;
  };
  for (int y = y0; y < y1; ++y) {
    int lab = 0;
    float pv = 0.f;
    int p = 0;
    if (x < W) {
      p = y * W + x;
      const int r = parent[p];
      if (r >= 0) {
        lab = (r == p) ? labels[p] : labels[r];  // roots were labelled by ccl_rank_kernel (0 if beyond capacity)
        pv = pfg[p];
      }
      if (r != p) labels[p] = lab;               // never rewrites a root's label (read by other threads)
    }
    unsigned long long todo = __ballot(lab != 0);
    while (todo) {
      const int leader = __ffsll((long long)todo) - 1;
      const int L = __shfl(lab, leader, 64);
      const bool in = (lab == L);
      const unsigned long long mask = __ballot(in);
      todo &= ~mask;
      const int cnt = __popcll(mask);
      const float ps = wave_sum(in ? pv : 0.f);
      int sx = in ? x : 0;
#pragma unroll
      for (int o = 32; o > 0; o >>= 1) sx += __shfl_xor(sx, o, 64);
      unsigned long long key = in ? (((unsigned long long)f2ord32(pv) << 32) | (0xffffffffu - (uint32_t)p)) : 0ull;
#pragma unroll
      for (int o = 32; o > 0; o >>= 1) {
        unsigned long long other = __shfl_xor(key, o, 64);
        key = other > key ? other : key;
      }
      const int x0 = x - lane;
      const int mnx = x0 + (__ffsll((long long)mask) - 1), mxx = x0 + 63 - __clzll((long long)mask);
      if (pl != L) {   // wave-uniform
        flush();
        pl = L; pcnt = 0; pmnx = mnx; pmny = y; pmxx = mxx; pmxy = y; psx = 0; psy = 0; pkey = 0; pps = 0.0;
      }
      pcnt += cnt;
      pmnx = min(pmnx, mnx); pmxx = max(pmxx, mxx); pmny = min(pmny, y); pmxy = max(pmxy, y);
      psx += (unsigned long long)sx;
      psy += (unsigned long long)y * (unsigned long long)cnt;
      pkey = key > pkey ? key : pkey;
      pps += (double)ps;
    }
  }
  flush();
}

__global__ void ccl_finalize_kernel(const int* counters, int cap, int W, const int* __restrict__ acc_i,
                                    const unsigned long long* __restrict__ acc_u, const double* __restrict__ acc_d,
                                    const int* __restrict__ fg_sum, double* __restrict__ tab, CclStride st) {
  counters += blockIdx.z * st.counters; acc_i += blockIdx.z * st.acc_i; acc_u += blockIdx.z * st.acc_u; acc_d += blockIdx.z * st.acc_d;
  tab += blockIdx.z * st.tab;
  if (fg_sum) fg_sum += blockIdx.z * st.fg_sum;
  const int nall = counters[0];
  const int n = nall < cap ? nall : cap;
  __shared__ double total_s;
  if (threadIdx.x == 0) {
    long long tot = 0;
    if (fg_sum) {
      tot = fg_sum[0];
    } else {
      for (int k = 0; k < n; ++k) tot += acc_i[k * 5];
    }
    total_s = (double)tot;
    tab[0] = (double)nall;
    tab[1] = (double)n;
    tab[2] = (double)tot;
  }
  __syncthreads();
  // util/utils.py:490: conf = sum(p * [label == j]) / (sum(pred) + 1e-6), evaluated in fp32 by numpy
  const float den = (float)total_s + 1e-6f;
  for (int k = threadIdx.x; k < n; k += blockDim.x) {
    double* t = tab + CC_HDR + (size_t)k * CC_STRIDE;
    const unsigned long long key = acc_u[k * 3 + 2];
    const int bidx = (int)(0xffffffffu - (uint32_t)(key & 0xffffffffull));
    t[0] = (double)acc_i[k * 5 + 0];
    t[1] = (double)acc_u[k * 3 + 0];
    t[2] = (double)acc_u[k * 3 + 1];
    t[3] = (double)acc_i[k * 5 + 1];
    t[4] = (double)acc_i[k * 5 + 2];
    t[5] = (double)acc_i[k * 5 + 3];
    t[6] = (double)acc_i[k * 5 + 4];
    t[7] = (double)((float)acc_d[k] / den);
    t[8] = (double)(bidx % W);
    t[9] = (double)(bidx / W);
    t[10] = (double)ord2f32((uint32_t)(key >> 32));
    t[11] = 0.0;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    // util/utils.py:510-515: first component with strictly larger confidence wins
    int best = 0;
    double bc = -1.0;
    for (int k = 0; k < n; ++k) {
      double c = tab[CC_HDR + (size_t)k * CC_STRIDE + 7];
      if (c > bc) {
        bc = c;
        best = k;
      }
    }
    tab[3] = (double)best;
  }
}

// pred u8 [H,W]; pfg fp32 [H,W]; labels int32 [H,W] (out); parent int32 [H*W] scratch; scratch: int32 area of
// 2 + cap + 5*cap ints, then 8-byte aligned 3*cap u64 + cap doubles (see protosam_amd/ops.py: CclWorkspace).
static int ccl_launch(const void* pred, const float* pfg, int H, int W, int cap, int* labels, int* parent, int* counters, int* roots,
                      int* acc_i, void* acc_u, double* acc_d, const int* fg_sum, double* tab, int B, const CclStride& st, hipStream_t s) {
  const int n = H * W;
  const uint8_t* pr = (const uint8_t*)pred;
  if (B == 1 || st.labels == (long long)n) {
    (void)hipMemsetAsync(labels, 0, (size_t)n * B * sizeof(int), s);
  } else {
    for (int b = 0; b < B; ++b) (void)hipMemsetAsync(labels + (size_t)b * st.labels, 0, (size_t)n * sizeof(int), s);
  }
  const unsigned Z = (unsigned)B;
  hipLaunchKernelGGL(ccl_init_kernel, dim3((W + 255) / 256, H, Z), dim3(256), 0, s, pr, H, W, parent, counters, st);
  hipLaunchKernelGGL(ccl_merge_kernel, dim3((W + 255) / 256, H, Z), dim3(256), 0, s, pr, H, W, parent, st);
  hipLaunchKernelGGL(ccl_flatten_kernel, dim3((n + 255) / 256, 1, Z), dim3(256), 0, s, n, parent, roots, cap, counters, st);
  hipLaunchKernelGGL(ccl_rank_kernel, dim3(1, 1, Z), dim3(256), cap * sizeof(int), s, roots, cap, counters, labels, acc_i,
                     (unsigned long long*)acc_u, acc_d, st);
  hipLaunchKernelGGL(ccl_stats_kernel, dim3((W + 255) / 256, (H + STAT_ROWS - 1) / STAT_ROWS, Z), dim3(256), 0, s, parent, H, W, pfg, labels,
                     acc_i, (unsigned long long*)acc_u, acc_d, st);
  hipLaunchKernelGGL(ccl_finalize_kernel, dim3(1, 1, Z), dim3(256), 0, s, counters, cap, W, acc_i,
                     (const unsigned long long*)acc_u, acc_d, fg_sum, tab, st);
  return psam_launch_status();
}

extern "C" int psam_ccl(const void* pred, const float* pfg, int H, int W, int cap, int* labels, int* parent, int* counters,
                        int* roots, int* acc_i, void* acc_u, double* acc_d, const int* fg_sum, double* tab, void* stream) {
  if (H <= 0 || W <= 0 || cap <= 0 || cap > 4096) return PSAM_ERR_ARG;
  CclStride st = {};
  return ccl_launch(pred, pfg, H, W, cap, labels, parent, counters, roots, acc_i, acc_u, acc_d, fg_sum, tab, 1, st, (hipStream_t)stream);
}

// B images in one launch chain (six launches for the whole batch instead of six per image; validation_protosam.py walks slices,
// ProtoSAM.forward_batch batches them): pred u8 [B,H,W] contiguous, pfg fp32 with `pfg_stride` floats between images (channel 1 of a
// [B,2,H,W] probability map: 2*H*W), every scratch array B times the single-image size, contiguous per image; fg_sum int32 [B] or
// NULL; tab fp64 [B][8 + 12*cap].
extern "C" int psam_ccl_batch(const void* pred, const float* pfg, long long pfg_stride, int B, int H, int W, int cap, int* labels,
                              int* parent, int* counters, int* roots, int* acc_i, void* acc_u, double* acc_d, const int* fg_sum,
                              double* tab, void* stream) {
  if (H <= 0 || W <= 0 || cap <= 0 || cap > 4096 || B <= 0 || B > 65535) return PSAM_ERR_ARG;
  const long long n = (long long)H * W;
  CclStride st = {n, pfg_stride, n, n, 2, cap, 5LL * cap, 3LL * cap, cap, 1, CC_HDR + (long long)CC_STRIDE * cap};
  return ccl_launch(pred, pfg, H, W, cap, labels, parent, counters, roots, acc_i, acc_u, acc_d, fg_sum, tab, B, st, (hipStream_t)stream);
}

// ---- negative point prompts (models/ProtoSAM.py:361-372 global, :395-419 per component) ---------------------------------
// keys[0]     = most confident background pixel among those with p_bg >= thr (the "global" negative point)
// keys[1 + k] = most confident background pixel of the ring {dilate_r(component k) minus component k}, where dilate_r is
//               r iterations of a 3x3 dilation = a (2r+1) x (2r+1) box (cv2.dilate(mask, ones(3,3), iterations=10), r = 10)
// key = (float bits of p_bg) << 32 | (0xFFFFFFFF - pixel index): the maximum key is the largest p_bg, first pixel in raster
// order on ties (torch.topk leaves tie order unspecified; same convention as the positive points); 0 = no such pixel.
// One workgroup = one 32x32 tile of one component (blockIdx.z - 1) or of the global search (blockIdx.z == 0); tiles outside
// the component's bounding box grown by r leave at once. Box dilation is separable: rows in LDS, then columns.
#define NP_T 32
#define NP_RMAX 10
__global__ __launch_bounds__(256) void neg_points_kernel(const int* __restrict__ labels, const float* __restrict__ pbg,
                                                         const double* __restrict__ tab, int H, int W, int r, float thr,
                                                         unsigned long long* __restrict__ keys) {
  constexpr int TW = NP_T + 2 * NP_RMAX;
  __shared__ unsigned char m[TW][TW];
  __shared__ unsigned char hd[TW][NP_T];
  __shared__ unsigned long long red[4];
  const int t = threadIdx.x;
  const int x0 = blockIdx.x * NP_T, y0 = blockIdx.y * NP_T;
  const int z = blockIdx.z;
  unsigned long long best = 0ull;
  if (z == 0) {
    for (int i = t; i < NP_T * NP_T; i += 256) {
      const int y = y0 + i / NP_T, x = x0 + i % NP_T;
      if (y < H && x < W) {
        const float v = pbg[(size_t)y * W + x];
        if (v >= thr) {
          const unsigned long long k = ((unsigned long long)__float_as_uint(v) << 32) | (0xFFFFFFFFu - (unsigned)(y * W + x));
          best = k > best ? k : best;
        }
      }
    }
  } else {
    const int k = z - 1;
    if (k >= (int)tab[1]) return;
    const double* row = tab + CC_HDR + (size_t)CC_STRIDE * k;
    const int bx0 = (int)row[3] - r, by0 = (int)row[4] - r, bx1 = (int)row[5] + r, by1 = (int)row[6] + r;
    if (x0 > bx1 || x0 + NP_T - 1 < bx0 || y0 > by1 || y0 + NP_T - 1 < by0) return;
    const int c = k + 1, span = NP_T + 2 * r;
    for (int i = t; i < span * span; i += 256) {
      const int ly = i / span, lx = i % span;
      const int y = y0 - r + ly, x = x0 - r + lx;
      m[ly][lx] = (y >= 0 && y < H && x >= 0 && x < W && labels[(size_t)y * W + x] == c) ? 1 : 0;
    }
    __syncthreads();
    for (int i = t; i < span * NP_T; i += 256) {
      const int ly = i / NP_T, cx = i % NP_T;
      unsigned char any = 0;
      for (int d = 0; d <= 2 * r; ++d) any |= m[ly][cx + d];
      hd[ly][cx] = any;
    }
    __syncthreads();
    for (int i = t; i < NP_T * NP_T; i += 256) {
      const int ty = i / NP_T, tx = i % NP_T;
      const int y = y0 + ty, x = x0 + tx;
      if (y >= H || x >= W || m[ty + r][tx + r]) continue;
      unsigned char any = 0;
      for (int d = 0; d <= 2 * r; ++d) any |= hd[ty + d][tx];
      if (any) {
        const float v = pbg[(size_t)y * W + x];
        const unsigned long long kk = ((unsigned long long)__float_as_uint(v) << 32) | (0xFFFFFFFFu - (unsigned)(y * W + x));
        best = kk > best ? kk : best;
      }
    }
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    const unsigned long long other = __shfl_xor(best, o, 64);
    best = other > best ? other : best;
  }
  if ((t & 63) == 0) red[t >> 6] = best;
  __syncthreads();
  if (t == 0) {
    for (int i = 1; i < 4; ++i) best = red[i] > best ? red[i] : best;
    if (best) atomicMax(keys + z, best);
  }
}
extern "C" int psam_neg_points(const int* labels, const float* pbg, const double* tab, int H, int W, int max_comp, int r,
                               float thr, unsigned long long* keys, void* stream) {
  if (H <= 0 || W <= 0 || max_comp < 0 || r < 0 || r > NP_RMAX) return PSAM_ERR_ARG;
  (void)hipMemsetAsync(keys, 0, sizeof(unsigned long long) * (max_comp + 1), (hipStream_t)stream);
  hipLaunchKernelGGL(neg_points_kernel, dim3((W + NP_T - 1) / NP_T, (H + NP_T - 1) / NP_T, max_comp + 1), dim3(256), 0,
                     (hipStream_t)stream, labels, pbg, tab, H, W, r, thr, keys);
  return psam_launch_status();
}
