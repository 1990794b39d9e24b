/* protosam_hip.h -- C ABI of libprotosam_hip.so (hand-written gfx950 / CDNA4 kernels for ProtoSAM's hot path).
 *
 * The reference (levayz/ProtoSAM) is pure Python on PyTorch and has no FFI layer: its drop-in boundary is the Python
 * class API (`ProtoSAM.forward`, `FewShotSeg`, `MultiProtoAsConv`, `sam_model_registry`, `SamPredictor`, ...; see
 * SURVEY.md 8b), mirrored by the `protosam_amd` package. This header is the native boundary underneath it: each entry
 * point replaces one chain of stock torch ops of the reference and cites it (paths relative to the reference repo).
 *
 * Conventions: plain pointers and sizes only (no torch types). All pointers are DEVICE pointers unless marked
 * "host". No ownership transfer: the caller allocates every buffer. `stream` is a hipStream_t (0 = default stream);
 * launches are asynchronous on it. Every function returns 0 on success, 1 = bad argument, 2 = launch error.
 * "half" = IEEE fp16 (`_Float16`). Matrices are row-major with the stated leading dimension in ELEMENTS.
 */
#ifndef PROTOSAM_HIP_H
#define PROTOSAM_HIP_H
#include <stddef.h>
#ifdef __cplusplus
extern "C" {
#endif

/* ---- contraction engine --------------------------------------------------------------------------------------
 * out[M,N] = epi(A[M,K] . W[N,K]^T + bias[N]); A, W half (K contiguous), fp32 accumulate on MFMA.
 * epilogue 0: half out;  1: half out = gelu_erf(.);  2: fp32 out = (resid ? resid[r,:] : 0) + (gamma ? gamma : 1)*(.)
 *   3: half out = relu(. + resid16[m,:]) with `resid` pointing at a HALF [*, ldr] map or null (conv + folded BatchNorm
 *   (+ identity) + ReLU of a torchvision ResNet Bottleneck; models/backbone/torchvision_backbones.py:19-23)
 *   resid row r = resid_mod ? m % resid_mod : m.  out row = out_seg ? (m/out_seg)*out_seg_stride + out_seg_off + m%out_seg : m.
 * N % 128 == 0, K % 64 == 0. Replaces every nn.Linear / 1x1 conv / patch-embed conv / ConvTranspose2d(2,2) GEMM:
 *   models/segment_anything/modeling/image_encoder.py:223-249 (qkv, proj), modeling/common.py:13-26 (MLPBlock),
 *   image_encoder.py:375-406 (PatchEmbed), :90-106 (neck), modeling/transformer.py:218-240 (image-side projections),
 *   modeling/mask_decoder.py:53-55 (first ConvTranspose2d); DINOv2 Attention/Mlp/PatchEmbed (hub model, call site
 *   models/grid_proto_fewshot.py:88-91). */
int psam_gemm_f16(const void* A, const void* W, const float* bias, void* out, const float* resid, const float* gamma,
                  int M, int N, int K, int lda, int ldw, int ldo, int ldr, int resid_mod, int out_seg,
                  int out_seg_stride, int out_seg_off, int epilogue, void* stream);
/* The packed qkv projection with a head-major result: out half [N / hd planes][M][hd] (plane = which*H + h of
 * `self.qkv(x).reshape(B, N, 3, H, hd)`, image_encoder.py:236-238), read by psam_attention_f16 / psam_relpos with
 * head_major = 1: a head's Q / K / V rows are then contiguous hd-vectors. Same arithmetic as epilogue 0. */
int psam_gemm_f16_heads(const void* A, const void* W, const float* bias, void* out, int M, int N, int K, int lda, int ldw,
                        int hd, void* stream);
/* psam_gemm_f16 with a LayerNorm folded into the GEMMs either side of it (no LayerNorm pass, no cast pass):
 *   epilogue 2 (x = resid + gamma*(a w^T + bias)) also writes out16 = half(x) [.., ld16] and stats: per row and 64-column
 *              group (sum, sum of squares) of x, float [M][N/64][2] (either may be null);
 *   epilogue 0 / 1 take A = half(x), W already multiplied by the LayerNorm weight, bias' = bias + W . ln_bias, and
 *              ln_mr = the buffer psam_ln_finalize fills: float [M][2] = (mean, rstd), then half [M][8] = {hi, hi, lo, 0 x 5} of
 *              -mean; ln_s = float [N] row sums of the fp16 W', then half [N][8] = {hi, lo, hi, 0 x 5} of the same sums (the MFMA
 *              operands of the assembly kernels' rank-1 correction acc -= mean (x) ln_s; ops.fold_layernorm builds ln_s):
 *              out = act(rstd * (acc - mean * ln_s[n]) + bias'[n]).
 * modeling/image_encoder.py:174-193 (norm1 -> attn.qkv, norm2 -> mlp.lin1); DINOv2 Block (norm1 / norm2). */
int psam_gemm_f16_ln(const void* A, const void* W, const float* bias, void* out, const float* resid, const float* gamma,
                     int M, int N, int K, int lda, int ldw, int ldo, int ldr, int resid_mod, int out_seg,
                     int out_seg_stride, int out_seg_off, int epilogue, void* out16, int ld16, float* stats,
                     const float* ln_mr, const float* ln_s, void* stream);
/* stats [M][D/64][2] -> mr: float [M][2] = (mean, 1/sqrt(var + eps)), biased variance (nn.LayerNorm), followed by half [M][8] =
 * {hi, hi, lo, 0, 0, 0, 0, 0} of -mean: 6 floats per row in all */
int psam_ln_finalize(const float* stats, int M, int D, float eps, float* mr, void* stream);


/* One slice through a residual Linear with FEW 256 x 256 tiles and a LONG K (SAM ViT-H mlp.lin2 of one image: 4096 x 1280 x 5120, 80 tiles
 * for 256 CUs), and the LayerNorm that follows it in the block stack, as one split-K launch of the assembly tile + one reduce pass:
 *   x[M,N] (fp32, in place) += A[M,K] . W[N,K]^T + bias;  out16 (optional half [M, ld16]) = LayerNorm(x; ln_w, ln_b, eps), or half(x) when
 *   ln_w is null.
 * `ks` K ranges per tile (ks * tiles workgroups side by side); ws: CALLER-OWNED fp32 scratch of >= ks * Mp * N elements, Mp = M rounded up
 * to a multiple of 256 (the library keeps no
 * state for this path: it may be captured into a graph); the ranges are summed in a fixed order (deterministic).
 * psam_gemm_splitk_ranges returns the ks this device would use for the shape (0: the shape does not pay, or the assembly kernel is not
 * loaded) - a count, not a status; psam_gemm_f16_splitk_ln takes exactly that ks (>= 2), N % 256 == 0, N <= 2048, K % 64 == 0.
 * modeling/common.py:13-26 (MLPBlock.lin2) + image_encoder.py:174-193 (x = x + mlp(...), then the next block's norm1). */
int psam_gemm_splitk_ranges(int M, int N, int K);
int psam_gemm_f16_splitk_ln(const void* A, const void* W, const float* bias, float* x, int M, int N, int K, int lda, int ldw, int ldx,
                            int ks, float* ws, const float* ln_w, const float* ln_b, float eps, void* out16, int ld16, void* stream);

/* Tile override for psam_gemm_f16: 0 auto (default; also env PSAM_GEMM_TILE), 1 = 128x128x64 (HIP), 11 = 256x256x64 persistent
 * 8-wave kernel (HIP), 15 = assembly kernels (csrc/gemm_asm_gen.py, the default large tile), 16 = half-tile ping-pong assembly
 * kernels (csrc/gemm_asm2_gen.py, experimental); a tile that cannot take the call's layout falls back (16 -> 15 -> 11 -> 1). */
int psam_gemm_set_tile(int tile);
/* Schedule variant of the assembly GEMM (tile 15): 0 = the shipped kernels; n > 0 selects the numbered experiment kernels of a
 * library built with `make GENFLAGS=--experiments` (csrc/gemm_asm_gen.py; a missing variant makes the next GEMM return an error). */
int psam_gemm_asm_variant(int variant);
/* Dispatch switches of psam_gemm_f16 (1 = on, the default; initial values also from the environment): "asm" (PSAM_GEMM_ASM) the
 * assembly kernels for the large tiles, "half_tiles" (PSAM_GEMM_HALF) the half-tile assembly kernels for shapes with few 256x256
 * tiles, "splitk" (PSAM_GEMM_SPLITK), "nsplit" (PSAM_GEMM_NSPLIT) the column split of one-slice fc1 shapes; "max_wgs"
 * (PSAM_GEMM_MAX_WGS, an integer, 0 = none) caps the persistent grids of the assembly kernels - two streams then run their GEMMs
 * side by side on disjoint CUs. Unknown name: error. */
int psam_gemm_set_option(const char* name, int value);
/* Device scratch for the split-K form of psam_gemm_f16 (fp32 partial sums, [ksplit][M][N]): used only when registered and
 * large enough; EPI 2 on few 256x256 tiles with K >= 2048 then runs `ksplit` workgroups per tile + one reduce pass
 * (deterministic). 16-byte aligned; the caller keeps it alive and must not share it between concurrently running streams.
 * ptr = NULL unregisters (no split-K). Env PSAM_GEMM_SPLITK=0 disables the path. */
int psam_gemm_set_workspace(void* ptr, size_t bytes);

/* Row LayerNorm, fp32 in; out_dtype 0: half out (+ optional fp32 copy y2), 1: fp32 out. Appends `zero_tail_rows`
 * all-zero rows. torch.nn.LayerNorm of image_encoder.py:174-193, transformer.py:133-144 and LayerNorm2d
 * (modeling/common.py:31-43) on token-major rows; DINOv2 norm1/norm2/norm. */
int psam_layernorm(const float* x, const float* w, const float* b, void* y, float* y2, int M, int D, int ldx, int ldy,
                   float eps, int out_dtype, int zero_tail_rows, void* stream);

/* Fused multi-head attention on the packed projection qkv half [B,N,3,H,hd] -> out half [B,N,H*hd]; hd in {64, 80}.
 * mode 0 global (DINOv2 Attention); mode 1 global + decomposed rel-pos (rel_h/rel_w fp32, gw == 64); mode 2 ws x ws
 * windows with the reference's zero padding (pad_row half [3,H,hd] = qkv bias) + rel-pos folded into the score MFMA
 * as 32 extra k-slots (relq half [B,H,N,2,32] from psam_relpos, or relq = null and rpack = psam_relpos' windowed table
 * pack: the query-side terms are then computed inside the kernel and never touch HBM).
 * mode 1 with rel_h = rel_w = null and rpack = psam_relpos' GLOBAL table pack: the decomposed rel-pos terms are computed inside the
 * assembly kernel (psam_gattn_asm_{80,64}_fused) for the shapes psam_attention_fused_relpos reports; PSAM_ERR_ARG elsewhere.
 * mode 0 with hd = 64 and N >= 128 (DINOv2 at 1297 / 5330 tokens) runs the assembly kernel psam_gattn_asm_64_norel: any token count,
 * the last key tile masked; models/grid_proto_fewshot.py:88-98 (the hub model's Attention.forward).
 * head_major = 0: qkv is token-major [B,N,3,H,hd]; 1: head-major [3,H,B*N,hd] as written by psam_gemm_f16_heads.
 * image_encoder.py:235-251 (Attention.forward), :254-300 (window_partition / unpartition), :337-372. */
int psam_attention_f16(const void* qkv, void* out, const float* rel_h, const float* rel_w, const void* relq,
                       const void* rpack, const void* pad_row, int B, int N, int H, int hd, float scale, int mode, int gh,
                       int gw, int ws, int head_major, void* stream);
/* kernel selection of psam_attention_f16 (A/B, tests; default 5). bit 0: V2 softmax of the HIP global kernels (0 = the serial round-1
 * form); bits 1-2: window kernel (0 attn_kernel, 1 wattn_kernel, 2 the assembly kernel of csrc/wattn_asm_gen.py where it applies -
 * rpack given, hd = 80, 14 x 14 windows of a 64 x 64 token map, token-major qkv - and the persistent wattn_p_kernel elsewhere, 3
 * wattn_p_kernel everywhere); bit 3: the register-staged HIP
 * global kernel; bit 4: the DMA-fed HIP global kernel everywhere; neither bit 3 nor 4: the assembly global kernel
 * (csrc/gattn_asm_gen.py) where it applies - rel-pos: hd = 80 or 64, N a multiple of 256; no bias: hd = 64, N >= 128; any head count
 * and batch - and the DMA-fed HIP kernel elsewhere. */
int psam_attention_set_variant(int v);
/* 1 when psam_attention_f16(mode 1) computes the rel-pos terms itself from `rpack` for this shape (64 x 64 token map, hd 80 / 64,
 * token-major qkv, the assembly kernels selected), else 0: the caller then runs psam_relpos first.
 * image_encoder.py:325-372 (add_decomposed_rel_pos) inside :235-251. Returns 0 / 1, not a status. */
int psam_attention_fused_relpos(int B, int N, int H, int hd, int gh, int gw);

/* rel_h[b,h,n,k] = q . Rh[qy - k + K-1], rel_w likewise (UNSCALED q), as an MFMA GEMM against the whole table followed by
 * a scatter. Rpack half [2 (h,w)][2 (hi,lo)][RP][HDP] (RP = 128 global / 32 windowed, zero padded).
 * global: rel_h, rel_w fp32 [B,H,N,64]. windowed: relq half [B,H,N,2,32] = hi/lo of (rel_h | rel_w | 0) / scale.
 * image_encoder.py:303-372 (get_rel_pos, add_decomposed_rel_pos). */
int psam_relpos(const void* qkv, const void* Rpack, float* rel_h, float* rel_w, void* relq, int B, int N, int H, int hd,
                int gw, int K, int windowed, float scale, int head_major, void* stream);

/* ---- ALP module ------------------------------------------------------------------------------------------------
 * Prototype bank from token-major support features sup fp32 [h*w, C] (row stride ld) and the foreground mask fp32
 * [MH,MW] (bmask optional, default 1 - mask): nearest resize, avg-pool coverage > thresh, pooled prototypes,
 * global masked-average prototype, safe_norm. bank fp32 [2*cap, C]; meta int[8] = {n_bg, n_fg, fg_mode, n_fg_cells}.
 * force_mode -1: reference FewShotSeg rule; 0 'mask'; 1 'gridconv+'; 2 'gridconv'.
 * models/alpmodule.py:97-159 (get_prototypes), :14-18 (safe_norm); models/grid_proto_fewshot.py:228-231,253-256. */
int psam_alp_bank(const float* sup, int ld, int h, int w, int C, const float* mask, const float* bmask, int MH, int MW,
                  int pool_w, int kernel_size, float thresh, float eps, float* bank, int cap, int* meta, int* slot_bg,
                  int* slot_fg, float* mres, int force_mode, void* stream);

/* pred[b, {bg,fg}, pix] = sum_p softmax_p(d) * d, d = sim_scale * cos(qry[pix], proto_p), fp32 MFMA.
 * models/alpmodule.py:57-94 (get_prediction_from_prototypes), :195. which_only -1: both banks. */
int psam_alp_sim(const float* qry, long long q_bstride, int ld, int B, int npix, int C, const float* bank, int cap,
                 const int* meta, float eps, float sim_scale, float* part, float* pred, int which_only, void* stream);

/* ---- resampling / packing -------------------------------------------------------------------------------------- */
/* F.interpolate(img,(S,S),'bilinear') + im2col of a PxP/stride-P conv -> half [B*(S/P)^2, Kpad].
 * models/grid_proto_fewshot.py:88-89 + DINOv2 PatchEmbed; also SAM PatchEmbed when H == S. */
int psam_patchify_bilinear(const float* img, int B, int C, int H, int W, int S, int P, int Kpad, void* out, void* stream);
/* F.interpolate(x, (OH,OW), 'bilinear', align_corners=False), fp32 planes. grid_proto_fewshot.py:272-273; ProtoSAM.py:592-594 */
int psam_bilinear_nchw(const float* in, int planes, int IH, int IW, int OH, int OW, float* out, void* stream);
/* The same resize in the conventions of the vendored SAM copies' postprocess_masks: mode 0 as above, 1 = bilinear
 * align_corners=True (SamBatched, modeling/sam.py:313-320), 2 = nearest (vendored Sam, modeling/sam.py:154-160). */
int psam_resize2d(const float* in, int planes, int IH, int IW, int OH, int OW, int mode, float* out, void* stream);
/* token-major feature-map resize fp32 [B][ih*iw,C] -> [B][oh*ow,C] (the 32x32 upsample of grid_proto_fewshot.py:96-98) */
int psam_bilinear_tokens(const float* in, long long in_bstride, int ld, int B, int ih, int iw, int C, int oh, int ow,
                         float* out, void* stream);
/* [bilinear to OHxOW] -> softmax(dim=1) -> argmax for 2-class logits; prob fp32 [B,2,OH,OW], pred u8, fg_sum int[B].
 * models/ProtoSAM.py:592-602 */
int psam_prob_argmax(const float* logits, int B, int IH, int IW, int OH, int OW, float* prob, void* pred, int* fg_sum,
                     void* stream);
int psam_broadcast_rows(const float* row, int D, float* out, int B, long long stride, long long off, void* stream);
/* per-image min/max (order-preserving uint32 pairs).  models/ProtoSAM.py:660 */
int psam_minmax(const float* x, int B, long long n_per_img, void* mm, void* stream);
/* ((x-min)/(max-min)*255).astype(uint8) -> (u8 - mean)/std -> im2col(16x16) half; mean3/std3 are HOST pointers.
 * models/ProtoSAM.py:651-660; modeling/sam.py:163-173; predictor.py:56-58,88. quantise 0 = ProtoMedSAM.py:203-205. */
int psam_sam_patchify(const float* img, const void* mm, int B, int S, int P, const float* mean3, const float* std3,
                      int quantise, void* out, void* u8out, void* stream);
/* (x - mean[c]) / std[c] on [B,3,plane]; in_u8: x is uint8. mean3/std3 HOST pointers. modeling/sam.py:163-168 */
int psam_normalize_chw(const void* x, int in_u8, int B, long long plane, const float* mean3, const float* std3,
                       float* y, void* stream);
/* im2col of the neck's 3x3/pad-1 conv on a token-major half map. image_encoder.py:98-104 */
int psam_im2col3x3(const void* in, int B, int H, int W, int C, void* out, void* stream);
int psam_cast_f16(const float* x, void* y, long long n, void* stream);
/* The element-wise passes of the reference-width mode of the SAM image encoder (every Linear of a block through psam_gemm_f32x3, fp32 in
 * and out): half -> fp32 (the attention output on its way to attn.proj, modeling/image_encoder.py:249; n % 8 == 0) and nn.GELU in its erf
 * form, in place on fp32 (MLPBlock, modeling/common.py:25-26; n % 4 == 0). */
int psam_cast_f32(const void* x, float* y, long long n, void* stream);
int psam_gelu_f32(float* x, long long n, void* stream);
/* fp32 -> fp16 pair hi = half(x), lo = half(x - hi) (write_hi = 0: hi is read, e.g. the copy a folded-LayerNorm GEMM wrote). Operands of
 * the split-fp16 GEMMs (hi W_hi + lo W_hi + hi W_lo) of the image encoder's neck: modeling/image_encoder.py:90-106. n % 8 == 0. */
int psam_split_f16(const float* x, void* hi, void* lo, long long n, int write_hi, void* stream);

/* ---- connected components + per-component statistics ------------------------------------------------------------
 * util/utils.py:474-494 (cv2.connectedComponentsWithStats, confidences), models/ProtoSAM.py:242-289 (bbox, most
 * confident point). tab fp64: [n_found, n_kept, sum(pred), argmax_conf, 0,0,0,0] then 12 doubles per component:
 * {area, sum_x, sum_y, min_x, min_y, max_x, max_y, conf, best_x, best_y, best_p, 0}. */
int psam_ccl(const void* pred, const float* pfg, int H, int W, int cap, int* labels, int* parent, int* counters,
             int* roots, int* acc_i, void* acc_u, double* acc_d, const int* fg_sum, double* tab, void* stream);
/* psam_ccl for B images in one chain of six launches (blockIdx.z = image): pred u8 [B,H,W], pfg with `pfg_stride` floats between
 * images, every scratch array B times the single-image size (contiguous per image), fg_sum int32 [B] or NULL, tab fp64
 * [B][8 + 12*cap]. Replaces the per-slice loop over util/utils.py:468-541 of a batch of slices. */
int psam_ccl_batch(const void* pred, const float* pfg, long long pfg_stride, int B, int H, int W, int cap, int* labels, int* parent,
                   int* counters, int* roots, int* acc_i, void* acc_u, double* acc_d, const int* fg_sum, double* tab, void* stream);

/* ---- SAM prompt encoder / mask decoder --------------------------------------------------------------------------- */
/* grouped fp32 y = act((x [+ x2]) W^T + b) (+ resid); act 1 = ReLU.  transformer.py:218-240, mask_decoder.py:154-176 */
int psam_small_linear(const float* x, const float* x2, const float* W, const float* b, const float* resid, float* y,
                      int G, int M, int N, int K, long long xg, long long wg, long long bg, long long yg, int ldx,
                      int ldy, int act, void* stream);
/* One fp32 linear y = x W^T + b (+ resid) with few rows and a long contraction (the two-way block's MLP output 2048 -> 256,
 * modeling/transformer.py:170-171, common.py:13-26) as `ks` K ranges in one launch + a fixed-order sum: parts fp32 [ks][M][N] is
 * caller-owned scratch. W [N, K] row-major, K % (64 ks) == 0, ks >= 2. */
int psam_small_linear_splitk(const float* x, const float* W, const float* b, const float* resid, float* y, float* parts, int M, int N,
                             int K, int ks, int ldx, int ldy, void* stream);
/* softmax(q k^T / sqrt(hd)) v with <= 16 keys (token self-attention; image->token attention). transformer.py:151-182 */
int psam_small_attention(const void* q, const float* k, const float* v, void* out, int B, int Tq, int Tk, int NH, int hd,
                         int ldq, int ldk, int ldv, int ldo, int q_f16, void* stream);
/* token -> image cross attention over Nk <= 4096 keys; K, V fp32 (kv_f32 = 1) or half.  transformer.py:163-167, 98-103 */
int psam_t2i_attention(const float* q, const void* K, const void* V, float* out, int B, int T, int Nk, int NH,
                       int kv_f32, void* stream);
/* The same with the 4096 keys of every (prompt set, head) split over S workgroups (1 <= S <= 16; one slice at a time a launch is 8-16
 * workgroups of latency-bound waves otherwise) and a second small launch that merges the partial (max, sum, weighted values) in the
 * order of the split index: `part` fp32 scratch of >= B * NH * S * T * 18 elements. Same reference lines as psam_t2i_attention. */
int psam_t2i_attention_split(const float* q, const void* K, const void* V, float* out, int B, int T, int Nk, int NH,
                             int kv_f32, int S, float* part, void* stream);
/* out = (a [+ a2[m % a2_mod]]) w^T + bias [+ resid], all fp32 on the exact-fp32 MFMA: the decoder's projections of the
 * 4096 image tokens (keys [+ key_pe]) and ConvTranspose2d #1 as a GEMM. K % 32 == 0, N % 64 == 0, lds % 4 == 0.
 * transformer.py:163-167,176-180,98-103,218-240 (k_proj / v_proj / q_proj / out_proj); mask_decoder.py:54,137 */
int psam_gemm_f32(const float* a, const float* a2, int a2_mod, const float* w, const float* bias, const float* resid,
                  float* out, int M, int N, int K, int lda, int ldw, int ldo, void* stream);
/* psam_gemm_f32 with a HEAD-MAJOR result: out fp32 [M / nk][N / hd][nk][hd] (image, head, key, channel) - the K / V projections of the
 * two-way transformer's token-to-image attention (modeling/transformer.py:228-230 on the 4096-token operand), read by psam_t2i_attention
 * with bit 1 of kv_f32 set. M % nk == 0, N % hd == 0, hd % 4 == 0; no residual. */
int psam_gemm_f32_heads(const float* a, const float* a2, int a2_mod, const float* w, const float* bias, float* out, int M, int N, int K,
                        int lda, int ldw, int nk, int hd, void* stream);

/* psam_gemm_f32 / psam_gemm_f32_heads at fp32 accuracy on the fp16 matrix pipe (three v_mfma_f32_32x32x16_f16 products on (hi, lo) fp16
 * halves, fp32 accumulation; the dropped lo x lo term is 2^-22 of a product): out[M,N] = (a [+ a2[m % a2_mod]]) @ (wh + wl)^T + bias
 * [+ resid], the product times acc_scale. a / a2 / bias / resid / out fp32; wh / wl fp16 [N, ldw]: the halves of the fp32 weight times a
 * power of two 2^s (so that the lo half of a small weight stays a normal fp16), split once by the caller (wh = half(w 2^s), wl =
 * half(w 2^s - wh)); acc_scale = 2^-s. nk > 0: head-major output [M / nk][N / hd][nk][hd], no residual. K % 32 == 0, N % 64 == 0,
 * every pointer 16-byte aligned. The image-token side of the two-way transformer (modeling/transformer.py:163-167,176-180,98-103,228-230)
 * and the first transposed convolution of the upscaling (modeling/mask_decoder.py:137) as a GEMM. */
int psam_gemm_f32x3(const float* a, const float* a2, int a2_mod, const void* wh, const void* wl, const float* bias, const float* resid,
                    float* out, int M, int N, int K, int lda, int ldw, int ldo, int nk, int hd, float acc_scale, void* stream);
/* y = [LayerNorm](x[src row] + add_vec); emits fp32 y, half y, half (y + pe[row % pe_mod]). With in_mod > 0 the input
 * is one [in_mod,256] embedding per image and prompt row/in_mod reads image img_of_prompt[prompt] (null: image 0).
 * mask_decoder.py:126-127; transformer.py:164,178,180 */
int psam_ln_pe(const float* x, const float* add_vec, const float* w, const float* b, const float* pe, float* y32,
               void* y16, void* ype16, int M, int in_mod, int pe_mod, float eps, int do_ln, const int* img_of_prompt,
               void* stream);
/* PromptEncoder.get_dense_pe (prompt_encoder.py:62-71,195-206), token-major fp32 [gh*gw, 256] */
int psam_dense_pe(const float* G, int gh, int gw, float* pe, void* stream);
/* output tokens ++ point / box-corner embeddings.  prompt_encoder.py:73-101,208-214; mask_decoder.py:121-123 */
int psam_prompt_tokens(const float* coords, const int* labels, const float* G, const float* type_emb,
                       const float* out_tok, int B, int Ns, float img_size, float* tokens, void* stream);
/* LayerNorm2d -> GELU -> ConvTranspose2d(64->32) -> GELU -> hyper-network product.  mask_decoder.py:53-59,137-144 */
int psam_upscale_tail(const float* u1, const float* lnw, const float* lnb, const float* W2r, const float* b2,
                      const float* hyper, float* masks, int B, int g, void* stream);
/* Sam.postprocess_masks first stage; variant 0 upstream (bilinear, align_corners=False), 1 vendored SamBatched
 * (align_corners=True, modeling/sam.py:313-320), 2 vendored Sam (nearest, sam.py:154-160). */
int psam_mask_upsample(const float* low, int planes, int IN, int MID, int variant, float* out, void* stream);
/* pred = OR_b(upsample(low[b,sel]) > thr) sampled by F.interpolate(..., 'nearest') to OUT.  models/ProtoSAM.py:669-676 */
int psam_mask_union(const float* low, int B, int C, int sel, int IN, int MID, int OUT, int variant, float thr,
                    float* pred, void* stream);

/* Candidate statistics of SamAutomaticMaskGenerator without the full-resolution masks: for plane p (prompt p / nsel,
 * channel first + p % nsel of low [B,C,IN,IN]) over y < H, x < W of the MID x MID up-sampling, stats int32 [B*nsel, 8] =
 * {count(v > thr+off), count(v > thr-off), count(v > thr), min_x, min_y, max_x, max_y, 0} (empty: min = INT_MAX, max = -1).
 * automatic_mask_generator.py:293-310; utils/amg.py:156-176 (calculate_stability_score), :303-346 (batched_mask_to_box) */
int psam_mask_stats(const float* low, int B, int C, int first, int nsel, int IN, int MID, int H, int W, int variant,
                    float thr, float off, int* stats, void* stream);
/* out uint8 [n,H,W] = upsample(low plane idx[i]) > thr (automatic_mask_generator.py:305, predictor.py:238-239); with
 * label uint8 [H,W]: counts int64 [n,3] = {tp, fp, fn} against it (models/SamWrapper.py:8-13 get_iou). */
int psam_mask_binarize(const float* low, const int* idx, int n, int IN, int MID, int H, int W, int variant, float thr,
                       unsigned char* out, const unsigned char* label, long long* counts, void* stream);
/* the same statistics (and the binary masks, out uint8 [n,H,W] or null) of MATERIALISED candidate planes fp32 [n,H,W]: the
 * generator's crop layers / images not at the model input size (second resize of postprocess_masks).
 * automatic_mask_generator.py:221-316; utils/amg.py:156-176, 303-346 */
int psam_plane_stats(const float* planes, int n, int H, int W, float thr, float off, int* stats, void* out, void* stream);

/* PromptEncoder.mask_downscaling for mask prompts: masks fp32 [n,4g,4g] -> dense embeddings fp32 token-major [n,g*g,256].
 * wts = c1w[4][4] c1b[4] n1w[4] n1b[4] c2w[16][4][2][2] c2b[16] n2w[16] n2b[16] c3w[256][16] c3b[256] (4684 floats).
 * prompt_encoder.py:51-59,102-105; common.py:31-43 (LayerNorm2d) */
int psam_mask_downscale(const float* masks, const float* wts, int n, int g, float eps, float* out, void* stream);

/* Negative point prompts: keys[0] = most confident background pixel with p_bg >= thr (ProtoSAM.py:361-372), keys[1+k] = most
 * confident background pixel of the ring dilate_r(component k) \ component k, r iterations of a 3x3 dilation
 * (ProtoSAM.py:395-419; labels / tab as written by psam_ccl). key = float_bits(p_bg) << 32 | (0xFFFFFFFF - y*W - x); 0 = none. */
int psam_neg_points(const int* labels, const float* pbg, const double* tab, int H, int W, int max_comp, int r, float thr,
                    unsigned long long* keys, void* stream);

/* Slice hand-off from a scan volume (raw NIfTI voxels [Z,H,W]; vol_dtype 0 int16, 1 float32, 2 uint8, 3 int32).
 * psam_volume_stats: out[0] = sum(x), out[1] = sum(x^2) in fp64, x = voxel*slope + inter  (MR_normalize / get_CT_statistics,
 *   dataloaders/dataset_utils.py:76-108).
 * psam_volume_slices: out fp32 [Z,tile,S,S] = tile copies of cv2.resize((x - mean) * inv_std, (S,S), INTER_LINEAR) per slice
 *   (mode 0), or cv2.INTER_NEAREST of the raw values for label volumes (mode 1).
 *   dataloaders/ManualAnnoDatasetv2.py:165-187 (read_dataset), :317-327 (tile_z_dim). */
int psam_volume_stats(const void* vol, int vol_dtype, long long n, float slope, float inter, double* out, void* stream);
int psam_volume_slices(const void* vol, int vol_dtype, int Z, int H, int W, float slope, float inter, float mean,
                       float inv_std, int S, int tile, int mode, float* out, void* stream);

/* Convolution front-end of the ResNet-101 encoder (models/backbone/torchvision_backbones.py:12-52; torchvision's
 * deeplabv3_resnet101 backbone, output stride 8): im2col on token-major (NHWC) half maps for any kernel / stride / dilation /
 * padding, the 7x7 stride-2 stem straight from the fp32 NCHW image, and MaxPool2d(3, 2, 1). The convolutions themselves are
 * psam_gemm_f16 with BatchNorm folded into weights and bias (epilogue 3). */
int psam_im2col(const void* in, int B, int H, int W, int C, int kh, int kw, int stride, int dil, int pad, int ldo, void* out,
                void* stream);
int psam_im2col_stem(const float* img, int B, int H, int W, int ldo, void* out, void* stream);
int psam_maxpool3x3s2(const void* in, int B, int H, int W, int C, void* out, void* stream);

/* ---- rotation test-time augmentation: ProtoSAM.forward(..., degrees_rotate != 0) ---------------------------------------------
 * Replaces util/utils.py:66-83 `rotate_tensor_no_crop` (torchvision `rotate(expand=True)` + `resize(antialias=True)`) and
 * util/utils.py:40-59 `reverse_tensor` (`resize` + `rotate(expand=False)` + centre crop), called from models/ProtoSAM.py:544
 * and :553. torchvision 0.15.2 (requirements.txt:65) is not in /root/reference: restated from its tensor code path
 * (affine grid + grid_sample NEAREST / aten _upsample_bilinear2d_aa), see oracle/rotate.py.
 * psam_rotate_nearest: planes [C,H,W] fp32 -> [C,outH,outW]; xg / yg = the base-grid linspace values of the (expanded)
 *   canvas, rt6 = host pointer to the 3x2 rescaled theta (row-major), crop_* = first canvas row / column kept.
 * psam_resize_aa: anti-aliased bilinear [C,H,W] -> [C,OH,OW] (tmp = fp32 scratch [C,H,OW]). */
int psam_rotate_nearest(const float* src, float* dst, const float* xg, const float* yg, const float* rt6, int C, int H, int W,
                        int crop_y, int crop_x, int outH, int outW, void* stream);
int psam_resize_aa(const float* src, float* tmp, float* dst, int C, int H, int W, int OH, int OW, void* stream);

#ifdef __cplusplus
}
#endif
#endif
