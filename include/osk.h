/* osk.h — C ABI of libosk_hip.so: the MI355X (gfx950) kernels of the Open-Sora denoise path.
 *
 * The reference (hpcaitech/Open-Sora v2.0, /root/reference) has no native code and no FFI: every GPU
 * kernel on its hot path is imported from a pip dependency (flash-attn, liger-kernel, cuBLAS/cuDNN via
 * torch).  Each entry point below therefore cites the reference *call site* whose arithmetic it replaces
 * (paths relative to /root/reference).  INTEGRATION.md shows the ctypes binding a reference maintainer
 * would add (it is the binding open_sora_amd/_C.py uses).
 *
 * Conventions
 *  - Plain pointers and sizes only; no torch types.  All pointers are DEVICE pointers unless noted.
 *  - `stream` is a hipStream_t passed as void*.  Every function only ENQUEUES work on `stream`:
 *    no allocation, no synchronisation, no global state in any compute entry point -> safe under hipGraph capture.
 *    (The only process-wide state are the two TOOL switches osk_gemm_tile_override / osk_attention_rows_override -- A/B timing aids
 *    that default to "by estimate" and are never set by the product path.)
 *  - bf16 tensors are passed as `const void*` / `void*` (16-bit storage); f32 as float*.
 *  - Strides are in ELEMENTS of the tensor's dtype.  "rows_per_batch" addressing: logical row m of a
 *    [B*L, ...] matrix lives at  base + (m / L) * batch_stride + (m % L) * row_stride, so that streams
 *    stored inside larger joint buffers need no copies.
 *  - Return value: 0 = OSK_OK, <0 = invalid argument / unsupported shape (nothing was launched),
 *    >0 = hipError_t from the launch.
 */
#ifndef OSK_H
#define OSK_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

/* (round 6 added osk_gemm_group_bf16 without changing any existing contract: the version stays 2.)
 * 2 (round 5): osk_copy_rows_bf16 added; and the contracts that changed under version 1 during round 4 are now versioned -- the
 * key order osk_v_transpose_bf16 bakes into V^T for head_dim 64 (the 16x16x32 order of head_dim 72: attention_fwd.hip), the entry
 * points osk_gemm_geglu_bf16 / osk_attention_short_bf16 / osk_attention_hd512_fwd_ws_bf16 / osk_causal_conv3d_gnin_ndhwc_bf16.
 * A binding must refuse a library whose version it was not written for (open_sora_amd/_C.py does). */
#define OSK_ABI_VERSION 2
int osk_abi_version(void);
/* name of the arch the library was compiled for ("gfx950") — host-only, no GPU needed */
const char* osk_arch(void);

/* ---- LayerNorm(no affine, eps) + adaLN modulate:  out = (1 + scale[b]) * LN(x) + shift[b]
 * replaces nn.LayerNorm + the bf16 elementwise chain at opensora/models/mmdit/layers.py:205-206,
 * 223-224, 248, 252, 311-312 and LastLayer layers.py:400.  x/out bf16 [B*L, D]; shift/scale f32,
 * element (b, d) at ptr[b * mod_batch_stride + d].  Statistics and modulate in f32, one rounding. */
int osk_ln_modulate_bf16(const void* x, int64_t x_batch_stride, int64_t x_row_stride,
                         void* out, int64_t out_batch_stride, int64_t out_row_stride,
                         const float* shift, const float* scale, int64_t mod_batch_stride,
                         int B, int L, int D, float eps, void* stream);

/* ---- C = epilogue(A @ W^T + bias) on MFMA bf16 (f32 accumulate).
 * replaces every nn.Linear on the path: layers.py:146-152,209-215,226-232,247-252,314-320,332-333,
 * 401; model.py:176-180,191.  A bf16 [M, K] (rows_per_batch addressing), W bf16 [N, K] row-major
 * (nn.Linear.weight layout), bias f32 [N] or NULL.
 * Epilogue, in f32, in this order:
 *   v = acc + bias[n];  if (n >= gelu_from) v = gelu_tanh(v);
 *   if (gate) v = res[m, n] + gate[b * gate_batch_stride + n] * v;      (res bf16, same addressing as C)
 *   C[m, n] = bf16(v)            (or f32 when out_f32 != 0)
 * K % 64 == 0 required; any M, N >= 1.  gelu_from = N disables GELU.  res may alias C. */
int osk_gemm_bf16(const void* A, int64_t a_batch_stride, int64_t a_row_stride, int a_rows_per_batch,
                  const void* W, int64_t w_row_stride, const float* bias,
                  void* C, int64_t c_batch_stride, int64_t c_row_stride, int c_rows_per_batch,
                  const void* res, const float* gate, int64_t gate_batch_stride,
                  int M, int N, int K, int gelu_from, int out_f32, void* stream);

/* reporting / tools: the tile kernel osk_gemm_bf16 picks for an [M, N, K] problem whose operands the large tiles support -- 2 = 256 x 256
 * (gemm256x_kernel), 1 = 256 x 128 (gemm256p_kernel), 0 = 128 x 128 (gemm_bf16_kernel) -- by estimated rounds of the chip x per-tile rate;
 * osk_gemm_tile_override(kind) forces one of them process-wide for same-process A/B timing (tools/gemm_tile_ab.py; -1 = back to the estimate;
 * never set in production code). */
int osk_gemm_tile_choice(int M, int N, int K);
int osk_gemm_tile_override(int tile_kind);

/* ---- TWO Linear layers that differ only in their operands and row count in ONE launch.
 * replaces the img-stream / txt-stream pairs of DoubleStreamBlockProcessor (layers.py:209-215 img_attn.qkv | txt_attn.qkv,
 * 247 img_attn.proj | 251 txt_attn.proj, 248 img_mlp | 252 txt_mlp): same N, K and epilogue kind, separate weights, inputs and
 * outputs.  Exactly osk_gemm_bf16(first...) followed by osk_gemm_bf16(second...) with out_f32 = 0 (the two must not depend on
 * each other); where both problems take the 256 x 256 tile kernel they are walked as one tile list, so the small text GEMM
 * (a third of the chip for one round when launched alone) fills the last round of the image GEMM. */
typedef struct OskGemmOperands {
  const void* A; int64_t a_batch_stride, a_row_stride; int a_rows_per_batch;
  const void* W; int64_t w_row_stride; const float* bias;
  void* C; int64_t c_batch_stride, c_row_stride; int c_rows_per_batch;
  const void* res; const float* gate; int64_t gate_batch_stride;
  int M;
} OskGemmOperands;
int osk_gemm_bf16_pair(const OskGemmOperands* first, const OskGemmOperands* second, int N, int K, int gelu_from, void* stream);

/* ---- a GROUP of up to four Linear problems that share K (at most two plain + two V^T tasks) on the 256 x 256 tile kernel (round 6).
 * replaces, per block, the QKV projection + the V re-layout of the reference's attention path: layers.py:209-220 (img / txt qkv
 * Linear + rearrange), :314-320 (linear1 + split + rearrange), math.py:22-36 (attention() permutes V to the kernel's layout) --
 * the V columns of the projection are written DIRECTLY as the key-major V^T operand of osk_attention_fwd_*_bf16, so the separate
 * osk_v_transpose_bf16 pass (read + write of V per block) disappears.  The plain tasks go out as one launch (two of equal N: one tile
 * list, the small text-stream problem fills the image problem's last round), the V^T tasks as one launch of the kernel's V^T form.
 * Plain task (vt_head_dim == 0): exactly osk_gemm_bf16(op..., N, K, gelu_from, out_f32 = 0), except that the physical columns
 *   [skip_from, skip_from + skip_len) of W / bias / C are neither computed nor stored (skip_len == 0: none; skip_from % 256 == 0,
 *   skip_len % 8 == 0, no gate): a single-stream block's linear1 without its V columns -- row layout [q | k | . | mlp].
 * V^T task (vt_head_dim = 64 / 72 / 128): op.A = the activations X bf16 [op.M = B * L rows, K] (rows_per_batch addressing,
 *   a_rows_per_batch = L), op.W = W_v bf16 [N = H * hd, K], op.bias = b_v f32 [N] | NULL, op.C = V^T base pointer (bf16),
 *   c_batch_stride = elements between batch items, c_row_stride = elements between (head, dim) rows (>= round_up(L, 64));
 *   res / gate NULL; gelu_from / skip_* ignored:
 *     C[b * c_batch_stride + n * c_row_stride + p] = bf16(sum_k X[b, key(p), k] W_v[n, k] + b_v[n])   for key(p) < L,  else 0,
 *   p < round_up(L, 64), key(p) = the position -> key map of osk_v_transpose_bf16 for this head_dim (inside every 64-key group) --
 *   i.e. the result of osk_gemm_bf16 into a [B, L, N] tensor followed by osk_v_transpose_bf16, with the same f32 accumulation
 *   and ONE rounding (the bias is added after the K loop instead of initialising the accumulators: results may differ from the
 *   two-kernel path in the last bf16 bit).  A stream stored behind another one on the key axis (img behind txt) passes
 *   C + its first position (a multiple of 64).
 * All tasks must qualify for the 256 x 256 tile kernel (M >= 256, N >= 128 (V^T: H * hd >= 256), operands within 4 GiB):
 * otherwise OSK_EUNSUPPORTED is returned and NOTHING is launched (the caller runs the single calls). */
typedef struct OskGemmTask {
  OskGemmOperands op;
  int N, gelu_from;
  int skip_from, skip_len;
  int vt_head_dim;
} OskGemmTask;
int osk_gemm_group_bf16(const OskGemmTask* tasks, int n_tasks, int K, void* stream);

/* ---- GEGLU up-projection (the STDiT-generation block's "GEGLU MLP" of BASELINE.json's
 * north_star; the mounted v2.0 reference has no GEGLU call site -- its MLP is Linear -> GELU(tanh) -> Linear, layers.py:277-281 --
 * so this entry is PARITY-UNPINNED: semantics = diffusers' FeedForward(activation_fn="geglu") EXCEPT the activation: diffusers'
 * GEGLU applies the exact (erf) F.gelu to the gate, this entry the tanh approximation the v2.0 blocks use everywhere
 * (|gelu_tanh - gelu_erf| <= ~1e-3 absolute, a systematic difference a checkpoint trained with diffusers GEGLU would see per MLP;
 * the fp64 test oracle uses the same tanh formula, so it does not measure this deviation -- an erf epilogue is one more class):
 *   C[m, j] = (A W_v^T + b_v)[m, j] * gelu_tanh((A W_g^T + b_g)[m, j]),   j < N_out
 * as ONE GEMM with N = 2 N_out whose epilogue multiplies value and gate in registers.  W_packed [2 N_out, K] / bias_packed
 * [2 N_out]: value and gate rows interleaved in blocks of 16 -- packed row 32 j2 + i (i < 16) = value row 16 j2 + i, packed row
 * 32 j2 + 16 + i = gate row 16 j2 + i (open_sora_amd/_C.py::geglu_pack builds it once at plan time); N_out % 16 == 0.
 * Shapes the 256 x 256 tile kernel does not take (M < 256, narrow N, ...) run the plain GEMM into `workspace` (>= M * 2 N_out * 2
 * bytes, 16-byte aligned) followed by a row kernel; without a workspace they return OSK_EUNSUPPORTED and launch nothing. */
int osk_gemm_geglu_bf16(const void* A, int64_t a_batch_stride, int64_t a_row_stride, int a_rows_per_batch,
                        const void* W_packed, int64_t w_row_stride, const float* bias_packed, void* C,
                        int64_t c_batch_stride, int64_t c_row_stride, int c_rows_per_batch, int M, int N_out, int K,
                        void* workspace, int64_t workspace_bytes, void* stream);

/* ---- FP8 (OCP e4m3fn) variant of the Linear GEMM: BASELINE configs[4] ("fp8 MFMA"); the reference itself runs its
 * nn.Linear layers (same call sites as osk_gemm_bf16) in bf16, so this is an opt-in mode.
 * osk_quantize_rows_fp8: dynamic per-row quantisation of a bf16 [M, K] activation (rows_per_batch addressing):
 *   scales[m] = absmax(x[m, :]) / 448  (1.0 for an all-zero row),  out8[m, k] = e4m3(clamp(x[m, k] * (448 / absmax)))
 *   out8 is contiguous [M, K] bytes.  K % 8 == 0.  Also used once per weight matrix at load time.
 * osk_gemm_fp8: C = epilogue(a_scale[m] * w_scale[n] * (A8 @ W8^T) + bias), the epilogue of osk_gemm_bf16, on
 *   v_mfma_f32_32x32x64_f8f6f4 (f32 accumulate).  Strides of A8 / W8 in bytes, multiples of 16; K % 128 == 0,
 *   M >= 256, N >= 128 (returns OSK_EUNSUPPORTED (-2) otherwise: such layers stay on osk_gemm_bf16). */
/* osk_ln_modulate_fp8: osk_ln_modulate_bf16 followed by osk_quantize_rows_fp8 in one pass (bit-identical to the pair:
 * the modulated row is rounded to bf16 first): out8 = e4m3 bytes [B*L, D] contiguous, scales f32 [B*L]. */
int osk_ln_modulate_fp8(const void* x, int64_t x_batch_stride, int64_t x_row_stride, void* out8, float* scales,
                        const float* shift, const float* scale, int64_t mod_batch_stride,
                        int B, int L, int D, float eps, void* stream);
int osk_quantize_rows_fp8(const void* x, int64_t x_batch_stride, int64_t x_row_stride, int rows_per_batch,
                          void* out8, float* scales, int M, int K, void* stream);
int osk_gemm_fp8(const void* A8, int64_t a_batch_stride, int64_t a_row_stride, int a_rows_per_batch,
                 const float* a_scale, const void* W8, int64_t w_row_stride, const float* w_scale,
                 const float* bias, void* C, int64_t c_batch_stride, int64_t c_row_stride, int c_rows_per_batch,
                 const void* res, const float* gate, int64_t gate_batch_stride,
                 int M, int N, int K, int gelu_from, int out_f32, void* stream);

/* ---- skinny matrix-vector batch (any Bv; launched in slices of <= 8 rows, K <= 16384):
 * out[b, n] (+)= act_in(x[b, :]) . W[n, :] + bias[n]
 * replaces Modulation (layers.py:184-191), MLPEmbedder (layers.py:91-99) and LastLayer.adaLN_modulation
 * (layers.py:396,399): weight-bandwidth bound.  One launch covers a LIST of layers that share x:
 * descriptor arrays (device memory, n_tasks entries each, one task = up to 64 consecutive rows of one
 * layer):  w_ptrs[i] -> bf16 W rows [rows, K];  b_ptrs[i] -> bf16 bias or 0;  out_cols[i] = column
 * offset in out;  n_rows[i].  x f32 [Bv, K], out f32 [Bv, out_batch_stride].
 * act_in: 0 none, 1 SiLU.  accumulate != 0 adds into out. */
int osk_gemv_tasks_bf16(const float* x, int64_t x_batch_stride, int Bv, int K,
                        const uint64_t* w_ptrs, const uint64_t* b_ptrs, const int32_t* out_cols,
                        const int32_t* n_rows, int n_tasks,
                        float* out, int64_t out_batch_stride, int act_in, int accumulate, void* stream);

/* ---- sinusoidal timestep embedding: out[b] = [cos(a) | sin(a)], a = time_factor * t[b] * exp(-ln(max_period) i / half)
 * replaces timestep_embedding, layers.py:68-88 (f32). */
int osk_timestep_embedding(const float* t, int B, int dim, float max_period, float time_factor,
                           float* out, void* stream);

/* ---- RoPE angle tables from integer (t, h, w) position ids.
 * replaces EmbedND / rope (layers.py:31-44, math.py:50-57; f64 angles) and LigerEmbedND / liger_rope
 * (layers.py:47-65, math.py:39-47; f32 angles).  ids f32 [n_rows, n_axes]; axes_dim[n_axes] host ints;
 * cos/sin f32 [n_rows, sum(axes_dim)/2] with pair index j ordered axis-major.  f32_angles selects the
 * liger arithmetic. */
int osk_rope_table(const float* ids, int64_t n_rows, int n_axes, const int32_t* axes_dim_host,
                   double theta, int f32_angles, float* cos_out, float* sin_out, void* stream);

/* ---- per-head RMSNorm (QK-norm) + RoPE on q and k, in place, inside a [B, L, ld] projection buffer.
 * replaces QKNorm/FusedRMSNorm (layers.py:114-135 -> liger RMSNorm "llama") followed by apply_rope
 * (math.py:60-65, rope_mode 0 = interleaved pairs (2j,2j+1)) or LigerRopeFunction (math.py:27,
 * rope_mode 1 = half-split pairs (j, j+hd/2)).
 * q row (b,l) head h at q + b*batch_stride + l*row_stride + h*hd (same for k).  Rows l < l_split use
 * (q_scale0,k_scale0) (the txt stream's norm weights), rows >= l_split use (q_scale1,k_scale1).
 * Scales bf16 [hd].  cos/sin f32 [*, L, hd/2] with batch stride cs_batch_stride (0 = shared).
 * Rounding points follow the reference: bf16(x*rrms) * bf16 scale -> bf16, rotate in f32, -> bf16.
 * Either q or k (not both) may be NULL: only the other one is processed (the sequence-parallel path norms
 * K first so its all-gather can start while the Q / MLP projections run).
 * q_mult multiplies the rotated q (not k) in f32 BEFORE its final rounding to bf16: pass 1.0 for the reference's
 * values, or softmax_scale * log2(e) together with q_prescaled = 1 in osk_attention_fwd_bf16 (the reference
 * multiplies the f32 scores by the scale after q.k; folding it into q's single rounding keeps the same
 * one-rounding error on q and lets the attention kernel exponentiate the MFMA output directly). */
int osk_qknorm_rope_bf16(void* q, void* k, int64_t batch_stride, int64_t row_stride,
                         const void* q_scale0, const void* k_scale0,
                         const void* q_scale1, const void* k_scale1, int l_split,
                         const float* cos_t, const float* sin_t, int64_t cs_batch_stride,
                         int B, int L, int H, int hd, int rope_mode, float eps, float q_mult, void* stream);

/* ---- V -> key-major transposed copy for the attention kernel's PV operand.
 * (internal layout, no reference counterpart: flash-attn does this transpose in shared memory.)
 * v row (b,l) head h at v + b*batch_stride + l*row_stride + h*hd.
 * vt [B, H, hd, Lp], Lp = round_up(L, 64), zero-filled for keys >= L.  The key order inside a 64-key tile is the one in
 * which the attention kernel of that head_dim holds P for its second product (an internal contract between the two calls):
 *   hd 128 (P.V on 32x32x16 MFMAs): inside every group of 16 keys the two middle quads are swapped (k0-3, k8-11, k4-7,
 *     k12-15) -- the order the score accumulators hand P back, no cross-lane shuffle;
 *   hd 72 and hd 64 (the same loop; P.V on 16x16x32 MFMAs, 80 instead of 96 padded rows): 16-byte chunk c of a tile row = 32-key half c / 4, MFMA
 *     lane row c % 4, holding keys {0-3, 8-11}, {16-19, 24-27}, {4-7, 12-15}, {20-23, 28-31} of that half for rows 0..3. */
int osk_v_transpose_bf16(const void* v, int64_t batch_stride, int64_t row_stride,
                         void* vt, int B, int L, int H, int hd, void* stream);

/* ---- flash attention forward, non-causal: out = softmax(q k^T * scale) v, bf16 I/O, f32 softmax.
 * replaces flash_attn_func (math.py:16-19,33) and _flash_attn_forward with LSE
 * (distributed.py:148-161).
 * q  row (b,i) head h at q + b*q_batch_stride + i*q_row_stride + h*hd           (i < Lq)
 * k  key j lives in segment s = j / seg_len, r = j % seg_len (sequence-parallel all-gather layout;
 *    n_seg = 1, seg_len = Lk for one GPU): k + s*k_seg_stride + b*k_batch_stride + r*k_row_stride + h*hd
 * vt as written by osk_v_transpose_bf16 per segment: vt + s*vt_seg_stride + ((b*H + h)*hd + d)*seg_lp + r'
 *    with seg_lp = round_up(seg_len, 64).
 * out row (b,i) head h at out + b*o_batch_stride + i*o_row_stride + h*hd (may alias the dead v slot).
 * lse f32 [B, H, Lq] (natural log of the softmax denominator incl. scale) or NULL.
 * q_prescaled != 0: q already carries scale * log2(e) (osk_qknorm_rope_bf16's q_mult) and `scale` is ignored;
 * q_prescaled == 0: the kernel applies `scale` itself (the head_dim-72 hand-scheduled kernel then re-rounds
 * scale*log2(e)*q to bf16 once per workgroup: one extra bf16 rounding of q).
 * kv_batches: 0 (or B) = every query batch has its own keys; otherwise query batch b attends to key/value batch
 * b % kv_batches (head-parallel sequence parallelism: the B * P received query chunks share B key sets).
 * hd in {64, 72, 128}. */
int osk_attention_fwd_bf16(const void* q, int64_t q_batch_stride, int64_t q_row_stride,
                           const void* k, int64_t k_seg_stride, int64_t k_batch_stride, int64_t k_row_stride,
                           const void* vt, int64_t vt_seg_stride,
                           void* out, int64_t o_batch_stride, int64_t o_row_stride,
                           float* lse, int B, int H, int Lq, int n_seg, int seg_len, int hd,
                           float scale, int q_prescaled, int kv_batches, void* stream);

/* The same call with a caller-owned workspace (16-byte aligned, osk_attention_workspace_bytes() is always enough).
 * A workgroup owns a CU for its whole key loop, so a launch takes ceil(units / CUs) rounds; with a workspace the
 * work units of the last, partial round are cut into up to 8 key parts (whole key segments when n_seg > 1, runs of
 * 64-key tiles otherwise) that fill the idle CUs, and a small merge kernel combines the parts by their LSE
 * (head_dim 72 / 128; other head dims and workspace == NULL: exactly osk_attention_fwd_bf16).  Partials are kept in
 * f32 and rounded to bf16 once; a split row differs from the unsplit one only through the bf16 rounding of P against
 * the reference max of its own key part (same error bound against the exact result). */
int64_t osk_attention_workspace_bytes(void);
/* reporting / tests: the number of key parts (1 = no split) osk_attention_fwd_ws_bf16 -- a call WITHOUT a score bound -- would use */
int osk_attention_tail_split_factor(int B, int H, int Lq, int n_seg, int seg_len, int hd, int64_t workspace_bytes);
/* reporting / tests (round 5): the launch shape of osk_attention_fwd_bounded_bf16 for these arguments, from the selection code the
 * call itself runs: returns the number of key parts of the last round's work units and stores the query rows per work unit (256,
 * or 512 = the wide head_dim 72 / 64 layout, chosen by estimated rounds of the chip -- few batch x head pairs keep 256). */
int osk_attention_launch_shape(int B, int H, int Lq, int n_seg, int seg_len, int hd, float score_bound,
                               int64_t workspace_bytes, int* rows_per_unit);
/* tools only: force the work-unit rows of bounded head_dim 72 / 64 calls process-wide (256, or 512 where the wide layout is legal) for
 * same-process A/B timing (tools/attn_layout_ab.py); 0 = back to the estimate.  Never set in production code. */
int osk_attention_rows_override(int rows_per_unit);
int osk_attention_fwd_ws_bf16(const void* q, int64_t q_batch_stride, int64_t q_row_stride,
                           const void* k, int64_t k_seg_stride, int64_t k_batch_stride, int64_t k_row_stride,
                           const void* vt, int64_t vt_seg_stride,
                           void* out, int64_t o_batch_stride, int64_t o_row_stride,
                           float* lse, int B, int H, int Lq, int n_seg, int seg_len, int hd,
                           float scale, int q_prescaled, int kv_batches, void* workspace, int64_t workspace_bytes, void* stream);

/* The same call with a caller-supplied BOUND on the scores: |scale * log2(e) * q . k| <= score_bound for every (query, key) of
 * the launch (log2 units; with q_prescaled the bound is on q . k as stored).  The denoiser knows one for free: q and k come out
 * of an RMS norm (layers.py:102-135), so |q| <= sqrt(hd) max|w_q| and |k| <= sqrt(hd) max|w_k| (RoPE is a rotation).  With a bound
 * in (0, 56], n_seg == 1 and seg_len % 64 == 0 the head_dim-72 / 128 kernels run their FAST body: the online-softmax reference
 * is the constant bound instead of a tracked maximum (P = exp2(s - bound) can neither overflow nor vanish), which removes the
 * per-tile max chains, the rescale path and the segment / ragged-tile bookkeeping from the issue-bound loop.  The result is the
 * same softmax (a different, equally valid reference point: same error bound against the exact result).  score_bound == 0, or
 * any call shape outside the above: exactly osk_attention_fwd_ws_bf16.  A bound that is violated is the caller's bug: scores
 * above it by more than ~100 overflow. */
int osk_attention_fwd_bounded_bf16(const void* q, int64_t q_batch_stride, int64_t q_row_stride,
                           const void* k, int64_t k_seg_stride, int64_t k_batch_stride, int64_t k_row_stride,
                           const void* vt, int64_t vt_seg_stride,
                           void* out, int64_t o_batch_stride, int64_t o_row_stride,
                           float* lse, int B, int H, int Lq, int n_seg, int seg_len, int hd,
                           float scale, int q_prescaled, int kv_batches, float score_bound,
                           void* workspace, int64_t workspace_bytes, void* stream);

/* ---- the bound taken from the OPERANDS, on the device (round 6).  The weight-derived bound above is sqrt(hd)-loose per side: a
 * checkpoint whose QK-norm scale vectors (layers.py:102-135) have a few large entries exceeds 56 on paper while its actual q / k rows
 * stay far below -- and without a bound the step falls to the slower general body.  Here the caller passes NO promise:
 * osk_rownorm2_max_bf16: out[b * H + h] = max over rows l of sum_d x[b, l, h, d]^2 of a bf16 [B, L, H * hd] view (f32, device
 *   memory; one read of the tensor; accumulate != 0 folds into the values already there -- several key segments, or ranks
 *   that max-reduce afterwards).  Run it on the q and on the k the attention call receives.
 * osk_attention_fwd_auto_bf16: osk_attention_fwd_ws_bf16 on q_prescaled operands (q carries scale * log2 e: the norms must be those
 *   of the values the MFMA multiplies), q_norm2_max f32 [B, H], k_norm2_max f32 [kv_batches or B, H].  Every work unit derives its
 *   own bound sqrt(q_norm2_max k_norm2_max) (Cauchy-Schwarz: always valid) and the call enqueues a PAIR of launches over the same
 *   grid: units whose bound is <= 56 run the FAST body with that bound as softmax reference, the others the general body; the
 *   workgroups of the other kind exit at once.  No host round trip, no global state, hipGraph-capturable.  Cost over a host-promised
 *   FAST call: the norm pass (HBM-bound) + one launch of early-exiting workgroups. */
int osk_rownorm2_max_bf16(const void* x, int64_t batch_stride, int64_t row_stride, int B, int L, int H, int hd,
                          float* out, int accumulate, void* stream);
int osk_attention_fwd_auto_bf16(const void* q, int64_t q_batch_stride, int64_t q_row_stride,
                           const void* k, int64_t k_seg_stride, int64_t k_batch_stride, int64_t k_row_stride,
                           const void* vt, int64_t vt_seg_stride,
                           void* out, int64_t o_batch_stride, int64_t o_row_stride,
                           float* lse, int B, int H, int Lq, int n_seg, int seg_len, int hd,
                           float scale, int q_prescaled, int kv_batches, const float* q_norm2_max, const float* k_norm2_max,
                           void* workspace, int64_t workspace_bytes, void* stream);

/* ---- fp8 P.V variant of the attention (opt-in fp8 mode, BASELINE configs[4]; head_dim 72 / 128): QK^T, the softmax and
 * the output as osk_attention_fwd_ws_bf16, but P (<= 2^8 by construction) and V^T are OCP e4m3 and a 64-key tile's P.V
 * is one v_mfma_f32_32x32x64_f8f6f4 per O^T row tile.
 * osk_v_transpose_fp8: V bf16 [B, L, H*hd] view -> vt8 e4m3 bytes [B, H, RP, Lp], RP = (hd + 1) rounded up to 16
 *   (80 / 144), Lp = L rounded up to 64: rows 0..hd-1 = clamp(V / scales[b*H + h]) in the key order the kernel's
 *   accumulator layout dictates, row hd = 1.0 for keys < L else 0 (softmax denominator / key validity), further rows 0.
 *   scales f32 [B, H] (device memory; typically absmax over the head / 448, computed by the caller).
 * osk_attention_fwd_pv8_bf16: vt8 from the call above (per key segment, vt8_seg_stride in BYTES), v_scale = the same
 *   scales indexed by (key batch, head); everything else as osk_attention_fwd_ws_bf16. */
/* osk_v_scale_fp8: scales[b*H + h] = absmax(V[b, :, h, :]) / 448 (1.0 for an all-zero head): the e4m3 scales
 * osk_v_transpose_fp8 / osk_attention_fwd_pv8_bf16 take.  (Under sequence parallelism max-reduce them over the ranks.) */
int osk_v_scale_fp8(const void* v, int64_t v_batch_stride, int64_t v_row_stride, float* scales, int B, int L, int H,
                    int hd, void* stream);
int osk_v_transpose_fp8(const void* v, int64_t v_batch_stride, int64_t v_row_stride, const float* scales, void* vt8,
                        int B, int L, int H, int hd, void* stream);
int osk_attention_fwd_pv8_bf16(const void* q, int64_t q_batch_stride, int64_t q_row_stride,
                               const void* k, int64_t k_seg_stride, int64_t k_batch_stride, int64_t k_row_stride,
                               const void* vt8, int64_t vt8_seg_stride, const float* v_scale,
                               void* out, int64_t o_batch_stride, int64_t o_row_stride,
                               float* lse, int B, int H, int Lq, int n_seg, int seg_len, int hd,
                               float scale, int q_prescaled, int kv_batches, void* workspace, int64_t workspace_bytes,
                               void* stream);

/* ---- short-sequence attention with an optional ALiBi bias (the STDiT-generation block's temporal
 * self-attention over T <= 64 frames, B * H * W independent sequences, "RoPE/ALiBi" in BASELINE.json's north_star).  The mounted
 * v2.0 reference passes alibi_slopes=None at every flash-attn call site (opensora/models/mmdit/math.py:22-36), so this entry is
 * PARITY-UNPINNED; semantics = flash-attn's documented `alibi_slopes`:
 *   out[b, i, h] = softmax_j(scale * q_i . k_j - alibi_slopes[h] * |i + Lk - Lq - j|) v_j ,   Lq, Lk <= 64
 * q, k, v, out bf16 [B, L, H * hd] views (row strides in elements, last dim contiguous), alibi_slopes f32 [H] or NULL (no bias).
 * One wave per (batch, head), registers only, HBM-bound; V is taken as is (no osk_v_transpose_bf16).  hd in {64, 72, 128};
 * longer sequences: OSK_EUNSUPPORTED (use osk_attention_fwd_bf16). */
int osk_attention_short_bf16(const void* q, int64_t q_batch_stride, int64_t q_row_stride, const void* k, int64_t k_batch_stride,
                             int64_t k_row_stride, const void* v, int64_t v_batch_stride, int64_t v_row_stride, void* out,
                             int64_t o_batch_stride, int64_t o_row_stride, const float* alibi_slopes, int B, int H, int Lq, int Lk,
                             int hd, float scale, void* stream);

/* name of the device kernel osk_attention_fwd_bf16 dispatches to for (hd, seg_len) (reporting only: bench.py labels its
 * roofline line and the rocprof stats with it). */
const char* osk_attention_kernel_name(int hd, int seg_len);
/* ... and of the loop BODY a call of osk_attention_fwd_bounded_bf16 with this segment layout and score bound runs (reporting /
 * tests only): "attn_asm72_kernel<FAST>" = the bounded body (no max tracking; needs 0 < score_bound <= 56 and, with several key
 * segments, segments of >= 3 tiles), "attn_asm72_kernel<general>" = the running-reference body (head_dim 64 runs the head_dim-72 kernels).  The
 * choice is data-dependent through score_bound (open_sora_amd/mmdit.py derives it from the QK-norm scale vectors of
 * opensora/models/mmdit/layers.py:113-135), so bench.py prints it next to the timing. */
const char* osk_attention_body_name(int hd, int n_seg, int seg_len, float score_bound);

/* ---- classifier-free-guidance combine + Euler step of the rectified-flow sampler (f32 math):
 *   v = u2 + g_img*(u - u2) + g_txt*(c - u);  x_out = x + dt * v
 * replaces opensora/utils/sampling.py:217-222.  pred bf16 [3, n] = (cond, uncond, uncond_2) chunks,
 * x bf16 [n], x_out bf16 [n].  g_img_vec (f32 [n]) overrides the scalar g_img when non-NULL
 * (temporal guidance ramp, sampling.py:209-216). */
int osk_cfg_euler_bf16(const void* pred, int64_t n, const void* x, void* x_out,
                       float g_txt, float g_img, const float* g_img_vec, float dt, void* stream);

/* ---- strided row copy: dst[j, b, l, 0:C] = src[j, b, l, 0:C] (bf16; element strides; src_batch_stride 0 broadcasts one item).
 * replaces the host-side tensor glue of a denoise step: `torch.cat([img, img, img])` of the CFG triple
 * (opensora/utils/sampling.py:196-201), the operand assembly in front of img_in / cond_in (opensora/models/mmdit/model.py:170-176),
 * and -- with n_chunks = P -- the two rearranges around the head all-to-all of the sequence-parallel attention
 * (opensora/models/mmdit/distributed.py:473-495: [B, L/P, P x Dg] <-> [P, B, L/P, Dg]; the token-major side has chunk stride Dg).
 * C and all strides multiples of 4 elements, pointers 8-byte aligned. */
int osk_copy_rows_bf16(const void* src, int64_t src_chunk_stride, int64_t src_batch_stride, int64_t src_row_stride, void* dst,
                       int64_t dst_chunk_stride, int64_t dst_batch_stride, int64_t dst_row_stride, int n_chunks, int B, int L, int C,
                       void* stream);

/* =====================================================================================================
 * Causal 3-D VAE (HunyuanVideo VAE, /root/reference/opensora/models/hunyuan_vae).  Activations are channels-last
 * NDHWC bf16 ([B, T, H, W, C] contiguous); the NCTHW <-> NDHWC conversion happens once at the module boundary.
 * ===================================================================================================== */

/* ---- CausalConv3d (+ optional fused nearest upsample in front, + optional fused residual add behind).
 * replaces CausalConv3d.forward = F.pad(x, (k/2,k/2,k/2,k/2,k-1,0), "replicate") + ChannelChunkConv3d
 * (hunyuan_vae/unet_causal_3d_blocks.py:82-96; vae/utils.py:153-190 — the channel chunking is unnecessary here:
 * 64-bit addressing), the interpolation of UpsampleCausal3D.forward (unet_causal_3d_blocks.py:136-150: frame 0
 * upsampled in H,W only, later frames in T,H,W) when up_t/up_hw != 0, the strided DownsampleCausal3D conv
 * (:175), the 1x1x1 conv_shortcut / quant_conv / post_quant_conv (ksize 1), and the `input + hidden` of
 * ResnetBlockCausal3D.forward (:257) when res != NULL.
 *   x   bf16 [B, T, H, W, Cin]   source grid BEFORE the virtual upsample; Cin = 8 * 2^j (pad 3 -> 8 channels)
 *   w   bf16 [Cout, w_row_stride], element [co][((dt*3 + dh)*3 + dw)*Cin + ci] = weight[co][ci][dt][dh][dw];
 *       rows zero-padded to w_row_stride >= round_up(ksize^3 * Cin, 64)
 *   bias f32 [Cout] or NULL;  res bf16 [B, To, Ho, Wo, Cout] or NULL;  out bf16 [B, To, Ho, Wo, Cout]
 *   To = (Tu-1)/stride_t + 1 with Tu = up_t ? 1 + 2(T-1) : T;  Ho = (Hu-1)/stride_h + 1 with Hu = up_hw ? 2H : H
 * f32 accumulate, one rounding (bias and residual added in f32).
 * Kernel by shape (same contract for all, chosen inside): 3 x 3 x 3, stride 1, Cin % 128 == 0, Cout >= 128 and whole 16 x 16 output
 * bricks -> the LDS sliding window (the halo brick of a 32-channel block resident in LDS, taps = immediate offsets; with or
 * without the fused upsample; two-frame tiles for Cout == 128); other Cin % 128 == 0, Cout >= 128 shapes -> the table-driven implicit
 * GEMM; Cout <= 4 with Cin == 128 -> a v_dot2_f32_bf16 reduction; everything else -> the 128 x 128 implicit-GEMM tile. */
int osk_causal_conv3d_ndhwc_bf16(const void* x, int B, int T, int H, int W, int Cin,
                                 const void* w, int64_t w_row_stride, const float* bias, int Cout, int ksize,
                                 int stride_t, int stride_h, int stride_w, int up_t, int up_hw,
                                 const void* res, void* out, int To, int Ho, int Wo, void* stream);

/* ---- the same convolution + the GroupNorm statistics of its OUTPUT in the epilogue: the first half of the nn.GroupNorm
 * that consumes this tensor next (ResnetBlockCausal3D.norm2 after conv1, the next block's norm1 / Attention.group_norm /
 * conv_norm_out after conv2 + residual or a Down/Upsample conv: unet_causal_3d_blocks.py:247-259) without the extra
 * read of the tensor that osk_groupnorm_stats_ndhwc_bf16 costs.
 *   gn_sums f64 [B, gn_groups, 2], ZEROED BY THE CALLER; on return (stream order) it holds what
 *   osk_groupnorm_stats_ndhwc_bf16(out, ...) would (sums of the bf16-rounded outputs; f32 partials per 256-voxel tile,
 *   f64 atomics across tiles) -- feed it to osk_groupnorm_apply_ndhwc_bf16.  The waves of a tile add their partials with
 *   LDS float atomics: the result is reproducible to f32 rounding of a tile's partial (~1e-7 relative), not bit for bit.
 * Only the large-tile kernels carry this epilogue: returns OSK_EUNSUPPORTED -- and launches NOTHING -- unless
 * Cin % 128 == 0, Cout >= 128, Cout % 32 == 0, Cout / gn_groups in {4, 8, 16}, >= 256 output voxels, out 16-byte aligned and
 * (B == 1 or To*Ho*Wo % 256 == 0); the caller then runs the plain conv + osk_groupnorm_stats_ndhwc_bf16. */
int osk_causal_conv3d_gn_ndhwc_bf16(const void* x, int B, int T, int H, int W, int Cin,
                                    const void* w, int64_t w_row_stride, const float* bias, int Cout, int ksize,
                                    int stride_t, int stride_h, int stride_w, int up_t, int up_hw,
                                    const void* res, void* out, int To, int Ho, int Wo,
                                    double* gn_sums, int gn_groups, void* stream);

/* ---- GroupNorm statistics: sums[b][g] = (sum, sum of squares) in f64 over S voxels x C/G channels.
 * first half of nn.GroupNorm(32, C, eps=1e-6) (unet_causal_3d_blocks.py:216,218; vae.py:115,229; diffusers
 * Attention.group_norm).  x bf16 [B, S, C]; sums f64 [B, G, 2] (zeroed inside, on the stream).  C in
 * {32..512} power of two, C/G <= 16. */
int osk_groupnorm_stats_ndhwc_bf16(const void* x, int B, int64_t S, int C, int G, double* sums, void* stream);

/* ---- GroupNorm apply (+ SiLU): y = bf16((x - mean_g) * rstd_g * gamma_c + beta_c); out = silu ? bf16(y*sigmoid(y)) : y
 * second half of nn.GroupNorm + nonlinearity (unet_causal_3d_blocks.py:250-254).  gamma/beta f32 [C]. */
int osk_groupnorm_apply_ndhwc_bf16(const void* x, const double* sums, const float* gamma, const float* beta,
                                   void* out, int B, int64_t S, int C, int G, float eps, int silu, void* stream);

/* ---- GroupNorm + SiLU folded into the CONSUMING convolution (round 4): the norm1 -> SiLU -> conv1 and norm2 -> SiLU -> conv2
 * chains of ResnetBlockCausal3D.forward (unet_causal_3d_blocks.py:247-256) without the normalised tensor ever existing in HBM.
 * osk_groupnorm_table_f32: sums (osk_groupnorm_stats_ndhwc_bf16 / a producing conv's gn_sums) + gamma, beta ->
 *   table f32 [B][C / 8][16]: per 8-channel chunk 8 scales a_c = rstd_g gamma_c, then 8 shifts d_c = beta_c - mean_g a_c
 *   (exactly the per-channel constants osk_groupnorm_apply_ndhwc_bf16 derives; 16-byte aligned).
 * osk_causal_conv3d_gnin_ndhwc_bf16: osk_causal_conv3d_ndhwc_bf16 (no upsample) of
 *   bf16(silu(bf16(x * a + d))) -- the same rounding points as apply(silu = 1) followed by the conv -- reading x itself: the
 *   sliding-window kernels fetch their halo pieces into registers, transform them in the MFMA shadows and write them to LDS.
 *   gn_sums != NULL: + the fused statistics of the OUTPUT as osk_causal_conv3d_gn_ndhwc_bf16 (gn_groups groups; zeroed by the caller).
 *   Shapes: 3 x 3 x 3, stride 1, Cin % 128 == 0, whole 16 x 16 output bricks, Cout >= 256 or (Cout == 128 and T >= 2); anything
 *   else returns OSK_EUNSUPPORTED and launches NOTHING -- the caller then runs apply + the plain conv. */
int osk_groupnorm_table_f32(const double* sums, const float* gamma, const float* beta, float* table, int B, int64_t S,
                            int C, int G, float eps, void* stream);
int osk_causal_conv3d_gnin_ndhwc_bf16(const void* x, const float* gn_in_table, int B, int T, int H, int W, int Cin,
                                      const void* w, int64_t w_row_stride, const float* bias, int Cout, int ksize,
                                      int stride_t, int stride_h, int stride_w, const void* res, void* out, int To, int Ho,
                                      int Wo, double* gn_sums, int gn_groups, void* stream);

/* ---- frame-causal masked softmax over attention score rows (mid-block attention, one head of dim C):
 *   probs[i][j] = softmax_j(scale * scores[i][j])  over keys j < (i / keys_per_frame + 1) * keys_per_frame,
 *   0 elsewhere (columns up to ld_probs are written, so probs is a K-padded GEMM operand).
 * replaces prepare_causal_attention_mask + the softmax inside diffusers Attention / SDPA
 * (unet_causal_3d_blocks.py:52-60,345-351).  keys_per_frame = 0: no mask.  scores f32, probs bf16. */
int osk_masked_softmax_f32_bf16(const float* scores, int64_t ld_scores, void* probs, int64_t ld_probs,
                                int Sq, int Sk, int keys_per_frame, float scale, void* stream);

/* ---- flash attention of the causal VAE's mid block: ONE head of dimension 512, frame-causal mask
 *   out[i] = softmax_j(scale * q[i].k[j]) v[j]  over keys j with  j / keys_per_frame <= i / keys_per_frame   (+ bias_v)
 * replaces diffusers Attention + prepare_causal_attention_mask inside UNetMidBlockCausal3D
 * (hunyuan_vae/unet_causal_3d_blocks.py:52-60,312-351): no S x S mask, no S x S score matrix in memory.
 * q, k, out: bf16 [B, S, 512] views (batch / row strides in elements, rows contiguous); vt: bf16 V^T [B][512][ld]
 * (natural key order, ld >= round_up(S, 32), ZERO beyond S) -- the layout the V projection GEMM V^T = W_v x^T writes;
 * bias_v f32 [512] | NULL is added after normalisation (softmax rows sum to one).  keys_per_frame = 0: no mask. */
int osk_attention_hd512_fwd_bf16(const void* q, int64_t q_batch_stride, int64_t q_row_stride, const void* k,
                                 int64_t k_batch_stride, int64_t k_row_stride, const void* vt, int64_t vt_batch_stride,
                                 int64_t vt_row_stride, const float* bias_v, void* out, int64_t out_batch_stride,
                                 int64_t out_row_stride, int B, int S, int keys_per_frame, float scale, void* stream);
/* ... with a caller-owned workspace (>= osk_attention_hd512_workspace_bytes(B, S), 16-byte aligned): the launch is only
 * ceil(S / 128) x B x 2 workgroups and its time is the key-tile chain of the last frame's query blocks, so with a workspace every
 * chain is cut into up to 4 runs of key tiles that run side by side (partial rows in f32 + LSE, combined by a merge kernel).
 * Same result up to the f32 rounding of the combination; NULL / too small a workspace = the single-chain launch. */
int osk_attention_hd512_fwd_ws_bf16(const void* q, int64_t q_batch_stride, int64_t q_row_stride, const void* k,
                                    int64_t k_batch_stride, int64_t k_row_stride, const void* vt, int64_t vt_batch_stride,
                                    int64_t vt_row_stride, const float* bias_v, void* out, int64_t out_batch_stride,
                                    int64_t out_row_stride, int B, int S, int keys_per_frame, float scale, void* workspace,
                                    int64_t workspace_bytes, void* stream);
int64_t osk_attention_hd512_workspace_bytes(int B, int S);

/* ---- tile cross-fade of the tiled VAE paths, in place in b (f32 math, one rounding):
 *   b[o, e, i] = a[o, Da - extent + e, i] * (1 - e/extent) + b[o, e, i] * (e/extent),  e < extent
 * replaces blend_v / blend_h / blend_t (hunyuan_vae/autoencoder_kl_causal_3d.py:360-382; a Python loop of `extent`
 * slice assignments there).  a, b: contiguous bf16 viewed as [outer, Da | Db, inner] along the blended axis
 * (NCTHW tiles: blend_t -> outer B*C, inner H*W; blend_v -> outer B*C*T, inner W; blend_h -> outer B*C*T*H, inner 1). */
int osk_blend_bf16(const void* a, void* b, int64_t outer, int Da, int Db, int extent, int64_t inner, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* OSK_H */
