/* lap_hip.h — C ABI of liblap_hip.so, the gfx950 (MI355X / CDNA4) kernel library
 * behind lap_amd's LAP-3B train step and policy-serving path.
 *
 * The reference (lihzha/lap) has no FFI: its hot path is a jax.jit program
 * (scripts/train.py:532-537 -> TrainingStepRunner.__call__ 329-419 ->
 * LAP.compute_loss src/lap/models/lap.py:380-602 -> gemma.Module
 * src/lap/models/backbones/gemma.py:396-531).  Each entry point below replaces
 * the XLA-fused computation of the cited reference lines.
 *
 * Conventions
 *   - every pointer is a DEVICE pointer unless stated otherwise; the caller
 *     owns all buffers, kernels never allocate;
 *   - `stream` is a hipStream_t passed as void* (NULL = default stream);
 *   - return value 0 = launched, LAP_ERR_ARG (1001) = rejected argument,
 *     anything else = hipError_t of the failed launch;
 *   - bf16 = IEEE bfloat16 bits, f32 = float, i32 = int32_t;
 *   - thread-safe for distinct streams; no global state.
 */
#ifndef LAP_HIP_H_
#define LAP_HIP_H_
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define LAP_ABI_VERSION 1
int lap_abi_version(void);

/* ---------------------------------------------------------------- GEMM ---- */
#define LAP_GEMM_OUT_F32  1   /* C is f32 (default bf16) */
#define LAP_GEMM_ACCUM    2   /* C += result (f32 output only) */
#define LAP_GEMM_GELU     4   /* tanh-GELU after bias (bf16 output only) */
#define LAP_GEMM_BIAS_F32 8   /* bias is f32 (default bf16) */
#define LAP_GEMM_PARTIALS 16  /* lap_gemm_bf16_ex only: leave the ksplit raw f32 partial products [ksplit][M][N] in `scratch`
                                 and skip the reduce/epilogue kernel (C unused) — consumed by the lap_fused_reduce_* kernels */

/* C[M,N] = epi(alpha * opA . opB): bf16 operands, f32 MFMA accumulate.
 * a_kc=1: A is [M][lda] (k contiguous); a_kc=0: A is [K][lda] (m contiguous).
 * b_kc=1: B is [N][ldb] (k contiguous); b_kc=0: B is [K][ldb] (n contiguous).
 * epi(v) = [gelu](v + bias[n]) + residual[m][n], optionally accumulated into C.
 * Replaces every jnp.dot / einsum projection of gemma.py:188-202,279-285,
 * 303-319 (lora.Einsum / FeedForward), Embedder.decode gemma.py:153-154 and the
 * Flax Dense / MultiHeadDotProductAttention projections of SigLIP
 * (siglip_gemma3.py:59-114), plus their transposes in the backward pass. */
int lap_gemm_bf16(const void* A, const void* B, void* C, const void* bias, const void* residual,
                  int M, int N, int K, int lda, int ldb, int ldc, int ldr, float alpha,
                  int a_kc, int b_kc, int flags, void* stream);

/* Ablation knob for kernel studies (tools/gemm_ablate.py); rejected (LAP_ERR_ARG) unless the library was built with
 * LAP_GEMM_EXPERIMENTAL=1.  bits: 1 = tile 10 issues no LDS-DMA inside its k-loop, 2 = no MFMA (results are garbage). */
int lap_gemm_set_debug(int bits);

/* Same with explicit scheduling knobs: tile = -1 (heuristic: 5 or 6) | 0 (128x128x64, 4 waves) | 1 (256x128x64, 3 stages)
 * | 2 (256x256x64, 8 waves) | 3 (256x256x32, 4 stages) | 4 (128x128x32, 4 stages) | 5 (256x256x64, 16 waves: production,
 * with the tail split of a poorly filled last round and the LDS-staged bf16 epilogue) | 6 (128x128x64, 8 waves, 2 blocks
 * per CU: production for small shapes) | 7 (256x128x64, 16 waves) | 8, 9 (8- / 16-wave ping-pong probes) | 10 (256x256x64,
 * 8 waves, software-pipelined with two fragment register sets; K % 64 == 0; stands in for 5 in production on the forward layout)
 * | 11 (probe: 10 with two barriers per k-tile) | 12 (256x256, 8 waves as two ping-pong groups over a ring of four 32-deep k-half
 * slots; K % 64 == 0; production for the data- / weight-gradient layouts) | 13 (probe: ping-pong for the forward layout).  All tiles give
 * the same bits for the same ksplit (k is accumulated in the same order).  ksplit > 1 splits K over grid.y:
 *  - with `scratch` (>= ksplit*M*N*4 bytes): f32 partials + a reduce/epilogue kernel, any output / epilogue
 *    (deterministic; used for GEMMs with too few output tiles to fill 256 CUs: skinny-M serving, small weights);
 *  - without scratch: f32 atomics onto C (requires LAP_GEMM_OUT_F32 | LAP_GEMM_ACCUM, no bias / residual). */
int lap_gemm_bf16_ex(const void* A, const void* B, void* C, const void* bias, const void* residual,
                     int M, int N, int K, int lda, int ldb, int ldc, int ldr, float alpha,
                     int a_kc, int b_kc, int flags, int tile, int ksplit, void* scratch, long long scratch_bytes,
                     void* stream);

/* Small exact-f32 GEMM (VALU, k-ordered fmaf chain):
 * C[M,N] = alpha * opA . opB (+ bias[n]) (+ C if accum).  Same layout flags as
 * above.  Used where the reference computes in float32: the SigLIP stem conv
 * (siglip_gemma3.py:401-408), nnx.Linear action_in_proj / time_mlp_* /
 * action_out_proj (lap.py:52-62,298) and their gradients. */
int lap_gemm_f32(const float* A, const float* B, float* C, const float* bias,
                 int M, int N, int K, int lda, int ldb, int ldc, float alpha,
                 int a_kc, int b_kc, int accum, void* stream);

/* ------------------------------------------------------- normalisation ---- */
/* RMSNorm / adaptive RMSNorm forward (gemma.py:113-131).
 * x,y: bf16 [rows][D]; scale: f32 [D] (plain) or NULL; mod: bf16
 * [rows/rows_per_sample][mod_ld >= 3*D] = (scale|shift|gate) from Dense(cond)
 * (adaptive; a column slice of the all-layers modulation matrix) or NULL.
 * rstd: f32 [rows] out (saved for backward), may be NULL. */
int lap_rmsnorm_fwd(const void* x, const float* scale, const void* mod, void* y, float* rstd,
                    int rows, int D, int rows_per_sample, int mod_ld, float eps, void* stream);
/* Backward: dx (bf16, overwritten or accumulated into when accum_dx) and
 * either dscale f32[D] or dmod f32 [B][dmod_ld] scale/shift thirds — both
 * accumulated with f32 atomics (caller zeroes); the gate third is untouched. */
int lap_rmsnorm_bwd(const void* x, const float* scale, const void* mod, const float* rstd, const void* dy,
                    void* dx, float* dscale, float* dmod,
                    int rows, int D, int rows_per_sample, int mod_ld, int dmod_ld, int accum_dx, void* stream);

/* LayerNorm (Flax nn.LayerNorm, eps 1e-6, f32 statistics; siglip_gemma3.py:93,104,167).
 * x,y bf16 [rows][D]; gamma,beta f32 [D]; mean,rstd f32 [rows] saved. */
int lap_layernorm_fwd(const void* x, const float* gamma, const float* beta, void* y, float* mean, float* rstd,
                      int rows, int D, float eps, void* stream);
int lap_layernorm_bwd(const void* x, const float* gamma, const float* mean, const float* rstd, const void* dy,
                      void* dx, float* dgamma, float* dbeta, int rows, int D, int accum_dx, void* stream);

/* ----------------------------------------------------------- elementwise -- */
/* RoPE + q-scale + head split (gemma.py:215-218,548-564).
 * qkv: bf16 [B*T_seg][(NH+2)*HD] (q heads | k | v) of one expert stream;
 * pos: i32 [B][T_total] joint positions, this stream's tokens start at
 * seg_off inside each sample.  Writes q [B*T_seg][NH*HD] (rotated, rounded to
 * bf16, then * q_scale in bf16), k [B*T_seg][HD] rotated, v [B*T_seg][HD]. */
int lap_rope_split_fwd(const void* qkv, const int32_t* pos, void* q, void* k, void* v,
                       int B, int T_seg, int T_total, int seg_off, int NH, int HD, float q_scale, void* stream);
/* Inverse: dq,dk,dv -> dqkv (same layouts). */
int lap_rope_split_bwd(const void* dq, const void* dk, const void* dv, const int32_t* pos, void* dqkv,
                       int B, int T_seg, int T_total, int seg_off, int NH, int HD, float q_scale, void* stream);

/* GeGLU (gemma.py:308-312): gu bf16 [rows][2*H] (gate | up), act = gelu_tanh(gate) * up, bf16 [rows][H]. */
int lap_geglu_fwd(const void* gu, void* act, int rows, int H, void* stream);
int lap_geglu_bwd(const void* gu, const void* dact, void* dgu, int rows, int H, void* stream);
/* tanh-GELU (SigLIP MlpBlock, siglip_gemma3.py:76): y = gelu(x); bwd dx = dy * gelu'(x). bf16. */
int lap_gelu_fwd(const void* x, void* y, long long n, void* stream);
int lap_gelu_bwd(const void* x, const void* dy, void* dx, long long n, void* stream);

/* Embedding gather (gemma.py:148-151,448): out[r] = bf16(table[tok[r]] * scale), table f32.
 * `table` holds vocabulary rows [row_lo, row_hi) (the whole table, or this rank's FSDP shard); tokens
 * outside the range produce zero rows.  Rows are written at dst row
 * (r / T) * dst_rows_per_sample + dst_off + r % T (prefix assembly, lap.py:150-166). */
int lap_embed_gather(const float* table, const int32_t* tok, void* out, int rows, int T, int D,
                     int dst_rows_per_sample, int dst_off, float scale, int row_lo, int row_hi, void* stream);
/* Backward: dtable[tok[r]] += scale * dout[row(r)] (f32 atomics). */
int lap_embed_scatter_add(float* dtable, const int32_t* tok, const void* dout, int rows, int T, int D,
                          int src_rows_per_sample, int src_off, float scale, void* stream);

/* y = x + u * gate[sample] (gemma.py:577-583), bf16; gate: bf16 [B][ldg] slice (third of mod) or NULL (plain add). */
int lap_gated_residual_fwd(const void* x, const void* u, const void* gate, void* y,
                           int rows, int D, int rows_per_sample, int ldg, void* stream);
/* du = dy * gate; dgate[b] (f32 [B][ldg_out]) = sum_rows dy*u. */
int lap_gated_residual_bwd(const void* dy, const void* u, const void* gate, void* du, float* dgate,
                           int rows, int D, int rows_per_sample, int ldg, int ldg_out, void* stream);

/* Column sums: out[c] += sum_r x[r][c]  (bias gradients). x bf16 [rows][ld], out f32 [cols] (caller zeroes). */
int lap_colsum_bf16(const void* x, float* out, int rows, int cols, int ld, void* stream);
/* Same for f32 input. */
int lap_colsum_f32(const float* x, float* out, int rows, int cols, int ld, void* stream);

/* Generic casts / copies / fills. */
int lap_cast_f32_to_bf16(const float* x, void* y, long long n, void* stream);
/* hi = bf16(x), lo = bf16(x - hi) for x f32 [rows][cols] (row stride ld); outputs [rows][ld_out], ld_out % 8 == 0, zero
   padded beyond `cols`.  Feeding (hi, lo) pairs to the bf16 MFMA GEMM with f32 accumulation reproduces the f32 SigLIP
   stem (siglip_gemma3.py:398-408: conv + bias in f32) to ~2^-17 relative at MFMA speed. */
int lap_split_f32_hilo(const float* x, int rows, int cols, int ld, void* hi, void* lo, int ld_out, void* stream);
int lap_cast_bf16_to_f32(const void* x, float* y, long long n, void* stream);
int lap_add_bf16(const void* a, const void* b, void* y, long long n, void* stream);
/* Strided 2-D copy of bf16 rows: dst[r*ldd + c] = src[r*lds + c], c < cols. */
int lap_copy2d_bf16(const void* src, void* dst, int rows, int cols, int lds, int ldd, void* stream);
/* Row re-blocking: dst row (r / T) * dst_rps + dst_off + r % T  <-  src row (r / T) * src_rps + src_off + r % T. */
int lap_copy_rows_bf16(const void* src, void* dst, int rows, int T, int D,
                       int src_rps, int src_off, int dst_rps, int dst_off, int accumulate, void* stream);

/* SigLIP stem helpers (siglip_gemma3.py:398-418): NHWC f32 image -> patch matrix f32 [B*GH*GW][P*P*C]
 * (row-major over (ph, pw, c) to match a Flax conv kernel reshaped [P,P,C,W]). */
int lap_im2col_patch(const float* img, float* out, int B, int H, int W, int C, int P, void* stream);
/* Train-time augmentation of f32 [B][H][W][3] images in [-1, 1] (model_adapter.py:118-151: augmax RandomCrop 95 % -> Resize
 * -> Rotate +-5 deg -> ColorJitter 0.2/0.2/0.2, [UPSTREAM-RECALL]) as one gather + per-pixel colour pass.  params f32 [B][12]:
 * crop offset x, y (pixels), crop width, height, cos, sin of the rotation, brightness, contrast, saturation in [-0.2, 0.2],
 * skip flag (non-zero: copy the sample unchanged), 2 unused.  out must not alias img. */
int lap_augment_images(const float* img, float* out, const float* params, int B, int H, int W, void* stream);
/* y[bf16] = x[f32] + posemb[f32][r % T]  (stem output + learned posemb, cast to bf16). */
int lap_add_posemb_cast(const float* x, const float* pos, void* y, int rows, int T, int D, void* stream);
/* Backward of the above: dx f32 = dy; dpos[t] += sum_b dy[b,t]. */
int lap_add_posemb_cast_bwd(const void* dy, float* dx, float* dpos, int rows, int T, int D, void* stream);

/* ------------------------------------------------------------ attention --- */
/* Joint multi-segment masked attention (gemma.py:234-272; Flax MHA for SigLIP).
 * Queries / keys of one sample are the concatenation of up to 2 segments
 * (expert streams, or KV-cache prefix + new suffix).  Segment s of Q/O holds
 * bf16 [B][len_s][NH*HD]; of K/V holds [B][len_s][NKV*HD] (NKV in {1, NH}).
 * Mask: allowed(i,j) = (qinfo[b][i] >> 24) & (kinfo[b][j] >> 24) != 0  &&
 *                      (kinfo[b][j] & 0xffffff) <= (qinfo[b][i] & 0xffffff);
 * NULL infos = no mask.  This encodes make_attn_mask + the LAP prefix/action
 * block structure (lap.py:303-364) without materialising [B,T,T].
 * logits = scale * q.k (Gemma: q already carries 1/sqrt(HD) from the RoPE kernel, scale = 1).
 * lse: f32 [B][NH][Tq] out. */
typedef struct {
  const void* q[2]; void* o[2];
  const void* k[2]; const void* v[2];
  int q_len[2]; int k_len[2];
  int q_rs[2]; int kv_rs[2]; int o_rs[2];   /* row strides in elements; 0 = packed (NH*HD / NKV*HD) */
  const int32_t* qinfo; const int32_t* kinfo;
  float* lse;
  float* scratch;                             /* f32 scratch for nsplit > 1: nsplit * B * Tq * NH * (HD + 1) floats */
  long long scratch_floats;
  float scale;                                /* logits = scale * q.k (1.0 when q is pre-scaled) */
  int nsplit;                                 /* > 1: key tiles are split over nsplit blocks per query tile (few-query
                                                 launches such as the batch-1 denoise step) and combined by a 2nd kernel */
  int B, NH, NKV, HD;
} lap_attn_fwd_args;
int lap_attention_fwd(const lap_attn_fwd_args* a, void* stream);

typedef struct {
  const void* q[2]; const void* o[2]; const void* d_o[2];
  const void* k[2]; const void* v[2];
  void* dq[2]; void* dk[2]; void* dv[2];
  int q_len[2]; int k_len[2];
  int q_rs[2]; int kv_rs[2]; int o_rs[2];   /* q/dq, k/v/dk/dv, o/dO row strides; 0 = packed */
  const int32_t* qinfo; const int32_t* kinfo;
  const float* lse;
  float* delta;              /* scratch f32 [B][NH][Tq] */
  float* scratch;            /* optional f32 scratch for hsplit > 1: 2 * hsplit * B * Tk * NKV * HD floats */
  long long scratch_floats;
  float scale;
  int hsplit;                /* > 1 (must divide NH/NKV): the query heads sharing a kv head are spread over hsplit
                                blocks per key tile; partial dK/dV are reduced by a second small kernel */
  int B, NH, NKV, HD;
  int stop_q1_to_k0;         /* stop_action_to_vlm_grad (gemma.py:242-269): no dK/dV from segment-1 queries into segment-0 keys */
} lap_attn_bwd_args;
int lap_attention_bwd(const lap_attn_bwd_args* a, void* stream);
/* Kernel-selection knob for tests / benchmarks (process-wide, not thread safe): -1 automatic (default);
   0 = generic padded-LDS kernels also for HD = 256; 1 = the HD = 256 LDS-DMA kernels.  Both agree to bf16
   rounding of the probabilities. */
int lap_attention_set_variant(int variant);

/* ---------------------------------------------------------------- fp8 GEMM ---- */
/* BASELINE.json config 5 (north_star: "fp8 weights/activations, CDNA4 fp8 MFMA"; the reference itself has bf16 only,
 * lap_config.py:24).  OCP e4m3, per-tensor scaling: q = e4m3(clamp(x * s, +-448)), s = 448 / amax(x).
 * lap_gemm_fp8: C[M,N] = alpha / (s_a * s_b) * A8[M,K] . B8[N,K]^T (+ residual) — both operands K-contiguous fp8 bytes,
 *   f32 accumulation on v_mfma_scale_f32_16x16x128_f8f6f4 (block scales 2^0); K % 128 == 0; flags: LAP_GEMM_OUT_F32,
 *   LAP_GEMM_ACCUM.  scale_a / scale_b: DEVICE scalars (as written by lap_quantize_fp8*), so no host round trip.
 *   Stands in for lap_gemm_bf16 on the forward (x8 . W8^T) and data-gradient (dy8 . W8t^T) products of the Gemma
 *   projections (gemma.py:188-202,279-285,303-319); weight gradients stay bf16. */
int lap_gemm_fp8(const void* A8, const void* B8, void* C, const void* residual, const float* scale_a, const float* scale_b,
                 int M, int N, int K, int lda, int ldb, int ldc, int ldr, float alpha, int flags, void* stream);
/* amax[0] = max(amax[0], max |x|) over a bf16 [rows][cols] view (row stride ld); the caller zeroes amax first. */
int lap_amax_bf16(const void* x, long long rows, int cols, long long ld, float* amax, void* stream);
/* out8[r][c] = e4m3(x[r][c] * 448 / amax[0]); the scale used is written to scale[0]. */
int lap_quantize_fp8(const void* x, long long rows, int cols, long long ld, const float* amax, void* out8, long long ldo,
                     float* scale, void* stream);
/* Weights Wt[rows = out][cols = in] (contiguous): the fp8 copy out8 [rows][cols] for the forward AND its transpose
 * out8_t [cols][rows] for the data gradient (so that both products are the K-contiguous NT form). */
int lap_quantize_fp8_weight(const void* w, int rows, int cols, const float* amax, void* out8, void* out8_t, float* scale,
                            void* stream);

/* ---------------------------------------- skinny-M fused projections (batch-1 denoise step) -- */
/* One launch per projection of a denoise-step layer (lap.py:634-667 -> gemma.py:336-387, action-expert stream only,
 * M = B x action_horizon rows, 64 per block tile): the 8 waves of a block split K and reduce through LDS in wave order
 * (no cross-block partial slabs, deterministic), weights are streamed once, and the neighbouring ops run in the
 * prologue / epilogue.  Shapes: D = 1024 (Gemma-300M width), HD = 256, K in {1024, 2048, 4096}; other shapes are
 * rejected with LAP_ERR_ARG and the caller uses lap_gemm_bf16_ex(LAP_GEMM_PARTIALS) + lap_fused_reduce_* instead.
 *
 * lap_serve_qkv_rope: h = adaRMS(x; mod) (gemma.py:113-131; mod = scale|shift|gate bf16 [.., 3D], row stride mod_ld,
 *   0 = one row for all samples) -> qkv = h @ wqkv^T (wqkv [(NH+2)*HD][D], gemma.py:188-202) -> RoPE with the hoisted
 *   sin / cos table + q scale + head split (gemma.py:215-218,548-564) -> q [M][NH*HD], k [M][HD], v [M][HD]. */
int lap_serve_qkv_rope(const void* x, const void* mod, int mod_ld, int rows_per_sample, const void* wqkv,
                       const float* rope_table, void* q, void* k, void* v, int M, int D, int NH, int HD, float q_scale,
                       float eps, void* stream);
/* lap_serve_gate_up: h = adaRMS(x; mod) -> [gate | up] = h @ wgu^T (wgu [2H][D]) -> act = bf16(gelu(gate)) * up
 *   (gemma.py:303-312) -> act [M][H]. */
int lap_serve_gate_up(const void* x, const void* mod, int mod_ld, int rows_per_sample, const void* wgu, void* act, int M,
                      int D, int H, float eps, void* stream);
/* lap_serve_proj_residual: out = x + bf16((a @ w^T) * gate[sample]) (attention out-projection / FFN down-projection with
 *   the gated residual of gemma.py:577-583; gate NULL: plain add).  a [M][K], w [N][K], x / out [M][N]. */
int lap_serve_proj_residual(const void* a, const void* w, const void* x, const void* gate, int gate_ld, int rows_per_sample,
                            void* out, int M, int N, int K, void* stream);
/* Head and tail of one Euler step (lap.py:655-672): tokens = bf16(x_t @ w_in^T + b_in) (action_in_proj, f32 nnx.Linear,
 *   lap.py:52; w_in [D][action_dim]);  and  v_t = adaRMS_final(x; mod) @ w_out^T + b_out (f32, w_out [action_dim][D]),
 *   x_t += dt * v_t, v_t optionally stored. */
/* Tuning knob (process-wide, not thread safe): feature tiles per block of lap_serve_proj_residual (1 default, 2). */
int lap_serve_set_variant(int feature_tiles);
int lap_serve_embed_actions(const float* x_t, const float* w_in, const float* b_in, void* tokens, int rows, int action_dim,
                            int D, void* stream);
int lap_serve_final_euler(const void* x, const void* mod, int mod_ld, int rows_per_sample, const float* w_out,
                          const float* b_out, float* x_t, float* v_t, int rows, int D, int action_dim, float dt, float eps,
                          void* stream);

/* ---------------------------------------- fused consumers of GEMM partials (serving) -- */
/* partials: f32 [ksplit][rows][cols] from lap_gemm_bf16_ex(LAP_GEMM_PARTIALS).  Each kernel sums the slabs, rounds to
 * bf16 like the GEMM would have, and applies the ops that follow the projection in gemma.py:336-387. */
/* table (optional, else NULL): sin / cos per (row, frequency), f32 [B*T_seg][HD/2][2] from lap_rope_table — the positions
 * of the action tokens are the same for all 10 x 18 projections of a denoise loop, so the transcendental work is hoisted. */
int lap_rope_table(const int32_t* pos, float* table, int B, int T_seg, int T_total, int seg_off, int HD, void* stream);
int lap_fused_reduce_rope_split(const float* partials, int ksplit, const int32_t* pos, const float* table, void* q, void* k,
                                void* v, int B, int T_seg, int T_total, int seg_off, int NH, int HD, float q_scale, void* stream);
int lap_fused_reduce_geglu(const float* partials, int ksplit, void* act, int rows, int H, void* stream);
/* xn = x + bf16(y * gate[sample]) (gate NULL: plain add); h = adaptive RMSNorm(xn; mod) when mod != NULL. */
int lap_fused_reduce_residual_norm(const float* partials, int ksplit, const void* x, const void* gate, int ldg,
                                   const void* mod, int mod_ld, void* xn, void* h, int rows, int D,
                                   int rows_per_sample, float eps, void* stream);

/* ----------------------------------------------------------------- loss --- */
/* Vocab-chunked cross entropy (lap.py:221-260). logits: f32 [rows][ldl] chunk covering vocab
 * [v0, v0+vc); state m,l (running max / sum-exp), tl (target logit) f32 [rows]. */
int lap_ce_chunk_update(const float* logits, int ldl, const int32_t* target, float* m, float* l, float* tl,
                        int rows, int v0, int vc, void* stream);
/* dlogits chunk (bf16 [rows][ldd]) = w[r] * (softmax - onehot), softmax from (m,l). */
int lap_ce_chunk_grad(const float* logits, int ldl, const int32_t* target, const float* m, const float* l,
                      const float* w, void* dlogits, int ldd, int rows, int v0, int vc, void* stream);

/* ------------------------------------------------------------ optimizer --- */
/* sumsq[0] += sum x^2 (f32 atomics; caller zeroes). */
int lap_sumsq_f32(const float* x, long long n, float* sumsq, void* stream);
/* Greedy decode step of sample_tokens (lap.py:719-724): out[r] = argmax_c x[r][c], lowest index among ties (jnp.argmax). */
int lap_argmax_rows_f32(const float* x, int rows, int n, int ld, int* out, void* stream);
/* Fused clip-by-global-norm + AdamW + EMA + bf16 weight refresh (train.py:363-396).
 * scalars (device f32[8]): [0]=sum of squared grads (global), [1]=lr, [2]=bias_corr1, [3]=bias_corr2,
 * [4]=ema_decay, [5]=ema_enabled (0/1).  clip = min(1, max_norm / (sqrt(s0)+1e-6)).
 * p,m,v,ema f32 [n]; g f32 [n]; p16 bf16 [n] out (may be NULL); ema may be NULL.  n must be even (8-byte accesses;
 * unit buffers are padded).  The kernel is held to 32 VGPRs so that its waves are co-resident with the 256x256 GEMM. */
int lap_adamw_ema(float* p, float* m, float* v, float* ema, const float* g, void* p16, long long n,
                  const float* scalars, float b1, float b2, float eps, float wd, float max_norm, void* stream);

/* ------------------------------------------------------- flow matching ---- */
/* x_t = t*eps + (1-t)*a ; u_t = eps - a  (lap.py:193-197). all f32 [B][n_per]; t f32 [B]. */
int lap_fm_mix(const float* noise, const float* actions, const float* t, float* x_t, float* u_t,
               int B, int n_per, void* stream);
/* posemb_sincos(t, D, min_period, max_period) -> f32 [B][D] (openpi pi0.posemb_sincos, UPSTREAM-RECALL). */
int lap_posemb_sincos(const float* t, float* out, int B, int D, float min_period, float max_period, void* stream);
/* swish: y = x * sigmoid(x), f32; bwd dx = dy * (s + x s (1-s)). */
int lap_swish_fwd(const float* x, float* y, long long n, void* stream);
int lap_swish_bwd(const float* x, const float* dy, float* dx, long long n, void* stream);
/* MSE (lap.py:298-299): per_sample[b] = mean_{n_per}((v-u)^2); dv = coef[b] * 2 (v-u) / n_per. f32. */
int lap_mse_fwd_bwd(const float* v, const float* u, const float* coef, float* per_sample, float* dv,
                    int B, int n_per, void* stream);
/* x += dt * v (Euler step, lap.py:667). f32. */
int lap_axpy_f32(float* x, const float* v, float dt, long long n, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* LAP_HIP_H_ */
