/* libfxhip -- C ABI of the MI355X (gfx950) kernels behind the flexynesis training hot path.
 *
 * The reference (BIMSBbioinfo/flexynesis v1.1.14) has NO native/FFI interface: its hot path is a
 * chain of ATen ops dispatched from nn.Modules.  Each entry point below
 * therefore replaces a group of reference ops; the reference file:line it stands in for is cited.
 * The reference-side binding a maintainer would add (ctypes) is shown in INTEGRATION.md.
 *
 * Conventions
 *  - every function returns int: 0 = ok, FX_EINVAL (-22) = bad argument, other <0 = -(hipError_t);
 *    fx_last_error_string() describes the last failure of the calling thread.  Never throws/aborts.
 *  - all tensor pointers are DEVICE pointers owned by the caller (e.g. the torch allocator); the
 *    library never allocates or frees device memory.  Scratch is passed in (see *_workspace_* queries).
 *  - every launcher takes the hipStream_t LAST, is asynchronous and stream-ordered; there is no global
 *    mutable state besides the thread-local error string, so calls are re-entrant across host threads.
 *  - matrices are fp32 row-major with an explicit leading dimension (elements); weights use torch's
 *    [out_features, in_features] layout; "B" is the batch (rows = samples, reference data.py:550).
 *  - randomness is EITHER a supplied tensor (parity mode) OR a Philox4x32-10 (seed, offset) pair; when a
 *    step-control block ``ctrl`` is passed too, its step counter is folded into the offset on the device.
 */
#ifndef FXHIP_H
#define FXHIP_H

#ifdef __cplusplus
extern "C" {
#endif

typedef struct ihipStream_t* fx_stream_t; /* == hipStream_t */

#define FX_EINVAL (-22)

/* step-control block: 64 floats in device memory, see flexynesis_amd/csrc/fx_common.h (enum FxCtrl) */
#define FX_CTRL_FLOATS 64
#define FX_CTRL_STEP 0
#define FX_CTRL_LR 1
#define FX_CTRL_CLIP_COEF 4
#define FX_CTRL_GNORM 5
#define FX_CTRL_BATCH_CURSOR 8

/* activation codes for fx_bn_act_* */
#define FX_ACT_NONE 0
#define FX_ACT_LEAKY 1 /* LeakyReLU(0.2) applied BEFORE BatchNorm (Encoder/Decoder, modules.py:25-34) */
#define FX_ACT_RELU 2  /* ReLU (+ optional dropout) applied AFTER BatchNorm (MLP, modules.py:145-148) */

/* GEMM layouts for fx_gemm_f32 */
#define FX_GEMM_NT 0 /* C[M,N] = A[M,K] . B[N,K]^T  : nn.Linear forward (modules.py:54,101,145,149)     */
#define FX_GEMM_NN 1 /* C[M,N] = A[M,K] . B[K,N]    : autograd grad-input  (dX = dY . W)                 */
#define FX_GEMM_TN 2 /* C[M,N] = A[K,M]^T . B[K,N]  : autograd grad-weight (dW = dY^T . X)               */

const char* fx_last_error_string(void);
int fx_version(void);
/* hipRuntimeGetVersion() of the process (major * 10^7 + minor * 10^5 + patch), -1 if unavailable */
int fx_hip_runtime_version(void);
/* SHA-256 (hex) of the HIP sources the library was built from; csrc/build.py rebuilds when it no longer matches */
const char* fx_source_hash(void);

/* ---- data: replaces MultiOmicDataset.__getitem__ + default_collate + per-batch H2D
 *      (data.py:980-995, main.py:289-298).  dst[r,:] = src[idx[r],:]; idx int64 on device.
 *      ctrl_cursor != NULL: idx is a [n_batches, cursor_stride] table and the row block
 *      idx + ctrl[FX_CTRL_BATCH_CURSOR] * cursor_stride is used (lets a captured hipGraph walk the epoch's permutation without new arguments). */
int fx_gather_rows(float* dst, const float* src, const long* idx, int n_rows, int n_cols, long ld_src, long ld_dst,
                   const float* ctrl_cursor, long cursor_stride, fx_stream_t stream);

/* ---- Linear layers (nn.Linear fwd / autograd bwd: modules.py:125-130,145,149; direct_pred.py:87-93) */
long fx_gemm_workspace_bytes(int M, int N, int K);
int fx_gemm_f32(int layout, float* C, const float* A, const float* B, const float* bias, int M, int N, int K, long lda,
                long ldb, long ldc, int accumulate, void* workspace, long workspace_bytes, fx_stream_t stream);
/* fused dW = dY^T.X  ->  clip  ->  Adam on W,m,v  (loss.backward + clip_grad_norm_ + Adam.step for one
 * weight: main.py:216-217, direct_pred.py:143).  dW is never materialised. */
int fx_linear_dw_adam_f32(float* W, float* adam_m, float* adam_v, const float* dY, const float* X, int batch, int n_out,
                          int k_in, long lddy, long ldx, long ldw, const float* ctrl, fx_stream_t stream);
int fx_colsum(float* out, const float* x, int B, int C, long ldx, fx_stream_t stream); /* bias gradients */

/* ---- split-bf16 ("bf16x3") variants for the WIDE layers: x ~= hi + lo in bf16, products evaluated as
 *      hi*hi + hi*lo + lo*hi on the bf16 MFMA with fp32 accumulation (3/16 of the fp32-MFMA cost, ~2^-16
 *      per-product error).  hi/lo buffers are bf16 (2 bytes/element), zero padded to a multiple of 32 in
 *      the contracted dimension.  Two layouts:
 *        K-BLOCKED (fx_split_bf16, fx_gather_split hi/lo; the X operand of fx_linear_fwd_bf16x3*):
 *          [ceil(C/32)][rows_padded][32], element (r, c) at ((c/32)*rows_padded + r)*32 + c%32, rows_padded a
 *          multiple of 128 with zero padding rows -- a 128-row K-step tile is 8 KB contiguous (64 full cache
 *          lines instead of 128 half lines; measured +22 % on the wide forward);
 *        TRANSPOSED row-major (fx_split_bf16_t, fx_gather_split hiT/loT; operands of fx_linear_dw_adam_bf16x3):
 *          [C][ld] with the contracted (batch) dimension contiguous, ld a multiple of 8.
 *      Same reference ops as fx_gemm_f32 NT / fx_linear_dw_adam_f32 above.  In fx_linear_fwd_bf16x3* the `ldx`
 *      argument is rows_padded of the K-blocked X. */
int fx_split_bf16(void* hi, void* lo, const float* x, int R, int C, long ldx, long rows_padded, fx_stream_t stream);
int fx_split_bf16_t(void* hiT, void* loT, const float* x, int R, int C, long ldx, long ldo, fx_stream_t stream);
long fx_linear_fwd_bf16x3_workspace_bytes(int M, int N, int K);
int fx_linear_fwd_bf16x3(float* Y, const void* xhi, const void* xlo, const float* W, const float* bias, int M, int N, int K,
                         long ldx, long ldw, long ldy, void* workspace, long workspace_bytes, fx_stream_t stream);
/* fx_linear_fwd_bf16x3 with explicit tuning (A/B experiments; same contraction): splitk 0 = auto, wave_cols 0|4 = 128x128
 * tile / 2 = 128x64, no_mt 0 = stacked rows (M > 128) and M <= 64 on the register-fragment kernel, 1 = one workgroup per 128-row M tile for every M,
 * 2 = the first multi-M-tile kernel (both operands staged through registers), 3 = the second (X by LDS-DMA), 4 = the register-fragment kernel also
 * for 65..128 rows; nt 1 = non-temporal W loads.  The library
 * reads no environment variables: callers that want A/B switches pass them here. */
int fx_linear_fwd_bf16x3_ex(float* Y, const void* xhi, const void* xlo, const float* W, const float* bias, int M, int N, int K,
                            long ldx, long ldw, long ldy, void* workspace, long workspace_bytes, int splitk, int wave_cols,
                            int no_mt, int nt, fx_stream_t stream);
/* dX[M,N] = dY[M,K] . W[K,N] with W = the weight [out = K, in = N] as stored: autograd's data-gradient mm through a WIDE
 * layer (the decoders' FC_output of supervised_vae / CrossModalPred, modules.py:89,101).  dY as a K-blocked split. */
int fx_linear_bwd_x_bf16x3(float* dX, const void* dyhi, const void* dylo, const float* W, int M, int N, int K,
                           long dy_rows_padded, long ldw, long lddx, void* workspace, long workspace_bytes, fx_stream_t stream);
int fx_linear_dw_adam_bf16x3(float* W, float* adam_m, float* adam_v, const void* dyT_hi, const void* dyT_lo,
                             const void* xT_hi, const void* xT_lo, int batch_padded, int n_out, int k_in, long lddy,
                             long ldx, long ldw, const float* ctrl, fx_stream_t stream);
/* fx_linear_dw_adam_bf16x3 with explicit tuning: tile_order 0 = auto, 1 = linear, 2 = XCD-partitioned / L2-blocked (results are
 * bit-identical); wave_cols 0|4 = 128x128 tile, 2 = 128x64; plain_loads 1 = default cache policy for W / m / v. */
int fx_linear_dw_adam_bf16x3_ex(float* W, float* adam_m, float* adam_v, const void* dyT_hi, const void* dyT_lo,
                                const void* xT_hi, const void* xT_lo, int batch_padded, int n_out, int k_in, long lddy,
                                long ldx, long ldw, const float* ctrl, int tile_order, int wave_cols, int plain_loads,
                                fx_stream_t stream);
/* The same optimiser step for one wide weight PLUS the wide-layer forward of the FOLLOWING training step
 * (nn.Linear of modules.py:145 / :28 applied to the next batch, main.py:289-298's next DataLoader item): while a tile of
 * W_new is in registers it is multiplied into the next batch xn (K-blocked split of <= 384 rows -- the triplet network
 * stacks anchor / positive / negative --, xn_rows_padded = 128, 256 or 384), so the next step does not read W again (24
 * instead of 28 bytes per parameter and step).  The partial sums land in
 * y_slabs [fx_linear_dw_adam_fwd_bf16x3_slabs(n_out, k_in)][next_rows][n_out]; fx_reduce_slabs adds them (+ bias) in a
 * fixed order.  W, m, v results are bit-identical to fx_linear_dw_adam_bf16x3.
 * flags: bit 0 = non-temporal W / m / v accesses; bits 1-2 = workgroup mapping (0 auto = 3 for large weights / 2 for stacked
 * batches, 1 plain, 2 row blocks interleaved over the XCDs with all their runs, 3 contiguous row-block ranges per XCD); bits 8-15 = runs per row block (0 = the library's plan,
 * fx_linear_dw_adam_fwd_bf16x3_slabs_ex(); the slab buffer must hold at least that many); bits 17-18 = issue-priority scheme of the two
 * workgroups a CU holds (0 = they take turns per tile, 1 = none, 2 / 3 = variants).
 * k_in % 4 == 0; descriptors are rebased per row block, so the weight itself may exceed 4 GiB. */
int fx_linear_dw_adam_fwd_bf16x3_slabs(int n_out, int k_in);
/* ... for explicit flags and padded batch: the row blocks are cut into runs so that all 512 workgroup slots of the chip are
 * taken (the last row blocks get one run more), per XCD with flag mapping 3; every slab of the buffer passed to the launch is
 * written (zeros where a row block has fewer runs), so a consumer may add up all of them */
int fx_linear_dw_adam_fwd_bf16x3_slabs_ex(int n_out, int k_in, int batch_padded, int flags);
int fx_linear_dw_adam_fwd_bf16x3(float* W, float* adam_m, float* adam_v, const void* dyT_hi, const void* dyT_lo,
                                 const void* xT_hi, const void* xT_lo, int batch_padded, int n_out, int k_in, long lddy,
                                 long ldx, long ldw, const float* ctrl, const void* xn_hi, const void* xn_lo,
                                 long xn_rows_padded, int next_rows, float* y_slabs, long y_slabs_bytes, int flags,
                                 fx_stream_t stream);
/* Plain-bf16 throughput mode of the wide kernels (the reference trains under torch.set_float32_matmul_precision("medium"),
 * main.py:24): pass NULL for every `lo` operand of fx_linear_fwd_bf16x3* / fx_linear_bwd_x_bf16x3 / fx_linear_dw_adam_bf16x3* /
 * fx_linear_dw_adam_fwd_bf16x3* and the contraction is hi . hi only (one bf16 MFMA product, fp32 accumulate; W is rounded to bf16
 * in-kernel), a third of the matrix work; master weights, Adam moments and everything narrow stay fp32.  The `lo` operands of one
 * call come together or not at all.
 * The fused step OUT OF PLACE: W / m / v are only read, the updated values go to W_dst / m_dst / v_dst (same shape and pitch) in
 * other memory partitions; the caller swaps sources and destinations for the next step. */
int fx_linear_dw_adam_fwd_bf16x3_oop(float* W_dst, float* m_dst, float* v_dst, const float* W, const float* adam_m,
                                     const float* adam_v, const void* dyT_hi, const void* dyT_lo, const void* xT_hi,
                                     const void* xT_lo, int batch_padded, int n_out, int k_in, long lddy, long ldx, long ldw,
                                     const float* ctrl, const void* xn_hi, const void* xn_lo, long xn_rows_padded, int next_rows,
                                     float* y_slabs, long y_slabs_bytes, int flags, fx_stream_t stream);
/* Y[M,N] = sum_z slabs[z][M][N] (+ bias[N]), summed in slab order (deterministic) */
int fx_reduce_slabs(float* Y, const float* slabs, const float* bias, int M, int N, long ldy, int n_slabs, long slab_stride,
                    fx_stream_t stream);
/* the same sum for MANY slabs of a SMALL output (the VAE's dz: 83 slabs of [128, 64]): 8 lanes per four outputs, each over a contiguous range
 * of slabs, combined in range order -- deterministic, but not fx_reduce_slabs' serial order.  N, ldy, slab_stride % 4 == 0, 16-byte bases. */
int fx_reduce_slabs_par(float* Y, const float* slabs, const float* bias, int M, int N, long ldy, int n_slabs, long slab_stride,
                        fx_stream_t stream);

/* ---- placement probe: one pass of fx_linear_dw_adam_fwd_bf16x3's W / m / v traffic pattern without the GEMMs, contents unchanged.  Its
 *      time depends on where the arrays landed in physical memory (400-494 us for the three [5000, 20000] arrays on one MI355X; 134-141 vs
 *      161-168 us for one array on its own: m = v = NULL): the host allocates candidate arrays one by one, times this with HIP events and
 *      keeps the fast ones (flexynesis_amd.engine.ParamStore, FX_PLACEMENT_TRIES). */
int fx_placement_probe(float* W, float* m, float* v, int n_out, int k_in, long ldw, fx_stream_t stream);
/* its out-of-place twin: reads W / m / v, writes the same values to W_dst / m_dst / v_dst (m, v and their destinations may be NULL
 * together): rates a (source partitions, destination partitions) layout for fx_linear_dw_adam_fwd_bf16x3_oop */
int fx_placement_probe_oop(float* W_dst, float* m_dst, float* v_dst, const float* W, const float* m, const float* v, int n_out,
                           int k_in, long ldw, fx_stream_t stream);

/* ---- launch-fusion variants (same reference ops, fewer passes): GEMMs that leave their split-K partial sums in
 *      slabs [splitk][M][N] for a consumer that reduces them in its own pass; Gram-norm Hadamard sum straight
 *      from two slab sets; cohort gather fused with the bf16 splits the wide-layer kernels consume. */
int fx_gemm_splitk(int M, int N, int K);
int fx_gemm_f32_slabs(int layout, float* slabs, const float* A, const float* B, int M, int N, int K, long lda, long ldb,
                      fx_stream_t stream);
int fx_linear_fwd_bf16x3_splitk(int M, int N, int K);
/* the slab-producing forward with the output tile chosen by the caller: wave_cols 0 | 4 = 128 x 128, 8 = 128 x 256 (at most 128 rows);
 * fx_linear_fwd_bf16x3_splitk_ex gives the number of slabs that choice writes */
int fx_linear_fwd_bf16x3_splitk_ex(int M, int N, int K, int wave_cols);
int fx_linear_fwd_bf16x3_slabs_ex(float* slabs, long slabs_bytes, const void* xhi, const void* xlo, const float* W, int M, int N,
                                  int K, long ldx, long ldw, int wave_cols, fx_stream_t stream);
int fx_linear_fwd_bf16x3_slabs(float* slabs, long slabs_bytes, const void* xhi, const void* xlo, const float* W, int M,
                               int N, int K, long ldx, long ldw, fx_stream_t stream);
int fx_bn_act_fwd_slabs(float* out, float* x_out, const float* slabs, int nslabs, long slab_stride, const float* lin_bias,
                        const float* gamma, const float* beta, float* running_mean, float* running_var, float* save_mean,
                        float* save_invstd, const float* mask, int B, int C, long ldx, long ldo, int pre_act, int post_act,
                        int train, float drop_p, unsigned long long seed, unsigned long long offset, const float* ctrl,
                        fx_stream_t stream);
int fx_gram_hadamard_blocks(long n);
int fx_gram_hadamard(double* slots, const float* slabs_x, int nslabs_x, const float* slabs_d, int nslabs_d, long n,
                     fx_stream_t stream);
/* fx_gather_split for up to 4 cohort layers (modalities) in ONE launch: the same index table, cursor and row count for all of
 * them (MultiOmicDataset.__getitem__ + default_collate over every layer, data.py:1015-1027). */
typedef struct fx_gather_split_desc {
  float* x; void* hi; void* lo; void* hiT; void* loT; const float* src; int n_cols; long ld_src, ldx, ldo, ldt;
} fx_gather_split_desc;
int fx_gather_split_group(const fx_gather_split_desc* descs, int n_layers, const long* idx, int n_rows, const float* ctrl_cursor,
                          long cursor_stride, fx_stream_t stream);
/* The batch-only factor X X^T of the Gram-identity gradient norm (clip_grad_norm_, main.py:216) for up to 4 modalities in
 * one launch: slabs[i][z] [R, R] = partial sum over K slice z < fx_gram_kb_slices(k_in[i]) of X_i X_i^T, from the K-blocked
 * split (hi, lo) of X_i [R <= 128 rows, k_in[i] columns] (fx_gather_split / fx_split_bf16 layout); three bf16 MFMA
 * products per term, fp32 accumulation.  fx_reduce_group adds the slabs in order. */
int fx_gram_kb_slices(int k_in);
int fx_gram_kb_group(const void* const* hi, const void* const* lo, float* const* slabs, const int* k_in, int n_modalities, int R,
                     fx_stream_t stream);
/* y[i][k] = sum_z slabs[i][z][k] (+ bias[i][k % bias_n[i]]) for k < len[i], slabs in order (deterministic), up to 4 jobs in
 * one launch.  len and bias_n multiples of 4, arrays 16-byte aligned; bias / bias_n may be NULL. */
int fx_reduce_group(float* const* y, const float* const* slabs, const float* const* bias, const long* len, const int* n_slabs,
                    const int* bias_n, int n_jobs, fx_stream_t stream);
/* Forward of an encoder TAIL for every modality in one launch (grid: column blocks x modalities), replacing per modality
 * fx_reduce_slabs -> fx_bn_act_fwd -> fx_gemm_f32 (+ its split-K reduce): the tail is "wide Linear output -> BatchNorm block
 * -> one or two small Linears" (MLP encoder, modules.py:145-149: BatchNorm1d -> ReLU -> Dropout -> layer_out; VAE encoder,
 * modules.py:25-41,47-56: LeakyReLU -> BatchNorm1d -> FC_mean, FC_var).  Per modality: x = sum of the wide Linear's partial-sum
 * slabs (+ lin_bias), in slab order (slabs NULL: x already holds the Linear's output); batch statistics (train) or running
 * statistics; out = block output (saved for the backward); part[k][blk] = out[:, blk's 64 columns] . W_k[:, those columns]^T,
 * k < n_up, one [B, L_k] partial product per column block (fx_enc_tail_blocks(H) of them), consumed by fx_fusion_fwd.
 * B <= 128, H % 4 == 0, L_k <= 128, every array 16-byte aligned.  The dropout stream is fx_bn_act_fwd's (seed, offset, ctrl). */
typedef struct fx_enc_tail_desc {
  const float* slabs; long slab_stride; const float* lin_bias;   /* [n_slabs][B][H] (stride in floats), bias [H] or NULL */
  float* x; float* out;                                            /* [B, H] contiguous */
  const float* gamma; const float* beta; float* running_mean; float* running_var; float* save_mean; float* save_invstd;
  const float* mask;                                               /* supplied dropout mask [B, H] or NULL */
  const float* W[2]; float* part[2];                               /* following Linears [L_k, H]; partial products */
  unsigned long long seed, offset;
  int n_slabs, H, n_up, L[2];
} fx_enc_tail_desc;
int fx_enc_tail_blocks(int H);
int fx_enc_tail_fwd(const fx_enc_tail_desc* descs, int n_modalities, int B, int pre_act, int post_act, int train, float drop_p,
                    const float* ctrl, fx_stream_t stream);
/* ecat[B, sum widths] = for every layer i the ordered sum of parts[i] [n_parts[i]][B][widths[i]] (+ part_bias[i]) -- the
 * concatenated encoder outputs of direct_pred.py:118-121 from fx_enc_tail_fwd's partial products -- and, when Wf is given,
 * emb = ecat Wf^T + bf (the fusion Linear, direct_pred.py:122-124; Wf [L, sum widths] contiguous, L <= 128).  One workgroup
 * per 4 rows.  widths: any positive integers (a latent size is one: config.py:8), sum <= 512; part i is [n_parts[i]][B][pitch] with
 * pitch = widths[i] rounded up to 4 (what fx_enc_tail_fwd writes for a following Linear whose rows are allocated up to that multiple,
 * zeros) and part_bias[i] readable up to that pitch; ecat rows hold the layers back to back (layer i from column sum of the widths
 * before it: unaligned when those are not multiples of 4 -- scalar stores then).  ecat may be NULL when only emb is wanted. */
int fx_fusion_fwd(float* emb, long ldemb, float* ecat, long ldecat, const float* const* parts, const int* n_parts,
                  const float* const* part_bias, const int* widths, int n_layers, const float* Wf, const float* bf, int B, int L,
                  fx_stream_t stream);
/* Two such layers over the same B rows in ONE launch (the VAE's mean = FC_mean(mcat), log_var = FC_log_var(vcat): supervised_vae.py:172-176):
 * every per-layer argument doubled (index 0 / 1); parts / n_parts / part_bias hold the first layer's n_layers entries, then the second's. */
int fx_fusion_fwd_pair(float* const* emb, const long* ldemb, float* const* ecat, const long* ldecat, const float* const* parts,
                       const int* n_parts, const float* const* part_bias, const int* widths, int n_layers, const float* const* Wf,
                       const float* const* bf, int B, int L, fx_stream_t stream);
/* x[r, :] = src[idx[r], :] (x optional) plus both splits in one pass (MultiOmicDataset.__getitem__ + default_collate,
 * data.py:1015-1027, and the operand preparation of the wide kernels).  hi / lo / hiT / loT 16-byte aligned, ldt % 8 == 0 and
 * >= n_rows rounded up to 32; 16-byte accesses when n_cols, ld_src and ldx are multiples of 4 (scalar otherwise). */
int fx_gather_split(float* x, void* hi, void* lo, void* hiT, void* loT, const float* src, const long* idx, int n_rows,
                    int n_cols, long ld_src, long ldx, long ldo /* rows_padded of the K-blocked hi/lo */, long ldt,
                    const float* ctrl_cursor, long cursor_stride, fx_stream_t stream);

/* ---- all supervisor heads in one launch each way.  A head is the reference MLP(latent -> hidden -> n_out)
 *      (modules.py:106-150: Linear -> BatchNorm1d -> ReLU -> Dropout -> Linear) applied to the [B, L] embedding
 *      (direct_pred.py:126-131, supervised_vae.py:190-196, triplet_encoder.py:160-164) and its autograd backward.
 *      `heads` is a HOST array of descriptors holding DEVICE pointers (copied into the kernel arguments at launch).
 *      Limits: B <= 128, L <= 128, hidden <= 32, n_out <= 32, <= 8 heads; larger shapes use the per-layer entry
 *      points above.  fx_heads_fwd: y1 (layer_1 output), a1 (block output), save_*, running stats (train) and
 *      `out`; dropout from `mask` (supplied 0/1 [B,hidden]) or Philox(seed, offset, step from ctrl), the same
 *      stream as fx_bn_act_fwd.  fx_heads_bwd: from `dout` (the loss kernels' output gradient) every parameter
 *      gradient of every head and dx (+)= sum over heads of the embedding gradient, heads summed in order. */
typedef struct fx_head_desc {
  const float* W1; const float* b1;            /* layer_1.weight [hidden, L], .bias [hidden] */
  const float* gamma; const float* beta;       /* batchnorm.weight / .bias [hidden] */
  float* running_mean; float* running_var;     /* updated in train mode */
  const float* W2; const float* b2;            /* layer_out.weight [n_out, hidden], .bias [n_out] or NULL */
  float* y1; float* a1;                        /* saved for backward: [B, hidden] each */
  float* save_mean; float* save_invstd;        /* [hidden] */
  float* out;                                  /* [B, n_out] */
  const float* mask;                           /* supplied dropout mask or NULL */
  const float* dout;                           /* backward input [B, n_out] */
  float* gW1; float* gb1; float* ggamma; float* gbeta; float* gW2; float* gb2;   /* gradients (gb2 NULL if no bias) */
  unsigned long long seed, offset;             /* Philox stream of this head's dropout */
  int hidden, n_out;
} fx_head_desc;
int fx_heads_fwd(const fx_head_desc* heads, int n_heads, const float* x, long ldx, int B, int L, int train, float drop_p,
                 const float* ctrl, fx_stream_t stream);
/* dx_scratch (optional): n_heads * B * L floats + one 32-bit counter, ZERO-FILLED ONCE by the caller and then owned by
 * these launches.  With it every head's share of dx is computed by its own workgroup and the last one to finish adds
 * the shares in head order (deterministic); without it one workgroup walks the heads one after the other. */
int fx_heads_bwd(const fx_head_desc* heads, int n_heads, const float* x, long ldx, float* dx, long lddx, int dx_accumulate,
                 int B, int L, float drop_p, void* dx_scratch, fx_stream_t stream);
/* The whole supervisor part of a TRAINING step in one launch (one workgroup per head): forward of every head, its loss value
 * and output gradient (kinds[i]: 0 masked MSE, 1 masked softmax-CE -- direct_pred.py:146-190 --, 2 Cox partial likelihood --
 * modules.py:265-305, labels = events, durations[i] --, scaled by exp(-logvars[i][0]) when given), backward of every head
 * (parameter gradients; dx (+)= the heads' embedding gradients added in head order) and the uncertainty-weighted total over
 * ALL n_terms named losses of the model (direct_pred.py:192-223; terms written by earlier launches -- triplet, MMD -- are read
 * from their slots; term_dlogvars / epoch_acc as fx_total_loss).  Replaces fx_heads_fwd -> per-head loss kernels ->
 * fx_total_loss -> fx_heads_bwd.  heads[i].dout receives the output gradient.  dx_scratch (REQUIRED): fx_heads_step_scratch_floats(n_heads,
 * B, L) floats, zero-filled once by the caller -- the shares of dx + arrival counter as fx_heads_bwd, then one block per head in which the
 * head's second workgroup (the weight-gradient role) keeps its own copies of the saved tensors: every shared output has ONE writer. */
long fx_heads_step_scratch_floats(int n_heads, int B, int L);
int fx_heads_step(const fx_head_desc* heads, int n_heads, const int* kinds, const float* const* labels,
                  const float* const* durations, const float* const* logvars, float* const* losses, const float* x, long ldx,
                  float* dx, long lddx, int dx_accumulate, int B, int L, float drop_p, const float* ctrl, void* dx_scratch,
                  int n_terms, int weighted, const float* const* term_losses, const float* const* term_logvars,
                  float* const* term_dlogvars, float* total_out, float* epoch_acc, fx_stream_t stream);

/* ---- the whole backward of an encoder tail "wide Linear -> BatchNorm block -> 1 or 2 small Linears" in one launch
 *      (MLP encoder, modules.py:145-149; VAE encoder, modules.py:25-41,47-56): autograd's mm for the small Linears'
 *      data gradient, the BatchNorm(+ReLU+Dropout | LeakyReLU) backward, mm + sum for the small Linears' weight / bias
 *      gradients, plus what the wide layer's fused optimiser kernel needs: dY as a transposed split-bf16 operand
 *      (fx_split_bf16_t layout) and this layer's share of the squared gradient norm <dY dY^T, gram_x> with
 *      gram_x = X X^T [B, B] (one double per workgroup into slots[0 .. fx_block_bwd_blocks(C))).
 *      dE / ldE / W / gW / gb / L are HOST arrays of n_up (1 or 2) entries: upstream gradient [B, L_k], weight
 *      [L_k, C], its gradient [L_k, C], bias gradient [L_k] or NULL.  Optional outputs may be NULL (dy, dyT_*, gram_x).
 *      B <= 128 rows per launch: one BatchNorm pass.  Stacked passes through the same weights (the triplet network's anchor /
 *      positive / negative rows) take one launch per pass on row-offset pointers, with accumulate = 1 from the second on
 *      (gW, gb, dgamma, dbeta, dbias are then added to; dy / dyT_* rows or columns of the pass are written). */
int fx_block_bwd_blocks(int C);
int fx_block_bwd(const float* const* dE, const long* ldE, const float* const* W, float* const* gW, float* const* gb,
                 const int* L, int n_up, const float* x, const float* out, const float* gamma, const float* save_mean,
                 const float* save_invstd, float* dgamma, float* dbeta, float* dbias, float* dy, void* dyT_hi, void* dyT_lo,
                 long ldt, const float* gram_x, double* slots, int B, int C, long ldx, long ldo, int pre_act, int post_act,
                 float drop_p, int accumulate, fx_stream_t stream);

/* fx_block_bwd that ALSO writes dY in fx_split_bf16's K-blocked layout [ceil(C / 32)][kb_rows][32] (hi, lo; rows kb_row0 .. kb_row0 +
 * round32(B) - 1 of it): a workgroup's 32 columns are exactly one K-block, so the split the stacked-rows Gram product dY dY^T needs
 * (triplet_encoder.py:298-330 under clip_grad_norm_, main.py:216) costs sixteen more bytes per thread here instead of a launch of
 * fx_split_bf16 over dY behind the last pass.  dy_kb_hi NULL = fx_block_bwd. */
int fx_block_bwd_ex(const float* const* dE, const long* ldE, const float* const* W, float* const* gW, float* const* gb,
                    const int* L, int n_up, const float* x, const float* out, const float* gamma, const float* save_mean,
                    const float* save_invstd, float* dgamma, float* dbeta, float* dbias, float* dy, void* dyT_hi, void* dyT_lo,
                    long ldt, const float* gram_x, double* slots, int B, int C, long ldx, long ldo, int pre_act, int post_act,
                    float drop_p, int accumulate, void* dy_kb_hi, void* dy_kb_lo, long kb_rows, int kb_row0, fx_stream_t stream);

/* fx_block_bwd for up to 4 independent encoder tails (one per modality) in ONE launch (grid: column blocks x tails); every
 * field has the meaning of the fx_block_bwd argument of the same name.  B, the activations and drop_p are common. */
typedef struct fx_block_bwd_desc {
  const float* dE[2]; long ldE[2]; const float* W[2]; float* gW[2]; float* gb[2]; int L[2]; int n_up;
  const float* x; const float* out; const float* gamma; const float* save_mean; const float* save_invstd;
  float* dgamma; float* dbeta; float* dbias; float* dy; void* dyT_hi; void* dyT_lo; long ldt;
  const float* gram_x; double* slots; int C; long ldx, ldo; int accumulate;
} fx_block_bwd_desc;
int fx_block_bwd_group(const fx_block_bwd_desc* descs, int n_tails, int B, int pre_act, int post_act, float drop_p,
                       fx_stream_t stream);

/* ---- small dense layers on the critical chain (fusion layer direct_pred.py:87-93,121-124; VAE FC_mean / FC_log_var
 *      supervised_vae.py:104-107,172-176): one forward launch, and ONE backward launch for the data, weight and bias
 *      gradients (autograd's two mm + sum).  W / gW contiguous [O, K]; R, O, K <= 4096; fixed summation order. */
int fx_small_linear_fwd(float* y, const float* x, const float* W, const float* b, int R, int O, int K, long ldx, long ldy,
                        fx_stream_t stream);
int fx_small_linear_bwd(float* dx, float* gW, float* gb, const float* dy, const float* x, const float* W, int R, int O, int K,
                        long ldx, long lddy, long lddx, int dx_accumulate, fx_stream_t stream);
/* Up to 4 such layers' backward in ONE launch (the VAE's FC_mean and FC_log_var: supervised_vae.py:172-176 backwards).  A job's upstream
 * gradient is dy[r, o] * dy_mul[r, o] when dy_mul is given: d log_var = dz * eps of the reparameterisation (:190-200) needs no launch. */
typedef struct fx_small_linear_job {
  float* dx; float* gW; float* gb; const float* dy; const float* dy_mul; const float* x; const float* W;
  int R, O, K, dx_accumulate;
  long ldx, lddy, ldmul, lddx;
} fx_small_linear_job;
int fx_small_linear_bwd_group(const void* jobs /* fx_small_linear_job[n], host */, int n, fx_stream_t stream);

/* ---- BatchNorm1d (+LeakyReLU before | +ReLU+Dropout after), train & eval (modules.py:25-34,145-148) */
int fx_bn_act_fwd(float* out, const float* x, const float* gamma, const float* beta, float* running_mean,
                  float* running_var, float* save_mean, float* save_invstd, const float* mask, float* mask_out, int B,
                  int C, long ldx, long ldo, int pre_act, int post_act, int train, float drop_p, unsigned long long seed,
                  unsigned long long offset, const float* ctrl, fx_stream_t stream);
int fx_bn_act_bwd(float* dx, float* dgamma, float* dbeta, float* dbias, const float* dout, const float* x,
                  const float* out, const float* gamma, const float* save_mean, const float* save_invstd, int B, int C,
                  long ldx, long ldo, long lddo, long lddx, int pre_act, int post_act, float drop_p, int accumulate,
                  fx_stream_t stream);

/* Eval-mode backward of the two blocks with respect to the block INPUT only (BatchNorm = affine map of its running
 * statistics, no dropout): the input-gradient path of the attributions the reference computes through Captum
 * (direct_pred.py:418-590: IntegratedGradients / GradientShap of forward_target under model.eval()). */
int fx_bn_eval_bwd(float* dx, const float* dout, const float* x, const float* out, const float* gamma, const float* running_var,
                   int B, int C, long ldx, long ldo, long lddo, long lddx, int pre_act, int post_act, fx_stream_t stream);

/* ---- elementwise pieces of supervised_vae (supervised_vae.py:187-200; modules.py:101-102) */
int fx_sigmoid(float* y, const float* x, long n, fx_stream_t stream);
int fx_sigmoid_bwd(float* dx, const float* dy, const float* y, long n, fx_stream_t stream);     /* dx = dy * y * (1 - y) */
/* y[r, :] = softmax(x[r, :]) for a small [B, C] logit matrix: predict()'s class probabilities (direct_pred.py:330-337) */
int fx_softmax_rows(float* y, const float* x, int B, int C, long ldx, long ldy, fx_stream_t stream);
int fx_reparam(float* z, float* eps_out, const float* mean, const float* log_var, const float* eps, long n,
               unsigned long long seed, unsigned long long offset, const float* ctrl, fx_stream_t stream);
int fx_mul(float* y, const float* a, const float* b, long n, fx_stream_t stream);
int fx_fill_normal(float* y, long n, unsigned long long seed, unsigned long long offset, const float* ctrl,
                   fx_stream_t stream);

/* ---- losses: raw value -> loss_out[0]; gradient of exp(-log_var)*extra_scale*loss -> grad output
 *      (log_var may be NULL = unweighted).  compute_loss direct_pred.py:146-190; cox modules.py:265-305;
 *      triplet triplet_encoder.py:178-194; MMD supervised_vae.py:494-550; total direct_pred.py:192-223 */
int fx_mse_masked(float* loss_out, float* dyhat, const float* yhat, const float* y, int B, long ld, long ldd,
                  const float* logvar, float extra_scale, fx_stream_t stream);
int fx_ce_masked(float* loss_out, float* dlogits, const float* logits, const float* y, int B, int C, long ld, long ldd,
                 const float* logvar, float extra_scale, fx_stream_t stream);
int fx_cox_ph(float* loss_out, float* dout, const float* out, const float* durations, const float* events, int B, long ld,
              long ldd, const float* logvar, float extra_scale, fx_stream_t stream);
int fx_triplet(float* loss_out, float* da, float* dp, float* dn, const float* a, const float* p, const float* n, int B,
               int L, long ld, float margin, const float* logvar, float extra_scale, fx_stream_t stream);
long fx_mmd_workspace_floats(int P, int B);
int fx_mmd_rows(float* row_sums, float* dz, const float* prior, const float* z, int P, int B, int L, long ldz,
                const float* logvar, float extra_scale, fx_stream_t stream);
/* the same with 16-byte row loads where the layout allows (L % 4 == 0, ldz % 4 == 0, aligned bases; same sums in the same order: bit-identical
 * results; 19 -> 9 us on an idle chip) and dz_overwrite: dz = term instead of dz += term (no zero-fill launch in front). */
int fx_mmd_rows_ex(float* row_sums, float* dz, const float* prior, const float* z, int P, int B, int L, long ldz,
                   const float* logvar, float extra_scale, int dz_overwrite, fx_stream_t stream);
int fx_recon_blocks(long n);
int fx_recon_sigmoid(float* partial, float* dlogits, float* xhat_out, const float* logits, const float* x, long n,
                     const float* logvar, float extra_scale, fx_stream_t stream);
/* The reconstruction term as the epilogue of the decoder's FC_output forward (reference supervised_vae.py:301-313 with modules.py:99-103):
 * ordered sum of the split-K slabs fx_linear_fwd_bf16x3_slabs left + bias -> sigmoid -> per-block sums of (x_hat - x)^2 (n_partial =
 * fx_recon_sigmoid_slabs_blocks) -> dlogits [B, F] fp32 and / or their K-blocked bf16 split (fx_split_bf16 layout, rows_padded rows):
 * one launch for fx_reduce_slabs + fx_recon_sigmoid + fx_split_bf16.  F % 4 == 0. */
int fx_recon_sigmoid_slabs_blocks(int B, int F);
int fx_recon_sigmoid_slabs(float* partial, float* dlogits, void* hi, void* lo, const float* slabs, int nslabs, long slab_stride,
                           const float* bias, const float* x, int B, int F, long rows_padded, const float* logvar, float extra_scale,
                           fx_stream_t stream);
int fx_mmd_finalize(float* loss_acc, const float* row_sums, int P, int B, const float* recon_partial, int n_partial,
                    float n_recon, float extra_scale, int accumulate, fx_stream_t stream);
int fx_total_loss(float* total_out, int n, int weighted, const float* const* losses, const float* const* logvars,
                  float* const* dlogvars, float* epoch_acc, fx_stream_t stream);

/* ---- optimiser: Lightning's clip_grad_norm_(1.0) + torch.optim.Adam(lr) (main.py:212-225, direct_pred.py:143) */
int fx_step_begin(float* ctrl, float lr, int n_batches, fx_stream_t stream);
/* x[0..n) *= scale[0], the scalar read on the device (nothing is written when it is 1): applies the upstream gradient
 * of loss.backward() to the materialised gradients without a host synchronisation */
int fx_scale_by(float* x, long n, const float* scale, fx_stream_t stream);
/* y[0..n) = value (zero_grad of an accumulator that no kernel of the step overwrites first) */
int fx_fill(float* y, long n, float value, fx_stream_t stream);
/* dst[0..n) = src[0..n): plain 16-byte-per-lane streaming copy (n % 4 == 0, 16-byte aligned).  bench.py times it to
 * quote the box's own read+write HBM rate next to the dominant kernel's roofline fraction. */
int fx_stream_copy(float* dst, const float* src, long n, fx_stream_t stream);
int fx_sumsq_blocks(long n);
int fx_sumsq(double* slots, const float* x, long n, fx_stream_t stream);
int fx_hadamard_sum(double* slot, const float* g1, const float* g2, long n, fx_stream_t stream);
int fx_clip_finalize(float* ctrl, const double* slots, int n_slots, float max_norm, fx_stream_t stream);
int fx_adam_flat(float* p, const float* g, float* m, float* v, long n, const float* ctrl,
                 const float* trainable /* optional 0/1 per element: 0 = requires_grad False, skipped (main.py:530-539,562-566) */,
                 fx_stream_t stream);

/* fx_clip_finalize and fx_adam_flat in ONE launch (every workgroup reduces the norm slots in the same fixed order; workgroup 0
 * publishes FXC_GNORM / FXC_CLIP_COEF in ctrl for the wide-weight kernels that follow): same results, one launch less */
int fx_adam_flat_clip(float* p, const float* g, float* m, float* v, long n, float* ctrl, const float* trainable,
                      const double* slots, int n_slots, float max_norm, fx_stream_t stream);

/* ---- device-side ingest of a raw omics matrix.  x is the matrix as the reference's HDF5
 *      importer reads it: contiguous [n_samples, n_features], samples as rows (h5_dataloader.py:88-116,
 *      csv_to_h5.py:13-21); dtype FX_IN_F32 (HDF5 path) or FX_IN_F64 (what pd.read_csv yields).  The four stages are
 *      the per-matrix arithmetic of DataImporter.cleanup_data / transform_data / normalize_data (data.py:360-452,
 *      519-545); the F-length decisions in between (variance quantile, NaN fraction, harmonize) stay on the host.
 *      "value" below = the entry after NaN -> med[column] (when med != NULL) and log1p (when log1p != 0).
 *      fx_col_moments: per column over the listed rows (rows == NULL: all n_rows), skipping NaN: count, mean and the
 *        sum of squared deviations m2 in fp64 (df.var(axis=1) = m2/(count-1), data.py:373; StandardScaler.fit =
 *        m2/count, data.py:527).  ws: fx_col_moments_workspace_bytes(n_rows, F) bytes, 8-byte aligned.
 *      fx_col_median: med_out[c] (an F-length vector indexed by column) = median of the non-NaN entries of each listed
 *        column c over all N rows (data.py:413-415); an even count averages the two middle values, fp32 input rounds
 *        the result to fp32 (the frame's dtype); an all-NaN column gives NaN.  Unlisted entries are left untouched.
 *      fx_row_moments: per sample, the ddof=1 variance over the listed columns after imputation (df.std(axis=0)**2,
 *        data.py:426).
 *      fx_ingest_transform: out[i, j] = fp32((value(x[rows[i], cols[j]]) - mean[j]) / scale[j]); mean/scale NULL = no
 *        scaling; rows / cols NULL = identity.  fp32 input rounds to fp32 after the subtraction as sklearn's in-place
 *        float32 transform does; fp64 input rounds once at the end (data.py:533-540,549). */
#define FX_IN_F32 0
#define FX_IN_F64 1
int fx_col_moments_chunks(int n_rows, int F);
long fx_col_moments_workspace_bytes(int n_rows, int F);
int fx_col_moments(const void* x, int dtype, long ldx, int n_rows, int F, const int* rows, const double* med, int log1p,
                   int* count, double* mean, double* m2, void* ws, fx_stream_t stream);
int fx_col_median(const void* x, int dtype, long ldx, int N, const int* cols, int n_cols, double* med_out,
                  fx_stream_t stream);
int fx_row_moments(const void* x, int dtype, long ldx, int N, const int* cols, int n_cols, const double* med,
                   double* var_out, fx_stream_t stream);
int fx_ingest_transform(const void* x, int dtype, long ldx, const int* rows, int n_rows, const int* cols, int n_cols,
                        const double* med, int log1p, const double* mean, const double* scale, float* out, long ldo,
                        fx_stream_t stream);

/* ---- graph-convolution encoder (flexGCN, modules.py:153-262; GNN, models/gnn_early.py:103-158).  Activations are
 *      [B, nodes, C] fp32 contiguous (node-major in a sample: what x.view(-1, C) and x.view(B, -1) see), 1 <= C <= 32.
 *      The convolutions are torch_geometric's GraphConv / SAGEConv / GCNConv (un-vendored): each is
 *        out = (A x) Wa^T [+ x Wr^T] + bias   with A the weighted adjacency the host builds once per graph
 *      (GC: weight 1; SAGE: 1/in-degree; GCN: self loops + D^-1/2 A D^-1/2), stored as CSR by target node for the
 *      forward and CSR by source node for the backward (dWa = dOut^T (A x), the forward's aggregate; dWr = dOut^T x;
 *      dx = A^T (dOut Wa) + dOut Wr, i.e. the transposed message passing runs at the layer's input width).
 *      fx_spmm_rows:    out[b, i, :] = sum_{e in [rowptr[i], rowptr[i+1])} w[e] * x[b, idx[e], :]   (not in place)
 *      fx_rowlin2:      out[r, :] (+)= a[r, :] Wa^T (+ b[r, :] Wb^T) (+ bias); trans != 0 applies W instead of W^T
 *                       (Wa then is [Ca, Cout]).  nn.Linear semantics per row, R = B * nodes rows.
 *      fx_rowlin_wgrad: dW[Cout, Cin] (+)= dy^T x, db[Cout] (+)= column sums of dy, summed in a fixed order.
 *      fx_bn_rows_fwd:  BatchNorm1d over the R rows (biased batch variance; running stats with momentum 0.1 and the
 *                       unbiased variance) -> act (0 relu, 1 sigmoid, 2 leakyrelu(0.01), 3 tanh, 4 gelu(erf)) ->
 *                       Dropout(drop_p) with an explicit mask [R, C] or Philox(seed, offset + step from ctrl).
 *      fx_bn_rows_bwd:  in place: da holds dL/d(out) on entry, dL/dx on return; dgamma / dbeta are written. */
int fx_gnn_row_blocks(long R);
int fx_spmm_rows(float* out, const float* x, const int* rowptr, const int* idx, const float* w, int B, int nodes, int C,
                 long n_edges /* = rowptr[nodes], a scheduling hint; 0 if unknown */, fx_stream_t stream);
int fx_rowlin2(float* out, const float* a, const float* Wa, int Ca, const float* b, const float* Wb, int Cb,
               const float* bias, long R, int Cout, int trans, int accumulate, fx_stream_t stream);
long fx_rowlin_wgrad_workspace_bytes(long R, int Cin, int Cout);
int fx_rowlin_wgrad(float* dW, float* db, const float* dy, const float* x, long R, int Cin, int Cout, int accumulate,
                    void* ws, fx_stream_t stream);
long fx_bn_rows_workspace_bytes(long R, int C);
int fx_bn_rows_fwd(float* out, const float* x, const float* gamma, const float* beta, float* running_mean,
                   float* running_var, float* save_mean, float* save_invstd, const float* mask, long R, int C, int act,
                   int train, float drop_p, unsigned long long seed, unsigned long long offset, const float* ctrl, void* ws,
                   fx_stream_t stream);
int fx_bn_rows_bwd(float* da, float* dgamma, float* dbeta, const float* x, const float* gamma, const float* beta,
                   const float* save_mean, const float* save_invstd, const float* mask, long R, int C, int act, float drop_p,
                   unsigned long long seed, unsigned long long offset, const float* ctrl, void* ws, fx_stream_t stream);

/* ---- index sampling on the device (csrc/fx_sampling.hip) ------------------------------------------------------------------
 * fx_randperm: out[i] = src[perm[i]] (src NULL: perm[i]) for a uniformly random permutation of 0..n-1 drawn from Philox4x32-10
 * (seed, offset): torch.randperm behind DataLoader(shuffle=True) (reference main.py:289-298).  n <= 4096: one launch; larger n
 * needs fx_randperm_scratch_bytes(n) bytes of scratch.
 * fx_triplet_sample: TripletMultiOmicDataset.__getitem__'s draws (reference data.py:1106-1131) for n anchors: positive = uniform
 * among the other members of the anchor's label group, negative = uniform member of a uniformly chosen other group.  gid[sample]
 * = group id, order = samples sorted by group, starts / counts per group, rank_in_group[sample]; *err_flag |= 1 if an anchor's
 * group has no other member (the reference raises). */
long fx_randperm_scratch_bytes(long n);
int fx_randperm(long* out, const long* src, long n, unsigned long long seed, unsigned long long offset, void* scratch,
                long scratch_bytes, fx_stream_t stream);
int fx_triplet_sample(long* pos, long* neg, const long* anchors, long n, const long* gid, const long* order, const long* starts,
                      const long* counts, const long* rank_in_group, int n_groups, unsigned long long seed,
                      unsigned long long offset, int* err_flag, fx_stream_t stream);

/* ---- host-side runtime (csrc/fx_runtime.hip): memory leases and the library's own hipGraph capture / replay -----------------
 * Leases: the engine sub-allocates the W / m / v arrays of every wide weight (reference: nn.Linear parameters of modules.py:26-41
 * and torch.optim.Adam's state, direct_pred.py:143) from long-lived pools; a range goes back to its pool when the LAST view of it
 * has gone.  fx_lease_wrap returns a DLPack DLManagedTensor* (1-D fp32, n elements at ptr; device_type 10 = ROCm, 1 = host) whose
 * deleter -- called by the importing framework from any thread -- pushes `id` onto the queue; fx_lease_drain pops released ids. */
void* fx_lease_queue_create(void);
void fx_lease_queue_destroy(void* queue);
void* fx_lease_wrap(void* queue, void* ptr, long long n, int device_type, int device_id, long long id);
void fx_lease_discard(void* managed);
int fx_lease_drain(void* queue, long long* ids, int max_ids);
long long fx_lease_outstanding(void* queue);
/* Graphs: everything launched on `stream` (and on streams forked from it through event waits) between fx_graph_begin and
 * fx_graph_end becomes one hipGraphExec (mode: 0 global, 1 thread-local, 2 relaxed); fx_graph_launch replays it on a stream.
 * Replaces torch.cuda.CUDAGraph around the reference's training_step / validation_step loop (main.py:212-225) on the engine's path. */
int fx_graph_begin(fx_stream_t stream, int mode);
int fx_graph_end(fx_stream_t stream, void** exec_out, int* n_nodes);
int fx_graph_abort(fx_stream_t stream);
int fx_graph_launch(void* exec, fx_stream_t stream);
int fx_graph_destroy(void* exec);
int fx_graph_capturing(fx_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* FXHIP_H */
