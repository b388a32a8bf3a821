/*
 * dlwp_hip.h -- C ABI of libdlwp_hip.so, the MI355X (gfx950) kernel library behind the
 * dlwpbench backbone rollout hot path.
 *
 * The reference (AnneLouisedb/dlwp-benchmark) is pure Python/PyTorch and has no FFI of its own;
 * every entry point below replaces a stretch of ATen calls inside a reference nn.Module.forward
 * (cited per function as file:line under src/dlwpbench/).  INTEGRATION.md shows the ctypes stub a
 * maintainer of the reference would add.
 *
 * Conventions
 *   - plain C: raw device pointers, explicit sizes, no torch / C++ types.
 *   - every function returns 0 on success, a negative dlwp_status otherwise; the message for the
 *     calling thread is available from dlwp_last_error().  No C++ exception crosses the ABI.
 *   - `stream` is a hipStream_t passed as void* (NULL = the default stream).  All work is
 *     enqueued asynchronously on it; nothing synchronises the device.
 *   - the caller owns every input / output / workspace buffer; the library owns plan handles.
 *     Plans are immutable after creation and may be shared between threads.
 *   - "dev" pointers are device memory, "host" pointers are host memory.
 */
#ifndef DLWP_HIP_H
#define DLWP_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef enum dlwp_status {
  DLWP_OK = 0,
  DLWP_ERR_INVALID_ARGUMENT = -1,
  DLWP_ERR_UNSUPPORTED = -2,
  DLWP_ERR_HIP = -3,
  DLWP_ERR_WORKSPACE = -4,
  DLWP_ERR_TIMEOUT = -5,     /* a fused kernel's inter-workgroup hand-off exceeded its spin bound (output poisoned) */
  DLWP_ERR_RANGE = -6        /* dlwp_fno2d_status only: an unchecked f16x3 launch produced a non-finite output */
} dlwp_status;

/* library version, major*10000 + minor*100 + patch */
int32_t dlwp_version(void);
/* message of the last failing call on this thread ("" if none) */
const char* dlwp_last_error(void);
/* number of visible HIP devices (<0 on error); does not create a context */
int32_t dlwp_device_count(void);

/* ------------------------------------------------------------------------------------------
 * FNO2d rollout  (reference models/fno/fno.py:12-106 `FNO2DModule`; the arithmetic it delegates
 * to neuralop.models.FNO -- fno.py:38-47 -- is restated in DESIGN.md / oracle/restate/fno.py)
 * ------------------------------------------------------------------------------------------ */
typedef struct dlwp_fno2d_plan dlwp_fno2d_plan;

typedef struct dlwp_fno2d_desc {
  int32_t in_channels;         /* constant + (prescribed + prognostic) * context            */
  int32_t hidden_channels;     /* fno.py:24  (kernel support: 32)                           */
  int32_t lifting_channels;    /* fno.py:25  (multiple of 16)                               */
  int32_t projection_channels; /* fno.py:26  (multiple of 16)                               */
  int32_t out_channels;        /* = prognostic_channels, fno.py:45 (<= 16)                  */
  int32_t n_layers;            /* fno.py:27                                                 */
  int32_t height, width;       /* grid; width must be a multiple of 64                      */
  int32_t n_rows;              /* kept spectral rows   (neuralop: min(H, n_modes[0]))       */
  int32_t n_cols;              /* kept rfft columns    (neuralop: n_modes[1]/2+1), <= 16    */
  const int32_t* rows_in;      /* host [n_rows]: un-shifted rfft row read by weight row r   */
  const int32_t* rows_out;     /* host [n_rows]: un-shifted row of out_fft it lands in      */
  float fwd_scale;             /* rfftn normalisation  (norm="forward": 1/(H*W))            */
  float inv_scale;             /* irfftn normalisation (norm="forward": 1)                  */
  /* weights, HOST pointers, reference (PyTorch) layouts, fp32 */
  const float* lift_w1;        /* [lifting, in]                                             */
  const float* lift_b1;        /* [lifting]                                                 */
  const float* lift_w2;        /* [hidden, lifting]                                         */
  const float* lift_b2;        /* [hidden]                                                  */
  const float* const* spec_w;  /* n_layers x [hidden(in), hidden(out), n_rows, n_cols, 2]   */
  const float* spec_b;         /* [n_layers, hidden]                                        */
  const float* const* skip_w;  /* n_layers x [hidden(out), hidden(in)]                      */
  const float* proj_w1;        /* [projection, hidden]                                      */
  const float* proj_b1;        /* [projection]                                              */
  const float* proj_w2;        /* [out, projection]                                         */
  const float* proj_b2;        /* [out]                                                     */
  /* ---- execution form, fixed for the life of the plan (no process-wide switches exist) ---- */
  int32_t precision_form;      /* 0 (default): the fp32 channel GEMMs run as "bf16x6" -- each fp32 operand split exactly
                                  into three bf16 parts, the six significant cross products accumulated in fp32 on the
                                  bf16 matrix pipe (fp32-GEMM accuracy) -- and the fused kernels;
                                  1: plain fp32-MFMA kernels and the unfused spectral path (independent cross-check);
                                  2: "f16x3" in the fused step kernel -- operands split into two f16 parts (22 significant
                                  bits, the weight residual stored scaled), three products on the f16 matrix instructions:
                                  fp32-GEMM accuracy for |activation| < 65504 at half the matrix instructions and split
                                  work.  A range whose output is not finite is repeated on the
                                  bf16x6 kernels (checked calls) or reported by dlwp_fno2d_status as DLWP_ERR_RANGE
                                  (unchecked = 1).  Kernels other than the fused step keep bf16x6. */
  int32_t launch_form;         /* 0 (default): fewest launches the shapes allow (whole rollout range in one persistent
                                  launch); 1: one launch per step; 2: three launches per step; 3: unfused kernels */
  int32_t on_timeout;          /* a fused launch whose hand-off spin ran out: 0 (default) re-run the range on the unfused
                                  kernels (no hand-offs) and return DLWP_OK; 1 return DLWP_ERR_TIMEOUT */
  int32_t unchecked;           /* 0 (default): every call that used fused kernels synchronises `stream` once and reads
                                  their fail word; 1: fully asynchronous calls, the caller polls dlwp_fno2d_status */
  int32_t debug_spin_limit;    /* 0: default bound (~0.1 s); > 0: spin bound of the hand-offs (test hook) */
} dlwp_fno2d_desc;

int32_t dlwp_fno2d_plan_create(dlwp_fno2d_plan** plan, const dlwp_fno2d_desc* desc, void* stream);
int32_t dlwp_fno2d_plan_destroy(dlwp_fno2d_plan* plan);
/* bytes of device workspace one call needs for `batch` samples */
size_t dlwp_fno2d_workspace_bytes(const dlwp_fno2d_plan* plan, int32_t batch);
/* Deferred check for plans created with unchecked = 1 (fully asynchronous calls): synchronises `stream` and returns
 * DLWP_ERR_TIMEOUT if any fused launch of this plan timed out since the previous status call (a plan-owned device counter
 * the kernels add to; reset here).  The outputs of such launches are poisoned with NaN.  DLWP_ERR_RANGE: an f16x3 launch
 * (precision_form 2) wrote a non-finite output since the previous status call. */
int32_t dlwp_fno2d_status(const dlwp_fno2d_plan* plan, void* stream);
/* statistics: fused launches of this plan that timed out so far (re-run or reported) */
uint32_t dlwp_fno2d_timeouts(const dlwp_fno2d_plan* plan);
/* statistics: f16x3 step ranges of this plan (precision_form 2) repeated on the bf16x6 kernels after a non-finite output */
uint32_t dlwp_fno2d_range_reruns(const dlwp_fno2d_plan* plan);

/* One backbone step WITHOUT the residual: y = fno(x).  Replaces `self.fno(x_t)` at fno.py:103.
 * x_dev [B, in, H, W], y_dev [B, out, H, W], both contiguous fp32. */
int32_t dlwp_fno2d_forward_f32(const dlwp_fno2d_plan* plan, const float* x_dev, float* y_dev,
                               int32_t batch, void* workspace_dev, size_t workspace_bytes,
                               void* stream);

/* Whole autoregressive rollout, device resident.  Replaces FNO2DModule.forward, fno.py:64-106
 * (loop + _prepare_inputs + residual + stack).  Tensors are contiguous fp32:
 *   constants_dev  [B, 1, Cc, H, W] or NULL (Cc = 0)
 *   prescribed_dev [B, T, Cp, H, W] or NULL (Cp = 0)
 *   prognostic_dev [B, T, Cg, H, W]
 *   out_dev        [B, T - context, Cg, H, W]
 * with Cc + (Cp + Cg) * context == plan in_channels and Cg == plan out_channels. */
int32_t dlwp_fno2d_rollout_f32(const dlwp_fno2d_plan* plan, const float* constants_dev,
                               int32_t n_const, const float* prescribed_dev, int32_t n_presc,
                               const float* prognostic_dev, int32_t n_prog, int32_t batch,
                               int32_t n_time, int32_t context, float* out_dev,
                               void* workspace_dev, size_t workspace_bytes, void* stream);

/* Rollout steps [step_begin, step_end) only (0 <= begin <= end <= T - context); steps before
 * step_begin must already be present in out_dev.  Lets the host overlap the all-gather of finished
 * time chunks with the remaining steps (dlwp_benchmark_amd/sharding.py). */
int32_t dlwp_fno2d_rollout_range_f32(const dlwp_fno2d_plan* plan, const float* constants_dev,
                                     int32_t n_const, const float* prescribed_dev, int32_t n_presc,
                                     const float* prognostic_dev, int32_t n_prog, int32_t batch,
                                     int32_t n_time, int32_t context, float* out_dev,
                                     void* workspace_dev, size_t workspace_bytes, void* stream,
                                     int32_t step_begin, int32_t step_end);

/* Same rollout with every kernel launch bracketed by a pair of HIP events on `stream`
 * (measurement aid for bench.py's roofline leg; synchronises the stream before returning).
 * Kernel classes: 0 lifting MLP, 1 fno_modes_kernel, 2 fno_layer_kernel, 3 projection MLP,
 * 4 an EMPTY bracket per step (what the event pair itself adds; subtract its average from the others).
 * class_ms[5]: summed event-to-event milliseconds, class_launches[5]: brackets per class. */
int32_t dlwp_fno2d_rollout_profiled_f32(const dlwp_fno2d_plan* plan, const float* constants_dev,
                                        int32_t n_const, const float* prescribed_dev,
                                        int32_t n_presc, const float* prognostic_dev,
                                        int32_t n_prog, int32_t batch, int32_t n_time,
                                        int32_t context, float* out_dev, void* workspace_dev,
                                        size_t workspace_bytes, void* stream, double* class_ms,
                                        int32_t* class_launches);

/* ------------------------------------------------------------------------------------------
 * SpectralConv2d  (reference models/unet/unet.py:19-69 + batchmul2d :15-17; PDE-Arena style:
 * un-normalised rfft2, rows [:m1] with weights1 and rows [-m1:] with weights2, cols [:m2],
 * irfft2).  x_dev [B, Ci, H, W] -> y_dev [B, Co, H, W], contiguous fp32.
 * ------------------------------------------------------------------------------------------ */
typedef struct dlwp_spectral_plan dlwp_spectral_plan;

int32_t dlwp_spectral_conv2d_plan_create(dlwp_spectral_plan** plan, int32_t in_channels,
                                         int32_t out_channels, int32_t height, int32_t width,
                                         int32_t modes1, int32_t modes2,
                                         const float* weights1_host, /* [Ci,Co,m1,m2,2] */
                                         const float* weights2_host, /* [Ci,Co,m1,m2,2] */
                                         void* stream);
/* The same operator with explicit kept rows and transform scales (SpectralCore of the FNO layers: row r of the
 * weights reads un-shifted rfft row rows_in[r] and writes row rows_out[r]; x_hat is scaled by fwd_scale, the inverse
 * by inv_scale) and weights that live on the DEVICE -- what a training step needs (reference
 * scripts/train.py:263-271 `loss.backward()` through models/unet/unet.py:46-69 / neuralop SpectralConv).
 * dlwp_spectral_conv2d_set_weights_dev packs weights_dev [Ci, Co, n_rows, n_cols, 2] (PyTorch layout, forward
 * operator) into the plan on `stream`; with adjoint != 0 it packs the conjugate transpose, which makes
 * dlwp_spectral_conv2d_f32 the BACKWARD-DATA pass (create that plan with rows_in and rows_out swapped). */
int32_t dlwp_spectral_conv2d_plan_create_ex(dlwp_spectral_plan** plan, int32_t in_channels, int32_t out_channels,
                                            int32_t height, int32_t width, int32_t n_rows, int32_t n_cols,
                                            const int32_t* rows_in, const int32_t* rows_out, float fwd_scale,
                                            float inv_scale, void* stream);
int32_t dlwp_spectral_conv2d_set_weights_dev(dlwp_spectral_plan* plan, const float* weights_dev, int32_t adjoint,
                                             void* stream);
int32_t dlwp_spectral_conv2d_plan_destroy(dlwp_spectral_plan* plan);
size_t dlwp_spectral_conv2d_workspace_bytes(const dlwp_spectral_plan* plan, int32_t batch);
int32_t dlwp_spectral_conv2d_f32(const dlwp_spectral_plan* plan, const float* x_dev, float* y_dev,
                                 int32_t batch, void* workspace_dev, size_t workspace_bytes,
                                 void* stream);

/* ------------------------------------------------------------------------------------------
 * Fused (shifted-)window attention, fp32.  Replaces everything between the qkv Linear and the proj
 * Linear of a transformer block:
 *   Swin : models/swintransformer/swin_transformer.py:217-251 (pad, roll, window_partition,
 *          WindowAttention.forward :122-154 without its two Linears, window_reverse, roll, crop) and
 *          the per-call shift-mask build :383-401
 *   Pangu: models/panguweather/panguweather.py:285-316 + EarthAttention3D.forward :176-211 without
 *          its two Linears, utils/shift_window_mask.py, utils/earth_position_index.py, utils/pad.py,
 *          utils/crop.py
 * qkv_dev  [B, L, 3, heads, head_dim]  output of the qkv Linear on the un-padded token sequence,
 *          L = grid[0]*grid[1]*grid[2] (a 2-D model uses grid[0] = window[0] = 1)
 * qkv_bias_dev [3*heads*head_dim] or NULL: value of q,k,v at zero-padded tokens (the reference
 *          pads before the Linear); required when padded != grid
 * table_dev  bias_mode 0: relative_position_bias_table [(2Wh-1)(2Ww-1), heads]
 *            bias_mode 1: earth_position_bias_table [wpl^2*wlat^2*(2wlon-1), types, heads],
 *                         types = (padded[0]/window[0]) * (padded[1]/window[1])
 * out_dev  [B, L, heads*head_dim] in the input token order (window reverse / roll back / crop done)
 * ------------------------------------------------------------------------------------------ */
typedef struct dlwp_wattn_desc {
  int32_t grid[3];        /* un-padded (pl, lat, lon)                                          */
  int32_t padded[3];      /* padded grid, multiple of window                                   */
  int32_t pad_lead[3];    /* zeros added in front / top / left                                 */
  int32_t window[3];
  int32_t shift_fwd[3];   /* torch.roll(x, shifts=-shift_fwd) before partitioning              */
  int32_t shift_back[3];  /* torch.roll(y, shifts=+shift_back) after window_reverse            */
  int32_t use_mask;       /* add the 0/-100 region mask                                        */
  int32_t mask_b1[3];     /* region id along a dim = (p >= mask_b1) + (p >= mask_b2), p being  */
  int32_t mask_b2[3];     /*   the coordinate in the shifted, padded frame                     */
  int32_t bias_mode;      /* 0 Swin relative position, 1 Pangu earth-specific                  */
  int32_t heads, head_dim;
  float scale;            /* qk scale (head_dim ** -0.5 unless overridden)                     */
  int32_t form;           /* dlwp_window_attn_f32 only: -1 by window size (default), 0 fp32 MFMA, 1 bf16x6 */
} dlwp_wattn_desc;

/* fp32-accurate window attention (the parity path), in one of two independent forms of the two contractions:
 * fp32 operands on v_mfma_f32_16x16x4_f32, or exact three-way bf16 splits of Q, K, V and P with six cross products each
 * on the bf16 matrix pipe ("bf16x6").  desc->form selects: 0 fp32 MFMA, 1 bf16x6, -1 by window size as measured
 * (>= 512 tokens per window: fp32 MFMA; smaller: bf16x6).  No process-wide switch exists.
 * tokens * 3 * heads * head_dim must stay below 2^31. */
/* bytes of device workspace the FAST path of the call below needs for this descriptor (bf16 != 0: for
 * dlwp_window_attn_bf16); 0 when the descriptor runs on the generic kernel, which needs none.  The fast path covers
 * 2-D windows (bias_mode 0, grid[0] = 1, no zero padding, window longitude extent a multiple of 16, head_dim a multiple
 * of 8 but not of 32, region boundaries along longitude on multiples of 16): every Swin block of the reference.  Its
 * workspace holds the window-ordered bf16 operand images a prep kernel writes per call.  Masked
 * (shifted-window) tiles are skipped there: exact unless a masked logit exceeds its row's unmasked maximum by > 83. */
size_t dlwp_window_attn_workspace_bytes(const dlwp_wattn_desc* desc, int32_t batch, int32_t bf16);
/* Diagnostics (synchronises `stream`): workgroups of the LAST fast-path call on this workspace whose scores left the
 * 2^+-100 exponent slack around their reference offset and were recomputed with the exact row maximum. */
int32_t dlwp_window_attn_fallbacks(const dlwp_wattn_desc* desc, int32_t batch, int32_t bf16, const void* workspace_dev,
                                   void* stream, int32_t* count);
/* workspace_dev may be NULL (or smaller than the bytes above): the generic kernel runs instead. */
int32_t dlwp_window_attn_f32(const dlwp_wattn_desc* desc, const float* qkv_dev,
                             const float* qkv_bias_dev, const float* table_dev, float* out_dev,
                             int32_t batch, void* workspace_dev, size_t workspace_bytes, void* stream);
/* Same interface (fp32 tensors in and out); Q, K, V and the softmax probabilities are rounded to bf16
 * and both products run on v_mfma_f32_16x16x32_bf16 with fp32 accumulation and fp32 softmax statistics
 * (the precision BASELINE.json names for the Swin / Pangu configs). */
int32_t dlwp_window_attn_bf16(const dlwp_wattn_desc* desc, const float* qkv_dev,
                              const float* qkv_bias_dev, const float* table_dev, float* out_dev,
                              int32_t batch, void* workspace_dev, size_t workspace_bytes, void* stream);

/* dlwp_window_attn_bf16 with bfloat16 TENSORS: qkv_dev [B, L, 3 C], qkv_bias_dev [3 C] and out_dev [B, L, C] are bf16 -- the
 * hand-over of a block in the bf16 form (the qkv Linear writes bf16, proj reads bf16: dlwp_linear_bf16_io).  Same arithmetic as
 * dlwp_window_attn_bf16 on bf16-rounded inputs.  Covered: the descriptors of the two fast paths (workspace as for
 * dlwp_window_attn_bf16); anything else returns DLWP_ERR_UNSUPPORTED (convert and call dlwp_window_attn_bf16).
 * 2-D (Swin) descriptors run WITHOUT the prep kernel: the attention kernel gathers Q, K, V from qkv_dev itself (window order and
 * roll through an LDS token map), so the images part of the workspace stays unused; shifted blocks launch one small key-norm
 * kernel in front.  dlwp_window_attn_fallbacks is maintained for shifted blocks only on this entry point. */
int32_t dlwp_window_attn_bf16_io(const dlwp_wattn_desc* desc, const void* qkv_dev, const void* qkv_bias_dev,
                                 const float* table_dev, void* out_dev, int32_t batch, void* workspace_dev,
                                 size_t workspace_bytes, void* stream);

/* Backward of dlwp_window_attn_f32 (csrc/window_attn_bwd.hip): what loss.backward() of reference scripts/train.py:263-271
 * runs through WindowAttention.forward (swin_transformer.py:122-154, :217-251) / EarthAttention3D.forward
 * (panguweather.py:176-211, :285-316), without their Linears.  Flash-style: scores are recomputed per 32 x 32 tile from
 * qkv and per-row statistics; no N x N tensor exists.  fp32 arithmetic.
 *   grad_out_dev      [B, L, C]      gradient of the attention output (same token order as out_dev)
 *   grad_qkv_dev      [B, L, 3 C]    written (zeroed inside; dq arrives through float atomics)
 *   grad_qkv_bias_dev [3 C] or NULL  gradient that reaches the qkv bias through ZERO-PADDED tokens (they carry q = k = v =
 *                                    bias); required when the descriptor pads, written (zeroed inside)
 *   grad_table_dev    like table_dev written (zeroed inside)
 *   workspace_dev     dlwp_window_attn_bwd_workspace_bytes(desc, batch) bytes: {row max, row sum, delta} per window row */
size_t dlwp_window_attn_bwd_workspace_bytes(const dlwp_wattn_desc* desc, int32_t batch);
int32_t dlwp_window_attn_bwd_f32(const dlwp_wattn_desc* desc, const float* qkv_dev, const float* qkv_bias_dev,
                                 const float* table_dev, const float* grad_out_dev, float* grad_qkv_dev,
                                 float* grad_qkv_bias_dev, float* grad_table_dev, int32_t batch, void* workspace_dev,
                                 size_t workspace_bytes, void* stream);

/* ------------------------------------------------------------------------------------------
 * AFNO2D frequency-domain mixing (reference models/fourcastnet/fourcastnet.py:87-121): complex
 * block-diagonal 2-layer MLP with ReLU, mode truncation and softshrink over the rfft2 spectrum.
 * xf_dev / yf_dev: interleaved complex64, CHANNELS-FIRST [B, C, H, Wf, 2] (Wf = W/2+1) -- the physical
 * layout torch.fft.rfft2(x_nhwc, dim=(1,2)) produces and irfft2 consumes without a copy; weights in the
 * reference layouts w1,w2 [2, nb, bs, bs], b1,b2 [2, nb, bs] (device pointers), bs in {4,8,16,32}.
 * yf is fully written (zeros outside the kept modes).
 * ------------------------------------------------------------------------------------------ */
int32_t dlwp_afno2d_mix_f32(const float* xf_dev, float* yf_dev, const float* w1_dev, const float* b1_dev,
                            const float* w2_dev, const float* b2_dev, int32_t batch, int32_t height,
                            int32_t wf, int32_t channels, int32_t num_blocks, float sparsity_threshold,
                            float hard_thresholding_fraction, void* stream);
/* Backward of the mixing between UNNORMALISED transforms (training; fourcastnet.py:96-121 under loss.backward()):
 * xf = R2C(x), gf = R2C(grad_y), both [B, C, H, Wf, 2] as above (Wf = the columns the transforms carry; `width` = W of the
 * real field, for the Hermitian weight of the Nyquist column).  Writes gxf (its C2R is grad_x) and the per-point factors of the
 * weight gradients -- xin, o1 (post-ReLU), d1, d2 (masked upstream gradients), same layout, zeros outside the kept modes:
 * dW1 = sum_p conj(xin) (x) d1, db1 = sum_p d1, dW2 = sum_p conj(o1) (x) d2, db2 = sum_p d2 per block (csrc/afno.hip). */
int32_t dlwp_afno2d_mix_bwd_f32(const float* xf_dev, const float* gf_dev, float* gxf_dev, float* xin_dev, float* o1_dev,
                                float* d1_dev, float* d2_dev, const float* w1_dev, const float* b1_dev, const float* w2_dev,
                                const float* b2_dev, int32_t batch, int32_t height, int32_t wf, int32_t channels,
                                int32_t num_blocks, int32_t width, float sparsity_threshold, float hard_thresholding_fraction,
                                float in_scale, float out_scale, void* stream);
/* Same with yf = out_scale * mix(in_scale * xf): lets the caller use UNNORMALISED transforms (below) and still get
 * the reference's norm="ortho" arithmetic (:87, :122; in_scale = out_scale = 1/sqrt(H*W)) without two elementwise
 * passes over the spectrum.  yf_dev may be xf_dev (in place): a point's channels are read before they are written and
 * no other point is read by the thread that writes it. */
int32_t dlwp_afno2d_mix_scaled_f32(const float* xf_dev, float* yf_dev, const float* w1_dev, const float* b1_dev,
                                   const float* w2_dev, const float* b2_dev, int32_t batch, int32_t height,
                                   int32_t wf, int32_t channels, int32_t num_blocks, float sparsity_threshold,
                                   float hard_thresholding_fraction, float in_scale, float out_scale, void* stream);

/* Batched unnormalised 2-D real FFTs of `batch` contiguous [H, W] planes (reference fourcastnet.py:87
 * `torch.fft.rfft2(x, dim=(1, 2))` and :122-123 `irfft2`, applied channels-first so that batch = B * C) through
 * hipFFT, without the clone, layout copies and scaling pass torch.fft adds per call.
 *   dlwp_rfft2_f32:  x_dev [batch][H][W] -> xf_dev [batch][H][W/2+1][2]
 *   dlwp_irfft2_f32: yf_dev [batch][H][W/2+1][2] -> y_dev [batch][H][W]; yf_dev is DESTROYED (C2R scratch). */
typedef struct dlwp_fft2_plan dlwp_fft2_plan;
int32_t dlwp_fft2_plan_create(dlwp_fft2_plan** out, int32_t batch, int32_t height, int32_t width);
int32_t dlwp_fft2_plan_destroy(dlwp_fft2_plan* plan);
int32_t dlwp_rfft2_f32(const dlwp_fft2_plan* plan, const float* x_dev, float* xf_dev, void* stream);
int32_t dlwp_irfft2_f32(const dlwp_fft2_plan* plan, float* yf_dev, float* y_dev, void* stream);

/* Hand-written 2-D real FFTs restricted to the columns the AFNO filter keeps (fourcastnet.py:85, :93-94, :124;
 * csrc/afno_fft.hip): one workgroup per [H][W] plane, rows as packed-real FFTs, columns on the LDS-resident image,
 * every complex FFT as two register passes (radix 4 / 8 / 16).  Unnormalised, like the hipFFT entry points above.
 *   dlwp_afno_rfft2_kept_f32:  x_dev [planes][H][W] -> spec_dev [planes][H][kept_cols][2]
 *   dlwp_afno_irfft2_kept_f32: spec_dev [planes][H][kept_cols][2] (columns >= kept_cols are zero; the imaginary part
 *                              of column 0 is ignored, c2r semantics) -> y_dev [planes][H][W]; spec_dev is preserved.
 * kept_cols = min(int((H/2+1) * hard_thresholding_fraction), W/2+1).  Instantiated grids: dlwp_afno_fft_supported. */
typedef struct dlwp_afno_fft_plan dlwp_afno_fft_plan;
int32_t dlwp_afno_fft_supported(int32_t height, int32_t width, int32_t kept_cols);
int32_t dlwp_afno_fft_plan_create(dlwp_afno_fft_plan** out, int32_t height, int32_t width, int32_t kept_cols, void* stream);
int32_t dlwp_afno_fft_plan_destroy(dlwp_afno_fft_plan* plan);
int32_t dlwp_afno_rfft2_kept_f32(const dlwp_afno_fft_plan* plan, const float* x_dev, float* spec_dev, int32_t planes,
                                 void* stream);
int32_t dlwp_afno_irfft2_kept_f32(const dlwp_afno_fft_plan* plan, const float* spec_dev, float* y_dev, int32_t planes,
                                  void* stream);

/* ------------------------------------------------------------------------------------------
 * CylinderPad(1) + Conv2d(3x3, padding 0) + bias + activation, input optionally given as two
 * channel segments (folds the preceding torch.cat).  Reference: utils/utils.py:11-26;
 * models/unet/unet.py:456-470, :512-525, :553; models/convlstm/convlstm.py:47-55, :94, :148-157.
 * x0_dev [B, c0, H, W], x1_dev [B, c1, H, W] or NULL (c1 = 0), weight_dev [cout, c0+c1, 3, 3],
 * bias_dev [cout] or NULL, y_dev [B, cout, H, W].  act: 0 none, 1 GELU(erf), 2 tanh, 3 ReLU, 4 SiLU.
 * ------------------------------------------------------------------------------------------ */
int32_t dlwp_conv3x3_cyl_f32(const float* x0_dev, int32_t c0, const float* x1_dev, int32_t c1,
                             const float* weight_dev, const float* bias_dev, float* y_dev, int32_t batch,
                             int32_t height, int32_t width, int32_t cout, int32_t act, void* stream);

/* The same convolution with two more fusions and either padding rule: pre_act is applied to the input while it is staged
 * (pre-activation residual blocks, unet.py:886 `h = act(norm1(x))` when norm1 is the identity), resid_dev [B, cout, H, W]
 * or NULL is added after the bias and before `act` (the block's shortcut, unet.py:901).  ring_table NULL: CylinderPad;
 * otherwise the HEALPix halo table of dlwp_conv3x3_hpx_f32 (batch = 12 * samples faces). */
int32_t dlwp_conv3x3_ex_f32(const float* x0_dev, int32_t c0, const float* x1_dev, int32_t c1, const float* weight_dev,
                            const float* bias_dev, const float* resid_dev, float* y_dev, int32_t batch, int32_t height,
                            int32_t width, int32_t cout, int32_t pre_act, int32_t act, const int32_t* ring_table, void* stream);

/* ------------------------------------------------------------------------------------------
 * fp32 Linear layers on the bf16 matrix pipe with the block's pointwise work fused (csrc/linear.hip): the qkv / proj /
 * fc1 / fc2 Linears, GELU and residual adds of the Swin and Pangu blocks (swin_transformer.py:21-39, :107-120, :254-262;
 * panguweather.py:176-211, :318-322).
 *   out[m][n] = act(sum_k x[m][k] W[n][k] + bias[n]) + resid[m][n],  act: 0 none, 1 exact-erf GELU
 * fp32 tensors, fp32-GEMM accuracy (exact three-way bf16 splits of both operands, six cross products, fp32 accumulation).
 * weight_dev [out, in] (nn.Linear layout) is split once by dlwp_linear_pack_f32 into `packed_dev`
 * (dlwp_linear_packed_bytes bytes, caller-owned; 0 = unsupported: in % 32 != 0 or out % 4 != 0); bias / resid may be NULL,
 * resid may alias out (in-place residual).
 * ------------------------------------------------------------------------------------------ */
size_t dlwp_linear_packed_bytes(int32_t out_features, int32_t in_features);
int32_t dlwp_linear_pack_f32(const float* weight_dev, int32_t out_features, int32_t in_features, void* packed_dev, void* stream);
int32_t dlwp_linear_f32(const float* x_dev, const void* packed_dev, const float* bias_dev, const float* resid_dev,
                        float* out_dev, int64_t rows, int32_t in_features, int32_t out_features, int32_t act, void* stream);
/* The same Linear with bf16 operands (x rounded to bf16 on the fly, the first bf16 image of the weight) and fp32
 * accumulation -- what torch.autocast(bfloat16) makes of nn.Linear; one matrix-pipe product instead of six. */
int32_t dlwp_linear_bf16(const float* x_dev, const void* packed_dev, const float* bias_dev, const float* resid_dev,
                         float* out_dev, int64_t rows, int32_t in_features, int32_t out_features, int32_t act, void* stream);
/* The same Linear in the "f16x3" form: both operands split exactly into two f16 parts (22 significant bits; the weight
 * residual stored scaled by 2^11, so any weight magnitude keeps them), three products on the f16 matrix instructions
 * instead of six bf16 ones.  fp32-GEMM accuracy for |x| < 65504 (an x beyond the f16 range turns into inf: the caller's
 * contract; LayerNorm / GELU / attention outputs are orders of magnitude inside it) and for activations that are not
 * all tiny (the residual of |x| < 0.125 is an f16 subnormal, absolute spacing 2^-24).  `packed_dev` comes from
 * dlwp_linear_pack_f16x3 (same byte count as dlwp_linear_packed_bytes). */
/* dlwp_linear_bf16 with a bf16 TENSOR on one side -- how the MLP of a block hands its hidden activation from fc1 to fc2 in the
 * bf16 form (swin_transformer.py:21-39 under autocast): x_is_bf16: x_dev is bf16 [rows][in] (the values dlwp_linear_bf16 would round
 * its fp32 input to: bit-identical result); out_is_bf16: out_dev is bf16 [rows][out], rounded to nearest even after bias / GELU,
 * resid_dev must be NULL.  At least one of the two flags.  dlwp_layernorm_prebias_bf16out (below the LayerNorm entry points)
 * produces such an input. */
int32_t dlwp_linear_bf16_io(const void* x_dev, const void* packed_dev, const float* bias_dev, const float* resid_dev,
                            void* out_dev, int64_t rows, int32_t in_features, int32_t out_features, int32_t act,
                            int32_t x_is_bf16, int32_t out_is_bf16, void* stream);
int32_t dlwp_linear_pack_f16x3(const float* weight_dev, int32_t out_features, int32_t in_features, void* packed_dev, void* stream);
int32_t dlwp_linear_f16x3(const float* x_dev, const void* packed_dev, const float* bias_dev, const float* resid_dev,
                          float* out_dev, int64_t rows, int32_t in_features, int32_t out_features, int32_t act, void* stream);

/* ------------------------------------------------------------------------------------------
 * The remaining U-Net / ModernUNet operators (csrc/conv2.hip), NCHW fp32, activations as above.
 *   dlwp_groupnorm_act_f32     y = act(GroupNorm(groups)(x)): unet.py:739 (+ GELU :761), :887-888; gamma / beta [C] or NULL
 *   dlwp_conv2d_f32            zero-padded Conv2d k x k, stride s: unet.py:583 (3x3 s2 p1), :584 / :879 / :450 (1x1);
 *                              optional pre_act on the input and resid_dev [N, cout, OH, OW] before `act`
 *   dlwp_conv_transpose2d_f32  ConvTranspose2d k x k, stride s, padding p (weight [cin, cout, k, k]): unet.py:719, :523
 *   dlwp_avgpool2x2_f32        AvgPool2d(2): unet.py:450
 * ------------------------------------------------------------------------------------------ */
int32_t dlwp_groupnorm_act_f32(const float* x_dev, const float* gamma_dev, const float* beta_dev, float* y_dev, int32_t batch,
                               int32_t channels, int32_t hw, int32_t groups, float eps, int32_t act, void* stream);
int32_t dlwp_conv2d_f32(const float* x_dev, const float* weight_dev, const float* bias_dev, const float* resid_dev, float* y_dev,
                        int32_t batch, int32_t cin, int32_t height, int32_t width, int32_t cout, int32_t k, int32_t stride,
                        int32_t pad, int32_t pre_act, int32_t act, void* stream);
int32_t dlwp_conv_transpose2d_f32(const float* x_dev, const float* weight_dev, const float* bias_dev, float* y_dev, int32_t batch,
                                  int32_t cin, int32_t height, int32_t width, int32_t cout, int32_t k, int32_t stride,
                                  int32_t pad, int32_t act, void* stream);
int32_t dlwp_avgpool2x2_f32(const float* x_dev, float* y_dev, int64_t planes, int32_t height, int32_t width, void* stream);

/* ------------------------------------------------------------------------------------------
 * HEALPix mesh.  Faces are folded into the batch, [(B*12), C, H, W], face index fastest
 * (reference models/unet/unet.py:413-426 `b c f h w -> (b f) c h w`).  The neighbour topology of
 * reference utils/healpix.py:165-368 (`HEALPixPadding`: rotated polar neighbours, synthesised corners) is
 * handed over as a device table of int32 pairs (a, b): source cells face*H*W + pixel inside the same
 * sample; b < 0 = copy a, else 0.5*a + 0.5*b.
 *   dlwp_healpix_pad_f32: the padding layer on its own, table [12][(H+2p)*(W+2p)][2], y [(B*12), C, H+2p, W+2p].
 *   dlwp_conv3x3_hpx_f32: HEALPixLayer(Conv2d 3x3) = HEALPixPadding(1) + Conv2d(padding 0) + bias +
 *   activation (healpix.py:69-114) in one kernel, ring table [12][(H+2)*(W+2)][2]; other arguments as
 *   dlwp_conv3x3_cyl_f32 with batch = n_faces = B*12.
 * ------------------------------------------------------------------------------------------ */
int32_t dlwp_healpix_pad_f32(const float* x_dev, float* y_dev, const int32_t* table_dev, int32_t n_faces,
                             int32_t channels, int32_t height, int32_t width, int32_t pad, void* stream);
int32_t dlwp_conv3x3_hpx_f32(const float* x0_dev, int32_t c0, const float* x1_dev, int32_t c1,
                             const float* weight_dev, const float* bias_dev, float* y_dev, int32_t n_faces,
                             int32_t height, int32_t width, int32_t cout, int32_t act,
                             const int32_t* ring_table_dev, void* stream);

/* ConvLSTM cell gate math (models/convlstm/convlstm.py:96-109): gates_dev [B, 4*hidden, H, W] in the
 * order (netin, igate, fgate, ogate), c_prev_dev [B, hidden, H, W] -> h_out_dev, c_out_dev. */
int32_t dlwp_convlstm_gates_f32(const float* gates_dev, const float* c_prev_dev, float* h_out_dev,
                                float* c_out_dev, int32_t batch, int32_t hidden, int32_t height,
                                int32_t width, void* stream);

/* LayerNorm over the last dimension of a token-major tensor x_dev [rows, channels] (channels % 4 == 0,
 * <= 2048): y = (x - mean) * rsqrt(var + eps) * gamma + beta, biased variance (torch.nn.LayerNorm).
 * Reference call sites: models/fourcastnet/fourcastnet.py:180-193, models/swintransformer/
 * swin_transformer.py:213,262,304,440,664, models/panguweather/panguweather.py:73,127,281,321. */
int32_t dlwp_layernorm_f32(const float* x_dev, const float* gamma_dev, const float* beta_dev, float* y_dev,
                           int64_t rows, int32_t channels, float eps, void* stream);
/* y = LayerNorm(x + pre_bias): pre_bias_dev [channels] (or NULL) is added before the statistics.  Lets the transformer
 * blocks (swin_transformer.py:254-262, panguweather.py:318-322: `x = shortcut + proj(attn)`, `x = x + mlp(norm2(x))`) run
 * their residual adds as the beta = 1 accumulation of the proj / fc2 GEMMs, in place on x, with the Linear biases
 * carried as ONE pending per-channel vector that only the LayerNorms (here) and the end of the layer ever apply. */
int32_t dlwp_layernorm_prebias_f32(const float* x_dev, const float* pre_bias_dev, const float* gamma_dev,
                                   const float* beta_dev, float* y_dev, int64_t rows, int32_t channels, float eps,
                                   void* stream);
/* The same LayerNorm with a bf16 result (y_bf16_dev [rows][channels], round to nearest even): the input of a bf16-form Linear
 * (dlwp_linear_bf16_io with x_is_bf16), which would round the fp32 result the same way -- bit-identical, half the bytes. */
int32_t dlwp_layernorm_prebias_bf16out(const float* x_dev, const float* pre_bias_dev, const float* gamma_dev,
                                       const float* beta_dev, void* y_bf16_dev, int64_t rows, int32_t channels, float eps,
                                       void* stream);

/* FourCastNet block glue fused with the layout change the FFT needs (models/fourcastnet/fourcastnet.py
 * :180-193 around AFNO2D :78-127).  x is token-major [B, tokens, C] ("NHWC"), y / f / l channel-major
 * [B, C, tokens] ("NCHW"); C % 4 == 0, C <= 256.
 *   dlwp_layernorm_nhwc_to_nchw_f32: y = LayerNorm1(x), written channel-major (:182 norm1 + the transpose
 *       torch.fft.rfft2(dim=(1,2)) would otherwise do with a strided copy)
 *   dlwp_afno_merge_f32: sum = f + l + x  (irfft2 output + AFNO2D "+ bias" :127 + first skip :187),
 *       norm = LayerNorm2(sum) (:191); both token-major.  sum_bias_dev [C] or NULL is added to the STORED sum only:
 *       the host passes mlp.fc2.bias so that `mlp(norm) + sum` (:192) becomes one GEMM with beta = 1.
 *       norm_nhwc_dev NULL (then gamma / beta may be NULL too): only the sum is produced -- for callers whose next
 *       kernel normalises on the fly (dlwp_token_mlp_f32 with ln_eps >= 0). */
int32_t dlwp_layernorm_nhwc_to_nchw_f32(const float* x_dev, const float* gamma_dev, const float* beta_dev,
                                        float* y_dev, int32_t batch, int64_t tokens, int32_t channels, float eps,
                                        void* stream);
int32_t dlwp_afno_merge_f32(const float* f_nchw_dev, const float* l_nchw_dev, const float* x_nhwc_dev,
                            const float* gamma_dev, const float* beta_dev, const float* sum_bias_dev,
                            float* sum_nhwc_dev, float* norm_nhwc_dev, int32_t batch, int64_t tokens, int32_t channels, float eps,
                            void* stream);

/* Patch embedding for 1x1 patches + position embedding (reference fourcastnet.py:530-543 `PatchEmbed` =
 * Conv2d(kernel = stride = patch) -> flatten(2).transpose(1, 2), and `x + pos_embed` at :286-288) in one pass:
 *   out[b][t][c] = bias[c] + pos[t][c] + sum_ci w[c][ci] x[b][ci][t]
 * x_dev [batch][in_channels][tokens] (NCHW with tokens = H*W), w_dev [channels][in_channels] (the conv weight with its
 * 1x1 kernel dims dropped), bias_dev [channels] or NULL, pos_dev [tokens][channels] or NULL, out_dev token-major.
 * in_channels <= 32, channels a power of two in [4, 256]; other shapes: DLWP_ERR_UNSUPPORTED. */
int32_t dlwp_patch_embed_1x1_f32(const float* x_dev, const float* w_dev, const float* bias_dev, const float* pos_dev,
                                 float* out_dev, int32_t batch, int32_t in_channels, int64_t tokens, int32_t channels,
                                 void* stream);

/* Head of a 1x1-patch token backbone (reference fourcastnet.py:144 `head = nn.Linear(embed_dim, out_chans * p1 * p2, bias=False)`, applied
 * at :296-303 with the rearrange "b h w (p1 p2 c_out) -> b c_out (h p1) (w p2)"; p1 = p2 = 1):
 *   out[b][co][t] = bias[co] + sum_c w[co][c] tokens[b][t][c]
 * tokens_dev [batch][tokens][channels] token-major, w_dev [out_channels][channels], bias_dev [out_channels] or NULL, out_dev
 * [batch][out_channels][tokens] channels-first (NCHW with tokens = H*W).  channels a multiple of 4, <= 256; out_channels <= 16. */
int32_t dlwp_patch_recover_1x1_f32(const float* tokens_dev, const float* w_dev, const float* bias_dev, float* out_dev,
                                   int32_t batch, int64_t tokens, int32_t channels, int32_t out_channels, void* stream);

/* `_prepare_inputs` of the rollout loop (reference swin_transformer.py:679-692 and its copies in fno.py:49-62, fourcastnet.py:294-307,
 * panguweather.py:442-455, unet.py:316-329): cat([constants[:, 0], prescribed window, prognostic window], dim = 1).  n_segments <= 8 blocks
 * [batch][seg_channels[i]][plane] whose samples lie seg_batch_strides[i] floats apart (views into the inputs / the trajectory buffer) are
 * copied into out_dev [batch][sum channels][plane], contiguous.  The three arrays are HOST arrays; plane a multiple of 4, 16-byte alignment. */
int32_t dlwp_concat_channels_f32(const float* const* seg_dev_ptrs, const int32_t* seg_channels, const int64_t* seg_batch_strides,
                                 int32_t n_segments, float* out_dev, int32_t batch, int64_t plane, void* stream);

/* Token MLP of the AFNO block (reference fourcastnet.py:41-57 `Mlp` = fc1 -> GELU -> fc2, called at :191-192 as
 * `x = mlp(norm2(x)) + residual`):  out[t] = resid[t] + b2 + W2 gelu(W1 n[t] + b1), all token-major [tokens][channels].
 * One launch; the [tokens][hidden] activation never reaches memory (both GEMMs on the bf16 matrix pipe as six-term
 * exact splits = fp32-GEMM accuracy).  channels == 64, hidden % 64 == 0, hidden <= 256 (weights are LDS resident);
 * anything else returns DLWP_ERR_UNSUPPORTED and the caller keeps its GEMM path.
 *   dlwp_token_mlp_packed_bytes: size of the packed-weight buffer (0 if the shape is unsupported)
 *   dlwp_token_mlp_pack_f32:     w1_dev [hidden][channels] (fc1.weight), w2_dev [channels][hidden] (fc2.weight) -> packed.
 *                                With ln_gamma_dev / ln_beta_dev [channels] (both or neither) the affine part of the
 *                                LayerNorm in front of fc1 (`norm2`, :191) is folded in: W1 diag(gamma), and
 *                                b1 + W1 beta (b1_dev [hidden] or NULL) is stored in the packed buffer.
 *                                merged_layout != 0: the k-slot order dlwp_afno_block_tail_f32 wants (below).
 *   dlwp_token_mlp_f32:          ln_eps < 0: n_dev is the fc1 input, b1_dev required.  ln_eps >= 0: n_dev is the
 *                                UN-normalised token (normally the same buffer as resid_dev); the kernel normalises
 *                                it (two-pass statistics over the channels) and takes b1 from a buffer packed WITH
 *                                gamma / beta (b1_dev ignored).  resid_dev, b2_dev may be NULL; out_dev may alias
 *                                resid_dev / n_dev (a wave reads all it needs of its 32 tokens before it writes). */
size_t dlwp_token_mlp_packed_bytes(int32_t channels, int32_t hidden);
int32_t dlwp_token_mlp_pack_f32(const float* w1_dev, const float* w2_dev, const float* ln_gamma_dev,
                                const float* ln_beta_dev, const float* b1_dev, int32_t channels, int32_t hidden,
                                int32_t merged_layout, void* packed_dev, void* stream);
int32_t dlwp_token_mlp_f32(const float* n_dev, const float* resid_dev, const void* packed_dev, const float* b1_dev,
                           const float* b2_dev, float* out_dev, int64_t tokens, int32_t channels, int32_t hidden,
                           float ln_eps, void* stream);
/* dlwp_token_mlp_f32 that ALSO emits next = LayerNorm(out; next_gamma, next_beta, next_eps) CHANNELS-FIRST,
 * next_cf_dev [tokens / tokens_per_sample][channels][tokens_per_sample]: what the following AFNO block computes first
 * (`norm1`, fourcastnet.py:182, + the layout rfft2 wants), taken from the accumulators instead of a separate pass over
 * `out`.  tokens_per_sample % 32 == 0 and tokens % tokens_per_sample == 0, else DLWP_ERR_UNSUPPORTED. */
int32_t dlwp_token_mlp_emit_norm_f32(const float* n_dev, const float* resid_dev, const void* packed_dev,
                                     const float* b1_dev, const float* b2_dev, float* out_dev, int64_t tokens,
                                     int32_t channels, int32_t hidden, float ln_eps, const float* next_gamma_dev,
                                     const float* next_beta_dev, float next_eps, float* next_cf_dev,
                                     int64_t tokens_per_sample, void* stream);

/* The whole tail of an AFNO block in ONE launch (reference fourcastnet.py:127 `+ bias`, :187 first skip, :191 `norm2`,
 * :41-57 `Mlp`, :192 second skip -- and, optionally, :182 `norm1` of the NEXT block):
 *   sum = f_cf + l_cf + x;  out = sum + b2 + W2 gelu(W1 LayerNorm(sum) + b1);  next_cf = LayerNorm_next(out) channels-first
 * f_cf_dev (irfft2 output) and l_cf_dev (norm1 output, the AFNO2D `bias` path) CHANNELS-FIRST [batch][channels][tokens_per_sample],
 * x_nhwc_dev / out_nhwc_dev token-major (may alias), packed_dev from dlwp_token_mlp_pack_f32 WITH ln_gamma / ln_beta and
 * merged_layout = 1, next_cf_dev (and its gamma / beta) NULL to skip the last part.  Replaces dlwp_afno_merge_f32 +
 * dlwp_token_mlp_f32 (+ dlwp_layernorm_nhwc_to_nchw_f32 of the next block).  channels == 64, tokens_per_sample % 32 == 0. */
int32_t dlwp_afno_block_tail_f32(const float* f_cf_dev, const float* l_cf_dev, const float* x_nhwc_dev,
                                 const void* packed_dev, const float* b2_dev, float* out_nhwc_dev, int32_t batch,
                                 int64_t tokens_per_sample, int32_t channels, int32_t hidden, float ln_eps,
                                 const float* next_gamma_dev, const float* next_beta_dev, float next_eps,
                                 float* next_cf_dev, void* stream);
/* The same block tail in the "f16x3" product form (exact two-part f16 splits of both operands, three products on the f16
 * matrix instructions instead of six bf16 ones; both weight images fit the LDS, nothing is re-read from global memory).
 * packed_dev from dlwp_token_mlp_pack_f16x3 (same arguments and byte count as dlwp_token_mlp_pack_f32).  fp32-GEMM
 * accuracy; its operands are LayerNorm and GELU outputs, orders of magnitude inside the f16 range. */
int32_t dlwp_token_mlp_pack_f16x3(const float* w1_dev, const float* w2_dev, const float* ln_gamma_dev,
                                  const float* ln_beta_dev, const float* b1_dev, int32_t channels, int32_t hidden,
                                  int32_t merged_layout, void* packed_dev, void* stream);
int32_t dlwp_afno_block_tail_f16x3(const float* f_cf_dev, const float* l_cf_dev, const float* x_nhwc_dev,
                                   const void* packed_dev, const float* b2_dev, float* out_nhwc_dev, int32_t batch,
                                   int64_t tokens_per_sample, int32_t channels, int32_t hidden, float ln_eps,
                                   const float* next_gamma_dev, const float* next_beta_dev, float next_eps,
                                   float* next_cf_dev, void* stream);

/* On-device evaluation sums (reference scripts/evaluate.py:786-821 `compute_metrics` + the
 * de-normalisation of :281-296): out_dev, target_dev [B, K, C, H, W]; climatology_dev [K, C, H, W] or NULL;
 * lat_weights_dev [H] (cos(lat)/mean(cos(lat))); scale_dev [C] (per-variable std) or NULL.
 * sums_dev: double [4, K, C], zeroed by the call:  0: sum w (s (out-tar))^2,  1: sum w s^2 (out-clim)(tar-clim),
 * 2: sum w (s (out-clim))^2,  3: sum w (s (tar-clim))^2  (1-3 only with a climatology).
 * RMSE[k,c] = sqrt(sums[0] / (B_total H W)),  ACC = sums[1] / sqrt(sums[2] sums[3]). */
int32_t dlwp_weighted_error_sums_f32(const float* out_dev, const float* target_dev, const float* climatology_dev,
                                     const float* lat_weights_dev, const float* scale_dev, double* sums_dev,
                                     int32_t batch, int32_t steps, int32_t channels, int32_t height, int32_t width,
                                     void* stream);
/* The same sums ADDED to the contents of sums_dev (the running sums of an evaluation over many batches: evaluate.py:786-821
 * accumulates before it takes the root); the caller zeroes sums_dev once. */
int32_t dlwp_weighted_error_sums_acc_f32(const float* out_dev, const float* target_dev, const float* climatology_dev,
                                         const float* lat_weights_dev, const float* scale_dev, double* sums_dev,
                                         int32_t batch, int32_t steps, int32_t channels, int32_t height, int32_t width,
                                         void* stream);

#ifdef __cplusplus
}
#endif
#endif /* DLWP_HIP_H */
