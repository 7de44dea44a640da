/* mvd_hip.h -- C ABI of libmvd_hip.so: the MI355X (gfx950) implementation of the
 * pananananas/MVD denoising hot path.
 *
 * The reference has no FFI of its own: its boundary is the Python object protocol of
 * MultiViewUNet.forward (/root/reference/src/models/mvd_unet.py:179-191) which calls
 * diffusers' UNet2DConditionModel (mvd_unet.py:318-326, image_encoder.py:105-110),
 * ImageCrossAttentionProcessor.__call__ (attention.py:48-188) and
 * CameraEncoder.{encode_cameras,apply_modulation} (camera_encoder.py:160-255).
 * This header is what a binding for that path binds instead (ctypes stub: INTEGRATION.md).
 *
 * Conventions
 *  - every pointer is a DEVICE pointer unless its name ends in _host;
 *  - no torch / C++ types; sizes are explicit; the caller owns every buffer;
 *  - work is issued asynchronously on the hipStream_t passed as `void* stream`;
 *  - every function returns 0 on success, <0 on error; mvd_last_error() returns the
 *    message of the calling thread's last failure;
 *  - one calling thread per engine handle.
 */
#ifndef MVD_HIP_H
#define MVD_HIP_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MVD_MAX_LEVELS 4

/* UNet2DConditionModel config subset (diffusers 0.32.2 names; SD-2.1 values in comments).
 * Replaces: UNet2DConditionModel.from_pretrained(...).config  (mvd_unet.py:46-53). */
typedef struct {
  int in_channels;                         /* 4 */
  int out_channels;                        /* 4 */
  int num_levels;                          /* 4 */
  int block_out_channels[MVD_MAX_LEVELS];  /* 320 640 1280 1280 */
  int num_heads[MVD_MAX_LEVELS];           /* 5 10 20 20 ("attention_head_dim") */
  int layers_per_block;                    /* 2 */
  int cross_attention_dim;                 /* 1024 */
  int norm_num_groups;                     /* 32 */
  float norm_eps;                          /* 1e-5 */
  /* MVD wrapper (mvd_unet.py:23-36) */
  int cam_output_dim;                      /* 1024 */
  int cam_hidden_dim;                      /* 512 */
  int simple_cam_encoder;                  /* 0 */
  float cam_modulation_strength;           /* 0.2 */
} mvd_config_t;

typedef struct mvd_engine mvd_engine_t;

const char* mvd_last_error(void);

/* ---- engine lifetime -------------------------------------------------------------- */
int mvd_engine_create(const mvd_config_t* cfg, mvd_engine_t** out);
int mvd_engine_destroy(mvd_engine_t* e);

/* Packed-weight registration.  `set` 0 = base_unet (+ adapter processors + camera encoder),
 * 1 = image_encoder.unet.  Slot names and layouts: DESIGN.md "Weight slots".
 * dtype: 0 = fp32, 1 = bf16.  The engine keeps the pointer; the caller keeps the memory alive.
 * Replaces: nn.Module.load_state_dict / .to(device) of mvd_unet.py:164-177, infer.py:67-76. */
int mvd_engine_set_weight(mvd_engine_t* e, int set, const char* slot, const void* ptr, int64_t numel, int dtype);
int mvd_engine_clear_weights(mvd_engine_t* e, int set);

/* Workspace: bytes needed for one forward at the given shape (dry run of the schedule). */
int64_t mvd_engine_workspace_bytes(mvd_engine_t* e, int batch, int height, int width, int text_len, int ref_batch);
/* Persistent bytes for the cached reference K/V (and kept feature maps when keep_features). */
int64_t mvd_engine_refcache_bytes(mvd_engine_t* e, int ref_batch, int height, int width, int keep_features);
int mvd_engine_bind_workspace(mvd_engine_t* e, void* ws, int64_t ws_bytes, void* refcache, int64_t refcache_bytes);

/* ---- the hot path ------------------------------------------------------------------ */
enum {
  MVD_USE_CAMERA = 1,       /* use_camera_conditioning and target_camera is not None (mvd_unet.py:241) */
  MVD_USE_IMAGE = 2,        /* use_image_conditioning and source_image_latents is not None (:269)      */
  MVD_REUSE_REF = 4,        /* reuse reference K/V computed by a previous call (inputs step-invariant)  */
  MVD_KEEP_FEATURES = 8     /* keep the 16 encoder feature maps for mvd_engine_get_feature               */
};

/* One MultiViewUNet.forward (mvd_unet.py:179-338).  All tensors fp32, contiguous. */
typedef struct {
  int batch;                 /* rows of `sample` (2B under classifier-free guidance)                    */
  int height, width;         /* latent spatial size (64x64 for 512x512 images)                           */
  int text_len;              /* 77                                                                       */
  const float* sample;       /* [batch][in_channels][H][W]                                               */
  const float* timesteps;    /* [batch] (already expanded; ints as floats)                               */
  const float* text;         /* [batch][text_len][cross_attention_dim] (already repeated, :233-237)      */
  const float* source_camera;/* [batch][cam_rows][4] or NULL                                             */
  const float* target_camera;
  int cam_rows;              /* 3 or 4 (Q8)                                                              */
  int cam_batch;             /* rows of the camera tensors: 0 or `batch`, or a divisor of `batch` (1 = torch's
                                (1,C,1,1) broadcast; under CFG the 2B latents share the B cameras: row b uses
                                camera b % cam_batch) -- pipeline.py:141-152                                   */
  const float* fourier_proj; /* [cam_output_dim][6*((cam_output_dim/2)/3)] the per-call random matrix Q1 */
  const float* source_latents;/* [ref_batch][in_channels][H][W] or NULL                                  */
  const float* encoder_text; /* [ref_batch][text_len][xdim]: text rows chosen per mvd_unet.py:278-285     */
  int ref_batch;
  int flags;
  float* out;                /* [batch][out_channels][H][W]                                              */
} mvd_forward_args_t;

int mvd_unet_forward(mvd_engine_t* e, const mvd_forward_args_t* args, void* stream);

/* Global Q2 statistics (SURVEY.md 8e mode ii; optional).  The adapter normalises the reference features with statistics
 * over (batch, channel) per pixel (attention.py:95-103), so a batch sharded over GPUs sees per-shard statistics unless the
 * shards exchange them.  The reference pass is therefore also available in two halves:
 *   mvd_engine_reference_encode   runs the image-encoder pass for `args` (uses height, width, text_len, ref_batch,
 *                                 source_latents, encoder_text), keeps the 16 raw feature maps in the reference cache (bind it
 *                                 with keep_features = 1) and writes the LOCAL per-pixel pair (mean, M2 = sum of squared
 *                                 deviations from that mean) over ref_batch x C of every feature to
 *                                 local_stats[mvd_engine_reference_pixels()][2], features concatenated in encoder order;
 *   (the host merges the ranks' pairs -- one all-gather of 215 KB at 64x64 -- into mean and k = 0.5 / max(std, 1e-6))
 *   mvd_engine_reference_finish   normalises with mean_k[pixels][2] = (mean, k) and projects to the adapter K/V.
 * Afterwards mvd_unet_forward with MVD_USE_IMAGE | MVD_REUSE_REF runs the main pass on that reference.  With one rank the
 * three calls reproduce the ordinary forward.  There is no counterpart in the reference (its DDP replicas normalise
 * locally); mvd_amd.distributed.merge_reference_stats is the host side. */
int64_t mvd_engine_reference_pixels(mvd_engine_t* e, int height, int width);
int mvd_engine_reference_encode(mvd_engine_t* e, const mvd_forward_args_t* args, float* local_stats, void* stream);
int mvd_engine_reference_finish(mvd_engine_t* e, const float* mean_k, void* stream);

/* hipGraph replay of whole forwards.  When enabled, a mvd_unet_forward whose argument block (every pointer, shape and flag),
 * bound buffers and starting reference-cache state were seen before is replayed as ONE hipGraphLaunch: the first call with
 * such a key runs normally, the second is stream-captured and instantiated, later ones replay (up to 8 graphs are kept;
 * registering weights, re-binding buffers or disabling the switch drops them).  The caller must therefore pass the SAME
 * device buffers every step and refresh their contents in place, on a non-default stream (a capture on the legacy stream is
 * illegal).  Ignored while profiling is on.  Mirrors SURVEY.md section 7 step 7; there is nothing like it in the reference. */
int mvd_engine_set_graph(mvd_engine_t* e, int enable);

/* N4 (training.py:60-65, config/train_config.yaml:43): when the image encoder's UNet weights are identical to the
 * base UNet's (frozen base), the encoder pass can read weight set 0 and set 1 need not be registered at all. */
int mvd_engine_share_encoder_weights(mvd_engine_t* e, int enable);

/* Per-kernel-class timing with HIP events on the launch stream (measurement only; off by default).
 * classes = kernels: 0..5, 12 lock-step GEMM/conv tile configs (gemm.hip); ping-pong 256x320 kernels (gemm_pp.hip): 7 dense,
 * 13 implicit-GEMM 3x3 convolution, 14 split-K, 6 GEGLU; 8..11 attention (1,2,4,8 waves); 16 groupnorm; 17 layernorm. */
int mvd_engine_set_profiling(mvd_engine_t* e, int enable);
int mvd_engine_profile_summary(mvd_engine_t* e, int cap, int* cls, int* launches, double* ms, double* flops, double* bytes);
/* Per-shape text table ("cls M N K tag launches ms tflops" per line) of the launches recorded since profiling was
 * enabled, written to a HOST buffer; call before mvd_engine_profile_summary (which resets the records).
 * Returns the number of bytes written (<0 on error). */
int mvd_engine_profile_shapes(mvd_engine_t* e, char* buf_host, int cap);

/* Number / shape / copy-out (NCHW fp32) of the encoder feature maps (image_encoder.py:36-84). */
int mvd_engine_num_features(mvd_engine_t* e);
int mvd_engine_feature_shape(mvd_engine_t* e, int idx, int* channels, int* height, int* width);
int mvd_engine_get_feature(mvd_engine_t* e, int idx, float* out_nchw, void* stream);
/* CameraEncoder.encode_cameras (camera_encoder.py:160-176): cameras [batch][cam_rows][4] -> out_emb [batch][cam_output_dim] */
int mvd_engine_encode_cameras(mvd_engine_t* e, const float* source_camera, const float* target_camera, int cam_rows,
                              int batch, const float* fourier_proj, float* out_emb, void* stream);
/* CameraEncoder.apply_modulation_to_tensor (camera_encoder.py:207-255) on x [batch][channels][hw] fp32.
 * Returns 1 (and writes nothing) when `name` is not a modulator -- the reference's silent identity (Q3). */
int mvd_engine_apply_modulation(mvd_engine_t* e, const char* name, const float* emb, int batch, const float* x_nchw,
                                int channels, int hw, float* out_nchw, void* stream);
/* Camera embedding of the last forward ([batch][cam_output_dim] fp32), for parity tests. */
int mvd_engine_get_camera_embedding(mvd_engine_t* e, float* out, void* stream);

/* ---- operator-level entry points (same kernels the engine schedules; used by tests) -- */
/* out[M][N] = alpha*(A[M][K] . W[N][K]^T + bias + rowvec[m/rows_per_batch]) + res ; bf16 A/W/res */
int mvd_op_linear(const void* a, const void* a2, int k1, int k2, const void* w, const float* bias, const float* rowvec,
                  int ld_rowvec, int rows_per_batch, const void* res, float alpha, int geglu, void* out, int out_f32,
                  int m, int n, int force_cfg, int splitk, float* splitk_ws /* splitk*m*n floats when splitk > 1 */,
                  void* stream);
/* out[M][N] = LayerNorm(x[M][K]; gamma, beta, eps) . W^T + b in ONE kernel (geglu = 1: followed by value*gelu(gate), width
 * N/2, rows interleaved as for mvd_op_linear).  w_folded = W.diag(gamma) in bf16, c1[n] = sum_k w_folded[n][k] (of the bf16
 * values, fp32), c2[n] = sum_k beta[k].W[n][k] + b[n]: mvd_amd/packing.py::fold_layernorm.  Shapes: K <= 1280, N % 320 == 0
 * and enough rows for the 256x320 tile grid (M*N >= 200 tiles); anything else returns an error (the engine then runs the
 * LayerNorm kernel + mvd_op_linear). */
int mvd_op_ln_linear(const void* x, int k, const void* w_folded, const float* c1, const float* c2, float eps, int geglu,
                     void* out, int m, int n, void* stream);
/* The X-stationary short-K form (mvd_amd/csrc/gemm_xs.hip; K = 320, the 64x64 level of SD-2.1 at many rows):
 *   out[M][units*32] = LN?(x[M][K]) . W^T + b (+ res)        geglu = 1: out[M][units*16] = value * gelu_erf(gate)
 * x rows stay in registers, w_packed is the fragment-ordered weight stream of mvd_amd/packing.py::pack_xs (bias included;
 * ln = 1: packed from the LayerNorm-folded pair of fold_layernorm).  csplit <= 0: heuristic column split.
 * Replaces, for those shapes, the Linear / GEGLU projections of diffusers' BasicTransformerBlock that
 * /root/reference/src/models/mvd_unet.py:318-326 reaches through UNet2DConditionModel.forward. */
int mvd_op_linear_xs(const void* x, int ldx, const void* w_packed, int m, int k, int units, int geglu, int ln, float ln_eps,
                     const void* res, int ldres, void* out, int ldo, int csplit, void* stream);
/* The weight-streaming form of a resnet 3x3 convolution on ONE image's small map (mvd_amd/csrc/conv_ws.hip: batch 1, the 8x8 and
 * 16x16 levels -- 15-65 MB of weights for a few GFLOP):  out[B*H*W][n] = conv3x3(x; stride 1, pad 1) (+ [sc | sc2] . Wsc^T) + bias
 * + rowvec[image] + res.  w in {8, 16, 32}, h*w % 64 == 0, B*h*w <= 1024, c % 128 == 0, sc_c1 / sc_c2 % 128 == 0, n % 16 == 0;
 * w_packed from mvd_amd/packing.py::pack_ws; variant 0 = the launcher's choice of block height (1: 64 pixels, 2: 128 pixels on a 16-wide map);
 * + 16: nearest-neighbour 2x upsampling in front of the convolution (diffusers' Upsample2D: output 2h x 2w, 16 or 32 wide, no shortcut).
 * No workspace, no split-K: each workgroup streams its 16-channel weight panel once.
 * Replaces, for those shapes, the Conv2d of diffusers' ResnetBlock2D that /root/reference/src/models/mvd_unet.py:318-326 reaches
 * through UNet2DConditionModel.forward (the reference's own batch-1 loop: /root/reference/infer.py:111-122). */
int mvd_op_conv3x3_ws(const void* x, int batch, int h, int w, int c, const void* w_packed, const float* bias, const float* rowvec,
                      int ld_rowvec, const void* res, const void* sc, const void* sc2, int sc_c1, int sc_c2, void* out, int n,
                      int variant, void* stream);
/* 3x3 conv (pad 1) on NHWC bf16 as implicit GEMM; optional fused 1x1 shortcut on (sc, sc2).  asym_pad = 1 (stride 2 only):
 * zero padding on the bottom/right edge only -- diffusers' VAE Downsample2D(padding=0). */
int mvd_op_conv3x3(const void* x, int batch, int in_h, int in_w, int cin, int stride, int upsample, int asym_pad, const void* w,
                   const float* bias, const float* rowvec, int ld_rowvec, const void* res, const void* sc, const void* sc2,
                   int sc_c1, int sc_c2, void* out, int cout, int force_cfg, int splitk, float* splitk_ws, void* stream);
/* softmax(scale * q.k^T).v per head of 64 channels.  scale == 0 selects the engine's form: q is already multiplied
 * by softmax_scale * log2(e) (the packed to_q / to_q_ref weight rows carry that factor, DESIGN.md "Weight slots"). */
int mvd_op_attention(const void* q, const void* k, const void* v, void* o, int batch, int heads, int nq, int nk, int ldq,
                     int ldk, int ldv, int ldo, float scale, void* stream);
/* Split-KV form (batch 1: too few (head, query block) pairs to fill the chip): the keys are cut into nsplit <= 8 ranges, one
 * workgroup each, merged in the kernel by the last workgroup to arrive.  Prescaled queries only (the scale == 0 form). */
int64_t mvd_op_attention_split_ws_bytes(int batch, int heads, int nq, int nsplit);
int mvd_op_attention_split(const void* q, const void* k, const void* v, void* o, int batch, int heads, int nq, int nk, int ldq,
                           int ldk, int ldv, int ldo, int nsplit, void* ws, void* stream);
int mvd_op_groupnorm(const void* x0, const void* x1, int c0, int c1, int batch, int hw, int groups, float eps,
                     const float* gamma, const float* beta, int silu, void* y, float* ws, void* stream);
int mvd_op_layernorm(const void* x, int rows, int c, float eps, const float* gamma, const float* beta, void* y,
                     void* stream);
int mvd_op_refnorm(const void* x, int batch, int hw, int c, void* y, void* stream);
int mvd_op_film(const void* x, int batch, int hw, int c, const float* scale, const float* shift, void* y, void* stream);
int mvd_op_conv_in(const void* x, int batch, int h, int w, int cin, const float* wt, const float* bias, int cout, void* y,
                   void* stream);
int mvd_op_conv_out(const void* x, int batch, int h, int w, int c, const void* wt, const float* bias, int cout, float* y,
                    void* stream);
int mvd_op_nchw_to_nhwc(const float* x, int batch, int c, int hw, const float* scale, const float* shift, void* y,
                        void* stream);
int mvd_op_nhwc_to_nchw(const void* x, int batch, int hw, int c, float* y, void* stream);
int mvd_op_f32_to_bf16(const float* x, int64_t n, void* y, void* stream);
/* One fp32 linear layer of the camera / time MLPs (camera_encoder.py:31-85; diffusers TimestepEmbedding): y[b][o] =
 * sum_k act(x[b][k]) W[o][k] + bias[o]; W fp32 [n][k], or bf16 with wbf16 = 1; act_in = 1 applies SiLU to the inputs. */
int mvd_op_skinny_linear(const float* x, int ldx, int batch, int k, const void* w, int wbf16, const float* bias, int n, int act_in,
                         float* y, int ldy, void* stream);
int mvd_gemm_num_configs(void);
/* What the calling thread's last GEMM/conv launch did: out[5] = {tile config, split-K, work items, workgroups, workgroups
 * per CU}; last attention launch: out[2] = {waves per workgroup, workgroups}.  Parity tests assert with these that the
 * persistent multi-tile path (work items > workgroups) is what ran at the benchmarked shapes. */
int mvd_debug_last_gemm_plan(int* out);
int mvd_debug_last_attention_plan(int* out);
/* 1 when the calling thread's last small-M split-K launch used the no-wait combine (requested, or chosen because the grid
 * cannot be resident at once), else 0. */
int mvd_debug_last_gemm_nowait(void);
/* The split-K factor the engine's schedule picks for a GEMM/conv of this size (1 = none). */
int mvd_debug_pick_splitk(int m, int n, int k, int geglu);
/* The same for a 3x3 convolution run as an implicit GEMM (k = 9 * Cin + fused shortcut channels). */
int mvd_debug_pick_splitk_conv(int m, int n, int k);
/* Small-M kernels (gemm_sm.hip, the batch-1 path): force_cfg = 100 + 10 * tile + ring depth in mvd_op_linear / mvd_op_conv3x3
 * (tiles 0..6 = 64x64, 128x64, 64x128, 128x128, 64x160, 128x160, 64x320; + 1000: the weight is in the blocked LDS-image
 * layout of mvd_amd.packing.block_weight).  With splitk > 1 these kernels combine the slices themselves: splitk_ws then
 * needs splitk*m*n floats + 4096 further 4-byte words (tile arrival counters, zeroed by the call). */
int mvd_gemm_sm_num_tiles(void);
/* Measurement hook: log2(waves per attention workgroup) for every later launch of this process; -1 = heuristic. */
int mvd_debug_set_attention_nw(int nw_log2);
/* Measurement / bisection switches of the engine's schedule: bit 0 no LayerNorm fold through the small-M kernels, bit 1 the
 * small-M kernels never split K, bit 2 small-M kernels off, bit 3 no split-KV attention, bit 4 the reference-encoder pass on the
 * caller's stream (one stream), bit 5 the single-stream launch policy (split-K / tile choice) also while two streams run,
 * bit 6 the side stream at default instead of highest priority (read when the stream is created).  0 = the product's behaviour. */
int mvd_debug_set_flags(int flags);

/* ---- denoising-loop helpers either side of the UNet, fp32 latents ------ */
/* DDPM ancestral step, coefficients from mvd_amd/scheduler.py (diffusers DDPMScheduler.step algebra):
 *   x0 = c0*model_out + c1*sample ; out = c2*x0 + c3*sample + sigma*noise.   Replaces pipeline.py:161. */
int mvd_op_ddpm_step(const float* model_out, const float* sample, const float* noise, float c0, float c1, float c2,
                     float c3, float sigma, float* out, int64_t n, void* stream);
/* classifier-free guidance combine of [uncond | cond] stacked on the batch dim (pipeline.py:156-158) */
int mvd_op_cfg_combine(const float* uncond_cond, float guidance_scale, float* out, int64_t n_half, void* stream);

/* ---- AutoencoderKL (SD-2.1 VAE) either side of the loop ------------------------------------ */
/* Replaces: vae.encode(x).latent_dist (pipeline.py:115) and vae.decode(z).sample (pipeline.py:171-176) of diffusers'
 * AutoencoderKL.  Slot names / layouts: DESIGN.md "VAE weight slots" (mvd_amd/vae.py packs a diffusers state dict). */
typedef struct {
  int in_channels;                         /* 3 */
  int latent_channels;                     /* 4 */
  int num_levels;                          /* 4 */
  int block_out_channels[MVD_MAX_LEVELS];  /* 128 256 512 512 */
  int layers_per_block;                    /* 2 */
  int norm_num_groups;                     /* 32 */
  float norm_eps;                          /* 1e-6 */
} mvd_vae_config_t;
typedef struct mvd_vae mvd_vae_t;
int mvd_vae_create(const mvd_vae_config_t* cfg, mvd_vae_t** out);
int mvd_vae_destroy(mvd_vae_t* v);
int mvd_vae_set_weight(mvd_vae_t* v, const char* slot, const void* ptr, int64_t numel, int dtype);
/* bytes for one encode (decode = 0: height/width of the IMAGE) or decode (decode = 1: height/width of the LATENT) */
int64_t mvd_vae_workspace_bytes(mvd_vae_t* v, int batch, int height, int width, int decode);
int mvd_vae_bind_workspace(mvd_vae_t* v, void* ws, int64_t ws_bytes);
/* image [batch][in_channels][H][W] fp32 in [-1,1] -> moments [batch][2*latent][H/f][W/f] fp32 = (mean | logvar) */
int mvd_vae_encode(mvd_vae_t* v, const float* image_nchw, int batch, int height, int width, float* moments, void* stream);
/* latents [batch][latent][h][w] fp32 (already divided by the scaling factor) -> image [batch][in_channels][f*h][f*w] fp32 */
int mvd_vae_decode(mvd_vae_t* v, const float* latents_nchw, int batch, int height, int width, float* image, void* stream);
/* DiagonalGaussianDistribution.sample(): out = (mean + exp(0.5*clamp(logvar,-30,20)) * noise) * scale; noise/out [batch][c][hw] */
int mvd_op_gaussian_sample(const float* moments, const float* noise, int batch, int channels, int hw, float scale, float* out,
                           void* stream);

#ifdef __cplusplus
}
#endif
#endif
