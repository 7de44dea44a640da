// Launcher interface of the hand-written gfx950 kernels (internal; the public C ABI is
// include/mvd_hip.h).  All tensors are device pointers; activations are token-major
// ("NHWC"): [batch][pixel][channel] bf16.  Every launcher returns 0 or a negative error
// code and records a message retrievable through mvd_last_error().
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "common.h"

// ---------------------------------------------------------------- GEMM / implicit conv
enum { MVD_A_DENSE = 0, MVD_A_CONV3 = 1 };

struct MvdASeg {          // one K segment of the A operand
  const bf16_t* p0;       // source 0 ([rows][c0] dense, or NHWC feature map)
  const bf16_t* p1;       // optional source 1 (channel concat), dense mode only
  int c0, c1;             // channels of each source (c1 == 0: single source)
  int mode;               // MVD_A_DENSE / MVD_A_CONV3
  int ksize;              // K extent: dense c0+c1, conv 9*c0
  int inH, inW;           // conv: stored input spatial size
  int stride;             // conv: 1 or 2
  int ups;                // conv: 1 = nearest 2x upsample fused in front of the conv
  int asym;               // conv, stride 2 only: 1 = zero padding on the bottom/right edge only (diffusers' VAE
                          // Downsample2D(padding=0): F.pad(x, (0,1,0,1)) then a pad-0 conv), 0 = the usual pad 1 all round
};

struct MvdGemmArgs {
  MvdASeg seg[2];
  int nseg;
  const bf16_t* W;        // [N][ldw] bf16, K contiguous
  int ldw;                // row stride of W in elements (>= Ktot)
  int M, N, Ktot;
  int rows_per_batch;     // pixels per batch element (conv geometry + per-batch epilogue vector)
  int outH, outW;         // conv output spatial size
  const float* bias;      // [N] fp32 or null
  const float* rowvec;    // [batch][ld_rowvec] fp32 added per batch (time-embedding projection) or null
  int ld_rowvec;
  const bf16_t* res;      // residual [M][ldres] bf16 or null
  int ldres;
  float alpha;            // out = alpha*(acc + bias + rowvec) + res
  int geglu;              // 1: W rows interleaved (16 value | 16 gate); out = value*gelu(gate), width N/2
  void* out;
  int ldo;
  int out_f32;            // 1: fp32 output, else bf16
  int splitk;             // > 1: K is split over `splitk` work items per tile; raw fp32 partial tiles go to `part`
  float* part;            // [splitk][M][N] fp32 partials (then mvd_launch_splitk_reduce applies the epilogue)
  int w_blocked;          // gemm_sm.hip only: W is stored as [N/32][K/64] blocks of 32 rows x 64 k in LDS-image order (packing.block_weight)
  int splitk_nowait;      // gemm_sm.hip split-K: 1 = the slice that arrives last combines the whole tile and nobody waits (launches that
                          // share the chip with another stream's kernels by design); 0 = the slices rendezvous for a BOUNDED time and each
                          // combines a share; the last arriver combines whatever was not claimed (safe under any residency)
  unsigned int* tile_cnt; // gemm_sm.hip split-K only: one ZEROED word per output tile (arrival count in bits 0-7, one claim bit per
                          // share above: the slices combine the partials in the kernel, no reduce launch; splitk <= 24)
  int dbg;                // probe builds only (-DMVD_PROBE, env MVD_GEMM_DEBUG): bit0 skip the output stores, bit1 skip the MFMAs
  // LayerNorm fold (ping-pong kernels only, see mvd_gemm_ln_fold_ok): A holds the UN-normalised rows x, W holds
  // W.diag(gamma), ln_c1[n] = sum_k W[n][k] (of the bf16 values), bias[n] = sum_k beta[k].W0[n][k] + b[n]; the kernel
  // accumulates the row sums / sums of squares of the A fragments it multiplies and its epilogue applies
  //   out = rstd[m] * (acc - mean[m] * ln_c1[n]) + bias[n]      ( = LayerNorm(x).W0^T + b )
  const float* ln_c1;     // [N] fp32 or null
  float ln_eps;
  // gemm_pp.hip tile walk (set by the launcher, mvd_gemm_pp_walk): 0 = an XCD's workgroups stride through its row-major tile
  // range (all column tiles of a row block side by side); c > 0 = the XCD's rows are walked in GROUPS of c column tiles -- the
  // group's weight panels stay in the XCD's L2 while the rows stream past (short-K GEMMs whose W exceeds the 4 MB L2)
  int walk_cg;
};

int mvd_launch_gemm(const MvdGemmArgs& a, hipStream_t s, int force_cfg = -1);
int mvd_gemm_pick_config(const MvdGemmArgs& a);   // tile config the heuristic gives this problem
// what the calling thread's last mvd_launch_gemm launched (tests assert that the persistent multi-tile path ran)
struct MvdLaunchPlan { int cfg, splitk, tiles, grid, per_cu, nowait; };
extern thread_local MvdLaunchPlan g_mvd_last_gemm;
// 256x320 "ping-pong" kernels (gemm_pp.hip): buffer-addressed LDS-DMA, two wave groups one phase apart.  Used for tile
// configs 6 (GEGLU) and 7 whenever every byte offset fits 32-bit buffer addressing; arguments validated by mvd_launch_gemm.
bool mvd_gemm_pp_applicable(const MvdGemmArgs& a);
int mvd_gemm_pp_walk(const MvdGemmArgs& a);        // column tiles per group of the ping-pong kernel's tile walk (0: row-major)
int mvd_debug_flags();                             // engine.hip: the measurement switches of mvd_debug_set_flags
// true when a problem with a.ln_c1 set can run (one dense source spanning the whole row, no residual / row vector /
// split-K, and a shape the heuristic gives to the ping-pong kernels); the engine falls back to ln_kernel + plain GEMM otherwise
bool mvd_gemm_ln_fold_ok(const MvdGemmArgs& a);
int mvd_launch_gemm_pp(const MvdGemmArgs& a, hipStream_t s);
// ring-pipelined 256x320 experiment (gemm_ring.hip, linked into probe builds only); arguments already validated by mvd_launch_gemm
int mvd_launch_gemm_ring(const MvdGemmArgs& a, hipStream_t s);
// small-M kernels (gemm_sm.hip): one work item per workgroup, NSTAGE-deep LDS-DMA ring, in-kernel split-K combine.
// tile: 0..6 = 64x64, 128x64, 64x128, 128x128, 64x160, 128x160, 64x320; nstage: ring depth (clamped to 2..8 and to 160 KB of LDS)
bool mvd_gemm_sm_applicable(const MvdGemmArgs& a, int tile);
int mvd_launch_gemm_sm(const MvdGemmArgs& a, hipStream_t s, int tile, int nstage);
// whether the engine should give this problem to the small-M kernels, and with which tile / ring depth / split-K
bool mvd_gemm_sm_plan(const MvdGemmArgs& a, int* tile, int* nstage, int* splitk);
// X-stationary short-K kernels (gemm_xs.hip): the activation rows stay in registers as the MFMA B operand (tokens on lanes), the
// weights stream through an LDS ring in the fragment-ordered layout of packing.pack_xs -- `units` 32-row tiles of
// (K / 16 + 1) x 1 KB (the last k-step carries the bias; GEGLU: units alternate gate | value of one 32-channel output tile).
struct MvdXsArgs {
  const bf16_t* x; int ldx;     // [M][ldx] bf16, the first K columns are the operand
  const bf16_t* w;              // packed weights: [units][K / 16 + 1][64][8] bf16
  int M, K, units;
  int geglu;                    // 1: out[M][units * 16] = value * gelu_erf(gate)
  int ln; float ln_eps;         // 1: LayerNorm the rows first (gamma / beta folded into w and its bias k-step: packing.fold_layernorm)
  const bf16_t* res; int ldres; // residual [M][ldres] bf16 or null (not with geglu)
  bf16_t* out; int ldo;
  int csplit;                   // column parts per row block (<= 0: mvd_gemm_xs_pick_csplit)
};
bool mvd_gemm_xs_applicable(const MvdXsArgs& a);
int mvd_gemm_xs_pick_csplit(const MvdXsArgs& a);
int mvd_launch_gemm_xs(const MvdXsArgs& a, hipStream_t s);
// ---------------------------------------------------------------- weight-streaming 3x3 convolution of small maps (conv_ws.hip)
// out[M][N] = conv3x3(x; stride 1, pad 1) (+ dense shortcut rows sc0 | sc1) + bias + rowvec[image] + res, one image's 8-, 16- or
// 32-wide map per 64- / 128-row block; weights host-packed by packing.pack_ws ([N / 16][round][wave][tap][64][8] bf16).
struct MvdWsArgs {
  const bf16_t* x; int B, H, W, C;            // NHWC bf16 input (H, W: the INPUT map)
  int ups;                                    // 1: nearest-neighbour 2x upsampling in front of the convolution (output 2H x 2W; no shortcut)
  const bf16_t* sc0; const bf16_t* sc1; int scc0, scc1;   // dense segment [M][scc0] | [M][scc1] (conv_shortcut fused along K) or null
  const bf16_t* w;                            // packed weights (mvd_conv_ws_packed_elems elements)
  const float* bias;                          // [N]
  const float* rowvec; int ld_rowvec;         // per-image row vector [B][ld_rowvec] (time embedding) or null
  const bf16_t* res; int ldres;               // residual [M][ldres] or null
  bf16_t* out; int ldo;
  int M, N;                                   // M = output pixels = B * H * W (x 4 with ups)
  int variant;                                // 0: the launcher's choice; 1 / 2: force 64- / 128-pixel blocks (tests, probes; conv_ws.hip)
};
bool mvd_conv_ws_applicable(const MvdWsArgs& a);
size_t mvd_conv_ws_packed_elems(int C, int sc, int N);
int mvd_launch_conv_ws(const MvdWsArgs& a, hipStream_t s);
#define MVD_OP_SPLITK_COUNTERS 4096   // tile counters behind the partials of an mvd_op_linear / mvd_op_conv3x3 split-K workspace
// sum the split-K partials and apply the GEMM epilogue (bias, row vector, alpha, residual) -> out
int mvd_launch_splitk_reduce(const MvdGemmArgs& a, hipStream_t s);
// split factor the engine should use for this problem (1 = none); needs splitk*M*N floats of workspace
int mvd_gemm_pick_splitk(const MvdGemmArgs& a);

// ---------------------------------------------------------------- attention (head_dim 64)
struct MvdAttnProblem {
  const bf16_t* q; const bf16_t* k; const bf16_t* v; bf16_t* o;
  int ldq, ldk, ldv, ldo;         // row strides in elements
  int64_t bsq, bsk, bsv, bso;     // batch strides in elements
  int nq, nk;                     // tokens per batch element
};
struct MvdAttnArgs {
  MvdAttnProblem p[2];            // up to two independent problems in one launch
  int nprob;
  int batch, heads;
  float scale;                    // softmax scale (1/sqrt(64))
  int prescaled;                  // 1: q is already multiplied by scale*log2(e) (host-folded into to_q); scale unused
  int dbg;                        // probe builds only (-DMVD_PROBE, env MVD_ATTN_DBG): ablation bits of the ping-pong kernel
  // split-KV (batch 1: too few (head, query block) pairs to fill the chip): the keys of every problem are cut into `nsplit`
  // ranges, one workgroup each; partial results (normalised O as bf16 + (running max, denominator) per query) go to
  // `split_ws` and the workgroup that arrives last on the (problem, batch, head, query block) counter merges them.
  int nsplit;                     // <= 1: off
  void* split_ws;                 // mvd_attention_split_ws_bytes(a) bytes
  unsigned int* split_cnt;        // mvd_attention_split_counters(a) ZEROED counters
};
// split the engine's heuristic gives this launch (1 = none) and what it then needs
int mvd_attention_pick_split(const MvdAttnArgs& a);
size_t mvd_attention_split_ws_bytes(const MvdAttnArgs& a, int nsplit);
int mvd_attention_split_counters(const MvdAttnArgs& a);
int mvd_launch_attention(const MvdAttnArgs& a, hipStream_t s);

// ---------------------------------------------------------------- normalisation
// GroupNorm over NHWC with optional 2-source channel concat; y = gn(x)*gamma+beta, optional SiLU.
// ws: fp32 scratch of at least batch*MVD_GN_MAXCHUNK*groups*2 floats.
#define MVD_GN_MAXCHUNK 256
int mvd_launch_groupnorm(const bf16_t* x0, const bf16_t* x1, int c0, int c1, int batch, int hw, int groups,
                         float eps, const float* gamma, const float* beta, int silu, bf16_t* y, float* ws,
                         hipStream_t s);
int mvd_launch_layernorm(const bf16_t* x, int rows, int c, float eps, const float* gamma, const float* beta,
                         bf16_t* y, hipStream_t s);
// Q2 reference normalisation: per pixel over (batch, channel), unbiased std, clamp 1e-6, *0.5
int mvd_launch_refnorm(const bf16_t* x, int batch, int hw, int c, bf16_t* y, hipStream_t s);
int mvd_launch_refstats(const bf16_t* x, int batch, int hw, int c, float* stats /*[hw][2]: mean, M2*/, hipStream_t s);
int mvd_launch_refapply(const bf16_t* x, int batch, int hw, int c, const float* mean_k /*[hw][2]*/, bf16_t* y, hipStream_t s);

// ---------------------------------------------------------------- elementwise / small ops
// NCHW fp32 -> NHWC bf16 with optional FiLM (scale/shift fp32 [batch][c])
int mvd_launch_nchw_to_nhwc(const float* x, int batch, int c, int hw, const float* scale, const float* shift,
                            int ld_ss, bf16_t* y, hipStream_t s);
int mvd_launch_film(const bf16_t* x, int batch, int hw, int c, const float* scale, const float* shift, int ld_ss,
                    bf16_t* y, hipStream_t s);
// conv_in: 3x3 pad 1, tiny Cin (<=8) -> Cout, NHWC bf16 in/out, weights fp32 [Cout][3][3][Cin]
int mvd_launch_conv_in(const bf16_t* x, int batch, int h, int w, int cin, const float* wt, const float* bias,
                       int cout, bf16_t* y, hipStream_t s);
// conv_out: 3x3 pad 1, C -> tiny Cout (<=8), NHWC bf16 in, NCHW fp32 out, weights bf16 [Cout][3][3][C]
int mvd_launch_conv_out(const bf16_t* x, int batch, int h, int w, int c, const bf16_t* wt, const float* bias,
                        int cout, float* y, hipStream_t s);
// conv_in as a GEMM: NCHW fp32 (+FiLM) -> im2col [batch*h*w][64] bf16, k = tap*c + ch, zero padded (c <= 7)
int mvd_launch_im2col_in(const float* x, int batch, int c, int h, int w, const float* scale, const float* shift, int ld_ss,
                         bf16_t* y, hipStream_t s);
// fp32 -> bf16 copy
int mvd_launch_f32_to_bf16(const float* x, int64_t n, bf16_t* y, hipStream_t s);
int mvd_launch_nhwc_to_nchw_f32(const bf16_t* x, int batch, int hw, int c, float* y, hipStream_t s);

// skinny fp32 linear: y[b][n] = act_out( sum_k act_in(x[b][k]) * W[n][k] + bias[n] ), b < 256
// W is fp32 (wbf16 == 0) or bf16 (wbf16 == 1).  act: 0 none, 1 SiLU
int mvd_launch_skinny_linear(const float* x, int ldx, int batch, int k, const void* w, int wbf16, const float* bias,
                             int n, int act_in, float* y, int ldy, hipStream_t s);
// grouped forms (the camera modulators batched): output feature o belongs to segment g = the first with o < seg.end[g]; its input
// row is x + g * xseg (every segment has the same K).  mvd_launch_layernorm_f32: row r uses gamma/beta + (r % nseg) * c.
struct MvdSegTable { int n; int end[16]; };
int mvd_launch_skinny_linear_grouped(const float* x, int ldx, int xseg, int batch, int k, const float* w, const float* bias, int n,
                                     const MvdSegTable& seg, float* y, int ldy, hipStream_t s);
// FiLM post-processing for all segments at once: raw [batch][seg.end[n-1]] with segment g = [scale_raw(dim_g) | shift_raw(dim_g)];
// outputs for segment g at out + out_rows * seg.end[g-1]: scale [out_rows][dim_g] then shift [out_rows][dim_g]
int mvd_launch_film_params_grouped(const float* raw, int batch, const MvdSegTable& seg, float strength, float* out, int out_rows,
                                   hipStream_t s);
// row LayerNorm in fp32 with optional SiLU
int mvd_launch_layernorm_f32(const float* x, int rows, int c, float eps, const float* gamma, const float* beta,
                             int silu, float* y, hipStream_t s, int nseg = 1);
// sinusoidal timestep embedding [cos|sin] (diffusers flip_sin_to_cos=True, freq_shift 0)
int mvd_launch_timestep_embedding(const float* t, int batch, int dim, float* y, hipStream_t s);
// camera front end: relative pose + Fourier features: cams [batch][rows(3|4)][4] fp32 ->
// rflat [batch][9], enc [batch][6*nfreq] (before the random projection)
int mvd_launch_camera_features(const float* src, const float* tgt, int batch, int cam_rows, int nfreq,
                               float max_freq, float* rflat, float* enc, hipStream_t s);
// FiLM post-processing: raw [batch][2*dim] -> scale = 2*sigmoid(raw[:dim])*k, shift = raw[dim:]*k; writes out_rows
// (>= batch) rows, row r from input row r % batch (camera batch broadcast over a larger sample batch)
int mvd_launch_film_params(const float* raw, int batch, int dim, float strength, float* scale, float* shift, int out_rows,
                           hipStream_t s);

// FiLM on an NCHW fp32 tensor (standalone CameraEncoder.apply_modulation); scale/shift [batch][c]
int mvd_launch_film_nchw_f32(const float* x, int batch, int c, int hw, const float* scale, const float* shift, float* y,
                             hipStream_t s);

void mvd_set_error(const char* fmt, ...);
