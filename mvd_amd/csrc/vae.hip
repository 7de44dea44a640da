// AutoencoderKL (the SD-2.1 VAE) encode / decode on the hot path's kernels -- SURVEY.md 8f row N3: the step either side
// of the denoising loop (/root/reference/src/models/pipeline.py:115-116 `vae.encode(x).latent_dist.sample()`,
// :171-176 `vae.decode(z).sample`).  diffusers-0.32.2 semantics restated (the reference reaches the VAE only through
// diffusers' StableDiffusionPipeline; parity unpinned like the UNet, oracle/vae.py):
//
//   Encoder  conv_in 3->C0 | per level: R x ResnetBlock2D (no time embedding), Downsample2D(padding=0) = zero pad
//            bottom/right + 3x3 stride-2 conv | mid: resnet, attention (1 head of C channels, GroupNorm, residual),
//            resnet | GroupNorm + SiLU | conv_out C->2*latent | quant_conv 1x1
//   Decoder  post_quant_conv 1x1 | conv_in latent->C | mid | per level (reversed): (R+1) resnets, nearest-2x + 3x3 conv |
//            GroupNorm + SiLU | conv_out C0->3
//
// Everything above 8 channels is NHWC bf16 through the implicit-GEMM conv / MFMA GEMM kernels (gemm.hip, gemm_pp.hip);
// the mid-block attention has head_dim = C (512), outside the 64-wide flash kernel, and runs as GEMMs per image:
// S = q.k^T / sqrt(C) (fp32) -> row softmax -> P.V with V^T produced directly by a GEMM with swapped roles (W_v . x^T),
// the value bias added after the product (softmax rows sum to one).
#include <math.h>
#include <stdio.h>
#include <string.h>

#include <string>
#include <unordered_map>
#include <vector>

#include "../../include/mvd_hip.h"
#include "kernels.h"

namespace {

struct VWeight { const void* p; int64_t numel; int dtype; };

struct VArena {
  char* base = nullptr;
  size_t cap = 0, off = 0, high = 0;
  bool dry = false;
  void* alloc(size_t bytes) {
    off = (off + 255) & ~size_t(255);
    void* p = dry ? (void*)(uintptr_t)(0x1000 + off) : (void*)(base + off);
    off += bytes;
    if (off > high) high = off;
    return p;
  }
};

struct VAct { bf16_t* p = nullptr; int B = 0, H = 0, W = 0, C = 0; int hw() const { return H * W; } int rows() const { return B * H * W; } };

// fp32 row softmax -> bf16 probabilities; one workgroup per row
__global__ __launch_bounds__(256) void softmax_rows_kernel(const float* __restrict__ s, int n, bf16_t* __restrict__ p) {
  __shared__ float red[4];
  const float* row = s + (size_t)blockIdx.x * n;
  bf16_t* out = p + (size_t)blockIdx.x * n;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  float mx = -3.0e38f;
  for (int i = threadIdx.x; i < n; i += 256) mx = fmaxf(mx, row[i]);
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o, 64));
  if (lane == 0) red[wave] = mx;
  __syncthreads();
  mx = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
  __syncthreads();
  float sum = 0.f;
  for (int i = threadIdx.x; i < n; i += 256) sum += __expf(row[i] - mx);
  sum = wave_sum(sum);
  if (lane == 0) red[wave] = sum;
  __syncthreads();
  const float inv = 1.0f / (red[0] + red[1] + red[2] + red[3]);
  for (int i = threadIdx.x; i < n; i += 256) out[i] = f2bf(__expf(row[i] - mx) * inv);
}

// 1x1 convolution on a tiny channel count (<= 8 in and out), NCHW fp32 -> NCHW fp32 (quant_conv / post_quant_conv)
__global__ void pointwise_small_kernel(const float* __restrict__ x, int cin, int cout, int hw, const float* __restrict__ w,
                                       const float* __restrict__ bias, float* __restrict__ y, long total) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;     // over (b, p)
  if (i >= total) return;
  const int p = i % hw;
  const long b = i / hw;
  float v[8];
  for (int c = 0; c < cin; ++c) v[c] = x[((size_t)b * cin + c) * hw + p];
  for (int o = 0; o < cout; ++o) {
    float a = bias[o];
    for (int c = 0; c < cin; ++c) a = fmaf(w[o * cin + c], v[c], a);
    y[((size_t)b * cout + o) * hw + p] = a;
  }
}

// DiagonalGaussianDistribution.sample(): mean + exp(0.5 * clamp(logvar, -30, 20)) * noise, times `scale`
__global__ void gaussian_sample_kernel(const float* __restrict__ mom, const float* __restrict__ noise, int c, int hw, float scale,
                                       float* __restrict__ out, long total) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;     // over (b, c, p)
  if (i >= total) return;
  const long chw = (long)c * hw;
  const long b = i / chw, r = i - b * chw;
  const float mean = mom[b * 2 * chw + r];
  const float lv = fminf(fmaxf(mom[b * 2 * chw + chw + r], -30.f), 20.f);
  out[i] = (mean + __expf(0.5f * lv) * noise[i]) * scale;
}

int vcheck(const char* what) {
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) { mvd_set_error("%s launch: %s", what, hipGetErrorString(e)); return -3; }
  return 0;
}

}  // namespace

struct mvd_vae {
  mvd_vae_config_t cfg;
  std::unordered_map<std::string, VWeight> w;
  VArena ar;
  void* ws_ptr = nullptr; int64_t ws_bytes = 0;
};

namespace {

#define VCHECK(x) do { int _r = (x); if (_r) return _r; } while (0)

struct VCtx {
  mvd_vae* v;
  hipStream_t s;
  bool dry;
  int err = 0;

  const VWeight* W(const std::string& name, int dtype, int64_t numel) {
    if (dry) return nullptr;
    auto it = v->w.find(name);
    if (it == v->w.end()) { mvd_set_error("vae: missing weight slot '%s'", name.c_str()); err = -10; return nullptr; }
    if (it->second.dtype != dtype || it->second.numel != numel) {
      mvd_set_error("vae: weight slot '%s': expected dtype %d numel %lld, got dtype %d numel %lld", name.c_str(), dtype, (long long)numel,
                    it->second.dtype, (long long)it->second.numel);
      err = -11; return nullptr;
    }
    return &it->second;
  }
  const bf16_t* WB(const std::string& n, int64_t numel) { auto* w = W(n, 1, numel); return w ? (const bf16_t*)w->p : nullptr; }
  const float* WF(const std::string& n, int64_t numel) { auto* w = W(n, 0, numel); return w ? (const float*)w->p : nullptr; }
  template <class T> T* alloc(size_t n) { return (T*)v->ar.alloc(n * sizeof(T)); }
  VAct act(int B, int H, int W_, int C) { VAct a; a.B = B; a.H = H; a.W = W_; a.C = C; a.p = alloc<bf16_t>((size_t)B * H * W_ * C); return a; }

  int gemm(MvdGemmArgs& g) {
    if (err) return err;
    const int S = mvd_gemm_pick_splitk(g);
    const size_t mark = v->ar.off;
    if (S > 1) { g.splitk = S; g.part = alloc<float>((size_t)S * g.M * g.N); }
    int r = 0;
    if (!dry) {
      r = mvd_launch_gemm(g, s);
      if (!r && S > 1) r = mvd_launch_splitk_reduce(g, s);
    }
    v->ar.off = mark;
    return r;
  }
  int linear(const bf16_t* a, int k, int M, const bf16_t* w, const float* bias, int N, const bf16_t* res, void* out, float alpha = 1.f,
             bool out_f32 = false) {
    MvdGemmArgs g; memset(&g, 0, sizeof(g));
    g.ldw = k;
    g.seg[0].p0 = a; g.seg[0].c0 = k; g.seg[0].mode = MVD_A_DENSE; g.seg[0].ksize = k;
    g.nseg = 1; g.W = w; g.M = M; g.N = N; g.Ktot = k; g.rows_per_batch = M; g.outH = 1; g.outW = M;
    g.bias = bias; g.res = res; g.ldres = N; g.alpha = alpha; g.out = out; g.ldo = N; g.out_f32 = out_f32;
    return gemm(g);
  }
  // 3x3 conv (pad 1; stride 2 = the VAE's bottom/right-padded downsampler; ups = nearest 2x in front), optional residual
  // or fused 1x1 shortcut on `sc`
  int conv3(const VAct& x, int stride, int ups, const bf16_t* w, const float* bias, const bf16_t* res, const bf16_t* sc, int scc, VAct& out) {
    MvdGemmArgs g; memset(&g, 0, sizeof(g));
    g.seg[0].p0 = x.p; g.seg[0].c0 = x.C; g.seg[0].mode = MVD_A_CONV3; g.seg[0].ksize = 9 * x.C;
    g.seg[0].inH = x.H; g.seg[0].inW = x.W; g.seg[0].stride = stride; g.seg[0].ups = ups; g.seg[0].asym = stride == 2 ? 1 : 0;
    g.nseg = 1; g.Ktot = 9 * x.C;
    if (sc) {
      g.seg[1].p0 = sc; g.seg[1].c0 = scc; g.seg[1].mode = MVD_A_DENSE; g.seg[1].ksize = scc; g.nseg = 2; g.Ktot += scc;
    }
    g.W = w; g.ldw = g.Ktot; g.M = out.rows(); g.N = out.C; g.rows_per_batch = out.hw(); g.outH = out.H; g.outW = out.W;
    g.bias = bias; g.res = res; g.ldres = out.C; g.alpha = 1.f; g.out = out.p; g.ldo = out.C;
    return gemm(g);
  }
  int groupnorm(const VAct& x, const float* g, const float* b, int silu, bf16_t* y) {
    if (err) return err;
    float* ws = alloc<float>((size_t)x.B * MVD_GN_MAXCHUNK * v->cfg.norm_num_groups * 2);
    if (dry) return 0;
    return mvd_launch_groupnorm(x.p, nullptr, x.C, 0, x.B, x.hw(), v->cfg.norm_num_groups, v->cfg.norm_eps, g, b, silu, y, ws, s);
  }

  // ResnetBlock2D without a time embedding: GN+SiLU -> conv1 -> GN+SiLU -> conv2 (+ x, or || 1x1 conv_shortcut(x))
  int resnet(const std::string& key, const VAct& x, int cout, VAct& out) {
    const int cin = x.C;
    const size_t mark = v->ar.off;
    VAct t1 = act(x.B, x.H, x.W, cin);
    VCHECK(groupnorm(x, WF(key + ".norm1.g", cin), WF(key + ".norm1.b", cin), 1, t1.p));
    VAct h1 = act(x.B, x.H, x.W, cout);
    VCHECK(conv3(t1, 1, 0, WB(key + ".conv1.w", (int64_t)cout * 9 * cin), WF(key + ".conv1.b", cout), nullptr, nullptr, 0, h1));
    VAct t2 = act(x.B, x.H, x.W, cout);
    VCHECK(groupnorm(h1, WF(key + ".norm2.g", cout), WF(key + ".norm2.b", cout), 1, t2.p));
    if (cin != cout) {
      VCHECK(conv3(t2, 1, 0, WB(key + ".conv2.w", (int64_t)cout * (9 * cout + cin)), WF(key + ".conv2.b", cout), nullptr, x.p, cin, out));
    } else {
      VCHECK(conv3(t2, 1, 0, WB(key + ".conv2.w", (int64_t)cout * 9 * cout), WF(key + ".conv2.b", cout), x.p, nullptr, 0, out));
    }
    v->ar.off = mark;
    return err;
  }

  // mid-block attention: one head of C channels over the H*W positions of each image
  int attention(const std::string& key, const VAct& x, VAct& out) {
    const int C = x.C, hw = x.hw(), M = x.rows();
    if (hw % 64 || C % 64) { mvd_set_error("vae attention: %d positions x %d channels must be multiples of 64", hw, C); return -1; }
    const size_t mark = v->ar.off;
    bf16_t* xn = alloc<bf16_t>((size_t)M * C);
    VCHECK(groupnorm(x, WF(key + ".norm.g", C), WF(key + ".norm.b", C), 0, xn));
    bf16_t* q = alloc<bf16_t>((size_t)M * C);
    bf16_t* k = alloc<bf16_t>((size_t)M * C);
    bf16_t* o = alloc<bf16_t>((size_t)M * C);
    VCHECK(linear(xn, C, M, WB(key + ".q.w", (int64_t)C * C), WF(key + ".q.b", C), C, nullptr, q));
    VCHECK(linear(xn, C, M, WB(key + ".k.w", (int64_t)C * C), WF(key + ".k.b", C), C, nullptr, k));
    bf16_t* vt = alloc<bf16_t>((size_t)C * hw);
    float* sc = alloc<float>((size_t)hw * hw);
    bf16_t* pr = alloc<bf16_t>((size_t)hw * hw);
    const bf16_t* wv = WB(key + ".v.w", (int64_t)C * C);
    const float* bv = WF(key + ".v.b", C);
    const float scale = 1.0f / sqrtf((float)C);
    for (int b = 0; b < x.B && !err; ++b) {
      const size_t o0 = (size_t)b * hw * C;
      VCHECK(linear(wv, C, C, xn + o0, nullptr, hw, nullptr, vt));                          // V^T = W_v . x^T   [C][hw]
      VCHECK(linear(q + o0, C, hw, k + o0, nullptr, hw, nullptr, sc, scale, true));          // S = q.k^T / sqrt(C)  fp32
      if (!dry) { hipLaunchKernelGGL(softmax_rows_kernel, dim3(hw), dim3(256), 0, s, sc, hw, pr); VCHECK(vcheck("vae softmax")); }
      VCHECK(linear(pr, hw, hw, vt, bv, C, nullptr, o + o0));                                // P.V + b_v
    }
    VCHECK(linear(o, C, M, WB(key + ".out.w", (int64_t)C * C), WF(key + ".out.b", C), C, x.p, out.p));   // to_out + residual
    v->ar.off = mark;
    return err;
  }

  int mid(const std::string& p, const VAct& x, VAct& out) {
    const int C = x.C;
    VAct r0 = act(x.B, x.H, x.W, C), a0 = act(x.B, x.H, x.W, C);
    VCHECK(resnet(p + ".resnets.0", x, C, r0));
    VCHECK(attention(p + ".attn", r0, a0));
    VCHECK(resnet(p + ".resnets.1", a0, C, out));
    return err;
  }
  // conv_in: NCHW fp32 -> im2col rows (K padded to 64) -> GEMM
  int conv_in(const std::string& key, const float* x_nchw, int B, int cin, int H, int W_, int cout, VAct& out) {
    bf16_t* col = alloc<bf16_t>((size_t)B * H * W_ * 64);
    if (!dry && !err) VCHECK(mvd_launch_im2col_in(x_nchw, B, cin, H, W_, nullptr, nullptr, 0, col, s));
    return linear(col, 64, B * H * W_, WB(key + ".w", (int64_t)cout * 64), WF(key + ".b", cout), cout, nullptr, out.p);
  }
  int norm_conv_out(const std::string& pfx, const VAct& x, int cout, float* y_nchw) {
    bf16_t* t = alloc<bf16_t>((size_t)x.rows() * x.C);
    VCHECK(groupnorm(x, WF(pfx + ".norm_out.g", x.C), WF(pfx + ".norm_out.b", x.C), 1, t));
    const bf16_t* w = WB(pfx + ".conv_out.w", (int64_t)cout * 9 * x.C);
    const float* b = WF(pfx + ".conv_out.b", cout);
    if (!dry && !err) VCHECK(mvd_launch_conv_out(t, x.B, x.H, x.W, x.C, w, b, cout, y_nchw, s));
    return err;
  }
  int pointwise(const std::string& key, const float* x, int B, int cin, int cout, int hw, float* y) {
    const float* w = WF(key + ".w", (int64_t)cout * cin);
    const float* b = WF(key + ".b", cout);
    if (dry || err) return err;
    const long total = (long)B * hw;
    hipLaunchKernelGGL(pointwise_small_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, x, cin, cout, hw, w, b, y, total);
    return vcheck("vae pointwise");
  }
};

int encode_impl(mvd_vae* v, const float* image, int B, int H, int W_, float* moments, hipStream_t s, bool dry) {
  const mvd_vae_config_t& c = v->cfg;
  const int n = c.num_levels, div = 1 << (n - 1);
  if (B <= 0 || H <= 0 || W_ <= 0 || H % div || W_ % div) { mvd_set_error("vae encode: image %dx%d must be divisible by %d", H, W_, div); return -1; }
  v->ar.dry = dry; v->ar.off = v->ar.high = 0;
  VCtx x{v, s, dry};
  VAct h = x.act(B, H, W_, c.block_out_channels[0]);
  VCHECK(x.conv_in("encoder.conv_in", image, B, c.in_channels, H, W_, c.block_out_channels[0], h));
  for (int i = 0; i < n; ++i) {
    const int co = c.block_out_channels[i];
    const std::string bk = "encoder.down_blocks." + std::to_string(i);
    for (int j = 0; j < c.layers_per_block; ++j) {
      VAct r = x.act(B, h.H, h.W, co);
      VCHECK(x.resnet(bk + ".resnets." + std::to_string(j), h, co, r));
      h = r;
    }
    if (i + 1 < n) {
      VAct d = x.act(B, h.H / 2, h.W / 2, co);
      VCHECK(x.conv3(h, 2, 0, x.WB(bk + ".down.w", (int64_t)co * 9 * co), x.WF(bk + ".down.b", co), nullptr, nullptr, 0, d));
      h = d;
    }
  }
  VAct m = x.act(B, h.H, h.W, h.C);
  VCHECK(x.mid("encoder.mid_block", h, m));
  const int L2 = 2 * c.latent_channels, hw = m.hw();
  float* pre = x.alloc<float>((size_t)B * L2 * hw);
  VCHECK(x.norm_conv_out("encoder", m, L2, pre));
  VCHECK(x.pointwise("quant_conv", pre, B, L2, L2, hw, moments));
  if (x.err) return x.err;
  if (!dry && v->ar.high > (size_t)v->ws_bytes) { mvd_set_error("vae encode: workspace too small"); return -4; }
  return 0;
}

int decode_impl(mvd_vae* v, const float* latents, int B, int h_, int w_, float* image, hipStream_t s, bool dry) {
  const mvd_vae_config_t& c = v->cfg;
  const int n = c.num_levels;
  if (B <= 0 || h_ <= 0 || w_ <= 0) { mvd_set_error("vae decode: bad shape"); return -1; }
  v->ar.dry = dry; v->ar.off = v->ar.high = 0;
  VCtx x{v, s, dry};
  const int L = c.latent_channels, cm = c.block_out_channels[n - 1];
  float* z = x.alloc<float>((size_t)B * L * h_ * w_);
  VCHECK(x.pointwise("post_quant_conv", latents, B, L, L, h_ * w_, z));
  VAct h = x.act(B, h_, w_, cm);
  VCHECK(x.conv_in("decoder.conv_in", z, B, L, h_, w_, cm, h));
  VAct m = x.act(B, h_, w_, cm);
  VCHECK(x.mid("decoder.mid_block", h, m));
  h = m;
  for (int i = 0; i < n; ++i) {
    const int co = c.block_out_channels[n - 1 - i];
    const std::string bk = "decoder.up_blocks." + std::to_string(i);
    for (int j = 0; j <= c.layers_per_block; ++j) {
      VAct r = x.act(B, h.H, h.W, co);
      VCHECK(x.resnet(bk + ".resnets." + std::to_string(j), h, co, r));
      h = r;
    }
    if (i + 1 < n) {
      VAct u = x.act(B, h.H * 2, h.W * 2, co);
      VCHECK(x.conv3(h, 1, 1, x.WB(bk + ".up.w", (int64_t)co * 9 * co), x.WF(bk + ".up.b", co), nullptr, nullptr, 0, u));
      h = u;
    }
  }
  VCHECK(x.norm_conv_out("decoder", h, c.in_channels, image));
  if (x.err) return x.err;
  if (!dry && v->ar.high > (size_t)v->ws_bytes) { mvd_set_error("vae decode: workspace too small"); return -4; }
  return 0;
}

}  // namespace

extern "C" {

int mvd_vae_create(const mvd_vae_config_t* cfg, mvd_vae_t** out) {
  if (!cfg || !out) { mvd_set_error("vae_create: null argument"); return -1; }
  if (cfg->num_levels < 2 || cfg->num_levels > MVD_MAX_LEVELS || cfg->layers_per_block < 1 || cfg->in_channels > 7 || cfg->latent_channels > 4 ||
      cfg->in_channels < 1 || cfg->latent_channels < 1) { mvd_set_error("vae_create: unsupported topology"); return -1; }
  for (int i = 0; i < cfg->num_levels; ++i)
    if (cfg->block_out_channels[i] % 64 || cfg->block_out_channels[i] % cfg->norm_num_groups) { mvd_set_error("vae_create: level %d: channels must be a multiple of 64 and of the group count", i); return -1; }
  mvd_vae* v = new mvd_vae();
  v->cfg = *cfg;
  *out = v;
  return 0;
}
int mvd_vae_destroy(mvd_vae_t* v) { delete v; return 0; }

int mvd_vae_set_weight(mvd_vae_t* v, const char* slot, const void* ptr, int64_t numel, int dtype) {
  if (!v || !slot || !ptr || numel <= 0 || dtype < 0 || dtype > 1) { mvd_set_error("vae_set_weight: bad argument"); return -1; }
  if ((uintptr_t)ptr & 15) { mvd_set_error("vae_set_weight: '%s' must be 16-byte aligned", slot); return -1; }
  v->w[slot] = VWeight{ptr, numel, dtype};
  return 0;
}

int64_t mvd_vae_workspace_bytes(mvd_vae_t* v, int batch, int height, int width, int decode) {
  if (!v) { mvd_set_error("vae_workspace_bytes: null handle"); return -1; }
  const int r = decode ? decode_impl(v, nullptr, batch, height, width, nullptr, nullptr, true)
                       : encode_impl(v, nullptr, batch, height, width, nullptr, nullptr, true);
  if (r) return r;
  return (int64_t)v->ar.high + 4096;
}

int mvd_vae_bind_workspace(mvd_vae_t* v, void* ws, int64_t ws_bytes) {
  if (!v || !ws || ws_bytes <= 0 || ((uintptr_t)ws & 255)) { mvd_set_error("vae_bind_workspace: bad argument (256-byte aligned buffer)"); return -1; }
  v->ws_ptr = ws; v->ws_bytes = ws_bytes;
  v->ar.base = (char*)ws; v->ar.cap = (size_t)ws_bytes;
  return 0;
}

int mvd_vae_encode(mvd_vae_t* v, const float* image_nchw, int batch, int height, int width, float* moments, void* stream) {
  if (!v || !image_nchw || !moments) { mvd_set_error("vae_encode: null argument"); return -1; }
  if (!v->ws_ptr) { mvd_set_error("vae_encode: workspace not bound"); return -1; }
  if (int r = encode_impl(v, nullptr, batch, height, width, nullptr, nullptr, true)) return r;       // size first: nothing is launched into a short buffer
  if (v->ar.high > (size_t)v->ws_bytes) { mvd_set_error("vae_encode: workspace too small: need %zu bytes, bound %lld", v->ar.high, (long long)v->ws_bytes); return -4; }
  return encode_impl(v, image_nchw, batch, height, width, moments, (hipStream_t)stream, false);
}

int mvd_vae_decode(mvd_vae_t* v, const float* latents_nchw, int batch, int height, int width, float* image, void* stream) {
  if (!v || !latents_nchw || !image) { mvd_set_error("vae_decode: null argument"); return -1; }
  if (!v->ws_ptr) { mvd_set_error("vae_decode: workspace not bound"); return -1; }
  if (int r = decode_impl(v, nullptr, batch, height, width, nullptr, nullptr, true)) return r;
  if (v->ar.high > (size_t)v->ws_bytes) { mvd_set_error("vae_decode: workspace too small: need %zu bytes, bound %lld", v->ar.high, (long long)v->ws_bytes); return -4; }
  return decode_impl(v, latents_nchw, batch, height, width, image, (hipStream_t)stream, false);
}

int mvd_op_gaussian_sample(const float* moments, const float* noise, int batch, int channels, int hw, float scale, float* out, void* stream) {
  if (!moments || !noise || !out || batch <= 0 || channels <= 0 || hw <= 0) { mvd_set_error("gaussian_sample: bad argument"); return -1; }
  const long total = (long)batch * channels * hw;
  hipLaunchKernelGGL(gaussian_sample_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, moments, noise, channels, hw,
                     scale, out, total);
  return vcheck("gaussian_sample");
}

}  // extern "C"
