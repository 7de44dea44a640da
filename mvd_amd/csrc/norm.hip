// Normalisation kernels (HBM-bound, 16-byte vectorised): GroupNorm(+SiLU) over NHWC with an
// optional two-source channel concat, LayerNorm, and the MVD reference-feature
// normalisation (per pixel over batch x channel, attention.py:95-103 of the reference).
#include <stdlib.h>
#include "kernels.h"

namespace {

MVD_DEVINL void unpack8(const u32x4 v, float* f) {
  f[0] = bflo(v[0]); f[1] = bfhi(v[0]); f[2] = bflo(v[1]); f[3] = bfhi(v[1]);
  f[4] = bflo(v[2]); f[5] = bfhi(v[2]); f[6] = bflo(v[3]); f[7] = bfhi(v[3]);
}
MVD_DEVINL u32x4 pack8(const float* f) {
  return u32x4{pack2bf(f[0], f[1]), pack2bf(f[2], f[3]), pack2bf(f[4], f[5]), pack2bf(f[6], f[7])};
}

// ---------------------------------------------------------------- GroupNorm: partial statistics
// grid (nchunk, batch); block = vec*R threads (vec = C/8 channel vectors, R rows in flight).
// Deterministic (fixed-order reductions) and CANCELLATION-SAFE: a thread accumulates sum / sum of squares of
// (x - shift) with shift = its own first element, turns them into (count, mean, M2 = sum (x - mean)^2), and partials
// are merged with the parallel-variance formula  M2 = sum_i M2_i + n_i (mean_i - mean)^2  (Chan et al.) -- never
// E[x^2] - mean^2, which loses every digit once |mean| >> std (trained SD channels).
// ws holds per (batch, chunk, group): mean, M2 (the count follows from the chunk geometry).
__global__ void gn_stats_kernel(const bf16_t* __restrict__ x0, const bf16_t* __restrict__ x1, int c0, int c1, int hw,
                                int groups, int rows_per_chunk, int R, float* __restrict__ ws) {
  extern __shared__ __attribute__((aligned(16))) float sh[];  // [R][C][2] = (mean, M2) per thread and channel
  const int C = c0 + c1, vec = C >> 3;
  const int b = blockIdx.y, chunk = blockIdx.x, nchunk = gridDim.x;
  const int v = threadIdx.x % vec, r = threadIdx.x / vec;
  const int row0 = chunk * rows_per_chunk;
  const int row1 = min(hw, row0 + rows_per_chunk);
  const int ch = v * 8;
  const bool first = ch < c0;
  const bf16_t* src = first ? x0 + (size_t)b * hw * c0 + ch : x1 + (size_t)b * hw * c1 + (ch - c0);
  const int ld = first ? c0 : c1;
  if (r < R) {
    float s[8], q[8], sft[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) { s[j] = 0.f; q[j] = 0.f; sft[j] = 0.f; }
    int n = 0;
    int row = row0 + r;
    if (row < row1) {   // the thread's first sample is the shift (its own deviation is 0): no extra memory round trip
      unpack8(*reinterpret_cast<const u32x4*>(src + (size_t)row * ld), sft);
      n = 1; row += R;
    }
    for (; row < row1; row += R) {
      float f[8];
      unpack8(*reinterpret_cast<const u32x4*>(src + (size_t)row * ld), f);
#pragma unroll
      for (int j = 0; j < 8; ++j) { const float d = f[j] - sft[j]; s[j] += d; q[j] = fmaf(d, d, q[j]); }
      ++n;
    }
    const float inv = n ? 1.0f / (float)n : 0.f;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      sh[(r * C + ch + j) * 2] = sft[j] + s[j] * inv;                       // mean of this thread's samples
      sh[(r * C + ch + j) * 2 + 1] = fmaxf(q[j] - s[j] * s[j] * inv, 0.f);  // their M2
    }
  }
  __syncthreads();
  const int cg = C / groups;
  if (threadIdx.x < groups) {
    const int g = threadIdx.x;
    const int nrows = row1 - row0;
    // thread r of the row loop saw rows row0 + r, row0 + r + R, ...: n_r = ceil((nrows - r) / R)
    float tot = 0.f, msum = 0.f;
    for (int rr = 0; rr < R; ++rr) {
      const int nr = nrows > rr ? (nrows - rr + R - 1) / R : 0;
      if (!nr) continue;
      float a = 0.f;
      for (int cc = 0; cc < cg; ++cc) a += sh[(rr * C + g * cg + cc) * 2];
      msum += a * (float)nr;
      tot += (float)nr * (float)cg;
    }
    const float mean = tot > 0.f ? msum / tot : 0.f;
    float m2 = 0.f;
    for (int rr = 0; rr < R; ++rr) {
      const int nr = nrows > rr ? (nrows - rr + R - 1) / R : 0;
      if (!nr) continue;
      for (int cc = 0; cc < cg; ++cc) {
        const float d = sh[(rr * C + g * cg + cc) * 2] - mean;
        m2 += sh[(rr * C + g * cg + cc) * 2 + 1] + (float)nr * d * d;
      }
    }
    float* o = ws + (((size_t)b * nchunk + chunk) * groups + g) * 2;
    o[0] = mean; o[1] = m2;
  }
}

// ---------------------------------------------------------------- GroupNorm: apply (+SiLU)
__global__ __launch_bounds__(256) void gn_apply_kernel(const bf16_t* __restrict__ x0, const bf16_t* __restrict__ x1,
                                                        int c0, int c1, int hw, int groups, int nchunk_stats,
                                                        int rows_per_chunk_stats, float eps,
                                                        const float* __restrict__ gamma,
                                                        const float* __restrict__ beta, int silu,
                                                        const float* __restrict__ ws, int rows_per_blk,
                                                        bf16_t* __restrict__ y) {
  extern __shared__ __attribute__((aligned(16))) float sh[];  // mean[groups] rstd[groups] a[C] b[C]
  const int C = c0 + c1, vec = C >> 3, cg = C / groups;
  const int b = blockIdx.y;
  float* s_mean = sh;
  float* s_rstd = sh + groups;
  float* s_a = sh + 2 * groups;
  float* s_b = s_a + C;
  {
    // fixed-order (deterministic) merge of the per-chunk (count, mean, M2): 8 lanes per group, xor-shuffle trees;
    // pass 1 the global mean, pass 2 M2 = sum M2_c + n_c (mean_c - mean)^2
    const int g = threadIdx.x >> 3, part = threadIdx.x & 7;
    const float* wg = ws + ((size_t)b * nchunk_stats * groups + (g < groups ? g : 0)) * 2;
    // (clamped: a chunk that starts at or beyond hw holds no rows -- the launcher never makes one, see nchunk below)
    auto rows_of = [&](int ck) { return max(0, min(hw, (ck + 1) * rows_per_chunk_stats) - ck * rows_per_chunk_stats); };
    float s = 0.f;
    if (g < groups)
      for (int ck = part; ck < nchunk_stats; ck += 8) s += wg[(size_t)ck * groups * 2] * (float)rows_of(ck);
#pragma unroll
    for (int o = 4; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
    const float mean = s / (float)hw;
    float q = 0.f;
    if (g < groups)
      for (int ck = part; ck < nchunk_stats; ck += 8) {
        const float d = wg[(size_t)ck * groups * 2] - mean;
        q += wg[(size_t)ck * groups * 2 + 1] + (float)rows_of(ck) * (float)cg * d * d;
      }
#pragma unroll
    for (int o = 4; o > 0; o >>= 1) q += __shfl_xor(q, o, 64);
    if (g < groups && part == 0) {
      s_mean[g] = mean;
      s_rstd[g] = rsqrtf(q / ((float)hw * (float)cg) + eps);
    }
  }
  __syncthreads();
  for (int c = threadIdx.x; c < C; c += blockDim.x) {
    const int g = c / cg;
    const float a = gamma[c] * s_rstd[g];
    s_a[c] = a;
    s_b[c] = beta[c] - s_mean[g] * a;
  }
  __syncthreads();
  const int row0 = blockIdx.x * rows_per_blk;
  const int nrow = min(hw, row0 + rows_per_blk) - row0;
  // element e = (row, 8-channel vector v); the (row, v) cursor advances by blockDim without a division per element,
  // and the per-channel affine comes out of LDS as four 16-byte reads
  int row = threadIdx.x / vec, v = threadIdx.x - row * vec;
  const int dr = blockDim.x / vec, dv = blockDim.x - dr * vec;
  while (row < nrow) {
    const int ch = v * 8;
    const size_t pix = (size_t)b * hw + row0 + row;
    const bf16_t* src = ch < c0 ? x0 + pix * c0 + ch : x1 + pix * c1 + (ch - c0);
    float f[8];
    unpack8(*reinterpret_cast<const u32x4*>(src), f);
    const f32x4 a0 = *reinterpret_cast<const f32x4*>(s_a + ch), a1 = *reinterpret_cast<const f32x4*>(s_a + ch + 4);
    const f32x4 b0 = *reinterpret_cast<const f32x4*>(s_b + ch), b1 = *reinterpret_cast<const f32x4*>(s_b + ch + 4);
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const float t0 = fmaf(f[j], a0[j], b0[j]), t1 = fmaf(f[j + 4], a1[j], b1[j]);
      f[j] = silu ? silu_f(t0) : t0;
      f[j + 4] = silu ? silu_f(t1) : t1;
    }
    *reinterpret_cast<u32x4*>(y + pix * C + ch) = pack8(f);
    v += dv; row += dr;
    if (v >= vec) { v -= vec; ++row; }
  }
}

// ---------------------------------------------------------------- GroupNorm in ONE pass (register-resident slice)
// A workgroup owns `gpw` consecutive groups of ONE image -- a channel slice of cs = gpw * C/groups channels (a multiple of
// 8) over all hw pixels -- and keeps it in registers (NV 16-byte vectors per thread): read once, exact two-pass statistics
// (mean, then sum (x - mean)^2: no cancellation, no shift), normalise + affine (+SiLU), write once.  Two trips over HBM
// instead of the three of gn_stats + gn_apply, and one launch instead of two (the 16x16 / 8x8 levels and batch 1 are
// launch-latency bound).  Thread t handles vector v = t % nvec of pixels t / nvec, t / nvec + npl, ...: its 8 channels
// stay fixed, so its affine constants live in registers and its elements fall into at most two groups (cg >= 8, even).
// Reductions are fixed-order (per-thread partials -> LDS -> one wave per group, xor-shuffle tree): bit-deterministic.
// Slices of one image sit on one XCD (blockIdx & 7 = image & 7 when the batch allows) so that the 128-byte lines they
// share are fetched / written back once by that XCD's L2.
template <int NV>
__global__ __launch_bounds__(1024) void gn_slice_kernel(const bf16_t* __restrict__ x0, const bf16_t* __restrict__ x1,
                                                         int c0, int c1, int batch, int hw, int groups, int gpw, float eps,
                                                         const float* __restrict__ gamma, const float* __restrict__ beta,
                                                         int silu, bf16_t* __restrict__ y) {
  __shared__ float2 part[1024];
  __shared__ int glo_of[1024];
  __shared__ float s_mean[32], s_rstd[32];
  const int C = c0 + c1, cg = C / groups, cs = gpw * cg, nvec = cs >> 3, nslice = groups / gpw;
  const int nthr = blockDim.x, npl = nthr / nvec;
  int b, slice;
  if ((batch & 7) == 0) { const int r = blockIdx.x >> 3; slice = r % nslice; b = (r / nslice) * 8 + (blockIdx.x & 7); }
  else { b = blockIdx.x / nslice; slice = blockIdx.x - b * nslice; }
  const int t = threadIdx.x;
  const int pl = t / nvec, v = t - pl * nvec;
  const bool active = pl < npl;
  const int ch = slice * cs + v * 8;                       // first channel of the thread's vector
  const int glo = (v * 8) / cg;                            // slice-local group of element 0
  const int kb = (glo + 1) * cg - v * 8;                   // elements j >= kb belong to group glo + 1 (kb >= 8: none)
  glo_of[t] = active ? glo : -4;

  // Buffer addressing: one descriptor per source image, a loop-invariant per-thread byte offset, the pixel step as a
  // scalar offset -- no per-vector 64-bit addresses (they would double the registers a vector costs).  Pixels >= hw and the
  // threads beyond npl * nvec fall outside num_records: their loads return zeros, their stores are dropped.
  constexpr unsigned OOB = 0x80000000u;
  __amdgpu_buffer_rsrc_t rs0 = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_t*>(x0 + (size_t)b * hw * c0), 0, hw * c0 * 2, 0x00020000);
  const unsigned vo0 = active && ch < c0 ? (unsigned)(pl * c0 + ch) * 2u : OOB;
  const int st0 = npl * c0 * 2;                           // (scalar) byte step between a thread's consecutive pixels
  u32x4 d[NV];
#pragma unroll
  for (int i = 0; i < NV; ++i) d[i] = __builtin_amdgcn_raw_buffer_load_b128(rs0, vo0, i * st0, 0);
  if (c1 && ch >= c0) {      // threads whose channels lie in the second source: same registers, loads under their exec mask
    __amdgpu_buffer_rsrc_t rs1 = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_t*>(x1 + (size_t)b * hw * c1), 0, hw * c1 * 2, 0x00020000);
    const unsigned vo1 = active ? (unsigned)(pl * c1 + ch - c0) * 2u : OOB;
    const int st1 = npl * c1 * 2;
#pragma unroll
    for (int i = 0; i < NV; ++i) d[i] = __builtin_amdgcn_raw_buffer_load_b128(rs1, vo1, i * st1, 0);
  }
  const int nval = active && pl < hw ? (hw - pl + npl - 1) / npl : 0;          // the thread's valid vectors are d[0 .. nval)
  const float inv_n = 1.f / ((float)hw * (float)cg);
  // fixed-order reduction of the per-thread (slot 0, slot 1) partials into one value per group
  auto reduce_groups = [&](float p0, float p1, float* out, bool to_rstd) {
    part[t] = float2{p0, p1};
    __syncthreads();
    const int wave = t >> 6, lane = t & 63, nwave = nthr >> 6;
    for (int g = wave; g < gpw; g += nwave) {
      float a = 0.f;
      for (int q = lane; q < nthr; q += 64) {
        const int gl = glo_of[q];
        const float2 pq = part[q];
        a += gl == g ? pq.x : 0.f;
        a += gl + 1 == g ? pq.y : 0.f;
      }
#pragma unroll
      for (int o = 32; o > 0; o >>= 1) a += __shfl_xor(a, o, 64);
      if (lane == 0) out[g] = to_rstd ? rsqrtf(a * inv_n + eps) : a * inv_n;
    }
    __syncthreads();
  };
  // The 8 channels of a vector are 4 dwords (channel pairs); cg is even, so a pair never straddles a group: dword w belongs
  // to slot 0 (group glo) iff 2w < kb.  Both passes keep one accumulator per dword and sort them into slots at the end --
  // no per-element select in the loops.
  typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2_t;
  const bf16x2_t ones = __builtin_bit_cast(bf16x2_t, 0x3f803f80u);
  // ---- pass A: mean (v_dot2c_f32_bf16 with (1, 1) sums a pair straight from the packed data; padding vectors are zeros)
  float pa[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    const unsigned w[4] = {d[i].x, d[i].y, d[i].z, d[i].w};
#pragma unroll
    for (int k = 0; k < 4; ++k) pa[k] = __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(bf16x2_t, w[k]), ones, pa[k], false);
  }
  float a0 = 0.f, a1 = 0.f;
#pragma unroll
  for (int k = 0; k < 4; ++k) { if (2 * k < kb) a0 += pa[k]; else a1 += pa[k]; }
  reduce_groups(a0, a1, s_mean, false);
  const float m0 = s_mean[glo], m1 = s_mean[glo + 1 < gpw ? glo + 1 : glo];
  // ---- pass B: sum of squared deviations
  float mw[4], pq[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int k = 0; k < 4; ++k) mw[k] = 2 * k < kb ? m0 : m1;
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    if (i < nval) {
      const unsigned w[4] = {d[i].x, d[i].y, d[i].z, d[i].w};
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const float lo = bflo(w[k]) - mw[k], hi = bfhi(w[k]) - mw[k];
        pq[k] = fmaf(lo, lo, pq[k]);
        pq[k] = fmaf(hi, hi, pq[k]);
      }
    }
  }
  float q0 = 0.f, q1 = 0.f;
#pragma unroll
  for (int k = 0; k < 4; ++k) { if (2 * k < kb) q0 += pq[k]; else q1 += pq[k]; }
  reduce_groups(q0, q1, s_rstd, true);
  if (!active) return;
  // ---- normalise + affine (+SiLU), store
  float ga[8], gb[8];
  {
    const f32x4 g0 = *reinterpret_cast<const f32x4*>(gamma + ch), g1 = *reinterpret_cast<const f32x4*>(gamma + ch + 4);
    const f32x4 b0 = *reinterpret_cast<const f32x4*>(beta + ch), b1 = *reinterpret_cast<const f32x4*>(beta + ch + 4);
    const float r0 = s_rstd[glo], r1 = s_rstd[glo + 1 < gpw ? glo + 1 : glo];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const float gm = j < 4 ? g0[j] : g1[j - 4], bt = j < 4 ? b0[j] : b1[j - 4];
      ga[j] = gm * (j < kb ? r0 : r1);
      gb[j] = bt - (j < kb ? m0 : m1) * ga[j];
    }
  }
  __amdgpu_buffer_rsrc_t rs_y = __builtin_amdgcn_make_buffer_rsrc(y + (size_t)b * hw * C, 0, hw * C * 2, 0x00020000);
  const unsigned voy = (unsigned)(pl * C + ch) * 2u;
  const int sty = npl * C * 2;
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    if (i < nval) {
      const unsigned w[4] = {d[i].x, d[i].y, d[i].z, d[i].w};
      u32x4 o;
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        float lo = fmaf(bflo(w[k]), ga[2 * k], gb[2 * k]), hi = fmaf(bfhi(w[k]), ga[2 * k + 1], gb[2 * k + 1]);
        if (silu) { lo = silu_f(lo); hi = silu_f(hi); }
        o[k] = pack2bf(lo, hi);
      }
      __builtin_amdgcn_raw_buffer_store_b128(o, rs_y, voy, i * sty, 0);
      asm volatile("s_nop 1" :: "v"(o));       // wait states hipcc leaves out behind a 16-byte store with an SGPR soffset (see gemm_pp.hip store16)
    }
    __builtin_amdgcn_sched_barrier(0);
  }
}

// ---------------------------------------------------------------- LayerNorm (bf16 rows)
// one wave per row, row kept in registers (C <= 64*8*MAXV), two-pass mean/variance.
template <int MAXV>
__global__ __launch_bounds__(256) void ln_kernel(const bf16_t* __restrict__ x, int rows, int c, float eps,
                                                  const float* __restrict__ gamma, const float* __restrict__ beta,
                                                  bf16_t* __restrict__ y) {
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  const int vec = c >> 3;
  float f[MAXV][8];
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < MAXV; ++i) {
    const int v = lane + i * 64;
    if (v < vec) {
      unpack8(*reinterpret_cast<const u32x4*>(x + (size_t)row * c + v * 8), f[i]);
#pragma unroll
      for (int j = 0; j < 8; ++j) s += f[i][j];
    }
  }
  const float mean = wave_sum(s) / (float)c;
  float q = 0.f;
#pragma unroll
  for (int i = 0; i < MAXV; ++i) {
    const int v = lane + i * 64;
    if (v < vec) {
#pragma unroll
      for (int j = 0; j < 8; ++j) { const float d = f[i][j] - mean; q = fmaf(d, d, q); }
    }
  }
  const float rstd = rsqrtf(wave_sum(q) / (float)c + eps);
#pragma unroll
  for (int i = 0; i < MAXV; ++i) {
    const int v = lane + i * 64;
    if (v < vec) {
      float o[8];
      const f32x4 g0 = *reinterpret_cast<const f32x4*>(gamma + v * 8), g1 = *reinterpret_cast<const f32x4*>(gamma + v * 8 + 4);
      const f32x4 b0 = *reinterpret_cast<const f32x4*>(beta + v * 8), b1 = *reinterpret_cast<const f32x4*>(beta + v * 8 + 4);
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        o[j] = (f[i][j] - mean) * rstd * g0[j] + b0[j];
        o[j + 4] = (f[i][j + 4] - mean) * rstd * g1[j] + b1[j];
      }
      *reinterpret_cast<u32x4*>(y + (size_t)row * c + v * 8) = pack8(o);
    }
  }
}

// ---------------------------------------------------------------- reference normalisation (Q2)
// one workgroup per pixel; statistics over (batch, channel), unbiased std.
__global__ __launch_bounds__(256) void refnorm_kernel(const bf16_t* __restrict__ x, int batch, int hw, int c,
                                                      bf16_t* __restrict__ y) {
  __shared__ float red[8];
  const int p = blockIdx.x;
  const int vec = c >> 3;
  const int total = batch * vec;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  auto block_sum = [&](float v) -> float {
    v = wave_sum(v);
    __syncthreads();
    if (lane == 0) red[wave] = v;
    __syncthreads();
    return red[0] + red[1] + red[2] + red[3];
  };
  float s = 0.f;
  for (int e = threadIdx.x; e < total; e += 256) {
    const int b = e / vec, v = e % vec;
    float f[8];
    unpack8(*reinterpret_cast<const u32x4*>(x + ((size_t)b * hw + p) * c + v * 8), f);
#pragma unroll
    for (int j = 0; j < 8; ++j) s += f[j];
  }
  const float n = (float)batch * (float)c;
  const float mean = block_sum(s) / n;
  float q = 0.f;
  for (int e = threadIdx.x; e < total; e += 256) {
    const int b = e / vec, v = e % vec;
    float f[8];
    unpack8(*reinterpret_cast<const u32x4*>(x + ((size_t)b * hw + p) * c + v * 8), f);
#pragma unroll
    for (int j = 0; j < 8; ++j) { const float d = f[j] - mean; q = fmaf(d, d, q); }
  }
  const float var = block_sum(q) / (n - 1.0f);
  const float k = 0.5f / fmaxf(sqrtf(var), 1e-6f);
  for (int e = threadIdx.x; e < total; e += 256) {
    const int b = e / vec, v = e % vec;
    float f[8];
    const size_t off = ((size_t)b * hw + p) * c + v * 8;
    unpack8(*reinterpret_cast<const u32x4*>(x + off), f);
#pragma unroll
    for (int j = 0; j < 8; ++j) f[j] = (f[j] - mean) * k;
    *reinterpret_cast<u32x4*>(y + off) = pack8(f);
  }
}

// Q2 with statistics that span several GPUs (SURVEY.md 8e mode ii): the same arithmetic cut in two.  refstats writes, per
// pixel, the LOCAL mean and the sum of squared deviations from it (M2) over (batch, channel); the host merges the ranks'
// pairs (Chan's parallel form -- no E[x^2] - mean^2 cancellation) into (mean, k = 0.5 / max(std, 1e-6)) and refapply
// normalises with those.  With one rank the pair (refstats -> merge -> refapply) reproduces refnorm_kernel.
__global__ __launch_bounds__(256) void refstats_kernel(const bf16_t* __restrict__ x, int batch, int hw, int c, float* __restrict__ stats) {
  __shared__ float red[8];
  const int p = blockIdx.x;
  const int vec = c >> 3;
  const int total = batch * vec;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  auto block_sum = [&](float v) -> float {
    v = wave_sum(v);
    __syncthreads();
    if (lane == 0) red[wave] = v;
    __syncthreads();
    return red[0] + red[1] + red[2] + red[3];
  };
  float s = 0.f;
  for (int e = threadIdx.x; e < total; e += 256) {
    const int b = e / vec, v = e % vec;
    float f[8];
    unpack8(*reinterpret_cast<const u32x4*>(x + ((size_t)b * hw + p) * c + v * 8), f);
#pragma unroll
    for (int j = 0; j < 8; ++j) s += f[j];
  }
  const float mean = block_sum(s) / ((float)batch * (float)c);
  float q = 0.f;
  for (int e = threadIdx.x; e < total; e += 256) {
    const int b = e / vec, v = e % vec;
    float f[8];
    unpack8(*reinterpret_cast<const u32x4*>(x + ((size_t)b * hw + p) * c + v * 8), f);
#pragma unroll
    for (int j = 0; j < 8; ++j) { const float d = f[j] - mean; q = fmaf(d, d, q); }
  }
  const float m2 = block_sum(q);
  if (threadIdx.x == 0) { stats[2 * p] = mean; stats[2 * p + 1] = m2; }
}

__global__ __launch_bounds__(256) void refapply_kernel(const bf16_t* __restrict__ x, int batch, int hw, int c,
                                                       const float* __restrict__ mean_k, bf16_t* __restrict__ y) {
  const int vec = c >> 3;
  const long total = (long)batch * hw * vec;
  for (long e = (long)blockIdx.x * 256 + threadIdx.x; e < total; e += (long)gridDim.x * 256) {
    const int p = (int)((e / vec) % hw);
    const float mean = mean_k[2 * p], k = mean_k[2 * p + 1];
    float f[8];
    unpack8(*reinterpret_cast<const u32x4*>(x + e * 8), f);
#pragma unroll
    for (int j = 0; j < 8; ++j) f[j] = (f[j] - mean) * k;
    *reinterpret_cast<u32x4*>(y + e * 8) = pack8(f);
  }
}

// ---------------------------------------------------------------- fp32 row LayerNorm (camera MLPs)
__global__ __launch_bounds__(256) void ln_f32_kernel(const float* __restrict__ x, int c, float eps,
                                                      const float* __restrict__ gamma, const float* __restrict__ beta,
                                                      int silu, float* __restrict__ y, int nseg) {
  __shared__ float red[8];
  const int row = blockIdx.x;
  gamma += (size_t)(row % nseg) * c; beta += (size_t)(row % nseg) * c;     // (grouped form: one affine per segment)
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  auto block_sum = [&](float v) -> float {
    v = wave_sum(v);
    __syncthreads();
    if (lane == 0) red[wave] = v;
    __syncthreads();
    return red[0] + red[1] + red[2] + red[3];
  };
  const float* xr = x + (size_t)row * c;
  float s = 0.f;
  for (int i = threadIdx.x; i < c; i += 256) s += xr[i];
  const float mean = block_sum(s) / (float)c;
  float q = 0.f;
  for (int i = threadIdx.x; i < c; i += 256) { const float d = xr[i] - mean; q = fmaf(d, d, q); }
  const float rstd = rsqrtf(block_sum(q) / (float)c + eps);
  for (int i = threadIdx.x; i < c; i += 256) {
    float t = (xr[i] - mean) * rstd * gamma[i] + beta[i];
    y[(size_t)row * c + i] = silu ? silu_f(t) : t;
  }
}

}  // namespace

static int check_launch(const char* what) {
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) { mvd_set_error("%s launch: %s", what, hipGetErrorString(e)); return -3; }
  return 0;
}

// One-pass form: applicable when a slice of gpw groups (the fewest whose channel count is a multiple of 8) of one image
// fits the registers of one workgroup, and either the slices are small or there are enough of them to fill the chip.
static bool launch_gn_slice(const bf16_t* x0, const bf16_t* x1, int c0, int c1, int batch, int hw, int groups, float eps,
                            const float* gamma, const float* beta, int silu, bf16_t* y, hipStream_t s) {
  static const int enable = MVD_ENV_INT("MVD_GN_SLICE", 1);
  const int C = c0 + c1, cg = C / groups;
  if (!enable || cg < 8 || (cg & 1)) return false;
  int gpw = 1;
  while (gpw <= groups && ((gpw * cg) % 8 || groups % gpw)) gpw <<= 1;
  if (gpw > groups) return false;
  const int cs = gpw * cg, nvec = cs / 8, nslice = groups / gpw;
  // a source boundary inside a vector cannot happen (c0 % 8 == 0); inside a slice it can and is handled per vector
  int threads = 1024, nv = 0;
  for (int th : {256, 512, 1024}) {       // the smallest block that keeps <= 4 vectors per thread, else the largest
    const int npl = th / nvec;
    if (npl < 1) continue;
    threads = th; nv = (hw + npl - 1) / npl;
    if (nv <= 4) break;
  }
  if (nv < 1 || nv > 21) return false;
  const long slice_bytes = (long)hw * cs * 2, nwg = (long)batch * nslice;
  // few big slices: the two-kernel form has more parallelism.  (Round 5 re-measured the one-pass form for one image's 64x64 map --
  // 8 slices of 327 KB, 13 launches fewer per pass: debug flag 2097152 -- on one box, three alternations: cfg2 4.27 -> 4.39 ms,
  // cfg3 cold 6.50 -> 6.61 ms.  Fewer launches, more time: the floor stays.)
  const long min_wg = (mvd_debug_flags() & 2097152) ? 8 : 128;
  if (slice_bytes > 96 * 1024 && nwg < min_wg) return false;
  const dim3 grid((unsigned)nwg), blk((unsigned)threads);
#define GN_SLICE(NVT) hipLaunchKernelGGL(gn_slice_kernel<NVT>, grid, blk, 0, s, x0, x1, c0, c1, batch, hw, groups, gpw, eps, gamma, beta, silu, y)
  if (nv <= 2) GN_SLICE(2); else if (nv <= 4) GN_SLICE(4); else if (nv <= 8) GN_SLICE(8); else if (nv <= 16) GN_SLICE(16); else GN_SLICE(21);
#undef GN_SLICE
  return true;
}

// (A one-launch cooperative form for a single image -- rows dealt to co-resident workgroups, in-kernel rendezvous -- was
//  built and measured in round 3: correct, but the rendezvous costs 12-15 us, more than the boundary + second launch of the
//  two-kernel form it would replace; profiles/r03_probe_groupnorm_cooperative.log.)
int mvd_launch_groupnorm(const bf16_t* x0, const bf16_t* x1, int c0, int c1, int batch, int hw, int groups, float eps,
                         const float* gamma, const float* beta, int silu, bf16_t* y, float* ws, hipStream_t s) {
  const int C = c0 + c1;
  if (!x0 || (c1 && !x1) || !y || !ws || !gamma || !beta || batch <= 0 || hw <= 0 || groups <= 0 || groups > 32 ||
      (C % groups) || (c0 % 8) || (c1 % 8) || C > 8192) {
    mvd_set_error("groupnorm: bad arguments (c0=%d c1=%d batch=%d hw=%d groups=%d)", c0, c1, batch, hw, groups);
    return -1;
  }
  if (launch_gn_slice(x0, x1, c0, c1, batch, hw, groups, eps, gamma, beta, silu, y, s)) return check_launch("gn_slice");
  const int vec = C / 8;
  if (vec > 1024) { mvd_set_error("groupnorm: C=%d too wide", C); return -1; }
  const int R = vec >= 256 ? 1 : 256 / vec;
  const int threads = ((vec * R + 63) / 64) * 64;
  // enough workgroups for the statistics pass at small batch, at least 8 rows per chunk
  static const int rows_target = MVD_ENV_INT("MVD_GN_ROWS", 128);
  int nchunk = hw / rows_target;
  // (every apply block re-merges the nchunk partials of its image, so more chunks than the statistics pass needs to be
  //  parallel cost more there than they win here: at batch 1, 64 x 64 x 320, a floor of 512 blocks -- 256 chunks -- made the
  //  apply kernel 22.7 us (rocprof); 64 blocks: GroupNorm 1.05 -> 0.88 ms per batch-1 forward, 128: 0.90)
#ifndef MVD_GN_MINBLOCKS
#define MVD_GN_MINBLOCKS 64
#endif
  if ((long)nchunk * batch < MVD_GN_MINBLOCKS) nchunk = (MVD_GN_MINBLOCKS + batch - 1) / batch;
  if (nchunk > hw / 8) nchunk = hw / 8;
  nchunk = nchunk < 1 ? 1 : (nchunk > MVD_GN_MAXCHUNK ? MVD_GN_MAXCHUNK : nchunk);
  const int rpc = (hw + nchunk - 1) / nchunk;
  nchunk = (hw + rpc - 1) / rpc;      // no empty trailing chunk (hw = 1296, 64 chunks: rpc 21 -> 62 chunks)
  const size_t sh1 = (size_t)R * C * 2 * sizeof(float);
  hipLaunchKernelGGL(gn_stats_kernel, dim3(nchunk, batch), dim3(threads), sh1, s, x0, x1, c0, c1, hw, groups, rpc, R, ws);
  if (int r = check_launch("gn_stats")) return r;
  // apply: ~16K elements per block, fewer when that would leave CUs idle (small batch)
  static const long per_blk_env = MVD_ENV_INT("MVD_GN_APPLY_ELEMS", 65536);
  long per_blk = per_blk_env;
  const long total_el = (long)batch * hw * C;
  if (total_el / per_blk < 1024) per_blk = total_el / 1024 < 2048 ? 2048 : total_el / 1024;
  int rows_per_blk = (int)((per_blk + C - 1) / C);
  if (rows_per_blk < 1) rows_per_blk = 1;
  if (rows_per_blk > hw) rows_per_blk = hw;
  const int nblk = (hw + rows_per_blk - 1) / rows_per_blk;
  const size_t sh2 = (size_t)(2 * groups + 2 * C) * sizeof(float);
  hipLaunchKernelGGL(gn_apply_kernel, dim3(nblk, batch), dim3(256), sh2, s, x0, x1, c0, c1, hw, groups, nchunk, rpc, eps, gamma,
                     beta, silu, ws, rows_per_blk, y);
  return check_launch("gn_apply");
}

int mvd_launch_layernorm(const bf16_t* x, int rows, int c, float eps, const float* gamma, const float* beta, bf16_t* y,
                         hipStream_t s) {
  if (!x || !y || !gamma || !beta || rows <= 0 || c <= 0 || (c % 8) || c > 64 * 8 * 4) {
    mvd_set_error("layernorm: bad arguments rows=%d c=%d", rows, c);
    return -1;
  }
  const int grid = (rows + 3) / 4;
  if (c <= 512) hipLaunchKernelGGL(ln_kernel<1>, dim3(grid), dim3(256), 0, s, x, rows, c, eps, gamma, beta, y);
  else if (c <= 1024) hipLaunchKernelGGL(ln_kernel<2>, dim3(grid), dim3(256), 0, s, x, rows, c, eps, gamma, beta, y);
  else hipLaunchKernelGGL(ln_kernel<4>, dim3(grid), dim3(256), 0, s, x, rows, c, eps, gamma, beta, y);
  return check_launch("layernorm");
}

int mvd_launch_refstats(const bf16_t* x, int batch, int hw, int c, float* stats, hipStream_t s) {
  if (!x || !stats || batch <= 0 || hw <= 0 || c <= 0 || (c % 8)) { mvd_set_error("refstats: bad arguments"); return -1; }
  hipLaunchKernelGGL(refstats_kernel, dim3(hw), dim3(256), 0, s, x, batch, hw, c, stats);
  return check_launch("refstats");
}
int mvd_launch_refapply(const bf16_t* x, int batch, int hw, int c, const float* mean_k, bf16_t* y, hipStream_t s) {
  if (!x || !y || !mean_k || batch <= 0 || hw <= 0 || c <= 0 || (c % 8)) { mvd_set_error("refapply: bad arguments"); return -1; }
  const long total = (long)batch * hw * (c >> 3);
  long grid = (total + 255) / 256; if (grid > 4096) grid = 4096;
  hipLaunchKernelGGL(refapply_kernel, dim3((unsigned)grid), dim3(256), 0, s, x, batch, hw, c, mean_k, y);
  return check_launch("refapply");
}
int mvd_launch_refnorm(const bf16_t* x, int batch, int hw, int c, bf16_t* y, hipStream_t s) {
  if (!x || !y || batch <= 0 || hw <= 0 || c <= 0 || (c % 8)) { mvd_set_error("refnorm: bad arguments"); return -1; }
  hipLaunchKernelGGL(refnorm_kernel, dim3(hw), dim3(256), 0, s, x, batch, hw, c, y);
  return check_launch("refnorm");
}

int mvd_launch_layernorm_f32(const float* x, int rows, int c, float eps, const float* gamma, const float* beta, int silu,
                             float* y, hipStream_t s, int nseg) {
  if (!x || !y || !gamma || !beta || rows <= 0 || c <= 0 || nseg < 1) { mvd_set_error("layernorm_f32: bad arguments"); return -1; }
  hipLaunchKernelGGL(ln_f32_kernel, dim3(rows), dim3(256), 0, s, x, c, eps, gamma, beta, silu, y, nseg);
  return check_launch("layernorm_f32");
}
