// bf16 MFMA GEMM / implicit-GEMM 3x3 convolution for gfx950 (MI355X).
//
//   out[M][N] = alpha * ( A[M][K] . W[N][K]^T + bias[N] + rowvec[batch(m)][N] ) + res[M][N]
//
// * A is either a dense token-major matrix (optionally the channel concat of two
//   sources) or the *virtual* im2col matrix of an NHWC feature map (3x3, pad 1, stride
//   1/2, optional fused nearest-2x upsample).  Up to two K segments are accumulated into
//   the same tile, which fuses a ResnetBlock2D's conv2 with its 1x1 conv_shortcut and an
//   attention out-projection with the adapter's to_out_ref.
// * W is [N][K] with K contiguous (the nn.Linear layout; conv weights are packed
//   [Cout][ky][kx][Cin] by the host).
// * v_mfma_f32_16x16x32_bf16 with the roles swapped (W rows feed the MFMA "A" operand) so
//   each lane ends up with 4 consecutive output channels of one row -> 8-byte stores.
// * Tiles are staged through LDS in 64-wide K slabs, XOR-swizzled at 16-byte granularity
//   ((row>>1)&7) so both the staging ds_write_b128 and the fragment ds_read_b128 are
//   bank-conflict free; double-buffered, global loads for slab t+1 are issued before the
//   MFMAs of slab t and written to LDS after them (one barrier per slab).
// * blockIdx -> tile mapping is XCD-aware: consecutive tiles (same A rows, different N
//   tile) land on the same XCD so the A slab is fetched from HBM once per XCD L2.
#include <stdlib.h>
#include <string.h>
#include "kernels.h"

thread_local MvdLaunchPlan g_mvd_last_gemm = {-1, 1, 0, 0, 0, 0};

namespace {

template <int BM_, int BN_, int WM_, int WN_>
struct Cfg {
  static constexpr int BM = BM_, BN = BN_, WM = WM_, WN = WN_;
  static constexpr int NT = 64 * WM * WN;
  static constexpr int WTM = BM / WM, WTN = BN / WN;
  static constexpr int TM = WTM / 16, TN = WTN / 16;
  static constexpr int A_CHUNKS = BM * 8, B_CHUNKS = BN * 8;
  static constexpr int A_IT = (A_CHUNKS + NT - 1) / NT;
  static constexpr int B_IT = (B_CHUNKS + NT - 1) / NT;
  static constexpr int ROWS_PER_IT = NT / 8;
  static constexpr int A_BYTES = BM * 128, B_BYTES = BN * 128;
  static constexpr int STAGE_BYTES = A_BYTES + B_BYTES;
  static constexpr int LDS_BYTES = 2 * STAGE_BYTES;
  // register budget: keep as many workgroups co-resident per CU as LDS allows (2nd __launch_bounds__ argument
  // = waves per SIMD); without it hipcc spends up to 512 registers per lane and halves the residency
  static constexpr int WG_PER_CU = (160 * 1024 / LDS_BYTES) > 4 ? 4 : (160 * 1024 / LDS_BYTES);
  static constexpr int MIN_WAVES = (WG_PER_CU * NT / 256) < 1 ? 1 : (WG_PER_CU * NT / 256 > 4 ? 4 : WG_PER_CU * NT / 256);
  static_assert(WTM % 16 == 0 && WTN % 16 == 0, "wave tile must be a multiple of the MFMA tile");
  static_assert(A_CHUNKS % NT == 0, "A slab must divide evenly over the threads");
};

MVD_DEVINL int swz_off(int row, int chunk) { return row * 128 + ((chunk ^ ((row >> 1) & 7)) << 4); }

// AMODE: 0 = dense A (one segment), 1 = implicit 3x3 conv, 2 = conv followed by a dense (1x1 shortcut) segment.
// The slab cursor (tap / channel offset) lives in scalar registers: it depends on kernel arguments only.
// GLDS: stage slabs with global_load_lds_dwordx4 (LDS-DMA: no VGPR round trip, no ds_write).  The LDS
// image is identical to the register-staged one: the DMA writes lane-linear, so the XOR swizzle is applied
// to the per-lane SOURCE column instead of the LDS address.  Out-of-image conv taps read a 16-byte zero buffer.
__device__ __attribute__((aligned(16))) unsigned int g_zero16[4] = {0u, 0u, 0u, 0u};

MVD_DEVINL void glds16(const void* gsrc, void* lds_wave_base) {
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)gsrc,
                                   (__attribute__((address_space(3))) void*)lds_wave_base, 16, 0, 0);
}

// The kernel is PERSISTENT: gridDim.x workgroups walk the tile list (XCD-contiguous chunks) and the K-slab
// pipeline runs straight across tile boundaries -- the first slab of the next tile is already in flight while
// the last slab of the current tile is multiplied and its epilogue runs, so short-K GEMMs (K = 320: five slabs)
// do not pay a load-latency prologue per tile.
// DBG: measurement instantiations (probe builds only, -DMVD_PROBE) honour a.dbg; product instantiations carry no such branches.
template <class C, int AMODE, bool GLDS, bool SPLITK, bool DBG>
__global__ __launch_bounds__(C::NT, C::MIN_WAVES) void gemm_kernel(const MvdGemmArgs a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave / C::WN, wn = wave % C::WN;
  const int ntn = a.N / C::BN;
  const int ntm = (a.M + C::BM - 1) / C::BM;
  const int lrow = tid >> 3;
  // 16-byte K chunk this thread fetches: register staging swizzles the LDS address, LDS-DMA the source column
  const int kc = GLDS ? ((tid & 7) ^ ((lrow >> 1) & 7)) : (tid & 7);
  const int wave_chunk0 = tid & ~63;   // first chunk id of this wave (LDS-DMA destination is wave-uniform)

  // ---- tile walk: XCD x (= blockIdx & 7) owns tiles [tstart, tstart + tcnt); its workgroups stride through them
  const int S = SPLITK ? a.splitk : 1;            // split-K: each tile is S work items over disjoint slab ranges
  const int ntiles = ntn * ntm * S;               // (work items)
  const int xcd = blockIdx.x & 7, xj = blockIdx.x >> 3;
  const int gx = (gridDim.x >> 3) + ((int)(gridDim.x & 7) > xcd ? 1 : 0);   // workgroups on this XCD
  const int tq = ntiles >> 3, tr = ntiles & 7;
  const int tstart = xcd < tr ? xcd * (tq + 1) : tr * (tq + 1) + (xcd - tr) * tq;
  const int tend = tstart + tq + (xcd < tr ? 1 : 0);
  int tile = tstart + xj;
  if (tile >= tend) return;

  constexpr bool HAS_CONV = AMODE != 0;
  const MvdASeg& cs = a.seg[0];                       // conv segment (AMODE 1, 2)
  const MvdASeg& ds = a.seg[AMODE == 2 ? 1 : 0];      // dense segment (AMODE 0, 2)
  const int conv_c = cs.c0, conv_inW = cs.inW, conv_ups = cs.ups;
  const int limH = conv_ups ? 2 * cs.inH : cs.inH, limW = conv_ups ? 2 * cs.inW : cs.inW;
  const bf16_t* conv_p = cs.p0;
  const bf16_t* dp0 = ds.p0;
  const bf16_t* dp1 = ds.p1;
  const int dc0 = ds.c0, dc1 = ds.c1;
  const int nkt_conv = HAS_CONV ? (9 * conv_c) / 64 : 0;
  const int nkt = a.Ktot / 64;

  // ---- loader state (belongs to the tile whose slabs are being fetched -- may run one tile ahead)
  int a_m[C::A_IT], a_pb[C::A_IT], a_yx[C::A_IT];   // a_yx = (oy*stride) | (ox*stride) << 16
  int ld_n0 = 0;
  auto setup_loader = [&](int work) {
    const int t = S == 1 ? work : work / S;
    const int m0 = (t / ntn) * C::BM;
    ld_n0 = (t % ntn) * C::BN;
#pragma unroll
    for (int i = 0; i < C::A_IT; ++i) {
      int m = m0 + lrow + i * C::ROWS_PER_IT;
      m = m < a.M ? m : a.M - 1;
      a_m[i] = m;
      a_pb[i] = 0; a_yx[i] = 0;
      if (HAS_CONV) {
        const int b = m / a.rows_per_batch;
        const int rem = m - b * a.rows_per_batch;
        const int oy = rem / a.outW, ox = rem - oy * a.outW;
        a_pb[i] = b * cs.inH * cs.inW;
        a_yx[i] = (oy * cs.stride + cs.asym) | ((ox * cs.stride + cs.asym) << 16);   // asym: the window starts AT (2oy, 2ox)
      }
    }
  };

  u32x4 ra[C::A_IT], rb[C::B_IT];
  // fetch slab lk of the loader's tile into stage st (or into registers); the slab position (tap, channel
  // offset) is derived from lk alone so that it stays in scalar registers
  auto load_slab = [&](int st, int lk) {
    unsigned char* sa = smem + st * C::STAGE_BYTES;
    unsigned char* sb = sa + C::A_BYTES;
    if (HAS_CONV && (AMODE == 1 || lk < nkt_conv)) {
      // conv K order is [channel slice][tap][64 channels]: the nine taps of one 64-channel slice are consecutive
      // slabs, so a workgroup re-reads the same ~50 KB of the feature map nine times from L2 instead of cycling
      // through the whole 3-row x C window (~250 KB per workgroup, > L2 per XCD with 64 resident workgroups)
      const int ld_cs = lk / 9;
      const int ld_tap = lk - ld_cs * 9;
      const int ld_cc = ld_cs << 6;
      const int dy = ld_tap / 3, dx = ld_tap - dy * 3;
      const int col = ld_cc + kc * 8;
#pragma unroll
      for (int i = 0; i < C::A_IT; ++i) {
        const int iy = (a_yx[i] & 0xffff) - 1 + dy, ix = (a_yx[i] >> 16) - 1 + dx;
        const bool ok = (unsigned)iy < (unsigned)limH && (unsigned)ix < (unsigned)limW;
        const int sy = conv_ups ? (iy >> 1) : iy, sx = conv_ups ? (ix >> 1) : ix;
        const bf16_t* p = conv_p + (size_t)(a_pb[i] + sy * conv_inW + sx) * conv_c + col;
        if (GLDS) {
          glds16(ok ? (const void*)p : (const void*)g_zero16, sa + (wave_chunk0 + i * C::NT) * 16);
        } else {
          u32x4 v = {0u, 0u, 0u, 0u};
          if (ok) v = *reinterpret_cast<const u32x4*>(p);
          ra[i] = v;
        }
      }
    } else {
      const int ld_cc = (lk - nkt_conv) << 6;
      const bool first = ld_cc < dc0;
      const bf16_t* base = first ? dp0 : dp1;
      const int ld = first ? dc0 : dc1;
      const int col = (first ? ld_cc : ld_cc - dc0) + kc * 8;
#pragma unroll
      for (int i = 0; i < C::A_IT; ++i) {
        const bf16_t* p = base + (size_t)a_m[i] * ld + col;
        if (DBG && GLDS && (a.dbg & 4)) continue;   // measurement aid: no A traffic
        if (GLDS) glds16(p, sa + (wave_chunk0 + i * C::NT) * 16);
        else ra[i] = *reinterpret_cast<const u32x4*>(p);
      }
    }
#pragma unroll
    for (int i = 0; i < C::B_IT; ++i) {
      const int row = lrow + i * C::ROWS_PER_IT;
      if (C::B_CHUNKS % C::NT == 0 || row < C::BN) {
        const bf16_t* p = a.W + (size_t)(ld_n0 + row) * a.ldw + lk * 64 + kc * 8;
        if (DBG && GLDS && (a.dbg & 8)) continue;   // measurement aid: no W traffic
        if (GLDS) glds16(p, sb + (wave_chunk0 + i * C::NT) * 16);
        else rb[i] = *reinterpret_cast<const u32x4*>(p);
      }
    }
  };
  auto commit_slab = [&](int st) {   // make the fetched slab visible in LDS stage st (before the barrier)
    if (GLDS) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); return; }
    unsigned char* sa = smem + st * C::STAGE_BYTES;
    unsigned char* sb = sa + C::A_BYTES;
    const int kcr = tid & 7;
#pragma unroll
    for (int i = 0; i < C::A_IT; ++i)
      *reinterpret_cast<u32x4*>(sa + swz_off(lrow + i * C::ROWS_PER_IT, kcr)) = ra[i];
#pragma unroll
    for (int i = 0; i < C::B_IT; ++i) {
      const int row = lrow + i * C::ROWS_PER_IT;
      if (C::B_CHUNKS % C::NT == 0 || row < C::BN) *reinterpret_cast<u32x4*>(sb + swz_off(row, kcr)) = rb[i];
    }
  };

  f32x4 acc[C::TM][C::TN];
  const int fr = lane & 15, fq = lane >> 4;
  const float alpha = a.alpha;

  // The accumulators of a tile START at bias + per-batch row vector (out = alpha*(A.W^T + bias + rowvec) + res),
  // so the epilogue needs no operand registers for them.  lane holds out[m][n..n+3]: m = tile row (lane&15),
  // n = 4*(lane>>4) + reg.
  auto init_acc = [&](int m0, int n0) {
    asm volatile("" : "+s"(m0), "+s"(n0));   // keep the address arithmetic here (not hoisted into live registers)
    const int nb = n0 + wn * C::WTN + fq * 4;
#pragma unroll
    for (int i = 0; i < C::TM; ++i) {
#pragma unroll
      for (int j = 0; j < C::TN; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
    if (SPLITK) return;
    if (a.bias) {
#pragma unroll
      for (int j = 0; j < C::TN; ++j) {
        const f32x4 bv = *reinterpret_cast<const f32x4*>(a.bias + nb + j * 16);
#pragma unroll
        for (int i = 0; i < C::TM; ++i) acc[i][j] = bv;
      }
    }
    if (C::TN <= 5 && a.rowvec && a.rows_per_batch % C::BM == 0) {
      // the whole tile lies in one batch element (every level but the 8x8 one): one vector for all rows, one load batch
      const float* rv = a.rowvec + (size_t)(m0 / a.rows_per_batch) * a.ld_rowvec + nb;
#pragma unroll
      for (int j = 0; j < C::TN; ++j) {
        const f32x4 r = *reinterpret_cast<const f32x4*>(rv + j * 16);
#pragma unroll
        for (int i = 0; i < C::TM; ++i) acc[i][j] += r;
      }
    } else if (C::TN <= 5 && a.rowvec) {   // (GEGLU-only configs never carry a row vector)
#pragma unroll
      for (int i = 0; i < C::TM; ++i) {
        int m = m0 + wm * C::WTM + i * 16 + fr;
        m = m < a.M ? m : a.M - 1;
        const float* rv = a.rowvec + (size_t)(m / a.rows_per_batch) * a.ld_rowvec + nb;
#pragma unroll
        for (int j = 0; j < C::TN; ++j) acc[i][j] += *reinterpret_cast<const f32x4*>(rv + j * 16);
      }
    }
  };

  // Residual loads are issued as one batch per row tile: while an LDS-DMA is in flight hipcc waits vmcnt(0) for
  // every ordinary load, so load-use-load-use would serialise the epilogue into dozens of memory round trips.
  auto epilogue = [&](int m0, int n0, int ks) {
    asm volatile("" : "+s"(m0), "+s"(n0));   // keep the address arithmetic here (not hoisted into live registers)
    const int nb = n0 + wn * C::WTN + fq * 4;
    if (SPLITK) {   // raw fp32 partial tile; bias / residual / activation are applied by the reduce kernel
      float* pp = a.part + (size_t)ks * a.M * a.N;
#pragma unroll
      for (int i = 0; i < C::TM; ++i) {
        const int m = m0 + wm * C::WTM + i * 16 + fr;
#pragma unroll
        for (int j = 0; j < C::TN; ++j)
          if (m < a.M) *reinterpret_cast<f32x4*>(pp + (size_t)m * a.N + nb + j * 16) = acc[i][j];
      }
      return;
    }
    // Residual rows are fetched for RG row tiles at a time: hipcc waits vmcnt(0) for them (an LDS-DMA is in flight),
    // which also drains the stores issued so far, so every load batch costs a full memory round trip -- 2 (dense) or
    // 4 (conv: fewer spare registers) per tile instead of one per row tile.
    constexpr int RG = (C::TM % 4 == 0 && AMODE == 0) ? 4 : (C::TM % 2 == 0 ? 2 : 1);
#pragma unroll
    for (int i0 = 0; i0 < C::TM; i0 += RG) {
      u32x2 res_r[RG][C::TN > 5 ? 1 : C::TN];
      if (!(C::TN > 5 || a.geglu) && a.res) {
#pragma unroll
        for (int g = 0; g < RG; ++g) {
          int mr = m0 + wm * C::WTM + (i0 + g) * 16 + fr;
          mr = mr < a.M ? mr : a.M - 1;
          const bf16_t* rp = a.res + (size_t)mr * a.ldres + nb;
#pragma unroll
          for (int j = 0; j < (C::TN > 5 ? 1 : C::TN); ++j) res_r[g][j] = *reinterpret_cast<const u32x2*>(rp + j * 16);
        }
      }
#pragma unroll
      for (int g = 0; g < RG; ++g) {
      const int i = i0 + g;
      const int m = m0 + wm * C::WTM + i * 16 + fr;
      const bool live = m < a.M;
      if (!(C::TN > 5 || a.geglu)) {       // configs with TN > 5 exist for GEGLU only
#pragma unroll
        for (int j = 0; j < (C::TN > 5 ? 1 : C::TN); ++j) {
          const int n = nb + j * 16;
          f32x4 v = acc[i][j] * alpha;
          if (a.res) {
            v[0] += bflo(res_r[g][j][0]); v[1] += bfhi(res_r[g][j][0]); v[2] += bflo(res_r[g][j][1]); v[3] += bfhi(res_r[g][j][1]);
          }
          if (!live || (DBG && (a.dbg & 1))) continue;
          if (a.out_f32) {
            *reinterpret_cast<f32x4*>(reinterpret_cast<float*>(a.out) + (size_t)m * a.ldo + n) = v;
          } else {
            u32x2 o = {pack2bf(v[0], v[1]), pack2bf(v[2], v[3])};
            *reinterpret_cast<u32x2*>(reinterpret_cast<bf16_t*>(a.out) + (size_t)m * a.ldo + n) = o;
          }
        }
      } else {
        if constexpr (C::TN % 2 == 0) {
#pragma unroll
          for (int j = 0; j < C::TN; j += 2) {
            const f32x4 v = acc[i][j], g = acc[i][j + 1];    // packed rows: 16 value | 16 gate (bias already in)
            if (!live || (DBG && (a.dbg & 1))) continue;
            const int no = (n0 + wn * C::WTN) / 2 + (j / 2) * 16 + fq * 4;
            u32x2 o = {pack2bf(v[0] * gelu_erf_f(g[0]), v[1] * gelu_erf_f(g[1])),
                       pack2bf(v[2] * gelu_erf_f(g[2]), v[3] * gelu_erf_f(g[3]))};
            *reinterpret_cast<u32x2*>(reinterpret_cast<bf16_t*>(a.out) + (size_t)m * a.ldo + no) = o;
          }
        }
      }
      }
    }
  };

  // slab range of a work item (all of K unless split-K); kept out of the slab loop: the integer divisions by
  // the runtime split factor are ~40 instructions each
  auto slab_range = [&](int work, int& k0, int& k1) {
    if (S == 1) { k0 = 0; k1 = nkt; return; }
    const int ks = work % S;
    k0 = (ks * nkt) / S;
    k1 = ((ks + 1) * nkt) / S;
  };
  int kt0, kt1;
  slab_range(tile, kt0, kt1);
  setup_loader(tile);
  load_slab(0, kt0);
  {
    const int tl0 = S == 1 ? tile : tile / S;
    init_acc((tl0 / ntn) * C::BM, (tl0 % ntn) * C::BN);
  }
  commit_slab(0);
  __syncthreads();
  int cur = 0;
  for (;;) {
    const int tl = S == 1 ? tile : tile / S;
    const int ks = S == 1 ? 0 : tile - tl * S;
    const int m0 = (tl / ntn) * C::BM, n0 = (tl % ntn) * C::BN;
    const int next_tile = tile + gx;
    const bool have_next = next_tile < tend;
    int nkt0 = 0, nkt1 = 0;
    if (have_next) slab_range(next_tile, nkt0, nkt1);
    for (int kt = kt0; kt < kt1; ++kt) {
      const bool last_k = kt + 1 == kt1;
      const bool more = !last_k || have_next;
      if (more) {
        if (last_k) setup_loader(next_tile);   // the loader runs ahead into the next work item
        load_slab(cur ^ 1, last_k ? nkt0 : kt + 1);
      }
      const unsigned char* sa = smem + cur * C::STAGE_BYTES;
      const unsigned char* sb = sa + C::A_BYTES;
#pragma unroll
      for (int s2 = 0; s2 < 2; ++s2) {
        if constexpr (C::TM > 4) {
          // tall wave tile (128 rows): 160 accumulator registers, so the A fragments are streamed two at a time
          // (sched_barrier keeps hipcc from hoisting all eight reads and spilling)
          bf16x8 wf[C::TN];
#pragma unroll
          for (int j = 0; j < C::TN; ++j)
            wf[j] = *reinterpret_cast<const bf16x8*>(sb + swz_off(wn * C::WTN + j * 16 + fr, s2 * 4 + fq));
#ifdef MVD_GEMM_NO_SWP
#pragma unroll
          for (int i = 0; i < C::TM; i += 2) {
            const bf16x8 a0 = *reinterpret_cast<const bf16x8*>(sa + swz_off(wm * C::WTM + i * 16 + fr, s2 * 4 + fq));
            const bf16x8 a1 = *reinterpret_cast<const bf16x8*>(sa + swz_off(wm * C::WTM + (i + 1) * 16 + fr, s2 * 4 + fq));
#pragma unroll
            for (int j = 0; j < C::TN; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[j], a0, acc[i][j], 0, 0, 0);
#pragma unroll
            for (int j = 0; j < C::TN; ++j) acc[i + 1][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[j], a1, acc[i + 1][j], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
          }
#else
          // software pipeline over BOTH 32-wide halves of the slab (8 steps of two row tiles): the next pair of A
          // fragments is read while the MFMAs of the current pair run, and during the last pair of the first half each
          // W fragment is replaced in place by its second-half successor as soon as its last MFMA has issued -- the
          // second half starts without waiting for LDS
          if (s2 == 1) continue;
          auto a_at = [&](int q) __attribute__((always_inline)) -> bf16x8 {   // q = half * TM + row tile
            return *reinterpret_cast<const bf16x8*>(sa + swz_off(wm * C::WTM + (q % C::TM) * 16 + fr, (q / C::TM) * 4 + fq));
          };
          bf16x8 a0 = a_at(0), a1 = a_at(1);
#pragma unroll
          for (int q = 0; q < 2 * C::TM; q += 2) {
            const int i = q % C::TM;
            bf16x8 n0 = a0, n1 = a1;
            if (q + 2 < 2 * C::TM) { n0 = a_at(q + 2); n1 = a_at(q + 3); }
            __builtin_amdgcn_sched_barrier(0);
            __builtin_amdgcn_s_setprio(1);
#pragma unroll
            for (int j = 0; j < C::TN; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[j], a0, acc[i][j], 0, 0, 0);
#pragma unroll
            for (int j = 0; j < C::TN; ++j) {
              acc[i + 1][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[j], a1, acc[i + 1][j], 0, 0, 0);
              if (q == C::TM - 2)   // last pair of the first half
                wf[j] = *reinterpret_cast<const bf16x8*>(sb + swz_off(wn * C::WTN + j * 16 + fr, 4 + fq));
            }
            __builtin_amdgcn_s_setprio(0);
            __builtin_amdgcn_sched_barrier(0);
            a0 = n0; a1 = n1;
          }
#endif
          continue;
        } else if constexpr (C::TN > 5) {
          // wide wave tile (160 columns, used for GEGLU where value/gate tiles must pair up): the W fragments are
          // streamed in pairs through ONE software pipeline over both 32-wide halves of the slab (the next pair is read
          // while the 2 x TM MFMAs of the current pair run; the A fragments of the second half replace those of the
          // first in place during its last pair)
          if (s2 == 1) continue;
          constexpr int NP = C::TN / 2;                        // W pairs per half
          auto w_at = [&](int q, int which) __attribute__((always_inline)) -> bf16x8 {   // q = half * NP + pair
            return *reinterpret_cast<const bf16x8*>(sb + swz_off(wn * C::WTN + (2 * (q % NP) + which) * 16 + fr, (q / NP) * 4 + fq));
          };
          bf16x8 af[C::TM];
#pragma unroll
          for (int i = 0; i < C::TM; ++i)
            af[i] = *reinterpret_cast<const bf16x8*>(sa + swz_off(wm * C::WTM + i * 16 + fr, fq));
          bf16x8 w0 = w_at(0, 0), w1 = w_at(0, 1);
#pragma unroll
          for (int q = 0; q < 2 * NP; ++q) {
            const int j = 2 * (q % NP);
            bf16x8 n0 = w0, n1 = w1;
            if (q + 1 < 2 * NP) { n0 = w_at(q + 1, 0); n1 = w_at(q + 1, 1); }
            __builtin_amdgcn_sched_barrier(0);
            __builtin_amdgcn_s_setprio(1);
#pragma unroll
            for (int i = 0; i < C::TM; ++i) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w0, af[i], acc[i][j], 0, 0, 0);
#pragma unroll
            for (int i = 0; i < C::TM; ++i) {
              acc[i][j + 1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w1, af[i], acc[i][j + 1], 0, 0, 0);
              if (q == NP - 1)   // last pair of the first half: this A fragment is done, fetch its second-half successor
                af[i] = *reinterpret_cast<const bf16x8*>(sa + swz_off(wm * C::WTM + i * 16 + fr, 4 + fq));
            }
            __builtin_amdgcn_s_setprio(0);
            __builtin_amdgcn_sched_barrier(0);
            w0 = n0; w1 = n1;
          }
          continue;
        }
        bf16x8 af[C::TM], wf[C::TN];
#pragma unroll
        for (int i = 0; i < C::TM; ++i)
          af[i] = *reinterpret_cast<const bf16x8*>(sa + swz_off(wm * C::WTM + i * 16 + fr, s2 * 4 + fq));
#pragma unroll
        for (int j = 0; j < C::TN; ++j)
          wf[j] = *reinterpret_cast<const bf16x8*>(sb + swz_off(wn * C::WTN + j * 16 + fr, s2 * 4 + fq));
        if (!(DBG && (a.dbg & 2))) {
#pragma unroll
          for (int i = 0; i < C::TM; ++i)
#pragma unroll
            for (int j = 0; j < C::TN; ++j)
              acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[j], af[i], acc[i][j], 0, 0, 0);
        } else {
#pragma unroll
          for (int i = 0; i < C::TM; ++i) asm volatile("" :: "v"(af[i]));
#pragma unroll
          for (int j = 0; j < C::TN; ++j) asm volatile("" :: "v"(wf[j]));
        }
      }
      if (last_k) {                            // next work item's first slab is in flight meanwhile
        epilogue(m0, n0, ks);
        if (have_next) {
          const int tn = S == 1 ? next_tile : next_tile / S;
          init_acc((tn / ntn) * C::BM, (tn % ntn) * C::BN);
        }
      }
      if (more) commit_slab(cur ^ 1);
      __syncthreads();
      cur ^= 1;
    }
    if (!have_next) break;
    tile = next_tile; kt0 = nkt0; kt1 = nkt1;
  }
}

// ---------------------------------------------------------------- split-K reduction + epilogue
__global__ __launch_bounds__(256) void splitk_reduce_kernel(const MvdGemmArgs a) {
  const long nvec = (long)a.M * (a.N >> 2);
  const int nv = a.N >> 2;
  for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < nvec; e += (long)gridDim.x * blockDim.x) {
    const int m = (int)(e / nv), n = (int)(e - (long)m * nv) * 4;
    f32x4 v = *reinterpret_cast<const f32x4*>(a.part + (size_t)m * a.N + n);
    for (int s = 1; s < a.splitk; ++s) v += *reinterpret_cast<const f32x4*>(a.part + ((size_t)s * a.M + m) * a.N + n);
    if (a.bias) v += *reinterpret_cast<const f32x4*>(a.bias + n);
    if (a.rowvec) v += *reinterpret_cast<const f32x4*>(a.rowvec + (size_t)(m / a.rows_per_batch) * a.ld_rowvec + n);
    v *= a.alpha;
    if (a.res) {
      const u32x2 r = *reinterpret_cast<const u32x2*>(a.res + (size_t)m * a.ldres + n);
      v[0] += bflo(r[0]); v[1] += bfhi(r[0]); v[2] += bflo(r[1]); v[3] += bfhi(r[1]);
    }
    if (a.out_f32) {
      *reinterpret_cast<f32x4*>(reinterpret_cast<float*>(a.out) + (size_t)m * a.ldo + n) = v;
    } else {
      u32x2 o = {pack2bf(v[0], v[1]), pack2bf(v[2], v[3])};
      *reinterpret_cast<u32x2*>(reinterpret_cast<bf16_t*>(a.out) + (size_t)m * a.ldo + n) = o;
    }
  }
}

struct CfgInfo { int bm, bn, tn_even; };

template <class C, int AMODE, bool GLDS, bool SPLITK, bool DBG>
int launch_mode3(const MvdGemmArgs& a, hipStream_t s) {
  static int per_cu = 0;   // resident workgroups per CU for this instantiation (LDS- and VGPR-limited)
  if (!per_cu) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_kernel<C, AMODE, GLDS, SPLITK, DBG>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, C::LDS_BYTES);
    if (e != hipSuccess) { mvd_set_error("gemm: hipFuncSetAttribute: %s", hipGetErrorString(e)); return -2; }
    int nb = 0;
    e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, gemm_kernel<C, AMODE, GLDS, SPLITK, DBG>, C::NT, C::LDS_BYTES);
    if (e != hipSuccess || nb < 1) nb = 1;
    per_cu = nb > 4 ? 4 : nb;
  }
  const int ntm = (a.M + C::BM - 1) / C::BM, ntn = a.N / C::BN;
  // persistent grid: as many workgroups as fit on the chip at once (LDS-limited), a multiple of the 8 XCDs
  int grid = 256 * per_cu;
  const int ntiles = ntm * ntn * (a.splitk > 1 ? a.splitk : 1);
  if (ntiles < grid) grid = ((ntiles + 7) / 8) * 8;
  g_mvd_last_gemm.tiles = ntiles; g_mvd_last_gemm.grid = grid; g_mvd_last_gemm.per_cu = per_cu;
  hipLaunchKernelGGL((gemm_kernel<C, AMODE, GLDS, SPLITK, DBG>), dim3(grid), dim3(C::NT), C::LDS_BYTES, s, a);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) { mvd_set_error("gemm launch: %s", hipGetErrorString(e)); return -3; }
  return 0;
}

template <class C, int AMODE, bool GLDS, bool SPLITK>
int launch_mode2(const MvdGemmArgs& a, hipStream_t s) {
#ifdef MVD_PROBE
  if (a.dbg) return launch_mode3<C, AMODE, GLDS, SPLITK, true>(a, s);
#endif
  return launch_mode3<C, AMODE, GLDS, SPLITK, false>(a, s);
}

template <class C, int AMODE, bool GLDS>
int launch_mode(const MvdGemmArgs& a, hipStream_t s) {
  if constexpr (GLDS) { if (a.splitk > 1) return launch_mode2<C, AMODE, true, true>(a, s); }
  return launch_mode2<C, AMODE, GLDS, false>(a, s);
}

// DMA_ONLY: the tile exists with LDS-DMA staging only (the 256x320 tile: its register-staged form spills 37-43 VGPRs to scratch
// memory and was never reachable in the product -- tools/lint_device_isa.py fails the build on any shipped kernel with scratch)
template <class C, bool DMA_ONLY = false>
int launch_cfg(const MvdGemmArgs& a, hipStream_t s, bool glds) {
  if (a.splitk > 1) glds = true;   // split-K exists for the LDS-DMA variants only
  if (glds || DMA_ONLY) {
    if (a.seg[0].mode == MVD_A_DENSE) return launch_mode<C, 0, true>(a, s);
    return a.nseg == 1 ? launch_mode<C, 1, true>(a, s) : launch_mode<C, 2, true>(a, s);
  }
  if constexpr (!DMA_ONLY) {
    if (a.seg[0].mode == MVD_A_DENSE) return launch_mode<C, 0, false>(a, s);
    return a.nseg == 1 ? launch_mode<C, 1, false>(a, s) : launch_mode<C, 2, false>(a, s);
  }
  return -1;
}

using C0 = Cfg<256, 160, 4, 2>;
using C1 = Cfg<256, 128, 4, 2>;
using C2 = Cfg<128, 160, 2, 2>;
using C3 = Cfg<128, 128, 2, 2>;
using C4 = Cfg<128, 64, 2, 2>;
using C5 = Cfg<64, 64, 2, 2>;
using C7 = Cfg<256, 320, 2, 4>;   // wave tile 128x80: 49 FLOP per LDS-read byte instead of 36
using C6 = Cfg<256, 320, 4, 2>;   // wave tile 64x160 (even number of 16-column tiles: GEGLU value/gate pairs)
// (a three-stage 256x160 experiment gave no gain: the kernel is LDS-read bound, not load-latency bound)
using C8 = Cfg<128, 320, 2, 4>;   // wave tile 64x80: twice the tiles of C7 for M = 8192 (deep levels) -> no split-K
const CfgInfo kCfgs[] = {{256, 160, 0}, {256, 128, 1}, {128, 160, 0}, {128, 128, 1}, {128, 64, 1}, {64, 64, 1}, {256, 320, 1}, {256, 320, 0},
                         {128, 320, 0}};
constexpr int kNumCfgs = 9;

}  // namespace

extern "C" int mvd_gemm_num_configs(void) { return kNumCfgs; }

// out[5] = {tile config, split-K factor, work items (tiles x split), workgroups launched, workgroups per CU} of the
// calling thread's last GEMM / conv launch
extern "C" int mvd_debug_last_gemm_plan(int* out) {
  if (!out) { mvd_set_error("last_gemm_plan: null argument"); return -1; }
  out[0] = g_mvd_last_gemm.cfg; out[1] = g_mvd_last_gemm.splitk; out[2] = g_mvd_last_gemm.tiles;
  out[3] = g_mvd_last_gemm.grid; out[4] = g_mvd_last_gemm.per_cu;
  return 0;
}

// 1 when the calling thread's last small-M split-K launch took the no-wait combine (asked for by the caller, or because its
// grid exceeded what the chip holds at once: gemm_sm.hip launch_sm3), else 0
extern "C" int mvd_debug_last_gemm_nowait(void) { return g_mvd_last_gemm.nowait; }

// Tile choice (tools/tune_gemm.py sweep on MI355X, profiles/r01_tune_gemm_B32_v2.log): the 256x320 tile wins wherever
// its grid (times a split-K of at most 2) can occupy the 256 CUs; below that, fall through 128x160 -> 128x128 ->
// 128x64 -> 64x64 until the grid has >= ~300 workgroups, else take the config with the most workgroups.
// split factor of the rule above (1: the rule does not apply)
static int deep_conv_split(const MvdGemmArgs& a, long t7) {
  static const int on = MVD_ENV_INT("MVD_GEMM_DEEP_CONV_SPLIT", 1);
  // (a.splitk: 0 = undecided, the engine asks mvd_gemm_pick_splitk first; a caller that fixed a shallow split -- the operator entry
  //  points with their default splitk = 1 -- keeps the tile that suits it: 32 unsplit 256x320 tiles would leave 224 CUs idle)
  if (!on || (mvd_debug_flags() & 262144) || a.seg[0].mode == MVD_A_DENSE || a.geglu || a.out_f32 || t7 < 24 || t7 > 64 || a.Ktot < 8192 || (a.splitk > 0 && a.splitk < 4)) return 1;
  int s = (int)(256 / t7);
  while (s > 1 && a.Ktot / 64 / s < 16) --s;
  return s > 16 ? 16 : s;
}

int mvd_gemm_pick_config(const MvdGemmArgs& a) {
  // 256x320 tile with 128x80 wave tiles (49 FLOP per LDS-read byte): the kernel is LDS-bandwidth bound, so this
  // is the fastest shape whenever its tile grid -- times a split-K of up to 8 -- can occupy the 256 CUs
  static const int use7 = MVD_ENV_INT("MVD_GEMM_BIG", 1);
  // (128x320 tiles instead of a two-way split-K at M = 8192: measured 1 % SLOWER end to end -- the W slab is
  //  re-fetched per 128 rows and the 64x80 wave tile reads more LDS per FLOP -- so it is an opt-in switch)
  static const int split_min_slabs = MVD_ENV_INT("MVD_GEMM_SPLIT_MINK", 4096) / 64;
  static const int use8 = MVD_ENV_INT("MVD_GEMM_C8", 0);
  if (use7 && a.N % 320 == 0 && a.M >= 1024) {
    const long t7 = (long)((a.M + 255) / 256) * (a.N / 320);
    if (a.geglu) { if (t7 >= 200) return 6; }
    // (a split of 2 at most: the fp32 partials of deeper splits cost more than the bigger tile gains)
    else if (t7 >= 200) return 7;
    else if (use8 && t7 * 2 >= 200) return 8;       // 128x320 tiles fill the chip without a split
    // too few 256x320 tiles (M = 8192 at the deep levels): a two-way split-K of the big tile only pays for long K
    // (K >= 4096: convolutions, ff2); shorter K goes to 128x160 tiles without a split -- and without the reduce pass
    // (measured at M 8192 x N 1280: K 1280 36 us vs 41 + 15 us, K 2560 67 vs 63 + 15 us, K 5120 a tie; end to end the
    //  rule is worth 0.2-0.5 %)
    else if (a.Ktot / 64 >= split_min_slabs && t7 * 2 >= 200) return 7;
    // the 3x3 convolutions of the 8x8 level at 32 images (M = 2048: 32 tiles, K = 11520 ... 23040): the big tile cut EIGHT ways
    // along K (256 workgroups, each slice >= 22 slabs) beats 512 work items of the 128x160 tile at split 4 by 7 / 16 / 3 %
    // (tools/tune_worst_shapes.py, profiles/r04_tune_worst_shapes.log) -- half the operand bytes per FLOP, same partial traffic
    else if (deep_conv_split(a, t7) > 1) return 7;
  }
  static const int order[] = {2, 3, 4, 5};
  int cfg = -1, first_valid = -1;
  long best_blocks = -1;
  for (int c : order) {
    if (a.N % kCfgs[c].bn) continue;
    if (a.geglu && !kCfgs[c].tn_even) continue;
    const long blocks = (long)((a.M + kCfgs[c].bm - 1) / kCfgs[c].bm) * (a.N / kCfgs[c].bn);
    if (first_valid < 0) {
      first_valid = c;
      // long K but too few tiles of the efficient shape: keep that tile, split-K supplies the parallelism
      if (blocks < 300 && !a.geglu && a.Ktot / 64 >= 16 && c <= 3 && a.M >= 512) return c;
    }
    if (blocks >= 300) return c;
    if (blocks > best_blocks) { best_blocks = blocks; cfg = c; }
  }
  return cfg;
}

int mvd_launch_gemm(const MvdGemmArgs& a, hipStream_t s, int force_cfg) {
  // ---- host-side shape validation: a wrong shape must never reach the kernel
  if (a.M <= 0 || a.N <= 0 || a.Ktot <= 0 || a.nseg < 1 || a.nseg > 2) { mvd_set_error("gemm: bad dims M=%d N=%d K=%d nseg=%d", a.M, a.N, a.Ktot, a.nseg); return -1; }
  int ksum = 0;
  for (int i = 0; i < a.nseg; ++i) {
    const MvdASeg& g = a.seg[i];
    if (g.mode == MVD_A_DENSE) {
      if (g.c0 % 64 || g.c1 % 64 || g.ksize != g.c0 + g.c1 || !g.p0 || (g.c1 && !g.p1)) { mvd_set_error("gemm: bad dense segment %d (c0=%d c1=%d ksize=%d)", i, g.c0, g.c1, g.ksize); return -1; }
    } else if (g.mode == MVD_A_CONV3) {
      if (g.c0 % 64 || g.c1 != 0 || g.ksize != 9 * g.c0 || !g.p0 || (g.stride != 1 && g.stride != 2) || (g.ups && g.stride != 1)) { mvd_set_error("gemm: bad conv segment %d", i); return -1; }
      if (g.asym && (g.asym != 1 || g.stride != 2 || (g.inH & 1) || (g.inW & 1))) { mvd_set_error("gemm: bottom/right-only padding needs stride 2 and an even input size"); return -1; }
      const int eh = g.ups ? 2 * g.inH : (g.stride == 2 ? (g.inH + 1) / 2 : g.inH);
      const int ew = g.ups ? 2 * g.inW : (g.stride == 2 ? (g.inW + 1) / 2 : g.inW);
      if (eh != a.outH || ew != a.outW || a.rows_per_batch != a.outH * a.outW || a.M % a.rows_per_batch) { mvd_set_error("gemm: conv geometry mismatch (in %dx%d out %dx%d rpb %d M %d)", g.inH, g.inW, a.outH, a.outW, a.rows_per_batch, a.M); return -1; }
    } else { mvd_set_error("gemm: bad mode"); return -1; }
    ksum += g.ksize;
  }
  if (a.nseg == 2 && !(a.seg[0].mode == MVD_A_CONV3 && a.seg[1].mode == MVD_A_DENSE)) { mvd_set_error("gemm: a second segment must be a dense segment after a conv segment"); return -1; }
  if (ksum != a.Ktot) { mvd_set_error("gemm: segment K sum %d != Ktot %d", ksum, a.Ktot); return -1; }
  if (a.ldw < a.Ktot || (a.ldw % 8) || !a.W) { mvd_set_error("gemm: bad weight stride ldw=%d (K=%d)", a.ldw, a.Ktot); return -1; }
  if (a.rows_per_batch <= 0) { mvd_set_error("gemm: rows_per_batch must be > 0"); return -1; }
  if (a.N % 64) { mvd_set_error("gemm: N=%d must be a multiple of 64", a.N); return -1; }
  if (a.splitk > 1 && (!a.part || a.geglu || a.splitk > (force_cfg >= 100 ? 64 : 16) || a.splitk > a.Ktot / 64)) { mvd_set_error("gemm: bad split-K request (splitk=%d)", a.splitk); return -1; }
  if (a.geglu && (a.out_f32 || a.res || a.rowvec)) { mvd_set_error("gemm: unsupported GEGLU epilogue combination"); return -1; }
  const int on = a.geglu ? a.N / 2 : a.N;
  if (a.ldo < on || (a.ldo % 4) || (a.res && (a.ldres % 4))) { mvd_set_error("gemm: bad leading dims"); return -1; }

  // force_cfg: -1 = heuristic; 0..5 = tile config with register staging; 8..13 = same tiles with LDS-DMA staging
  static const int default_glds = MVD_ENV_INT("MVD_GEMM_GLDS", 1);
  bool glds = default_glds != 0;
#ifdef MVD_PROBE
  static const int dbg = MVD_ENV_INT("MVD_GEMM_DEBUG", 0);
  if (dbg) const_cast<MvdGemmArgs&>(a).dbg = dbg;
#endif
  int cfg = force_cfg;
  // force_cfg >= 100: the small-M kernels of gemm_sm.hip, 100 + 10 * tile + ring depth (0: default 4)
  if (cfg >= 1000) { const_cast<MvdGemmArgs&>(a).w_blocked = 1; cfg -= 1000; }    // (+1000: W in the blocked LDS-image layout)
  if (a.w_blocked && cfg < 100) { mvd_set_error("gemm: the blocked weight layout is read by the small-M kernels only"); return -1; }
  if (cfg >= 100) {
    const int tile = (cfg - 100) / 10, ns = (cfg - 100) % 10;
    if (a.splitk > 1 && !a.tile_cnt) { mvd_set_error("gemm: the small-M kernels combine split-K in the kernel and need tile counters"); return -1; }
    g_mvd_last_gemm.cfg = cfg; g_mvd_last_gemm.splitk = a.splitk > 1 ? a.splitk : 1;
    return mvd_launch_gemm_sm(a, s, tile, ns ? ns : 4);
  }
#ifdef MVD_PROBE
  // probe builds only: force_cfg 15 = the ring-pipelined 256x320 experiment (gemm_ring.hip, not part of the product
  // library); MVD_GEMM_RING=1 routes every plain 256x320 launch to it
  static const int use_ring = MVD_ENV_INT("MVD_GEMM_RING", 0);
  if (cfg == 15) {
    if (a.N % 320 || a.geglu || a.out_f32) { mvd_set_error("gemm: the ring kernel needs N %% 320 == 0, bf16 output, no GEGLU"); return -1; }
    return mvd_launch_gemm_ring(a, s);
  }
#endif
  // force_cfg 16 / 17 = the lock-step (round-1) form of tile configs 6 / 7; 6 / 7 and the heuristic take the ping-pong kernels
  bool legacy = MVD_ENV_INT("MVD_GEMM_LEGACY", 0) != 0;
  if (cfg == 16 || cfg == 17) { legacy = true; cfg -= 10; }
  if (cfg == 14) { glds = true; cfg = 8; }       // force_cfg 14 = the 128x320 tile (LDS-DMA only)
  else if (cfg >= 8 && cfg < 14) { glds = true; cfg -= 8; } else if (cfg >= 0 && cfg < 6) { glds = false; }
  if (cfg < 0) cfg = mvd_gemm_pick_config(a);
  if (a.ln_c1) {   // LayerNorm fold: exists in the ping-pong kernels only (callers ask mvd_gemm_ln_fold_ok first)
    if ((cfg != 6 && cfg != 7) || legacy || a.dbg || !mvd_gemm_ln_fold_ok(a)) { mvd_set_error("gemm: LayerNorm fold not available for M=%d N=%d K=%d cfg=%d", a.M, a.N, a.Ktot, cfg); return -1; }
  }
  if (cfg < 0 || cfg >= kNumCfgs || a.N % kCfgs[cfg].bn || (a.geglu && !kCfgs[cfg].tn_even)) { mvd_set_error("gemm: no tile config for N=%d geglu=%d cfg=%d", a.N, a.geglu, cfg); return -1; }
  g_mvd_last_gemm.cfg = cfg; g_mvd_last_gemm.splitk = a.splitk > 1 ? a.splitk : 1;
#ifndef MVD_PROBE
  // tile configs the heuristic never picks (256x160, 256x128, 128x320) and the lock-step forms of 6 / 7 as a forced choice
  // exist in probe builds only (tools/build_variant.py <tag> -DMVD_PROBE)
  if (cfg == 0 || cfg == 1 || cfg == 8 || force_cfg == 16 || force_cfg == 17) { mvd_set_error("gemm: tile config %d exists in probe builds only", force_cfg >= 0 ? force_cfg : cfg); return -1; }
#endif
  switch (cfg) {
#ifdef MVD_PROBE
    case 0: return launch_cfg<C0>(a, s, glds);
    case 1: return launch_cfg<C1>(a, s, glds);
    case 8: return launch_cfg<C8>(a, s, true);
#endif
    case 2: return launch_cfg<C2>(a, s, glds);
    case 3: return launch_cfg<C3>(a, s, glds);
    case 4: return launch_cfg<C4>(a, s, glds);
    case 6:
      if (!a.geglu || a.seg[0].mode != MVD_A_DENSE || a.splitk > 1) { mvd_set_error("gemm: tile config 6 is GEGLU-only"); return -1; }
      if (!legacy && !(a.dbg & ~32) && mvd_gemm_pp_applicable(a)) return mvd_launch_gemm_pp(a, s);
      return launch_mode2<C6, 0, true, false>(a, s);
    case 7:
      if (!legacy && !(a.dbg & ~32) && !a.geglu && mvd_gemm_pp_applicable(a)) return mvd_launch_gemm_pp(a, s);
#ifdef MVD_PROBE
      if (use_ring && !a.out_f32 && !a.dbg) return mvd_launch_gemm_ring(a, s);
#endif
      return launch_cfg<C7, true>(a, s, true);
    default: return launch_cfg<C5>(a, s, glds);
  }
}

// Split-K heuristic: GEMMs whose tile grid cannot fill the chip (deep levels: M = batch * 64 pixels) but whose K
// is long are cut along K so that ~512 work items exist; the partials are summed by splitk_reduce_kernel.
int mvd_gemm_pick_splitk(const MvdGemmArgs& a) {
  if (a.geglu) return 1;
  const int cfg = mvd_gemm_pick_config(a);
  if (cfg < 0) return 1;
  const long tiles = (long)((a.M + kCfgs[cfg].bm - 1) / kCfgs[cfg].bm) * (a.N / kCfgs[cfg].bn);
  const int nkt = a.Ktot / 64;
  if (cfg == 7 && deep_conv_split(a, tiles) > 1) return deep_conv_split(a, tiles);
  if (cfg == 7 || cfg == 8) return (tiles >= 200 || nkt < 16) ? 1 : 2;   // one 115-147 KB workgroup per CU
  if (tiles >= 256 || nkt < 16) return 1;
  long s = 512 / tiles;                                       // two workgroups per CU
  if (s > nkt / 8) s = nkt / 8;
  // the fp32 partials (written and read back: 8 S M N bytes) against the weight bytes the split spreads over more CUs
  // (2 N K): up to 4 always, deeper while the partials stay below the weights -- the 8x8 / 16x16 levels at small batch
  // are pure weight streaming (M = 64, K = 11520: 29 MB of weights, 20 tiles), which a 4-way split leaves on 80 CUs
  long cap = (long)a.Ktot / (4L * a.M);
  cap = cap < 4 ? 4 : (cap > 16 ? 16 : cap);
  if (s > cap) s = cap;
  return s < 2 ? 1 : (int)s;
}

// the split factor the engine's schedule would use for a GEMM / conv of this size (tests drive mvd_op_* with it)
extern "C" int mvd_debug_pick_splitk(int m, int n, int k, int geglu) {
  MvdGemmArgs a; memset(&a, 0, sizeof(a));
  a.M = m; a.N = n; a.Ktot = k; a.geglu = geglu;
  return mvd_gemm_pick_splitk(a);
}
// the same for a 3x3 convolution (implicit GEMM: K = 9 * Cin + shortcut channels) -- the 8x8-level rule is for convolutions only
extern "C" int mvd_debug_pick_splitk_conv(int m, int n, int k) {
  MvdGemmArgs a; memset(&a, 0, sizeof(a));
  a.M = m; a.N = n; a.Ktot = k; a.seg[0].mode = MVD_A_CONV3;
  return mvd_gemm_pick_splitk(a);
}

int mvd_launch_splitk_reduce(const MvdGemmArgs& a, hipStream_t s) {
  if (a.splitk < 2 || !a.part || (a.N & 3)) { mvd_set_error("splitk_reduce: bad arguments"); return -1; }
  const long nvec = (long)a.M * (a.N >> 2);
  int grid = (int)((nvec + 255) / 256);
  if (grid > 2048) grid = 2048;
  hipLaunchKernelGGL(splitk_reduce_kernel, dim3(grid), dim3(256), 0, s, a);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) { mvd_set_error("splitk_reduce launch: %s", hipGetErrorString(e)); return -3; }
  return 0;
}
