// Small-M bf16 MFMA GEMM / implicit-GEMM 3x3 convolution for gfx950: the batch-1 path (M <= ~4096 rows: one image's
// 64x64 .. 8x8 feature maps, /root/reference/infer.py:111-122 runs exactly this).
//
//   out[M][N] = alpha * ( A[M][K] . W[N][K]^T + bias[N] + rowvec[batch(m)][N] ) + res[M][N]        (same contract as gemm.hip)
//
// At these sizes a GEMM is a few microseconds of arithmetic and the kernels of gemm.hip / gemm_pp.hip (built for M = 32 images)
// spend their time elsewhere: a persistent tile walk with ONE slab of prefetch is 5-45 dependent memory round trips per work
// item, split-K pays a second launch for the reduction, and the tile grid rarely matches the 256 CUs.  This kernel is built
// for latency instead:
//
// * ONE work item (output tile x K slice) per workgroup, grid = tiles x split: no tile loop, no loader-ahead bookkeeping.
// * An NSTAGE-deep LDS ring filled by buffer-addressed LDS-DMA: the first NSTAGE-1 K slabs are requested before anything is
//   waited for, one raw s_barrier per slab, counted vmcnt (the DMAs of the next NSTAGE-2 slabs stay in flight across it).
//   NSTAGE is a launch parameter (ring bytes = dynamic LDS), so K = 320 keeps the whole operand in flight at once.
// * bias / row vector / residual are fetched in the prologue, under the first slabs' latency, not after the last MFMA.
// * Split-K combines IN the kernel: every slice stores its fp32 partial tile write-through (sc1) and arrives on a per-tile
//   counter; the tile's slices wait for each other (they are all resident: the grid never exceeds what the chip holds) and
//   EVERY slice then sums a share of the tile's 16x16 sub-tiles over the slices, in slice order (bit-deterministic), and runs
//   the epilogue on it -- no reduce launch, no serial tail through one workgroup (the agent-scope hand-off of the CDNA4
//   guide: sc1 stores + vmcnt(0) + barrier + one relaxed agent atomic, sc1-load poll, sc1 loads of the partials, no fence).
// * Operand layouts, the LDS image (128-byte rows, 16-byte chunks XOR-swizzled by (row >> 1) & 7 on the SOURCE side) and
//   the MFMA roles (W rows feed the "A" operand of v_mfma_f32_16x16x32_bf16: a lane ends up with four consecutive output
//   channels of one row) are those of gemm.hip.
#include <stdlib.h>
#include <string.h>
#include "kernels.h"

namespace {

constexpr unsigned OOB = 0x80000000u;                   // voffset of a load that must return zeros (>= num_records)
typedef __attribute__((address_space(3))) void lds_void;

MVD_DEVINL void dma16(__amdgpu_buffer_rsrc_t rsrc, unsigned char* lds_wave_base, unsigned voff, unsigned soff) {
  __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (lds_void*)lds_wave_base, 16, (int)voff, (int)soff, 0, 0);
}

template <int N> MVD_DEVINL void wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }
// at most `ahead` groups of L loads each may still be in flight (the counter is 6 bits: a smaller number waits for more)
template <int L> MVD_DEVINL void wait_groups(int ahead) {
  constexpr int C1 = L > 63 ? 63 : L, C2 = 2 * L > 63 ? 63 : 2 * L, C3 = 3 * L > 63 ? 63 : 3 * L, C4 = 4 * L > 63 ? 63 : 4 * L,
                C5 = 5 * L > 63 ? 63 : 5 * L, C6 = 6 * L > 63 ? 63 : 6 * L;
  switch (ahead) {
    case 0: wait_vm<0>(); break;
    case 1: wait_vm<C1>(); break;
    case 2: wait_vm<C2>(); break;
    case 3: wait_vm<C3>(); break;
    case 4: wait_vm<C4>(); break;
    case 5: wait_vm<C5>(); break;
    default: wait_vm<C6>(); break;
  }
}

constexpr int SM_MAX_STAGES = 8;

// 256 threads = 4 waves as 2 x 2; wave tile (BM/2) x (BN/2).  AMODE: 0 dense A (1-2 sources), 1 implicit 3x3 conv (stride 1/2,
// fused nearest-2x upsample, bottom/right-only padding), 2 conv followed by a dense (1x1 shortcut) K segment.
// LNF: LayerNorm of the A rows folded in (MvdGemmArgs::ln_c1, as gemm_pp.hip: W carries gamma, out = rstd[m] * (acc - mean[m] *
// c1[n]) + c2[n]); the row sums / sums of squares are accumulated from the A fragments the MFMAs consume -- the two waves
// that share a row block take every other row tile each and trade (rstd, -rstd * mean) through LDS behind the ring.
template <int BM, int BN, int AMODE, bool SPLITK, bool GEGLU, bool LNF = false>
__global__ __launch_bounds__(256) void gemm_sm_kernel(const MvdGemmArgs a, const int nstage) {
  constexpr int WTM = BM / 2, WTN = BN / 2, TM = WTM / 16, TN = WTN / 16;
  constexpr int A_IT = BM / 32, B_IT = BN / 32, L = A_IT + B_IT;          // LDS-DMA instructions per wave per slab
  constexpr int A_BYTES = BM * 128, STAGE_BYTES = (BM + BN) * 128;
  constexpr bool HAS_CONV = AMODE != 0;
  static_assert(WTM % 16 == 0 && WTN % 16 == 0 && BM % 32 == 0 && BN % 32 == 0, "tile shape");
  static_assert(!GEGLU || (TN % 2 == 0 && AMODE == 0 && !SPLITK), "GEGLU: value/gate column tiles pair up inside a wave");
  static_assert(!LNF || (AMODE == 0 && !SPLITK), "the LayerNorm fold is a dense, unsplit form");
  constexpr int TS = LNF ? TM : 1;                                         // row tiles whose (half-K) statistics this wave accumulates
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 1, wn = wave & 1;
  const int lrow = tid >> 3;                                      // 0..31: row inside a 32-row DMA block
  const int kc = (tid & 7) ^ ((lrow >> 1) & 7);                   // source chunk (the XOR swizzle lives on the source side)
  const int fr = lane & 15, fq = lane >> 4;

  // ---- work item: logical id w = (tile, k slice), slices of a tile adjacent; XCD x gets a contiguous range of ids
  const int S = SPLITK ? a.splitk : 1;
  const int ntn = a.N / BN;
  const int work = xcd_remap((int)blockIdx.x, (int)gridDim.x);
  const int tl = S == 1 ? work : work / S;
  const int ks = S == 1 ? 0 : work - tl * S;
  const int m0 = (tl / ntn) * BM, n0 = (tl % ntn) * BN;
  const int nkt = a.Ktot / 64;
  const int kt0 = S == 1 ? 0 : (ks * nkt) / S;
  const int kt1 = S == 1 ? nkt : ((ks + 1) * nkt) / S;
  const int nk = kt1 - kt0;

  const MvdASeg& cs = a.seg[0];                       // conv segment (AMODE 1, 2)
  const MvdASeg& ds = a.seg[AMODE == 2 ? 1 : 0];      // dense segment (AMODE 0, 2)
  const int nkt_conv = HAS_CONV ? (9 * cs.c0) / 64 : 0;
  const int conv_c2 = cs.c0 * 2;                      // bytes per input pixel
  const int conv_rowB = cs.inW * conv_c2;             // bytes per input row
  const bool conv_ups = HAS_CONV && cs.ups;
  const int dc0 = ds.c0, dc1 = ds.c1;

  // ---- buffer descriptors (scalar).  The conv descriptor starts one row + one pixel BEFORE the feature map so that tap
  // (dy, dx) is a non-negative scalar offset from a per-lane base; nothing below the map is ever dereferenced.
  __amdgpu_buffer_rsrc_t rs_w = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_t*>(a.W), 0, (int)((size_t)a.N * a.ldw * 2), 0x00020000);
  __amdgpu_buffer_rsrc_t rs_c = rs_w, rs_d0 = rs_w, rs_d1 = rs_w;
  if (HAS_CONV) {
    const int shift = conv_rowB + conv_c2;
    const size_t bytes = (size_t)(a.M / a.rows_per_batch) * cs.inH * cs.inW * conv_c2;
    rs_c = __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<char*>(const_cast<bf16_t*>(cs.p0)) - shift, 0, (int)(bytes + shift), 0x00020000);
  }
  if (AMODE == 0 || AMODE == 2) {
    rs_d0 = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_t*>(ds.p0), 0, (int)((size_t)a.M * dc0 * 2), 0x00020000);
    if (dc1) rs_d1 = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_t*>(ds.p1), 0, (int)((size_t)a.M * dc1 * 2), 0x00020000);
  }
  // per-lane offsets, fixed for the whole work item.  Rows >= M are clamped to the last row (their outputs are not stored);
  // the row term lives in the VGPR offset, so every access stays inside its tensor whatever the range check covers.
  const unsigned voff_w = (unsigned)lrow * (unsigned)a.ldw * 2u + kc * 16;
  unsigned a_d0[A_IT], a_d1[A_IT];    // dense: row * pitch + chunk per source
  unsigned a_base[A_IT];              // conv: byte offset of the window's top-left tap (shifted origin) + chunk
  int a_yx[A_IT];                     // conv: parity bits of (ys, xs) | in-image mask of the 9 taps << 2
#pragma unroll
  for (int i = 0; i < A_IT; ++i) {
    int m = m0 + 32 * i + lrow;
    m = m < a.M ? m : a.M - 1;
    a_d0[i] = (unsigned)m * (unsigned)dc0 * 2u + kc * 16;
    a_d1[i] = (unsigned)m * (unsigned)dc1 * 2u + kc * 16;
    a_base[i] = 0; a_yx[i] = 0;
    if (HAS_CONV) {
      const int limH = conv_ups ? 2 * cs.inH : cs.inH, limW = conv_ups ? 2 * cs.inW : cs.inW;
      const int b = m / a.rows_per_batch;
      const int rem = m - b * a.rows_per_batch;
      const int oy = rem / a.outW, ox = rem - oy * a.outW;
      const int ys = oy * cs.stride + cs.asym, xs = ox * cs.stride + cs.asym;   // asym: the window starts AT (2oy, 2ox)
      int okm = 0;
#pragma unroll
      for (int t = 0; t < 9; ++t) {
        const int iy = ys - 1 + t / 3, ix = xs - 1 + t % 3;
        okm |= ((unsigned)iy < (unsigned)limH && (unsigned)ix < (unsigned)limW) ? (1 << t) : 0;
      }
      a_yx[i] = (ys & 1) | ((xs & 1) << 1) | (okm << 2);
      // top-left tap (ys-1, xs-1) in source coordinates; under the fused nearest-2x upsample the source row of upsampled row r
      // is r >> 1 (arithmetic), the parity-dependent +1 of the middle tap is added per load
      const int ty = conv_ups ? ((ys - 1) >> 1) : ys - 1, tx = conv_ups ? ((xs - 1) >> 1) : xs - 1;
      a_base[i] = (unsigned)((b * cs.inH * cs.inW + (ty + 1) * cs.inW + (tx + 1)) * conv_c2 + kc * 16);
    }
  }

  // LDS-DMAs of K slab lk into ring stage st: A rows 32 i + 8 wave .. + 7 by load i of this wave, W rows likewise
  auto issue = [&](int st, int lk) {
    unsigned char* sa = smem + st * STAGE_BYTES + wave * 1024;
    unsigned char* sb = sa + A_BYTES;
    if (HAS_CONV && (AMODE != 2 || lk < nkt_conv)) {
      // K order [64-channel slice][tap][64 channels] (gemm.hip): slice = lk / 9, tap = lk % 9
      const int sl = lk / 9, tap = lk - sl * 9;
      const int dy = tap / 3, dx = tap - dy * 3;
      unsigned soff = (unsigned)(sl * 128);
      if (!conv_ups) soff += (unsigned)(dy * conv_rowB + dx * conv_c2);
      else soff += (unsigned)((dy == 2 ? conv_rowB : 0) + (dx == 2 ? conv_c2 : 0));
#pragma unroll
      for (int i = 0; i < A_IT; ++i) {
        const bool ok = (a_yx[i] >> (2 + tap)) & 1;
        unsigned vo = a_base[i];
        if (conv_ups) {   // middle tap: +1 source row / pixel iff the upsampled coordinate ys-1 / xs-1 is odd, i.e. ys / xs even
          if (dy == 1) vo += (a_yx[i] & 1) ? 0u : (unsigned)conv_rowB;
          if (dx == 1) vo += (a_yx[i] & 2) ? 0u : (unsigned)conv_c2;
        }
        dma16(rs_c, sa + i * 4096, ok ? vo : OOB, soff);
      }
    } else {
      const int cc = (lk - nkt_conv) << 6;             // first K column of the slab inside the dense segment
      const bool first = cc < dc0;
      const unsigned col2 = (unsigned)((first ? cc : cc - dc0) * 2);
#pragma unroll
      for (int i = 0; i < A_IT; ++i) {
        if (first) dma16(rs_d0, sa + i * 4096, a_d0[i], col2);
        else dma16(rs_d1, sa + i * 4096, a_d1[i], col2);
      }
    }
    if (a.w_blocked) {
      // W stored as the LDS image itself: [N/32][K/64] blocks of 32 rows x 128 bytes (swizzle applied), 4 KB contiguous each --
      // one DMA instruction of the workgroup copies one block, a work item's K slice of a 32-row band is ONE contiguous range
      const unsigned so = ((unsigned)(n0 >> 5) * (unsigned)nkt + (unsigned)lk) * 4096u;
#pragma unroll
      for (int i = 0; i < B_IT; ++i) dma16(rs_w, sb + i * 4096, (unsigned)tid * 16u, so + (unsigned)i * (unsigned)nkt * 4096u);
    } else {
      const unsigned ldw2 = (unsigned)a.ldw * 2u;
      const unsigned so = (unsigned)n0 * ldw2 + (unsigned)lk * 128u;
#pragma unroll
      for (int i = 0; i < B_IT; ++i) dma16(rs_w, sb + i * 4096, voff_w, so + (unsigned)(32 * i) * ldw2);
    }
  };

  // ---- prologue: the epilogue's operands are requested FIRST (plain loads, nothing is computed from them before the
  // epilogue: they are older than every LDS-DMA, so the counted waits below stay exact and hipcc has no reason to drain the
  // ring for them), then the first NSTAGE-1 slabs.
  const int nb = n0 + wn * WTN + fq * 4;                 // this lane's first column inside column tile 0 of the wave
  const bool final_here = !SPLITK;                       // (SPLITK: the reducer fetches the epilogue operands itself)
  const bool tile_rv = !GEGLU && a.rowvec && a.rows_per_batch % BM == 0;   // the tile lies inside one batch element
  f32x4 cb[TN], rvv[GEGLU ? 1 : TN], c1v[LNF ? TN : 1];
#pragma unroll
  for (int j = 0; j < TN; ++j) cb[j] = f32x4{0.f, 0.f, 0.f, 0.f};
  constexpr int RN = GEGLU ? 1 : TN;
  u32x2 res_r[TM][RN];
  auto fetch_epilogue_operands = [&]() {
    if (a.bias) {
#pragma unroll
      for (int j = 0; j < TN; ++j) cb[j] = *reinterpret_cast<const f32x4*>(a.bias + nb + j * 16);
    }
    if constexpr (LNF) {
#pragma unroll
      for (int j = 0; j < TN; ++j) c1v[j] = *reinterpret_cast<const f32x4*>(a.ln_c1 + nb + j * 16);
    }
    if (tile_rv) {
      const float* rv = a.rowvec + (size_t)(m0 / a.rows_per_batch) * a.ld_rowvec + nb;
#pragma unroll
      for (int j = 0; j < RN; ++j) rvv[j] = *reinterpret_cast<const f32x4*>(rv + j * 16);
    }
    if (!GEGLU && a.res) {
#pragma unroll
      for (int i = 0; i < TM; ++i) {
        int mr = m0 + wm * WTM + i * 16 + fr;
        mr = mr < a.M ? mr : a.M - 1;
        const bf16_t* rp = a.res + (size_t)mr * a.ldres + nb;
#pragma unroll
        for (int j = 0; j < RN; ++j) res_r[i][j] = *reinterpret_cast<const u32x2*>(rp + j * 16);
      }
    }
  };
  if (final_here) fetch_epilogue_operands();
  const int pre = nk < nstage - 1 ? nk : nstage - 1;
  for (int p = 0; p < pre; ++p) issue(p, kt0 + p);

  f32x4 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  // ---- main loop: slab t lives in ring stage t % NSTAGE.  One barrier per slab: behind it every wave's DMAs of slab t have
  // landed (each waited for its own) and every wave has finished reading slab t-1, whose stage the next request re-fills.
  float row_s[TS], row_q[TS];
#pragma unroll
  for (int q = 0; q < TS; ++q) { row_s[q] = 0.f; row_q[q] = 0.f; }
  const int frag_off = fr * 128 + ((fq ^ ((fr >> 1) & 7)) << 4);
  int st = 0, st_fill = pre == nstage - 1 ? nstage - 1 : 0;
  for (int t = 0; t < nk; ++t) {
    const int left = nk - 1 - t;                                   // slabs behind this one
    wait_groups<L>(left < nstage - 2 ? left : nstage - 2);
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
    if (t + nstage - 1 < nk) {
      issue(st_fill, kt0 + t + nstage - 1);
      st_fill = st_fill + 1 == nstage ? 0 : st_fill + 1;
    }
    const unsigned char* sa0 = smem + st * STAGE_BYTES + wm * (WTM * 128);
    const unsigned char* sb0 = sa0 - wm * (WTM * 128) + A_BYTES + wn * (WTN * 128);
#pragma unroll
    for (int half = 0; half < 2; ++half) {
      // every fragment row of a wave is fr + a multiple of 16, so the swizzle term is that of fr; the halves differ in chunk bit 2
      const unsigned char* sa = sa0 + (frag_off ^ (half * 64));
      const unsigned char* sb = sb0 + (frag_off ^ (half * 64));
      bf16x8 af[TM], wf[TN];
#pragma unroll
      for (int j = 0; j < TN; ++j) wf[j] = *reinterpret_cast<const bf16x8*>(sb + j * 2048);
#pragma unroll
      for (int i = 0; i < TM; ++i) af[i] = *reinterpret_cast<const bf16x8*>(sa + i * 2048);
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[j], af[i], acc[i][j], 0, 0, 0);
      if constexpr (LNF) {
        // The two waves of a row block split the statistics by k HALF (wave wn takes the 32-deep half wn of every slab, all TM
        // row tiles): every register index stays a compile-time constant.  (Splitting by row tile -- af[2 q + wn] -- made hipcc
        // index the fragment array dynamically through scratch memory.)  This lane: 8 of the half's 32 k of row fr.
        if (half == wn) {
          typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2_t;
          const bf16x2_t ones = __builtin_bit_cast(bf16x2_t, 0x3f803f80u);
#pragma unroll
          for (int q = 0; q < TM; ++q) {
            const u32x4 x4 = __builtin_bit_cast(u32x4, af[q]);
            const unsigned dw[4] = {x4.x, x4.y, x4.z, x4.w};
#pragma unroll
            for (int e = 0; e < 4; ++e) {
              const bf16x2_t x = __builtin_bit_cast(bf16x2_t, dw[e]);
              row_q[q] = __builtin_amdgcn_fdot2_f32_bf16(x, x, row_q[q], false);
              row_s[q] = __builtin_amdgcn_fdot2_f32_bf16(x, ones, row_s[q], false);
            }
          }
        }
      }
    }
    st = st + 1 == nstage ? 0 : st + 1;
  }

  // ---- split-K: publish the partial tile, take a ticket; the last slice to arrive combines all of them in slice order
  if constexpr (SPLITK) {
    const size_t slab = (size_t)a.M * a.N;                              // floats per slice
    __amdgpu_buffer_rsrc_t rs_p = __builtin_amdgcn_make_buffer_rsrc(a.part, 0, (int)((size_t)S * slab * 4), 0x00020000);
    unsigned pvo[TM];
#pragma unroll
    for (int i = 0; i < TM; ++i) {
      const int m = m0 + wm * WTM + i * 16 + fr;
      pvo[i] = m < a.M ? (unsigned)(((size_t)m * a.N + nb) * 4) : OOB;  // rows >= M: the store is dropped, the load reads zeros
    }
    const unsigned pso = (unsigned)((size_t)ks * slab * 4);
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int j = 0; j < TN; ++j) {
        const u32x4 v = __builtin_bit_cast(u32x4, acc[i][j]);
        __builtin_amdgcn_raw_buffer_store_b128(v, rs_p, (int)pvo[i] + j * 64, (int)pso, 16);   // aux 16 = sc1: write-through
        asm volatile("s_nop 1" ::"v"(v));
      }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                   // every storing wave drains its own stores
    __builtin_amdgcn_s_barrier();
    // Rendezvous of the tile's S slices, then EVERY slice combines a share of the tile: the 16x16 sub-tiles are dealt
    // round-robin to the S x 4 waves, a wave sums its sub-tile over the slices in slice order (S loads in flight per lane, one
    // memory round trip) and runs the epilogue on it.  (A last-arriver-only combine reads S x tile bytes through ONE workgroup
    // after everybody else has left: 21.5 vs 18.4 us on the 8x8-level convolutions.)
    //
    // The rendezvous must not depend on WHO ELSE is on the chip (another process's or another stream's kernels may hold the
    // CUs the missing slices need -- ADVICE r3): the tile's word carries the arrival count in bits 0-7 and a CLAIM bit per
    // share in bits 8..8+S.  A slice waits a BOUNDED time for the others; if they all arrive it claims its own share
    // (fetch_or) and combines it, otherwise it simply leaves.  The slice that draws the LAST ticket never waits: it combines
    // its own share and then claims, with ONE fetch_or, every share nobody has claimed by then (owners that gave up, or that
    // are still on their way from the poll to the claim -- their own fetch_or then loses).  Every share is combined exactly
    // once by a live workgroup under any residency, nothing is poisoned, and each sub-tile's sum is in slice order whoever
    // computes it: the result stays bit-deterministic.
    unsigned* flag = reinterpret_cast<unsigned*>(smem);                 // (the one LDS array; the ring is dead by now)
    typedef __attribute__((address_space(1))) unsigned int gu32;
    gu32* const c = (gu32*)(a.tile_cnt + tl);
    const unsigned all_shares = (1u << S) - 1u;
    if (tid == 0) {
      const unsigned ticket = __hip_atomic_fetch_add(c, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) & 0xffu;
      const bool last = ticket == (unsigned)(S - 1);
      unsigned mine = 0;
      if (a.splitk_nowait) {
        // no-wait form (a launch that shares the chip with another stream's kernels by design): the last ticket combines the
        // whole tile, the others leave at once
        mine = last ? all_shares : 0u;
      } else {
        bool all = last;
        for (int spins = 0; !all && spins < 4096; ++spins) {
          __builtin_amdgcn_s_sleep(8);
          all = (__hip_atomic_load(c, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) & 0xffu) >= (unsigned)S;
        }
        if (all) {
          const unsigned bit = 1u << (8 + ks);
          if (!(__hip_atomic_fetch_or(c, bit, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) & bit)) mine = 1u << ks;
        }
      }
      flag[0] = mine; flag[1] = (last && !a.splitk_nowait) ? 1u : 0u;
    }
    __syncthreads();
    unsigned mine = flag[0];
    const bool rescuer = flag[1] != 0u;
    constexpr int UN = BN / 16, U = (BM / 16) * UN;                     // 16x16 sub-tiles of the tile
    const float alpha_s = a.alpha;
    for (int pass = 0; pass < 2; ++pass) {
    if (pass == 1) {                                                    // the last arriver: whatever nobody has claimed
      if (!rescuer) break;
      __syncthreads();
      if (tid == 0) flag[0] = (~__hip_atomic_fetch_or(c, all_shares << 8, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >> 8) & all_shares;
      __syncthreads();
      mine = flag[0];
    }
    if (!mine) continue;
    for (int u = wave; u < U; u += 4) {
      if (!((mine >> ((u >> 2) % S)) & 1u)) continue;                   // sub-tile u belongs to share (u / 4) mod S
      const int ti = u / UN, tj = u - ti * UN;
      const int m = m0 + ti * 16 + fr, n = n0 + tj * 16 + fq * 4;
      const bool live = m < a.M;
      const unsigned vo = live ? (unsigned)(((size_t)m * a.N + n) * 4) : OOB;
      // epilogue operands of this sub-tile first (plain loads of tensors nobody writes in this launch)
      f32x4 add = f32x4{0.f, 0.f, 0.f, 0.f};
      u32x2 rr = {0u, 0u};
      const int mc = live ? m : a.M - 1;
      if (a.bias) add = *reinterpret_cast<const f32x4*>(a.bias + n);
      f32x4 rvv2 = f32x4{0.f, 0.f, 0.f, 0.f};
      if (a.rowvec) rvv2 = *reinterpret_cast<const f32x4*>(a.rowvec + (size_t)(mc / a.rows_per_batch) * a.ld_rowvec + n);
      if (a.res) rr = *reinterpret_cast<const u32x2*>(a.res + (size_t)mc * a.ldres + n);
      f32x4 sum = f32x4{0.f, 0.f, 0.f, 0.f};
      for (int s0 = 0; s0 < S; s0 += 8) {
        u32x4 p[8];
#pragma unroll
        for (int q = 0; q < 8; ++q) {
          const int sq = s0 + q < S ? s0 + q : S - 1;                  // (beyond S: re-read the last slice, not added)
          p[q] = __builtin_amdgcn_raw_buffer_load_b128(rs_p, (int)vo, (int)((size_t)sq * slab * 4), 16);   // sc1: not from this CU's L1
        }
#pragma unroll
        for (int q = 0; q < 8; ++q)
          if (s0 + q < S) sum += __builtin_bit_cast(f32x4, p[q]);
      }
      f32x4 v = (sum + add + rvv2) * alpha_s;
      if (a.res) { v[0] += bflo(rr[0]); v[1] += bfhi(rr[0]); v[2] += bflo(rr[1]); v[3] += bfhi(rr[1]); }
      if (!live) continue;
      if (a.out_f32) {
        *reinterpret_cast<f32x4*>(reinterpret_cast<float*>(a.out) + (size_t)m * a.ldo + n) = v;
      } else {
        const u32x2 o = {pack2bf(v[0], v[1]), pack2bf(v[2], v[3])};
        *reinterpret_cast<u32x2*>(reinterpret_cast<bf16_t*>(a.out) + (size_t)m * a.ldo + n) = o;
      }
    }
    }
    return;
  }

  // ---- LNF: finish the row statistics.  The four lanes fr, fr + 16, fr + 32, fr + 48 hold the k quarters of a row's half; the
  // two waves of a row block trade their halves through the exchange area behind the ring ([k half][row of the tile] ->
  // (sum, sum of squares)) and both finish (rstd, -rstd * mean) in the same order: half 0 + half 1.
  float2 lnst[LNF ? TM : 1];
  if constexpr (LNF) {
    float2* xch = reinterpret_cast<float2*>(smem + nstage * STAGE_BYTES);
#pragma unroll
    for (int q = 0; q < TM; ++q) {
      float sm = row_s[q], sq = row_q[q];
      sm += __shfl_xor(sm, 16); sq += __shfl_xor(sq, 16);
      sm += __shfl_xor(sm, 32); sq += __shfl_xor(sq, 32);
      if (fq == 0) xch[wn * BM + wm * WTM + q * 16 + fr] = float2{sm, sq};
    }
    __syncthreads();
    const float invk = 1.f / (float)a.Ktot;
#pragma unroll
    for (int i = 0; i < TM; ++i) {
      const float2 p0 = xch[wm * WTM + i * 16 + fr], p1 = xch[BM + wm * WTM + i * 16 + fr];
      const float mean = (p0.x + p1.x) * invk;
      const float var = fmaxf((p0.y + p1.y) * invk - mean * mean, 0.f);
      const float sc = rsqrtf(var + a.ln_eps);
      lnst[i] = float2{sc, -sc * mean};
    }
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int j = 0; j < TN; ++j) acc[i][j] = acc[i][j] * lnst[i].x + c1v[j] * lnst[i].y;     // (+ c2 = cb below)
  }

  // ---- epilogue
  const float alpha = a.alpha;
  const bool row_rv = !GEGLU && a.rowvec && a.rows_per_batch % BM != 0;   // a tile straddling batch elements: per-row vectors
#pragma unroll
  for (int i = 0; i < TM; ++i) {
    const int m = m0 + wm * WTM + i * 16 + fr;
    const bool live = m < a.M;
    if constexpr (GEGLU) {
#pragma unroll
      for (int j = 0; j < TN; j += 2) {
        const f32x4 v = acc[i][j] + cb[j], g = acc[i][j + 1] + cb[j + 1];   // packed rows: 16 value | 16 gate
        const int no = (n0 + wn * WTN) / 2 + (j / 2) * 16 + fq * 4;
        const u32x2 o = {pack2bf(v[0] * gelu_erf_f(g[0]), v[1] * gelu_erf_f(g[1])), pack2bf(v[2] * gelu_erf_f(g[2]), v[3] * gelu_erf_f(g[3]))};
        if (live) *reinterpret_cast<u32x2*>(reinterpret_cast<bf16_t*>(a.out) + (size_t)m * a.ldo + no) = o;
      }
    } else {
      const float* rvp = row_rv ? a.rowvec + (size_t)((live ? m : a.M - 1) / a.rows_per_batch) * a.ld_rowvec + nb : nullptr;
#pragma unroll
      for (int j = 0; j < TN; ++j) {
        f32x4 v = acc[i][j] + cb[j];
        if (tile_rv) v += rvv[j];
        if (row_rv) v += *reinterpret_cast<const f32x4*>(rvp + j * 16);
        v *= alpha;
        if (a.res) { v[0] += bflo(res_r[i][j][0]); v[1] += bfhi(res_r[i][j][0]); v[2] += bflo(res_r[i][j][1]); v[3] += bfhi(res_r[i][j][1]); }
        if (!live) continue;
        const int n = nb + j * 16;
        if (a.out_f32) {
          *reinterpret_cast<f32x4*>(reinterpret_cast<float*>(a.out) + (size_t)m * a.ldo + n) = v;
        } else {
          const u32x2 o = {pack2bf(v[0], v[1]), pack2bf(v[2], v[3])};
          *reinterpret_cast<u32x2*>(reinterpret_cast<bf16_t*>(a.out) + (size_t)m * a.ldo + n) = o;
        }
      }
    }
  }
}

struct SmTile { int bm, bn; };
const SmTile kSmTiles[] = {{64, 64}, {128, 64}, {64, 128}, {128, 128}, {64, 160}, {128, 160}, {64, 320}};
constexpr int kNumSmTiles = 7;

template <int BM, int BN, int AMODE, bool SPLITK, bool GEGLU, bool LNF = false>
int launch_sm3(const MvdGemmArgs& a, int nstage, hipStream_t s) {
  constexpr int STAGE_BYTES = (BM + BN) * 128;
  if (LNF && nstage * STAGE_BYTES + BM * 16 > 160 * 1024) --nstage;
  const int lds = nstage * STAGE_BYTES + (LNF ? BM * 16 : 0);
  static bool lds_set[16] = {};                         // per device (one process may drive several)
  int dev = 0;
  (void)hipGetDevice(&dev);
  if (!lds_set[dev & 15]) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_sm_kernel<BM, BN, AMODE, SPLITK, GEGLU, LNF>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    if (e != hipSuccess) { mvd_set_error("gemm_sm: hipFuncSetAttribute: %s", hipGetErrorString(e)); return -2; }
    lds_set[dev & 15] = true;
  }
  const int ntm = (a.M + BM - 1) / BM, ntn = a.N / BN;
  const int S = a.splitk > 1 ? a.splitk : 1;
  const int grid = ntm * ntn * S;
  // (no residency CONDITION on a split-K grid: the in-kernel rendezvous is bounded and the last arriver combines whatever the
  //  others left, so slices that are not co-resident -- a grid beyond the chip, another tenant on the GPU -- cost time, not
  //  correctness.  But the time is ~1 ms of polling per waiting slice (4096 x s_sleep 8) before the missing slices can start, so
  //  the occupancy query stays as a HINT: a grid that cannot be resident at once -- a forced split through mvd_op_* -- takes the
  //  no-wait combine, where the last arriver sums the whole tile and nobody polls.  ADVICE r4.)
  MvdGemmArgs b = a;
  if constexpr (SPLITK) {
    static int resident[16][9] = {};                    // per device and ring depth: workgroups the chip holds at once
    int& cap = resident[dev & 15][nstage & 7];
    if (!cap) {
      int occ = 0, ncu = 0;
      if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, gemm_sm_kernel<BM, BN, AMODE, SPLITK, GEGLU, LNF>, 256, lds) != hipSuccess || occ < 1) occ = 1;
      if (hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || ncu < 1) ncu = 256;
      cap = occ * ncu;
    }
    if (grid > cap) b.splitk_nowait = 1;
  }
  g_mvd_last_gemm.tiles = grid; g_mvd_last_gemm.grid = grid; g_mvd_last_gemm.per_cu = 160 * 1024 / lds;
  g_mvd_last_gemm.nowait = b.splitk_nowait;
  hipLaunchKernelGGL((gemm_sm_kernel<BM, BN, AMODE, SPLITK, GEGLU, LNF>), dim3(grid), dim3(256), lds, s, b, nstage);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) { mvd_set_error("gemm_sm launch: %s", hipGetErrorString(e)); return -3; }
  return 0;
}

template <int BM, int BN>
int launch_sm2(const MvdGemmArgs& a, int nstage, hipStream_t s) {
  const bool sk = a.splitk > 1;
  if (a.geglu) {
    if constexpr ((BN / 32) % 2 == 0) return a.ln_c1 ? launch_sm3<BM, BN, 0, false, true, true>(a, nstage, s) : launch_sm3<BM, BN, 0, false, true>(a, nstage, s);
    else { mvd_set_error("gemm_sm: GEGLU needs an even number of 16-column tiles per wave"); return -1; }
  }
  if (a.ln_c1) return launch_sm3<BM, BN, 0, false, false, true>(a, nstage, s);
  if (a.seg[0].mode == MVD_A_DENSE) return sk ? launch_sm3<BM, BN, 0, true, false>(a, nstage, s) : launch_sm3<BM, BN, 0, false, false>(a, nstage, s);
  if (a.nseg == 1) return sk ? launch_sm3<BM, BN, 1, true, false>(a, nstage, s) : launch_sm3<BM, BN, 1, false, false>(a, nstage, s);
  return sk ? launch_sm3<BM, BN, 2, true, false>(a, nstage, s) : launch_sm3<BM, BN, 2, false, false>(a, nstage, s);
}

}  // namespace

// Every byte offset the kernel forms must fit 32-bit buffer addressing (and stay below OOB = 2^31).
bool mvd_gemm_sm_applicable(const MvdGemmArgs& a, int tile) {
  if (tile < 0 || tile >= kNumSmTiles) return false;
  const size_t lim = (size_t)1 << 31;
  if (a.N % kSmTiles[tile].bn || a.Ktot % 64) return false;
  if (a.ln_c1) {   // LayerNorm fold: one dense source spanning the whole row, a bias (c2), no residual / row vector / split-K
    const MvdASeg& g = a.seg[0];
    if (a.nseg != 1 || g.mode != MVD_A_DENSE || g.c1 || a.splitk > 1 || a.res || a.rowvec || a.out_f32 || !a.bias || a.Ktot != g.c0) return false;
  }
  if (a.w_blocked && a.ldw != a.Ktot) return false;
  if (a.geglu && ((kSmTiles[tile].bn / 32) % 2 || a.splitk > 1 || a.seg[0].mode != MVD_A_DENSE)) return false;
  if ((size_t)a.N * a.ldw * 2 >= lim) return false;
  if (a.splitk > 1 && ((size_t)a.splitk * a.M * a.N * 4 >= lim || !a.tile_cnt || !a.part || a.splitk > 24)) return false;   // (24 claim bits beside the arrival count)
  for (int i = 0; i < a.nseg; ++i) {
    const MvdASeg& g = a.seg[i];
    if (g.mode == MVD_A_DENSE) {
      if ((size_t)a.M * (g.c0 > g.c1 ? g.c0 : g.c1) * 2 >= lim) return false;
    } else {
      const size_t bytes = (size_t)(a.M / a.rows_per_batch) * g.inH * g.inW * g.c0 * 2 + (size_t)(g.inW + 1) * g.c0 * 2;
      if (bytes + (size_t)3 * g.inW * g.c0 * 2 >= lim) return false;
      if (2 * g.inH >= 32768 || 2 * g.inW >= 32768) return false;
    }
  }
  return true;
}

extern "C" int mvd_gemm_sm_num_tiles(void) { return kNumSmTiles; }

// Which problems the small-M kernels take, and how (tile, ring depth, split-K): rules read off the batch-1 sweep
// tools/tune_sm.py (profiles/r03_tune_sm_b1.log: every GEMM / conv shape of a batch-1 forward, cold weights, warm
// activations, against the launch the M = 32-images heuristic of gemm.hip makes).  One 256-thread workgroup per work item:
// * no split while the tile grid reaches ~200 workgroups or K is short (a slice needs >= 12 slabs to pay for the combine);
//   else tiles x split ~ 256 workgroups (one per CU);
// * 64x64 tiles by default (most work items); taller / wider tiles where 64x64 would give > ~600 work items or the weight
//   operand is the traffic (convolutions at 32x32 and 16x16: 64x160 / 64x128, the activation slab is re-read per column tile).
bool mvd_gemm_sm_plan(const MvdGemmArgs& a, int* tile, int* nstage, int* splitk) {
  const int M = a.M, N = a.N, nkt = a.Ktot / 64;
  if (M > 4608 || a.Ktot % 64 || N % 64) return false;
  // beyond one image's 32x32 map only SMALL problems: with work for every CU the persistent 128x160 / 256x320 kernels move
  // half the operand bytes per FLOP (the 8x8 level of a 32-image batch, M = 2048: 6.2 ms through these kernels against
  // 2.9 ms -- cfg4 kernel classes, round 3)
  if (M > 1024 && 2.0 * M * (double)N * a.Ktot > 12e9) return false;
  const bool conv = a.seg[0].mode == MVD_A_CONV3;
  if (conv && M >= 4096) return false;          // 64x64-level convolutions: the 128x160 / 256x320 kernels are as fast or faster
  int t = 0;
  if (a.geglu) {
    t = M >= 4096 ? 2 : (M >= 1024 ? (N % 320 == 0 ? 6 : 2) : (M >= 256 ? 3 : 0));
    if ((t == 2 || t == 3) && N % 128) t = 0;
  } else if (conv) {
    if (M >= 1024) t = (N % 160 == 0 && N >= 640 && nkt >= 80) ? 4 : 0;
    else if (M >= 256) t = N % 128 == 0 ? 2 : 0;
  } else {
    const long tiles0 = (long)((M + 63) / 64) * (N / 64);
    if (M > 64 && M <= 128 && N % 128 == 0) t = 3;                 // (text K/V: every W row read once)
    else if (tiles0 > 640 && M >= 2048) t = 1;
    else if (tiles0 > 320 && M <= 1024 && M > 128 && N % 128 == 0) t = 2;
  }
  const long tiles = (long)((M + kSmTiles[t].bm - 1) / kSmTiles[t].bm) * (N / kSmTiles[t].bn);
  int S = 1;
  if (!a.geglu && !a.ln_c1 && tiles < 200) {
    S = (int)(256 / tiles);
    if (S > nkt / 12) S = nkt / 12;
    S = S < 1 ? 1 : (S > 16 ? 16 : S);
  }
  *tile = t; *splitk = S;
  *nstage = a.geglu ? (t == 0 ? 4 : (M >= 4096 ? 2 : 3)) : ((S == 1 && nkt <= 12) ? 3 : 4);
  MvdGemmArgs b = a; b.splitk = 1; b.w_blocked = 0;
  return mvd_gemm_sm_applicable(b, t);
}

// tile: index into {64x64, 128x64, 64x128, 128x128, 64x160, 128x160, 64x320}; nstage: ring depth 2..8, clamped to what 160 KB
// of LDS hold (arguments already validated by mvd_launch_gemm)
int mvd_launch_gemm_sm(const MvdGemmArgs& a, hipStream_t s, int tile, int nstage) {
  if (!mvd_gemm_sm_applicable(a, tile)) { mvd_set_error("gemm_sm: tile %d does not take M=%d N=%d K=%d geglu=%d splitk=%d", tile, a.M, a.N, a.Ktot, a.geglu, a.splitk); return -1; }
  const int stage_bytes = (kSmTiles[tile].bm + kSmTiles[tile].bn) * 128;
  int maxst = 160 * 1024 / stage_bytes;
  if (maxst > SM_MAX_STAGES) maxst = SM_MAX_STAGES;
  if (nstage > maxst) nstage = maxst;
  if (nstage < 2) nstage = 2;
  switch (tile) {
    case 0: return launch_sm2<64, 64>(a, nstage, s);
    case 1: return launch_sm2<128, 64>(a, nstage, s);
    case 2: return launch_sm2<64, 128>(a, nstage, s);
    case 3: return launch_sm2<128, 128>(a, nstage, s);
    case 4: return launch_sm2<64, 160>(a, nstage, s);
    case 5: return launch_sm2<128, 160>(a, nstage, s);
    default: return launch_sm2<64, 320>(a, nstage, s);
  }
}
