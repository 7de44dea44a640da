// Weight-streaming 3x3 convolution for gfx950: the resnet convolutions of the 8x8, 16x16 and 32x32 levels of ONE image (batch 1:
// M = 64 / 256 / 1024 output pixels, N = 1280, K = 9 * 1280 .. 9 * 2560 + 2560 -- 15-65 MB of weights for 2-8 GFLOP).  Such a launch
// is a pure WEIGHT STREAM: the implicit-GEMM kernels (gemm_sm.hip) cut it along K to find 240+ workgroups, re-gather the
// activation map nine times per column tile through the LDS-DMA and pay a cross-workgroup split-K rendezvous -- 21-35 us for
// a stream that takes 7 us at the HBM rate (profiles/r03_tune_sm_b1.log; warm weights change it by 13 %: it is not the memory).
//
//   out[m][n] = sum_{tap, c} X[pixel(m) + tap][c] W[n][c][tap]  (+ sum_c SC[m][c] Wsc[n][c])  + bias[n] + rowvec[img][n] + res[m][n]
//
// One workgroup = 64 (or, on a 16-wide map, 128) consecutive output pixels (whole map rows of one image) x 16 output channels
// over the WHOLE K:
// * no split-K between workgroups: the four waves of a workgroup split K among themselves -- wave w owns channels
//   [32 w, 32 w + 32) of every 128-channel ROUND -- and are fully independent until one reduction through LDS at the end
//   (fixed order: bit-deterministic).  No barrier, no atomic, no workspace in the main loop.
// * THE MAP SLICE IS LOADED ONCE PER ROUND, NOT ONCE PER TAP: a wave DMAs its 32-channel quarter of the input rows the 64
//   output pixels touch (64 / W + 2 map rows, rows outside the image read as zeros) into a private LDS slab; the nine taps
//   are nine shifted ds_read_b128 patterns of that slab (a tap beyond the map edge points at a zero row).  16-byte chunks
//   are XOR-swizzled by the pixel index on the GLOBAL side of the DMA so that the operand reads are conflict free.
// * THE WEIGHTS ARE HOST-PACKED IN FRAGMENT ORDER (packing.pack_ws: [column tile][round][wave][tap][lane][8 bf16]): a wave's
//   k-step is ONE 1 KB LDS-DMA of contiguous global memory into its private 18-slot ring, the fragment read is
//   ds_read_b128 at slot + 16 * lane.  A wave keeps up to 2 rounds of weights + a slab in flight (~25 KB, ~100 KB per CU):
//   enough outstanding bytes for one CU to pull its 0.4-1.2 MB share at the rate its LDS-DMA issue allows.  Waits are counted
//   by hand (s_waitcnt vmcnt(N): loads and LDS-DMAs retire in order), one per round.
// * MFMA v_mfma_f32_16x16x32_bf16 with the WEIGHT fragment as the A operand (row = output channel) and the pixel fragment
//   as B (column = pixel): a lane ends with four consecutive channels of one pixel -- one 8-byte store.
// The dense shortcut of a channel-changing resnet (conv2 | conv_shortcut concatenated along K) is a second pass of the same
// loop with one "tap" and no shift.
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include "../../include/mvd_hip.h"
#include <type_traits>
#include "kernels.h"

namespace {

typedef __attribute__((address_space(3))) void lds_void;

MVD_DEVINL void ws_dma16(__amdgpu_buffer_rsrc_t rsrc, unsigned char* lds_wave_base, unsigned voff, unsigned soff) {
  __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (lds_void*)lds_wave_base, 16, (int)voff, (int)soff, 0, 0);
}
#ifdef WS_LAG_TEST     // diagnosis builds: idle cycles between the counted wait and the first read of what it waited for
#define WS_WAIT_VM(n) asm volatile("s_waitcnt vmcnt(%0)\n\ts_sleep 16" ::"n"(n) : "memory")
#else
#define WS_WAIT_VM(n) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(n) : "memory")
#endif

constexpr int WS_RING = 18;                    // 1 KB weight slots per wave: two rounds of nine taps

// WD: OUTPUT map width; UPS: nearest-neighbour 2x upsampling in front of the convolution (the input map is WD / 2 wide)
template <int WD, int RB, bool UPS> struct WsGeom {
  static constexpr int NBUF = 2;               // slab buffers per wave
  static constexpr int BM = 16 * RB;           // output pixels per workgroup (RB 16-pixel blocks)
  static constexpr int RO = BM / WD;           // output map rows per workgroup
  static constexpr int WI = UPS ? WD / 2 : WD; // input map width
  static constexpr int SR = UPS ? RO / 2 + 2 : RO + 2;   // input rows a block touches: its own + one halo row above and below
  static constexpr int SPX = SR * WI;          // slab pixels
  static constexpr int NSL = (SPX + 15) / 16;  // 1 KB DMA pieces per slab (16 pixels x 64 bytes); on a 12-wide map the last piece
                                               // runs 8 pixels past the slab's rows (real, clamped loads that nothing reads)
  static constexpr int SPXP = NSL * 16;        // slab pixels incl. that padding: the zero row sits behind them
  static constexpr int SLAB = SPXP * 64 + 64;  // + the zero row
  static constexpr int WAVE_BYTES = WS_RING * 1024 + NBUF * SLAB;
  static_assert(BM % WD == 0 && 4 * WAVE_BYTES <= 160 * 1024 && (!UPS || RO % 2 == 0), "geometry");
};

// One pass over `R` rounds of 128 input channels with T taps each.  svo[i]: the lane's global byte offset of slab piece i
// for round 0 (always a real address); `xstep`: bytes a round advances in the source; aoff[t][rb]: the lane's LDS byte offset
// (inside a slab) of its B-operand fragment for tap t and 16-pixel block rb; `wso`: scalar byte offset of this wave's weight
// stream (round rd, tap t at wso + (rd * 4 * T + t) * 1024).
// Two slab buffers, weights two rounds ahead.  Issue order W(rd, *) ... slab(rd) ... W(rd + 1, *): ONE counted wait per round --
// slab(rd) has only W(rd + 1, *) behind it, and with it every W(rd, *) has landed.  The last two rounds, which have less behind
// them, are separate instantiations (LA) with their own constants: no dummy out-of-range pieces to keep a count uniform (a
// first version had them and gave one intermittent wrong result -- whether an all-out-of-range LDS-DMA retires in order is not
// something to rest a wait on).  (A three-buffer form with the slab two rounds ahead as well measured the same: round 4.)
template <int T, int NSL, int SLAB, int RB>
MVD_DEVINL void ws_pass(const __amdgpu_buffer_rsrc_t rs_x, const unsigned (&svo)[NSL], unsigned xstep, int R,
                         const __amdgpu_buffer_rsrc_t rs_w, unsigned wso, const int (&aoff)[T][RB], f32x4 (&acc)[RB],
                         unsigned char* wbase, int lane) {
  unsigned char* const ring = wbase;
  unsigned char* const slabs = wbase + WS_RING * 1024;
  auto issue_slab = [&](int rd) {
    unsigned char* dst = slabs + (rd & 1) * SLAB;
    const unsigned so = (unsigned)rd * xstep;
#pragma unroll
    for (int i = 0; i < NSL; ++i) ws_dma16(rs_x, dst + i * 1024, svo[i], so);
  };
  auto issue_w = [&](int rd, int t, int slot) {
    ws_dma16(rs_w, ring + slot * 1024, (unsigned)lane * 16u, wso + (unsigned)((rd * 4 * T + t) * 1024));
  };
  auto round = [&](int rd, auto la_tag) {
    constexpr int LA = decltype(la_tag)::value;            // how many of the rounds rd + 1, rd + 2 exist
#ifdef WS_SC_DRAIN      // diagnosis builds: the one-tap passes drain every round
    if constexpr (T == 1) WS_WAIT_VM(0); else WS_WAIT_VM(LA >= 1 ? T : 0);
#else
    WS_WAIT_VM(LA >= 1 ? T : 0);
#endif
    if constexpr (LA >= 1) issue_slab(rd + 1);             // buffer (rd + 1) & 1: last read in round rd - 1, reads completed
    const unsigned char* sl = slabs + (rd & 1) * SLAB;
    const int half = (rd & 1) * T;
    bf16x8 wf, xf[RB];
    auto read_frags = [&](int t, bf16x8& w_, bf16x8 (&x_)[RB]) {
      w_ = *reinterpret_cast<const bf16x8*>(ring + (half + t) * 1024 + lane * 16);
#pragma unroll
      for (int rb = 0; rb < RB; ++rb) x_[rb] = *reinterpret_cast<const bf16x8*>(sl + aoff[t][rb]);
    };
    read_frags(0, wf, xf);
#pragma unroll
    for (int t = 0; t < T; ++t) {
      bf16x8 wn, xn[RB];
      if (t + 1 < T) read_frags(t + 1, wn, xn);
#pragma unroll
      for (int rb = 0; rb < RB; ++rb) acc[rb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf, xf[rb], acc[rb], 0, 0, 0);
      // Refill slot (half + t) for round rd + 2 -- but only once its fragment `wf` has RETURNED from LDS.  The asm "uses" wf (the
      // compiler puts the lgkmcnt wait for that ds_read in front of it) and its memory clobber keeps the LDS-DMA behind it.
      // Without it hipcc hoists the DMA to right behind the ds_read's ISSUE (no dependence it can see); a DMA whose 1 KB comes out of
      // this CU's L1 -- a second row block of the same column tile -- can then land before a read that co-resident waves of
      // another stream's kernel have delayed in the LDS queue: one k-chunk of the tile multiplies the wrong weights.  Seen only
      // in the two-stream forward, in the one-tap (fused shortcut) passes, where read and refill sit in the same step
      // (profiles/r04_probe_conv_ws_in_situ.log).
      // The comment travels into the compiler's assembly output: tools/lint_device_isa.py (--conv-ws-fences, run by
      // __graft_entry__.build() and a CPU test) finds the ds_read that filled the named registers and fails unless an
      // `s_waitcnt lgkmcnt` that retires it stands between that read and this fence.
      asm volatile("; MVD_REFILL_FENCE %0" ::"v"(__builtin_bit_cast(u32x4, wf)) : "memory");
      if constexpr (LA == 2) issue_w(rd + 2, t, half + t);
      if (t + 1 < T) {
        wf = wn;
#pragma unroll
        for (int rb = 0; rb < RB; ++rb) xf[rb] = xn[rb];
      }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  };
#pragma unroll
  for (int t = 0; t < T; ++t) issue_w(0, t, t);
  issue_slab(0);
  if (R > 1) {
#pragma unroll
    for (int t = 0; t < T; ++t) issue_w(1, t, T + t);
  }
  int rd = 0;
  for (; rd + 2 < R; ++rd) round(rd, std::integral_constant<int, 2>{});
  if (rd + 1 < R) { round(rd, std::integral_constant<int, 1>{}); ++rd; }
  round(rd, std::integral_constant<int, 0>{});
}

template <int WD, int RB, bool UPS>
__global__ __launch_bounds__(256) void conv_ws_kernel(const MvdWsArgs a) {
  using G = WsGeom<WD, RB, UPS>;
  constexpr int NSL = G::NSL, SLAB = G::SLAB, BM = G::BM, NBUF = G::NBUF;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int li = lane & 15, lh = lane >> 4;

  // ---- work item: column tile ct (16 channels), row block blk (BM pixels).  The row blocks of one column tile get block
  // ids that are equal mod 8 (one XCD): the weight panel they share is fetched once into that L2.
  const int nb = a.M / BM, nct = a.N / 16;
  const int g = blockIdx.x / (8 * nb), rem = blockIdx.x - g * 8 * nb;
  const int blk = rem >> 3, ct = g * 8 + (rem & 7);
  if (ct >= nct) return;
  const int OH = UPS ? 2 * a.H : a.H;                       // output map rows (a.H, a.W: the INPUT map)
  const int hw = OH * WD, m0 = blk * BM;
  const int img = m0 / hw, oy0 = (m0 - img * hw) / WD;
  const int iy_lo = UPS ? (oy0 - 1) >> 1 : oy0 - 1;         // input row of slab row 0 (may be -1: loaded from row 0, never read)

  unsigned char* const wbase = smem + wave * G::WAVE_BYTES;
  // zero rows of the slabs (a lane writes 4 bytes: 16 lanes per row)
  if (lane < 16 * NBUF) *reinterpret_cast<unsigned*>(wbase + WS_RING * 1024 + (lane >> 4) * SLAB + G::SPXP * 64 + (lane & 15) * 4) = 0u;

  const int Rc = a.C / 128, Rs0 = a.scc0 / 128, Rs1 = a.scc1 / 128;
  const unsigned tile_bytes = (unsigned)((Rc * 9 + Rs0 + Rs1) * 4) * 1024u;            // packed bytes of one column tile
  const __amdgpu_buffer_rsrc_t rs_w = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_t*>(a.w), 0, (int)((size_t)nct * tile_bytes), 0x00020000);
  f32x4 acc[RB];
#pragma unroll
  for (int rb = 0; rb < RB; ++rb) acc[rb] = f32x4{0.f, 0.f, 0.f, 0.f};

  // ---- the convolution: rounds of 128 channels x 9 taps
  {
    const __amdgpu_buffer_rsrc_t rs_x = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_t*>(a.x), 0, (int)((size_t)a.B * a.H * G::WI * a.C * 2), 0x00020000);
    unsigned svo[NSL];
#pragma unroll
    for (int i = 0; i < NSL; ++i) {
      const int q = 16 * i + (lane >> 2), cp = lane & 3;               // slab pixel, chunk POSITION in its 64-byte row
      const int c = cp ^ ((q >> 2) & 3);                                // chunk held there
      // a halo row outside the image is loaded from the nearest map row and never read (aoff points those taps at the zero
      // row): every lane of every piece is a real load.  An all-out-of-range LDS-DMA piece may retire ahead of its turn, and
      // the counted waits rest on in-order retirement (seen as run-to-run differences beside a second stream's kernels).
      int iy = iy_lo + q / G::WI;
      const int ix = q % G::WI;
      iy = iy < 0 ? 0 : (iy >= a.H ? a.H - 1 : iy);
      svo[i] = (unsigned)((((size_t)img * a.H + iy) * G::WI + ix) * a.C + 32 * wave + 8 * c) * 2u;
    }
    int aoff[9][RB];
#pragma unroll
    for (int t = 0; t < 9; ++t) {
      const int dy = t / 3, dx = t - dy * 3;
#pragma unroll
      for (int rb = 0; rb < RB; ++rb) {
        const int p = 16 * rb + li;                                     // output pixel inside the block
        // the tap's pixel in the (upsampled) map the convolution sees, then the input pixel behind it
        const int uy = oy0 + p / WD + dy - 1, ux = p % WD + dx - 1;
        const bool inside = (unsigned)ux < (unsigned)WD && (unsigned)uy < (unsigned)OH;
        const int iy = UPS ? uy >> 1 : uy, ix = UPS ? ux >> 1 : ux;
        const int q = (iy - iy_lo) * G::WI + ix;                        // slab pixel
        aoff[t][rb] = inside ? q * 64 + 16 * (lh ^ ((q >> 2) & 3)) : G::SPXP * 64 + 16 * lh;
      }
    }
    const unsigned wso = (unsigned)ct * tile_bytes + (unsigned)(wave * 9) * 1024u;
    ws_pass<9, NSL, SLAB, RB>(rs_x, svo, 256u, Rc, rs_w, wso, aoff, acc, wbase, lane);
  }
  // ---- the dense shortcut segment(s): one "tap", the block's own rows
  for (int sgm = 0; sgm < 2; ++sgm) {
    const bf16_t* p = sgm ? a.sc1 : a.sc0;
    const int cc = sgm ? a.scc1 : a.scc0, Rs = sgm ? Rs1 : Rs0;
    if (!p || !Rs) continue;
    const __amdgpu_buffer_rsrc_t rs_s = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_t*>(p), 0, (int)((size_t)a.M * cc * 2), 0x00020000);
    unsigned svo[RB];
#pragma unroll
    for (int i = 0; i < RB; ++i) {
      const int q = 16 * i + (lane >> 2), cp = lane & 3;
      const int c = cp ^ ((q >> 2) & 3);
      svo[i] = (unsigned)(((size_t)(m0 + q)) * cc + 32 * wave + 8 * c) * 2u;
    }
    int aoff[1][RB];
#pragma unroll
    for (int rb = 0; rb < RB; ++rb) {
      const int q = 16 * rb + li;
      aoff[0][rb] = q * 64 + 16 * (lh ^ ((q >> 2) & 3));
    }
    const unsigned wso = (unsigned)ct * tile_bytes + (unsigned)((Rc * 9 + (sgm ? Rs0 : 0)) * 4 + wave) * 1024u;
    ws_pass<1, RB, SLAB, RB>(rs_s, svo, 256u, Rs, rs_w, wso, aoff, acc, wbase, lane);
  }

  // ---- reduce the four waves' partial tiles in wave order (wave v finishes the pixel blocks v, v + 4, ...), epilogue, store
  __syncthreads();                                         // every wave is done with its ring: the area is reused below
  f32x4* red = reinterpret_cast<f32x4*>(smem);             // [wave][rb][lane]
#pragma unroll
  for (int rb = 0; rb < RB; ++rb) red[(wave * RB + rb) * 64 + lane] = acc[rb];
  __syncthreads();
#pragma unroll
  for (int j = 0; j < (RB + 3) / 4; ++j) {
    const int rb = wave + 4 * j;
    if (rb >= RB) break;                                   // (RB = 3, 6: the 12- and 24-wide maps)
    f32x4 v = red[(0 * RB + rb) * 64 + lane];
#pragma unroll
    for (int w = 1; w < 4; ++w) v += red[(w * RB + rb) * 64 + lane];
    const int m = m0 + 16 * rb + li, n = 16 * ct + 4 * lh;
    v += *reinterpret_cast<const f32x4*>(a.bias + n);
    if (a.rowvec) v += *reinterpret_cast<const f32x4*>(a.rowvec + (size_t)img * a.ld_rowvec + n);
    if (a.res) {
      const u32x2 rr = *reinterpret_cast<const u32x2*>(a.res + (size_t)m * a.ldres + n);
      v[0] += bflo(rr[0]); v[1] += bfhi(rr[0]); v[2] += bflo(rr[1]); v[3] += bfhi(rr[1]);
    }
    const u32x2 o = {pack2bf(v[0], v[1]), pack2bf(v[2], v[3])};
    *reinterpret_cast<u32x2*>(a.out + (size_t)m * a.ldo + n) = o;
  }
}

// variants: 1 = 64-pixel blocks (8-, 16-, 32-wide maps); 2 = 128-pixel blocks (16-wide maps: half the weight re-reads of variant 1);
// 3 = 48- / 96-pixel blocks (12- / 24-wide maps)
template <int WD, int RB, bool UPS>
int launch_ws(const MvdWsArgs& a, hipStream_t s, bool* lds_set) {
  using G = WsGeom<WD, RB, UPS>;
  const int lds = 4 * G::WAVE_BYTES;
  if (!*lds_set) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_ws_kernel<WD, RB, UPS>), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    if (e != hipSuccess) { mvd_set_error("conv_ws: hipFuncSetAttribute: %s", hipGetErrorString(e)); return -2; }
    *lds_set = true;
  }
  const int nb = a.M / G::BM, nct = a.N / 16;
  const int grid = ((nct + 7) / 8) * 8 * nb;
  hipLaunchKernelGGL((conv_ws_kernel<WD, RB, UPS>), dim3(grid), dim3(256), lds, s, a);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) { mvd_set_error("conv_ws launch: %s", hipGetErrorString(e)); return -3; }
  return 0;
}

bool g_ws_lds_set[16][12];

}  // namespace

// The variant a launch takes (0: the shape is not taken).  `a.variant` > 0 forces one (tests, probes).
static int ws_variant(const MvdWsArgs& a) {
  const int ow = a.ups ? 2 * a.W : a.W, hw = (a.ups ? 4 : 1) * a.H * a.W;          // output map width / pixels
  const bool ok1 = a.ups ? (ow == 16 || ow == 32) && hw % 64 == 0 : (ow == 8 || ow == 16 || ow == 32) && hw % 64 == 0;
  const bool ok2 = ow == 16 && hw % 128 == 0;
  // variant 3 (round 5): the maps of a 96 x 96 latent (the reference's 768 x 768 default, infer.py:187) -- 12 wide in 48-pixel
  // blocks (4 rows), 24 wide in 96-pixel blocks (4 rows); the 12 -> 24 upsampling form.  (48 wide is beyond try_ws's work-item cap.)
  const bool ok3 = a.ups ? ow == 24 && hw % 96 == 0 : (ow == 12 && hw % 48 == 0) || (ow == 24 && hw % 96 == 0);
  if (a.variant == 1) return ok1 ? 1 : 0;
  if (a.variant == 2) return ok2 ? 2 : 0;
  if (a.variant == 3) return ok3 ? 3 : 0;
  if (a.variant) return 0;
  if (ok3) return 3;
  if (ok2 && a.M >= 256) return 2;      // one 16x16 map or more: 128-pixel blocks halve the weight re-reads and fit one wave of workgroups
  return ok1 ? 1 : 0;
}

bool mvd_conv_ws_applicable(const MvdWsArgs& a) {
  if (!a.x || !a.w || !a.bias || !a.out || a.B <= 0 || a.H <= 0 || a.W <= 0) return false;
  const int hw = (a.ups ? 4 : 1) * a.H * a.W;
  if (a.M != a.B * hw || a.M > 1024 || !ws_variant(a)) return false;
  if (a.ups && (a.scc0 || a.scc1)) return false;
  if (a.C % 128 || a.C <= 0 || a.N % 16 || a.N <= 0) return false;
  if ((a.scc0 % 128) || (a.scc1 % 128) || (a.scc0 && !a.sc0) || (a.scc1 && !a.sc1) || (a.scc1 && !a.scc0)) return false;
  if ((a.ldo % 4) || a.ldo < a.N || (a.res && ((a.ldres % 4) || a.ldres < a.N))) return false;
  if (a.rowvec && (a.ld_rowvec % 4)) return false;
  const size_t lim = (size_t)1 << 31;
  if ((size_t)a.M * a.C * 2 >= lim || (size_t)a.M * (a.scc0 > a.scc1 ? a.scc0 : a.scc1) * 2 >= lim) return false;
  if ((size_t)a.N / 16 * (size_t)((a.C / 128) * 9 + a.scc0 / 128 + a.scc1 / 128) * 4096 >= lim) return false;
  return true;
}

size_t mvd_conv_ws_packed_elems(int C, int sc, int N) { return (size_t)(N / 16) * (size_t)((C / 128) * 9 + sc / 128) * 4 * 512; }

int mvd_launch_conv_ws(const MvdWsArgs& a, hipStream_t s) {
  if (!mvd_conv_ws_applicable(a)) { mvd_set_error("conv_ws: shape not taken (B=%d H=%d W=%d C=%d sc=%d+%d N=%d M=%d ups=%d variant=%d)", a.B, a.H, a.W, a.C, a.scc0, a.scc1, a.N, a.M, a.ups, a.variant); return -1; }
  int dev = 0;
  (void)hipGetDevice(&dev);
  bool* f = g_ws_lds_set[dev & 15];
  switch ((a.ups ? 1000 : 0) + ws_variant(a) * 100 + (a.ups ? 2 * a.W : a.W)) {
    case 108: return launch_ws<8, 4, false>(a, s, f + 0);
    case 116: return launch_ws<16, 4, false>(a, s, f + 1);
    case 132: return launch_ws<32, 4, false>(a, s, f + 2);
    case 216: return launch_ws<16, 8, false>(a, s, f + 3);
    case 1116: return launch_ws<16, 4, true>(a, s, f + 4);
    case 1132: return launch_ws<32, 4, true>(a, s, f + 5);
    case 1216: return launch_ws<16, 8, true>(a, s, f + 6);
    case 312: return launch_ws<12, 3, false>(a, s, f + 7);
    case 324: return launch_ws<24, 6, false>(a, s, f + 8);
    case 1324: return launch_ws<24, 6, true>(a, s, f + 9);
  }
  mvd_set_error("conv_ws: no kernel for variant %d at map width %d", ws_variant(a), a.W);
  return -1;
}

extern "C" int mvd_op_conv3x3_ws(const void* x, int batch, int h, int w, int c, const void* w_packed, const float* bias,
                                 const float* rowvec, int ld_rowvec, const void* res, const void* sc0, const void* sc1,
                                 int scc0, int scc1, void* out, int n, int variant, void* stream) {
  MvdWsArgs a; memset(&a, 0, sizeof(a));
  a.ups = variant >= 16;                 // (variant + 16: nearest-neighbour 2x upsampling in front of the convolution)
  variant &= 15;
  a.x = (const bf16_t*)x; a.B = batch; a.H = h; a.W = w; a.C = c; a.M = batch * h * w * (a.ups ? 4 : 1); a.N = n; a.variant = variant;
  a.w = (const bf16_t*)w_packed; a.bias = bias; a.rowvec = rowvec; a.ld_rowvec = ld_rowvec;
  a.res = (const bf16_t*)res; a.ldres = n; a.sc0 = (const bf16_t*)sc0; a.sc1 = (const bf16_t*)sc1; a.scc0 = scc0; a.scc1 = scc1;
  a.out = (bf16_t*)out; a.ldo = n;
  return mvd_launch_conv_ws(a, (hipStream_t)stream);
}
