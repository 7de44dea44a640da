// 256x320-tile bf16 MFMA GEMM / implicit-GEMM 3x3 convolution for gfx950, "ping-pong" form.
//
//   out[M][N] = alpha * ( A[M][K] . W[N][K]^T + bias[N] + rowvec[batch(m)][N] ) + res[M][N]
//
// Same contract, operand layouts, LDS image and epilogues as gemm.hip (see its header); what differs is how one
// workgroup spends its time.  gemm.hip's 256x320 kernel ran its eight waves in lock step: every wave computed the
// 64-bit global addresses of its nine LDS-DMA loads (~130 vector instructions per 64-deep K slab, several of them
// quarter-rate integer multiplies), then all waves multiplied, then all met at one barrier -- the matrix pipe sat
// idle through every address-and-read phase (55 % busy, profiles/r01_pmc_sq_counters_attention_gemm.txt).  Here:
//
// * BUFFER-ADDRESSED LDS-DMA (buffer_load_dwordx4 ... lds): the per-lane byte offset of a load is loop invariant
//   (row-in-block * row pitch + 16-byte chunk) and everything that changes from slab to slab / row block to row
//   block / tile to tile is a SCALAR offset.  A dense slab costs no vector instruction at all; a convolution tap
//   costs a bounds compare and a select per load (out-of-image taps get an offset beyond num_records, for which
//   the hardware returns zeros -- no zero buffer, no 64-bit address arithmetic).
// * TWO WAVE GROUPS ONE PHASE APART: waves 0-3 and waves 4-7 (the two waves of each SIMD) run the same program,
//   but the second group executes one extra s_barrier up front.  A slab is four phases, each closed by a barrier:
//       R0: ds_read the fragments of k 0..31 into registers, issue LDS-DMAs of the NEXT slab
//       M0: 40 MFMAs out of registers (s_setprio 1)
//       R1: ds_read the fragments of k 32..63, issue the rest of the next slab's loads
//       M1: 40 MFMAs, then wait for the group's loads
//   so at any time one wave of a SIMD is in an MFMA-only phase while its partner reads LDS / issues loads: the matrix
//   pipe always has a wave to serve.  DMAs stay in flight across barriers (raw s_barrier, explicit waits); the epilogues
//   of the two groups run in one common phase E per tile.
//
// Hazard bookkeeping (g = barrier generation; group 0 passes its p-th program barrier at g = p, group 1 at p + 1;
// slab t occupies program barriers 4t .. 4t+3):
// A group loads the A rows only it reads; W (read by both) is loaded by group 0.
//   WAR  stage (t+1)&1 is re-filled by DMAs issued in R0(t) / R1(t): group 0 from (4t-1, 4t] on, group 1 from (4t, 4t+1].
//        Its last readers are the R1(t-1) phases, closed (with lgkmcnt(0)) by barriers 4t-2 / 4t-1.
//   RAW  W of slab t+1 is first read by group 0 in R0(t+1), after barrier 4t+3; group 0 waits for its DMAs before that
//        barrier (end of its M1(t)).  Group 1's A rows are first read in ITS R0(t+1); it waits at the end of its M1(t).
#include <stdlib.h>
#include <string.h>
#include "kernels.h"

namespace {

constexpr int BN = 320, NT = 512;
constexpr int B_BYTES = BN * 128, B_IT = BN * 8 / NT;   // 5 W-side DMA instructions per wave-slice per slab
constexpr unsigned OOB = 0x80000000u;                   // voffset of a load that must return zeros (>= num_records)

typedef __attribute__((address_space(3))) void lds_void;

// lane id computed on the spot (2 vector ops) and opaque to the optimiser: whatever is derived from it is neither hoisted
// out of a loop nor kept live across one
MVD_DEVINL int fresh_lane() {
  int l = (int)__builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u));
  asm volatile("" : "+v"(l));
  return l;
}

// 16-byte buffer store + the wait states hipcc does not insert.  A store of more than 64 bits reads its data registers
// over several cycles; overwriting them in the next instruction corrupts the last lanes' data.  hipcc's hazard
// recognizer skips this case whenever the store's soffset is an SGPR -- as it always is here -- which the older ISAs
// allowed; on gfx950 it is not safe: in the LayerNorm-fold epilogue a v_pk_mul_f32 directly behind a
// buffer_store_dwordx4 replaced bf16 pairs of lanes 12..15 / 28..31 / ... by halves of the fp32 product (NaNs in the
// output).  The asm READS the data registers, so whatever overwrites them is ordered behind the two wait states.
MVD_DEVINL void store16(u32x4 v, __amdgpu_buffer_rsrc_t rsrc, int voff, int soff) {
  // (default cache policy: with the non-temporal hint, aux = 2, the L2 stops merging the four waves' 160-byte row pieces into
  //  whole lines -- dense class 10.1 -> 12.2 ms/step, fused-LayerNorm 3.4 -> 5.5, 460 -> 425 fwd/s on the same box)
  __builtin_amdgcn_raw_buffer_store_b128(v, rsrc, voff, soff, 0);
  asm volatile("s_nop 1" :: "v"(v));
}

MVD_DEVINL int swz_off(int row, int chunk) { return row * 128 + ((chunk ^ ((row >> 1) & 7)) << 4); }

// one 16-byte-per-lane LDS-DMA: LDS destination = wave-uniform base + lane * 16
MVD_DEVINL void dma16(__amdgpu_buffer_rsrc_t rsrc, unsigned char* lds_wave_base, unsigned voff, unsigned soff) {
  __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (lds_void*)lds_wave_base, 16, (int)voff, (int)soff, 0, 0);
}

// WM x WN wave grid: 2 x 4 (wave tile 128 x 80) for plain epilogues, 4 x 2 (64 x 160: value/gate column tiles pair
// up inside a wave) for GEGLU.  AMODE: 0 dense, 1 conv, 2 conv + dense (1x1 shortcut) segment, 3 conv behind a fused
// nearest-2x upsample.
// BM = 256 rows per tile (the default) or 128 (levels whose 256-row grid cannot fill the chip: twice the tiles, wave tile 64 x 80).
//
// LNF: LayerNorm of the A rows folded in (MvdGemmArgs::ln_c1; W carries gamma, the epilogue applies
//   out = rstd[m] * (acc - mean[m] * c1[n]) + c2[n]).  The row statistics are accumulated from the very fragments the MFMAs
// consume (v_dot2c_f32_bf16 in the shadow of the matrix pipe), so the normalised activation is never written or re-read
// and there is no LayerNorm kernel.  The WN waves that share a block of rows split its row tiles between them (TM / WN
// each) and trade (rstd, -rstd * mean) through a 6 KB exchange area behind the two LDS stages; the tile's column
// constants c1 / c2 are LDS-DMA'd into the same area at the tile's first slab (they are only needed by its epilogue, and
// 2 x TN x 4 registers for them do not exist in this kernel).
constexpr int XCH_C1 = 0, XCH_C2 = 2048, XCH_STATS = 4096, XCH_BYTES = 6144;

template <int BM, int WM, int WN, int AMODE, bool SPLITK, bool LNF = false>
__global__ __launch_bounds__(NT, 2) void gemm_pp_kernel(const MvdGemmArgs a) {
  static_assert(!LNF || (AMODE == 0 && !SPLITK), "the LayerNorm fold is a dense, unsplit form");
  constexpr int A_BYTES = BM * 128, STAGE_BYTES = A_BYTES + B_BYTES;
  constexpr int A_IT = BM * 8 / NT;                       // A-side DMA instructions per wave per slab
  static_assert(A_IT == 4, "the A loader below is written for 256-row tiles");
  constexpr int WTM = BM / WM, WTN = BN / WN, TM = WTM / 16, TN = WTN / 16;
  constexpr bool GEGLU = WM == 4;
  constexpr bool HAS_CONV = AMODE != 0;
  constexpr bool UPS = AMODE == 3;
  static_assert((WTM * 128) % 2048 == 0 && (WTN % 16) == 0, "fragment rows of a wave share one swizzle pattern");
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);     // scalar: LDS-DMA bases and the group test stay on the SALU
  const int grp = wave >> 2;                                      // ping-pong group (SIMD partners are waves w, w + 4)
  const int wm = wave / WN, wn = wave % WN;
  const int ntn = a.N / BN;
  const int ntm = (a.M + BM - 1) / BM;
  const int lrow = tid >> 3;                                      // 0..63: row inside a 64-row DMA block (W loads)
  const int kc = (tid & 7) ^ ((lrow >> 1) & 7);                   // source chunk (the XOR swizzle lives on the source side)
  // A loads: EACH GROUP LOADS THE ROWS IT MULTIPLIES (group g = row block wm's half: rows 128 g .. 128 g + 127 of the tile, the
  // 64-row blocks 2g and 2g + 1).  Load i of wave w (w4 = w & 3) is the 8-row slice w4 + 4 (i >> 1) of block 2g + (i & 1);
  // slices w4 and w4 + 4 have the same swizzle parity, so kc above serves both.  Nobody but group g reads those rows, so
  // group 1 may issue half of its A loads as late as its R1 and wait for them at the end of its M1 (one phase of cover,
  // like group 0) -- its R0, the long pole against group 0's 640-cycle MFMA phase, carries two loads instead of four.
  const int w4 = wave & 3;
  const int arow = w4 * 8 + (lane >> 3);                          // row inside the block of the i < 2 loads; i >= 2: + 32

  // ---- tile walk (as gemm.hip): XCD x owns a contiguous range of work items, its workgroups stride through it
  const int S = SPLITK ? a.splitk : 1;
  const int ntiles = ntn * ntm * S;
  const int xcd = blockIdx.x & 7, xj = blockIdx.x >> 3;
  const int gx = (gridDim.x >> 3) + ((int)(gridDim.x & 7) > xcd ? 1 : 0);
  const int tq = ntiles >> 3, tr = ntiles & 7;
  const int tstart = xcd < tr ? xcd * (tq + 1) : tr * (tq + 1) + (xcd - tr) * tq;
  const int tend = tstart + tq + (xcd < tr ? 1 : 0);
  // (A contiguous sub-range per workgroup -- the N tiles of a row block one after the other instead of side by side -- was
  //  measured for every form: dense +3 %, fused-LayerNorm +6 %, GEGLU +7 % slower; convolutions and split-K within +-1 %.)
  // Column-group walk (a.walk_cg = c > 0; unsplit launches whose XCD ranges are whole row blocks): local index i of the XCD's
  // range -> group g = i / (rows * c), row (i % (rows * c)) / c, column g * c + i % c.  32 workgroups of an XCD then sit on
  // 32 / c row blocks x c column tiles: the c weight panels are fetched once per XCD and group, not once per row-block pair.
  const int wcg = SPLITK ? 0 : a.walk_cg;
  const int xrows = (tend - tstart) / ntn, xrow0 = tstart / ntn;
  auto tile_mn = [&](int t, int& m0, int& n0) {
    if (wcg) {
      const int li = t - tstart, per = xrows * wcg;
      const int g = li / per, rem = li - g * per;
      const int r = rem / wcg;
      m0 = (xrow0 + r) * BM; n0 = (g * wcg + (rem - r * wcg)) * BN;
    } else { m0 = (t / ntn) * BM; n0 = (t % ntn) * BN; }
  };
  const int tile_first = tstart + xj, tile_end = tend, tile_step = gx;
  int tile = tile_first;
  if (tile >= tile_end) return;

  const MvdASeg& cs = a.seg[0];                       // conv segment (AMODE 1, 2)
  const MvdASeg& ds = a.seg[AMODE == 2 ? 1 : 0];      // dense segment (AMODE 0, 2)
  const int nkt_conv = HAS_CONV ? (9 * cs.c0) / 64 : 0;
  const int nkt = a.Ktot / 64;
  const int conv_c2 = cs.c0 * 2;                      // bytes per input pixel
  const int conv_rowB = cs.inW * conv_c2;             // bytes per input row
  constexpr bool conv_ups = UPS;
  const int limH = conv_ups ? 2 * cs.inH : cs.inH, limW = conv_ups ? 2 * cs.inW : cs.inW;
  const int dc0 = ds.c0, dc1 = LNF ? 0 : ds.c1;     // (LNF: one source)

  // ---- buffer descriptors (scalar).  The conv descriptor starts one row + one pixel BEFORE the feature map so
  // that tap (dy, dx) is a non-negative scalar offset (dy * row + dx * pixel) from a per-lane base; nothing below
  // the map is ever dereferenced (those taps are out of the image and take the OOB offset).
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
  // loop-invariant per-lane offsets: row-in-block * pitch + chunk.  Rows >= M of a dense source lie beyond num_records
  // and read as zeros (their outputs are never stored).
  const unsigned voff_w = (unsigned)lrow * (unsigned)a.ldw * 2u + kc * 16;
  const unsigned voff_d0 = (unsigned)arow * (unsigned)dc0 * 2u + kc * 16;
  const unsigned voff_d1 = (unsigned)arow * (unsigned)dc1 * 2u + kc * 16;

  // ---- loader state of the tile whose slabs are being fetched (may run one tile ahead of the multiplying tile)
  int ld_m0 = 0, ld_n0 = 0;
  unsigned a_base[A_IT];      // conv: byte offset of the window's top-left tap (shifted origin), + chunk
  int a_yx[A_IT];             // conv: (oy*stride) | (ox*stride) << 16 | in-image mask of the 9 taps << 22 (bit 22 + tap)
  auto setup_loader = [&](int work) {
    const int t = S == 1 ? work : work / S;
    tile_mn(t, ld_m0, ld_n0);
    if (HAS_CONV) {
#pragma unroll
      for (int i = 0; i < A_IT; ++i) {
        int m = ld_m0 + (2 * grp + (i & 1)) * 64 + arow + (i >> 1) * 32;
        m = m < a.M ? m : a.M - 1;
        const int b = m / a.rows_per_batch;
        const int rem = m - b * a.rows_per_batch;
        const int oy = rem / a.outW, ox = rem - oy * a.outW;
        const int ys = oy * cs.stride + cs.asym, xs = ox * cs.stride + cs.asym;   // asym: the window starts AT (2oy, 2ox)
        // which of the nine taps fall inside the image: worked out once per tile (the slab loop then tests one bit per
        // load instead of re-deriving both coordinates and comparing them -- the read phases are the long pole)
        int okm = 0;
#pragma unroll
        for (int t = 0; t < 9; ++t) {
          const int iy = ys - 1 + t / 3, ix = xs - 1 + t % 3;
          okm |= ((unsigned)iy < (unsigned)limH && (unsigned)ix < (unsigned)limW) ? (1 << t) : 0;
        }
        a_yx[i] = (ys & 0x7ff) | ((xs & 0x7ff) << 11) | (okm << 22);
        // top-left tap (ys-1, xs-1) in source coordinates; with the fused nearest-2x upsample the source row of
        // upsampled row r is r >> 1 (arithmetic), the parity-dependent +1 of the middle tap is added per load
        const int ty = conv_ups ? ((ys - 1) >> 1) : ys - 1, tx = conv_ups ? ((xs - 1) >> 1) : xs - 1;
        a_base[i] = (unsigned)((b * cs.inH * cs.inW + (ty + 1) * cs.inW + (tx + 1)) * conv_c2 + kc * 16);
      }
    }
  };

  // LDS-DMAs of slab lk (of the loader's tile) into stage st.  Each group loads the 128 A rows it multiplies (four loads per
  // wave, see above); the W operand -- whose offsets are scalar apart from one invariant per-lane term -- is loaded by GROUP 0
  // ALONE (both slices w and w+4 of each block), spread over its two read phases.  Per slab group 0 issues 4 + 3 loads in R0
  // and 7 in R1, group 1 two A loads in R0 and two in R1: the read phases are what the partner's 640-cycle MFMA phase waits for.
  auto issue_a = [&](int st, int lk, int i0, int i1) {
    // LDS image of the A operand: [64-row block][8-row slice][8 rows][128 B]; load i -> block 2g + (i & 1), slice w4 + 4 (i >> 1)
    unsigned char* sa = smem + st * STAGE_BYTES + (2 * grp) * 8192 + w4 * 1024;
    if (HAS_CONV && (AMODE != 2 || lk < nkt_conv)) {
      // K order [channel slice][tap][64 channels] (gemm.hip): slice = lk / 9, tap = lk % 9
      const int sl = lk / 9, tap = lk - sl * 9;
#ifdef PP_ABLATE_A_TAPS   // round-5 timing-only ablation (WRONG RESULTS): the A operand is fetched for tap 0 of every 64-channel slice only --
      if (tap != 0) return;   // what an activation patch held in LDS across its nine taps could save at most
#endif
      const int dy = tap / 3, dx = tap - dy * 3;
      unsigned soff = (unsigned)(sl * 128);
      if (!conv_ups) soff += (unsigned)(dy * conv_rowB + dx * conv_c2);
      else soff += (unsigned)((dy == 2 ? conv_rowB : 0) + (dx == 2 ? conv_c2 : 0));
#pragma unroll
      for (int i = 0; i < A_IT; ++i) {
        const bool ok = (a_yx[i] >> (22 + tap)) & 1;
        unsigned vo = a_base[i];
        if (conv_ups) {   // middle tap: +1 source row/pixel iff the upsampled coordinate (ys-1 / xs-1) is odd, i.e. ys / xs even
          if (dy == 1) vo += (a_yx[i] & 1) ? 0u : (unsigned)conv_rowB;
          if (dx == 1) vo += (a_yx[i] & 0x800) ? 0u : (unsigned)conv_c2;
        }
        if (i >= i0 && i < i1) dma16(rs_c, sa + (i & 1) * 8192 + (i >> 1) * 4096, ok ? vo : OOB, soff);
      }
    } else {
      const int cc = (lk - nkt_conv) << 6;             // first K column of the slab inside the dense segment
      const bool first = LNF || cc < dc0;
      const int pitch = (first ? dc0 : dc1) * 2;
      const unsigned col2 = (unsigned)((first ? cc : cc - dc0) * 2);
#pragma unroll
      for (int i = 0; i < A_IT; ++i) {
        if (i < i0 || i >= i1) continue;
        const unsigned soff = (unsigned)(ld_m0 + (2 * grp + (i & 1)) * 64 + (i >> 1) * 32) * (unsigned)pitch + col2;
        if (first) dma16(rs_d0, sa + (i & 1) * 8192 + (i >> 1) * 4096, voff_d0, soff);
        else dma16(rs_d1, sa + (i & 1) * 8192 + (i >> 1) * 4096, voff_d1, soff);
      }
    }
  };
  // W loads q0 <= q < q1 of group 0's ten per slab: q -> (64-row block q >> 1, 8-row slice wave + 4 * (q & 1))
  constexpr int W_Q = 2 * B_IT, W_Q_R0 = 3;   // (1: R0 shorter in isolation, but 7 -> 9 late loads expose their latency in situ: conv +2 %, split-K +5 % slower)
  auto issue_w = [&](int st, int lk, int q0, int q1) {
    unsigned char* sb = smem + st * STAGE_BYTES + A_BYTES + wave * 1024;
    const unsigned ldw2 = (unsigned)a.ldw * 2u;
    const unsigned so = (unsigned)ld_n0 * ldw2 + (unsigned)lk * 128u;
#pragma unroll
    for (int q = 0; q < W_Q; ++q)
      if (q >= q0 && q < q1) dma16(rs_w, sb + (q >> 1) * 8192 + (q & 1) * 4096, voff_w, so + (unsigned)((q >> 1) * 64 + (q & 1) * 32) * ldw2);
  };

  f32x4 acc[TM][TN];
  // LNF: for the wave's TS row tiles (wn * TS + t), this lane's share (its 8 of every 32 k) of sum x and sum x^2 of row fr
  constexpr int TS = TM / WN;
  static_assert(!LNF || (TS * WN == TM && WM * TM * 16 * 8 <= XCH_BYTES - XCH_STATS && BN * 4 <= XCH_C2 - XCH_C1), "exchange area layout");
  float row_s[LNF ? TS : 1], row_q[LNF ? TS : 1];
  unsigned char* const xch = smem + 2 * STAGE_BYTES;
  const int fr = lane & 15, fq = lane >> 4;
  const float alpha = a.alpha;

  // Accumulators start at ZERO; bias and the per-batch row vector are added by the epilogue, whose loads all precede its
  // stores.  (gemm.hip starts the accumulators at bias + row vector, i.e. loads them AFTER the previous tile's stores:
  // vmcnt counts loads and stores together, in order, so such a load can only be waited for by draining every store in
  // front of it -- a full HBM write round trip per tile and wave at the top of the next slab.)
  auto init_acc = [&]() {
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int j = 0; j < TN; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    if constexpr (LNF) {
#pragma unroll
      for (int t = 0; t < TS; ++t) { row_s[t] = 0.f; row_q[t] = 0.f; }
    }
  };
  // LNF: column constants of tile column n0 -> exchange area (wave 0: c1, wave 1: c2; 320 floats = 1.25 wave-loads each,
  // columns beyond N read as zeros).  Issued in the first R0 of a tile, i.e. behind the barrier that closed the previous
  // tile's epilogue (their last reader); waited for with the slab's other DMAs.
  auto issue_consts = [&](int n0) {
    if constexpr (LNF) {
      if (wave < 2) {
        __amdgpu_buffer_rsrc_t rs_k = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(wave == 0 ? a.ln_c1 : a.bias), 0, a.N * 4, 0x00020000);
        unsigned char* dst = xch + (wave == 0 ? XCH_C1 : XCH_C2);
        const unsigned vo = (unsigned)fresh_lane() * 16u;
        dma16(rs_k, dst, vo, n0 * 4);
        dma16(rs_k, dst + 1024, vo, n0 * 4 + 1024);
      }
    }
  };
  // LNF: finish the wave's TS row tiles -> exchange area [row block wm][row tile][row fr] (rstd, -rstd * mean).  The four
  // lanes fr, fr+16, fr+32, fr+48 hold the four k-quarters of a row.  E[x^2] - mean^2 in fp32 over one row (K <= 1280).
  auto stats_publish = [&]() {
    if constexpr (LNF) {
      const float invk = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, 1.f / (float)a.Ktot)));   // (an SGPR)
#pragma unroll
      for (int t = 0; t < TS; ++t) {
        float sm = row_s[t], sq = row_q[t];
        sm += __shfl_xor(sm, 16); sq += __shfl_xor(sq, 16);
        sm += __shfl_xor(sm, 32); sq += __shfl_xor(sq, 32);
        const float mean = sm * invk;
        const float var = fmaxf(sq * invk - mean * mean, 0.f);
        const float sc = rsqrtf(var + a.ln_eps);
        if (fq == 0) *reinterpret_cast<float2*>(xch + XCH_STATS + (((wm * TM + wn * TS + t) * 16 + fr) << 3)) = float2{sc, -sc * mean};
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    }
  };
  // column constants of a tile: bias + (row vector of the tile's batch element, when the tile lies inside one)
  auto load_colconst = [&](int m0, int n0, f32x4 (&cb)[TN]) {
    const int nb = n0 + wn * WTN + fq * 4;
#pragma unroll
    for (int j = 0; j < TN; ++j) cb[j] = f32x4{0.f, 0.f, 0.f, 0.f};
    if (a.bias) {
#pragma unroll
      for (int j = 0; j < TN; ++j) cb[j] = *reinterpret_cast<const f32x4*>(a.bias + nb + j * 16);
    }
    if (!GEGLU && a.rowvec && a.rows_per_batch % BM == 0) {
      const float* rv = a.rowvec + (size_t)(m0 / a.rows_per_batch) * a.ld_rowvec + nb;
#pragma unroll
      for (int j = 0; j < TN; ++j) cb[j] += *reinterpret_cast<const f32x4*>(rv + j * 16);
    }
  };

  // ---- epilogue.  Every load / store is BUFFER addressed: the per-lane byte offset (row fr of a 16-row tile, column
  // group) is tile invariant, the tile / row-tile / column-tile position is a scalar offset, and rows >= M fall beyond
  // num_records (loads return zeros, stores are dropped) -- no address arithmetic, no per-row predication.
  // A lane owns 4 consecutive channels (8 bytes) of one row per 16-column tile; v_permlane16_swap pairs up two adjacent
  // column tiles so that a lane holds 8 consecutive channels: one 16-byte access per lane, 64 contiguous bytes per row
  // and instruction, half the memory instructions (the store tail is issue bound, not bandwidth bound).
  //   before: lane (fr, fq) has tile j cols 4fq..4fq+3 in `x`, tile j+1 cols 4fq..4fq+3 in `y`
  //   after : fq 0: tile j cols 0-7 | fq 1: tile j+1 cols 0-7 | fq 2: tile j cols 8-15 | fq 3: tile j+1 cols 8-15  (x | y)
  auto swap_pair = [&](u32x2& x, u32x2& y) {
    auto r0 = __builtin_amdgcn_permlane16_swap(x[0], y[0], false, false);
    auto r1 = __builtin_amdgcn_permlane16_swap(x[1], y[1], false, false);
    x[0] = r0[0]; y[0] = r0[1]; x[1] = r1[0]; y[1] = r1[1];
  };
  const int pair_col = (fq & 1) * 16 + (fq >> 1) * 8;      // column of the lane's 16-byte piece inside a pair of column tiles
  auto epilogue = [&](int m0, int n0, int ks) {
    asm volatile("" : "+s"(m0), "+s"(n0));
    const int row0 = m0 + wm * WTM;                        // (scalar) first row of the wave tile
    if (SPLITK) {   // raw fp32 partial tile; bias / residual are applied by the reduce kernel
      __amdgpu_buffer_rsrc_t rs_p = __builtin_amdgcn_make_buffer_rsrc(a.part + (size_t)ks * a.M * a.N, 0, (int)((size_t)a.M * a.N * 4), 0x00020000);
      const int vo = (fr * a.N + fq * 4) * 4;
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
          store16(__builtin_bit_cast(u32x4, acc[i][j]), rs_p, vo, ((row0 + i * 16) * a.N + n0 + wn * WTN + j * 16) * 4);
      return;
    }
    __amdgpu_buffer_rsrc_t rs_o = __builtin_amdgcn_make_buffer_rsrc(a.out, 0, (int)((size_t)a.M * a.ldo * 2), 0x00020000);
    // (LNF is out of registers: its per-lane epilogue offsets are re-derived here from an opaque copy of the lane id so that
    //  they are not hoisted out of the tile loop -- a hoisted value gets spilled, and a scratch reload inside the slab loop
    //  comes with a vmcnt(0) that drains the DMA stream)
    int vo16, vo8, efr = fr, efq = fq;
    if constexpr (LNF) {
      const int el = fresh_lane();
      efr = el & 15; efq = el >> 4;
      vo16 = (efr * a.ldo + (efq & 1) * 16 + (efq >> 1) * 8) * 2; vo8 = (efr * a.ldo + efq * 4) * 2;
    } else {
      vo16 = (fr * a.ldo + pair_col) * 2; vo8 = (fr * a.ldo + fq * 4) * 2;
      // (round 4, timing-only experiment: the same stores aimed at 8 rows x 128 B per instruction instead of 16 rows x 64 B --
      //  wrong addresses, an upper bound of what a staged whole-line epilogue could buy here: K = 640 dense -4 %, N = 2560 -7 %,
      //  K >= 1280 and the convolutions -0...2 %; not worth re-laying the tile through LDS: tools/probe_pp_stores.py)
    }
    if constexpr (GEGLU) {
      // column tiles (j, j+1) = (value, gate) of ONE 16-wide output tile; output tiles are then paired for 16-byte stores
      constexpr int NO = TN / 2;                           // output column tiles per wave
      const int oc0 = (n0 + wn * WTN) / 2;
      auto gate = [&](const f32x4& v, const f32x4& g) -> u32x2 {
        return u32x2{pack2bf(v[0] * gelu_erf_f(g[0]), v[1] * gelu_erf_f(g[1])), pack2bf(v[2] * gelu_erf_f(g[2]), v[3] * gelu_erf_f(g[3]))};
      };
      if constexpr (LNF) {
        // same store order as the plain form; the (c1, c2) of a column tile are read from the exchange area at every use
        // (2 x TN x 4 registers for them do not exist next to 160 accumulators; the LDS is idle in this phase)
        const unsigned char* cc = xch + (wn * WTN + efq * 4) * 4;
#pragma unroll
        for (int i = 0; i < TM; ++i) {
          const float2 st = *reinterpret_cast<const float2*>(xch + XCH_STATS + (((wm * TM + i) * 16 + efr) << 3));
          auto lin = [&](int j) -> f32x4 {
            return acc[i][j] * st.x + (*reinterpret_cast<const f32x4*>(cc + XCH_C1 + j * 64) * st.y + *reinterpret_cast<const f32x4*>(cc + XCH_C2 + j * 64));
          };
          u32x2 o[NO];
#pragma unroll
          for (int q = 0; q < NO; ++q) o[q] = gate(lin(2 * q), lin(2 * q + 1));
          const int so = ((row0 + i * 16) * a.ldo + oc0) * 2;
#pragma unroll
          for (int q = 0; q + 1 < NO; q += 2) {
            swap_pair(o[q], o[q + 1]);
            store16(u32x4{o[q][0], o[q][1], o[q + 1][0], o[q + 1][1]}, rs_o, vo16, so + q * 32);
          }
          if (NO & 1) __builtin_amdgcn_raw_buffer_store_b64(o[NO - 1], rs_o, vo8, so + (NO - 1) * 32, 0);
        }
      } else {
        f32x4 cb[TN];
        load_colconst(m0, n0, cb);
#pragma unroll
        for (int i = 0; i < TM; ++i) {
          u32x2 o[NO];
#pragma unroll
          for (int q = 0; q < NO; ++q) o[q] = gate(acc[i][2 * q] + cb[2 * q], acc[i][2 * q + 1] + cb[2 * q + 1]);
          const int so = ((row0 + i * 16) * a.ldo + oc0) * 2;
#pragma unroll
          for (int q = 0; q + 1 < NO; q += 2) {
            swap_pair(o[q], o[q + 1]);
            store16(u32x4{o[q][0], o[q][1], o[q + 1][0], o[q + 1][1]}, rs_o, vo16, so + q * 32);
          }
          if (NO & 1) __builtin_amdgcn_raw_buffer_store_b64(o[NO - 1], rs_o, vo8, so + (NO - 1) * 32, 0);
        }
      }
    } else if constexpr (LNF) {
      // no residual / row vector in this form.  Same store order as the plain epilogue (a row tile's 160 bytes per row go out
      // back to back); the tile's column constants come from the exchange area once per epilogue.
      const unsigned char* cc = xch + (wn * WTN + efq * 4) * 4;
      f32x4 k1[TN], k2[TN];
#pragma unroll
      for (int j = 0; j < TN; ++j) {
        k1[j] = *reinterpret_cast<const f32x4*>(cc + XCH_C1 + j * 64);
        k2[j] = *reinterpret_cast<const f32x4*>(cc + XCH_C2 + j * 64);
      }
#pragma unroll
      for (int i = 0; i < TM; ++i) {
        const float2 st = *reinterpret_cast<const float2*>(xch + XCH_STATS + (((wm * TM + i) * 16 + efr) << 3));
        const int so = ((row0 + i * 16) * a.ldo + n0 + wn * WTN) * 2;
        auto one = [&](int j) -> u32x2 {
          const f32x4 v = (acc[i][j] * st.x + (k1[j] * st.y + k2[j])) * alpha;
          return u32x2{pack2bf(v[0], v[1]), pack2bf(v[2], v[3])};
        };
#pragma unroll
        for (int j = 0; j + 1 < TN; j += 2) {
          u32x2 oa = one(j), ob = one(j + 1);
          swap_pair(oa, ob);
          store16(u32x4{oa[0], oa[1], ob[0], ob[1]}, rs_o, vo16, so + j * 32);
        }
        if (TN & 1) __builtin_amdgcn_raw_buffer_store_b64(one(TN - 1), rs_o, vo8, so + (TN - 1) * 32, 0);
      }
    } else {
      const bool has_res = a.res != nullptr;
      __amdgpu_buffer_rsrc_t rs_r = rs_o;
      if (has_res) rs_r = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_t*>(a.res), 0, (int)((size_t)a.M * a.ldres * 2), 0x00020000);
      const int vr16 = (fr * a.ldres + pair_col) * 2, vr8 = (fr * a.ldres + fq * 4) * 2;
      constexpr int NP = TN / 2;                           // column-tile pairs (+ one single tile when TN is odd)
      f32x4 cb[TN];
      load_colconst(m0, n0, cb);
      const bool row_rv = a.rowvec && a.rows_per_batch % BM != 0;   // a tile straddling batch elements (maps < 256 pixels): per-row vectors
      // residual rows are fetched for RG row tiles at a time (one memory round trip per batch, not per row tile)
      constexpr int RG = HAS_CONV ? 2 : 4;
#pragma unroll
      for (int i0 = 0; i0 < TM; i0 += RG) {
        u32x4 rp[RG][NP > 0 ? NP : 1];
        u32x2 rs1[RG];
        if (has_res) {
#pragma unroll
          for (int g = 0; g < RG; ++g) {
            const int so = ((row0 + (i0 + g) * 16) * a.ldres + n0 + wn * WTN) * 2;
#pragma unroll
            for (int q = 0; q < NP; ++q) rp[g][q] = __builtin_amdgcn_raw_buffer_load_b128(rs_r, vr16, so + q * 64, 0);
            if (TN & 1) rs1[g] = __builtin_amdgcn_raw_buffer_load_b64(rs_r, vr8, so + (TN - 1) * 32, 0);
          }
        }
#pragma unroll
        for (int g = 0; g < RG; ++g) {
          const int i = i0 + g;
          const int so = ((row0 + i * 16) * a.ldo + n0 + wn * WTN) * 2;
          int mrow = row0 + i * 16 + fr;
          mrow = mrow < a.M ? mrow : a.M - 1;
          const float* rvp = row_rv ? a.rowvec + (size_t)(mrow / a.rows_per_batch) * a.ld_rowvec + n0 + wn * WTN + fq * 4 : nullptr;
          auto finish = [&](f32x4 v, int j, u32x2 r) -> u32x2 {
            v += cb[j];
            if (row_rv) v += *reinterpret_cast<const f32x4*>(rvp + j * 16);
            v *= alpha;
            if (has_res) { v[0] += bflo(r[0]); v[1] += bfhi(r[0]); v[2] += bflo(r[1]); v[3] += bfhi(r[1]); }
            return u32x2{pack2bf(v[0], v[1]), pack2bf(v[2], v[3])};
          };
#pragma unroll
          for (int q = 0; q < NP; ++q) {
            u32x2 ra = {0u, 0u}, rb = {0u, 0u};
            if (has_res) {   // the 16-byte residual piece is in the paired layout: the same swap (an involution) un-pairs it
              ra = u32x2{rp[g][q][0], rp[g][q][1]}; rb = u32x2{rp[g][q][2], rp[g][q][3]};
              swap_pair(ra, rb);
            }
            u32x2 oa = finish(acc[i][2 * q], 2 * q, ra), ob = finish(acc[i][2 * q + 1], 2 * q + 1, rb);
            swap_pair(oa, ob);
            store16(u32x4{oa[0], oa[1], ob[0], ob[1]}, rs_o, vo16, so + q * 64);
          }
          if (TN & 1) __builtin_amdgcn_raw_buffer_store_b64(finish(acc[i][TN - 1], TN - 1, has_res ? rs1[g] : u32x2{0u, 0u}), rs_o, vo8, so + (TN - 1) * 32, 0);
        }
      }
    }
  };

  auto slab_range = [&](int work, int& k0, int& k1) {
    if (S == 1) { k0 = 0; k1 = nkt; return; }
    const int ks = work % S;
    k0 = (ks * nkt) / S;
    k1 = ((ks + 1) * nkt) / S;
  };

  // fragment registers of ONE 32-deep half slab; read in an R phase, consumed by the M phase behind the barrier
  // Every fragment row of a wave is fr + a multiple of 16, so the XOR swizzle ((row >> 1) & 7) is that of fr: ONE per-lane
  // byte offset per half slab (the halves differ by chunk bit 2 = 64 bytes), the row tile is an immediate offset and the
  // stage / wave position is scalar.
  bf16x8 af[TM], wf[TN];
  u32x4 sf[LNF ? TS : 1];      // LNF: a second copy of the A fragments of the wave's own statistics row tiles (wn * TS + t)
  const int frag_off = fr * 128 + ((fq ^ ((fr >> 1) & 7)) << 4);
  auto read_frags = [&](int st, int half) {
    const unsigned char* sa = smem + st * STAGE_BYTES + wm * (WTM * 128) + (frag_off ^ (half * 64));
    const unsigned char* sb = smem + st * STAGE_BYTES + A_BYTES + wn * (WTN * 128) + (frag_off ^ (half * 64));
#pragma unroll
    for (int j = 0; j < TN; ++j) wf[j] = *reinterpret_cast<const bf16x8*>(sb + j * 2048);
#pragma unroll
    for (int i = 0; i < TM; ++i) af[i] = *reinterpret_cast<const bf16x8*>(sa + i * 2048);
    if constexpr (LNF) {   // (a scalar offset picks the tiles: cheaper than selecting them out of af[] with 3 v_cndmask per dword)
#pragma unroll
      for (int t = 0; t < TS; ++t) sf[t] = *reinterpret_cast<const u32x4*>(sa + (wn * TS + t) * 2048);
    }
  };
  auto mfma_half = [&]() {
    __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int j = 0; j < TN; ++j)
        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[j], af[i], acc[i][j], 0, 0, 0);
    if constexpr (LNF) {
      // 8 * TS dot products, independent of the MFMAs: they issue in the shadow of the matrix pipe
      typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2;
      const bf16x2 ones = __builtin_bit_cast(bf16x2, 0x3f803f80u);
#pragma unroll
      for (int t = 0; t < TS; ++t) {
        const unsigned d[4] = {sf[t].x, sf[t].y, sf[t].z, sf[t].w};
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const bf16x2 x = __builtin_bit_cast(bf16x2, d[e]);
          row_q[t] = __builtin_amdgcn_fdot2_f32_bf16(x, x, row_q[t], false);
          row_s[t] = __builtin_amdgcn_fdot2_f32_bf16(x, ones, row_s[t], false);
        }
        asm volatile("" : "+v"(row_q[t]), "+v"(row_s[t]));   // pinned to this phase (hipcc otherwise sinks both halves' sums behind M1)
      }
    }
    __builtin_amdgcn_s_setprio(0);
  };
  // phase boundary: nothing moves across it at compile time; the barrier itself is the raw s_barrier (no implied
  // vmcnt(0): LDS-DMAs stay in flight across it)
#ifdef MVD_PROBE
  // probe builds, MVD_GEMM_DEBUG & 32: shader-clock stamps at every phase boundary of waves 0 and 4 of the first 64
  // workgroups into the (otherwise unused) split-K partial buffer: [wg][grp][512] -- tools/probe_pp_stamps.py
  int stamp_n = 0;
  long long* stamp_p = ((a.dbg & 32) && a.part && blockIdx.x < 64 && (wave & 3) == 0 && lane == 0)
                           ? reinterpret_cast<long long*>(a.part) + ((size_t)blockIdx.x * 2 + grp) * 512 : nullptr;
#endif
  auto phase_end = [&]() {
    __builtin_amdgcn_sched_barrier(0);
#ifdef MVD_PROBE
    if (stamp_p && stamp_n < 511) stamp_p[1 + stamp_n++] = (long long)__builtin_amdgcn_s_memtime();   // arrival at the barrier
#endif
    __builtin_amdgcn_s_barrier();
#ifdef MVD_PROBE
    if (stamp_p && stamp_n < 511) { stamp_p[1 + stamp_n++] = (long long)__builtin_amdgcn_s_memtime(); stamp_p[0] = stamp_n; }   // release
#endif
    __builtin_amdgcn_sched_barrier(0);
  };

  // ---- prologue: first slab of the first work item, by all waves together
  int kt0, kt1;
  slab_range(tile, kt0, kt1);
  setup_loader(tile);
  issue_a(0, kt0, 0, A_IT);
  if (grp == 0) issue_w(0, kt0, 0, W_Q);
  issue_consts(ld_n0);
  init_acc();
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  phase_end();
  if (grp == 1) phase_end();             // the stagger: group 1 runs one phase behind group 0 from here on

  // A finished tile's epilogue (bias / residual loads, ~24 stores: several thousand cycles of memory round trips) is its
  // own phase "E" in which BOTH groups run theirs at the same time (group 0 idles one phase in front of it, group 1 one
  // phase behind it, so the one-phase stagger survives; the fragment registers are dead in E, which the epilogue's
  // temporaries need).  Run back to back instead (each group's epilogue under one 700-cycle MFMA phase of the other), the two
  // epilogues SERIALISE: in-kernel stamps showed 8-21 thousand cycles per group per tile, 50-100 % on top of a five-slab
  // tile (profiles/r02_probe_pp_phase_stamps.log).
  int cur = 0;
  bool pend = false;                    // the previous work item's epilogue is still to run
  int pm0 = 0, pn0 = 0, pks = 0;
  for (;;) {
    const int tl = S == 1 ? tile : tile / S;
    const int ks = S == 1 ? 0 : tile - tl * S;
    int m0, n0;
    tile_mn(tl, m0, n0);
    const int next_tile = tile + tile_step;
    const bool have_next = next_tile < tile_end;
    int nkt0 = 0, nkt1 = 0;
    if (have_next) slab_range(next_tile, nkt0, nkt1);
    for (int kt = kt0; kt < kt1; ++kt) {
      const bool last_k = kt + 1 == kt1;
      const bool more = !last_k || have_next;
      const int nlk = last_k ? nkt0 : kt + 1;          // the slab being fetched
      if (pend) {                                      // ---- E: both groups' epilogues in one common phase
        if (grp == 0) phase_end();                     //   group 0 idles through group 1's last M1 ...
        if constexpr (LNF) { stats_publish(); phase_end(); }
        epilogue(pm0, pn0, pks);
        init_acc();
        phase_end();
        if (grp == 1) phase_end();                     //   ... group 1 through group 0's first R0: the stagger is back
        pend = false;
      }
      // ---- R0
      if (LNF && kt == kt0 && tile != tile_first) issue_consts(n0);   // (the first tile's were issued by the prologue)
      read_frags(cur, 0);
      if (more) {
        if (last_k) setup_loader(next_tile);           // the loader runs ahead into the next work item
        if (grp == 0) { issue_a(cur ^ 1, nlk, 0, A_IT); issue_w(cur ^ 1, nlk, 0, W_Q_R0); }
        else issue_a(cur ^ 1, nlk, 0, A_IT / 2);
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      phase_end();
      // ---- M0
      mfma_half();
      phase_end();
      // ---- R1
      read_frags(cur, 1);
      if (grp == 0) {
        if (more) issue_w(cur ^ 1, nlk, W_Q_R0, W_Q);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      } else {
        if (more) issue_a(cur ^ 1, nlk, A_IT / 2, A_IT);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      }
      phase_end();
      // ---- M1
      mfma_half();
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the group's loads (R0 + R1) have landed: one MFMA phase of cover
      phase_end();
      if (last_k) { pend = true; pm0 = m0; pn0 = n0; pks = ks; }
      cur ^= 1;
    }
    if (!have_next) break;
    tile = next_tile; kt0 = nkt0; kt1 = nkt1;
  }
  // the last work item's.  (LNF: statistics are traded inside a group -- a wave's row block wm determines its group -- so
  // it does not matter that group 1 is still in its last M1 when group 0 publishes.)
  if constexpr (LNF) { stats_publish(); phase_end(); }
  epilogue(pm0, pn0, pks);
  if (grp == 0) phase_end();             // balance the extra barrier group 1 executed up front
}

template <int BM, int WM, int WN, int AMODE, bool SPLITK, bool LNF = false>
int launch_pp(const MvdGemmArgs& a, hipStream_t s) {
  constexpr int LDS_BYTES = 2 * (BM * 128 + B_BYTES) + (LNF ? XCH_BYTES : 0);
  static bool init[16] = {};                            // per device
  int dev = 0;
  (void)hipGetDevice(&dev);
  if (!init[dev & 15]) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_pp_kernel<BM, WM, WN, AMODE, SPLITK, LNF>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES);
    if (e != hipSuccess) { mvd_set_error("gemm_pp: hipFuncSetAttribute: %s", hipGetErrorString(e)); return -2; }
    init[dev & 15] = true;
  }
  const int ntm = (a.M + BM - 1) / BM, ntn = a.N / BN;
  MvdGemmArgs b = a;
  b.walk_cg = SPLITK ? 0 : mvd_gemm_pp_walk(a);
  int grid = 256;                                         // one 144 KB (112 KB at BM = 128) workgroup per CU
  const int ntiles = ntm * ntn * (a.splitk > 1 ? a.splitk : 1);
  if (ntiles < grid) grid = ((ntiles + 7) / 8) * 8;
  g_mvd_last_gemm.tiles = ntiles; g_mvd_last_gemm.grid = grid; g_mvd_last_gemm.per_cu = 1;
  hipLaunchKernelGGL((gemm_pp_kernel<BM, WM, WN, AMODE, SPLITK, LNF>), dim3(grid), dim3(NT), LDS_BYTES, s, b);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) { mvd_set_error("gemm_pp launch: %s", hipGetErrorString(e)); return -3; }
  return 0;
}

template <int BM, int WM, int WN>
int launch_pp_mode(const MvdGemmArgs& a, hipStream_t s) {
  const bool sk = a.splitk > 1;
  if (a.seg[0].mode == MVD_A_DENSE) return sk ? launch_pp<BM, WM, WN, 0, true>(a, s) : launch_pp<BM, WM, WN, 0, false>(a, s);
  if (a.seg[0].ups) {
    if (a.nseg != 1) { mvd_set_error("gemm_pp: an upsampling convolution takes no shortcut segment"); return -1; }
    return sk ? launch_pp<BM, WM, WN, 3, true>(a, s) : launch_pp<BM, WM, WN, 3, false>(a, s);
  }
  if (a.nseg == 1) return sk ? launch_pp<BM, WM, WN, 1, true>(a, s) : launch_pp<BM, WM, WN, 1, false>(a, s);
  return sk ? launch_pp<BM, WM, WN, 2, true>(a, s) : launch_pp<BM, WM, WN, 2, false>(a, s);
}

}  // namespace

// Column-group walk of the ping-pong kernel (MvdGemmArgs::walk_cg): for dense, unsplit launches whose weight operand does not
// fit an XCD's 4 MB L2 beside the activation rows in flight, while every XCD owns >= 8 whole row blocks (M >= 16384: the 32x32
// level at 32 images).  PMC before: the 32x32-level GEGLU (N 5120, K 640: W = 6.5 MB) read 13x its A + W -- with all 16 column
// tiles of two row blocks side by side every XCD re-fetched every weight panel for every pair of row blocks.  c = the most
// column tiles whose panels take <= 1.75 MB (the rest of the L2 holds 32 / c activation panels and the streaming output).
// MVD_GEMM_PP_WALK=0 / debug flag 131072 turn it off (A/B).
int mvd_gemm_pp_walk(const MvdGemmArgs& a) {
  static const int on = MVD_ENV_INT("MVD_GEMM_PP_WALK", 1);
  if (!on || (mvd_debug_flags() & 131072) || a.walk_cg < 0) return 0;
  if (a.walk_cg > 0) return a.walk_cg;                         // forced (tests / probes)
  if (a.seg[0].mode != MVD_A_DENSE || a.nseg != 1 || a.splitk > 1) return 0;
  const int ntm = (a.M + 255) / 256, ntn = a.N / BN;
  if (ntm % 8 || ntm < 64 || ntn < 8) return 0;
  const size_t panel = (size_t)BN * a.Ktot * 2;
  if (panel * ntn <= (size_t)(5 << 19)) return 0;              // W <= 2.5 MB: it stays in L2 as it is
  int c = (int)(((size_t)7 << 18) / panel);                    // 1.75 MB of panels
  if (c < 2) return 0;
  while (c > 1 && (ntn % c || 32 % c)) --c;
  return c >= 2 ? c : 0;
}

// Every byte offset the kernel forms must fit the 32-bit buffer addressing (and stay below OOB = 2^31).
bool mvd_gemm_pp_applicable(const MvdGemmArgs& a) {
  constexpr int BM = 256;
  const size_t lim = (size_t)1 << 31;
  if (a.N % BN || a.Ktot % 64 || a.out_f32) return false;      // (fp32 outputs exist at M = batch only: gemm.hip)
  if ((size_t)a.N * a.ldw * 2 >= lim) return false;
  if ((size_t)(a.M + BM) * a.ldo * 2 >= lim || (a.res && (size_t)(a.M + BM) * a.ldres * 2 >= lim)) return false;
  if (a.splitk > 1 && (size_t)(a.M + BM) * a.N * 4 >= lim) return false;
  for (int i = 0; i < a.nseg; ++i) {
    const MvdASeg& g = a.seg[i];
    if (g.mode == MVD_A_DENSE) {
      if ((size_t)(a.M + BM) * (g.c0 > g.c1 ? g.c0 : g.c1) * 2 >= lim) return false;
    } else {
      const size_t bytes = (size_t)(a.M / a.rows_per_batch) * g.inH * g.inW * g.c0 * 2 + (size_t)(g.inW + 1) * g.c0 * 2;
      if (bytes + (size_t)3 * g.inW * g.c0 * 2 >= lim) return false;
      if (2 * g.inH >= 32768 || 2 * g.inW >= 32768) return false;     // (int arithmetic on pixel coordinates)
    }
  }
  return true;
}

// geglu selects the 4 x 2 wave grid (arguments already validated by mvd_launch_gemm).
// (A 128-row instantiation of the same kernel -- BM = 128, wave tile 64 x 80, for the 16x16 / 8x8 levels -- was measured
//  against the 128x160 lock-step tiles those levels use: no faster, profiles/r02_probe_pp128.log; not instantiated.)
int mvd_launch_gemm_pp(const MvdGemmArgs& a, hipStream_t s) {
  if (a.ln_c1) {
    const MvdASeg& g = a.seg[0];
    if (a.nseg != 1 || g.mode != MVD_A_DENSE || g.c1 || a.splitk > 1 || a.res || a.rowvec || !a.bias) {
      mvd_set_error("gemm_pp: the LayerNorm fold takes one dense source, a bias, no residual / row vector / split-K"); return -1;
    }
    return a.geglu ? launch_pp<256, 4, 2, 0, false, true>(a, s) : launch_pp<256, 2, 4, 0, false, true>(a, s);
  }
  if (a.geglu) {
    if (a.seg[0].mode != MVD_A_DENSE || a.splitk > 1) { mvd_set_error("gemm_pp: GEGLU needs a dense, unsplit problem"); return -1; }
    return launch_pp<256, 4, 2, 0, false>(a, s);
  }
  return launch_pp_mode<256, 2, 4>(a, s);
}

bool mvd_gemm_ln_fold_ok(const MvdGemmArgs& a) {
  const MvdASeg& g = a.seg[0];
  if (a.nseg != 1 || g.mode != MVD_A_DENSE || g.c1 || a.splitk > 1 || a.res || a.rowvec || a.out_f32) return false;
  if (a.Ktot != g.c0 || a.Ktot > 640) return false;      // (the 64x64 and 32x32 levels; the small-M kernels fold at every level)
  // measured (tools/probe_lnfold.py, profiles/r02_probe_lnfold.log): the fold saves 36-47 us per launch at C = 320 and 5-19 us
  // at C = 640 -- except for the GEGLU form at C = 640, which loses 9 us (its epilogue fetches the column constants from LDS
  // at every use): that one keeps ln_kernel + the plain kernel
  if (a.geglu && a.Ktot > 320) return false;
  const int cfg = mvd_gemm_pick_config(a);
  return (a.geglu ? cfg == 6 : cfg == 7) && mvd_gemm_pp_applicable(a);
}
