// "X-stationary" short-K bf16 MFMA GEMM for gfx950: the dense projections of the 64x64 level (K = 320, M = 131072 rows at
// 32 pairs), which the 256x320 ping-pong tile runs at 0.31-0.39 of the MFMA peak and at neither roof.
//
//   out[M][N] = LN?(X)[M][K] . W[N][K]^T + bias[N]  (+ res[M][N])          or, GEGLU:  out = value * gelu_erf(gate)
//
// Why another kernel.  At K = 320 a 256x320 output tile is five 64-deep slabs: 12.8 k cycles of MFMA against a 164 KB
// store that, at the chip's HBM rate (~8 B/clk/CU), takes 20 k cycles -- and the ping-pong kernel stores in a phase of its
// own, re-fetches the A block once per column tile (GEGLU: 516 MB read for 84 MB of A by PMC) and runs its GELU with the
// matrix pipe idle.  Here the roles are turned round:
//
// * THE ACTIVATION ROWS STAY IN REGISTERS.  The MFMA is v_mfma_f32_32x32x16_bf16 with the WEIGHT tile as the A operand
//   (row m = output channel) and the activation block as the B operand (column n = token on the lane).  A wave owns 64
//   tokens (two 32-token column tiles); their K = 320 values are 2 x 20 B-operand fragments = 160 VGPRs, loaded ONCE from
//   HBM straight into the operand layout (lane (r, h) takes bytes [32 s + 16 h, +16) of row r for k-step s).  A never
//   touches LDS and is never re-read.
// * THE WEIGHTS STREAM THROUGH A THREE-STAGE LDS RING in units of one 32-row tile = 20 fragments of 1 KB + one bias
//   fragment, host-packed in FRAGMENT ORDER (packing.pack_xs: [unit][k-step][lane][8 bf16]): an LDS-DMA piece is 1 KB of
//   contiguous global memory, a fragment read is ds_read_b128 at base + 16 * lane (conflict free), and no address is ever
//   computed.  The four waves of a workgroup consume the same unit at the same time, one raw s_barrier per unit; every W
//   fragment read feeds two MFMAs.  W of one problem is 0.2-1.6 MB: it lives in the XCD's L2.
// * BIAS RIDES THE MATRIX PIPE: k-step 20 of a unit holds (bias_hi, bias_lo) as bf16 at k = 0, 1 and multiplies a constant
//   operand that is 1.0 there -- one extra MFMA of 21, no vector instruction, no bias buffer.
// * THE ROWS OF A UNIT ARE PERMUTED ON THE HOST so that the accumulator layout (row = (reg & 3) + 8 (reg >> 2) + 4 h) puts
//   16 CONSECUTIVE output channels into the 16 registers of a lane: a lane stores 32 contiguous bytes of its token's row
//   with two 16-byte buffer stores and no cross-lane exchange.
// * LAYERNORM IN REGISTERS: the whole row is resident, so mean / rstd come from 2 v_dot2c per dword and the fragments are
//   normalised in place (bf16((x - mean) rstd)); W carries gamma, the bias k-step carries W.beta + b (packing.fold_layernorm).
// * GEGLU: units alternate gate | value; gelu(gate) is computed in the gate unit's epilogue and kept in registers, the value
//   unit's epilogue multiplies, packs and stores -- the GELU's ~70 issue cycles per element run beside the OTHER
//   workgroup's MFMAs (two workgroups per CU, one wave of each per SIMD).
//
// Workgroup = 4 waves = 256 rows, one work item (row block x column range) each, two per CU (<= 256 VGPRs, 63 KB of LDS).
// Memory operations are counted by hand (s_waitcnt vmcnt(N): LDS-DMAs, loads and stores retire in order).
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include "../../include/mvd_hip.h"
#include "kernels.h"

namespace {

typedef __attribute__((address_space(3))) void lds_void;

MVD_DEVINL void xs_dma16(__amdgpu_buffer_rsrc_t rsrc, unsigned char* lds_wave_base, unsigned voff, unsigned soff) {
  __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (lds_void*)lds_wave_base, 16, (int)voff, (int)soff, 0, 0);
}
MVD_DEVINL void xs_dma4(__amdgpu_buffer_rsrc_t rsrc, unsigned char* lds_wave_base, unsigned voff, unsigned soff) {
  __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (lds_void*)lds_wave_base, 4, (int)voff, (int)soff, 0, 0);
}
// 16-byte buffer store + the wait states hipcc does not insert for an SGPR soffset (see store16() in gemm_pp.hip)
MVD_DEVINL void xs_store16(u32x4 v, __amdgpu_buffer_rsrc_t rsrc, int voff, int soff) {
  __builtin_amdgcn_raw_buffer_store_b128(v, rsrc, voff, soff, 0);
  asm volatile("s_nop 1" :: "v"(v));
}

#define XS_WAIT_VM(n) asm volatile("s_waitcnt vmcnt(" #n ")" ::: "memory")

template <int KS, bool GEGLU, bool RES, bool LN>
__global__ __launch_bounds__(256, 2) void gemm_xs_kernel(const MvdXsArgs a) {
  // ring depth: 3 units (GEGLU: 2 -- its store staging holds both row tiles of a 128-byte group, 8 KB per wave, and two workgroups
  // share 160 KB; a unit's LDS-DMAs are then issued one unit ahead instead of two: ~4000 cycles of cover for an L2 hit)
  constexpr int NW = 4, RT = 2, NS = GEGLU ? 2 : 3;
  constexpr int STG = GEGLU ? 8192 : 4096;              // store staging per wave
  constexpr int UNIT = (KS + 1) * 1024;                 // KS operand k-steps + the bias k-step
  constexpr int P = KS / NW;                            // 1 KB DMA pieces per wave and unit (+ one 256-byte piece of the bias k-step)
  static_assert(KS % NW == 0 && !(GEGLU && RES) && !(RES && LN), "shape of the instantiations");
  // vector-memory operations a wave issues per iteration BEHIND its DMA issue (the epilogue's loads and stores)
  constexpr int E = 2 * RT + (RES ? 2 * RT : 0);         // (per unit; issued in the odd unit's epilogue for both units of a group)
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int r = lane & 31, h = lane >> 5;
  // ablation builds (tools/build_xs_variant.sh <tag> -DXS_DBG=<bits>): 1 no stores, 2 no MFMAs, 4 no W stream, 8 no X loads, 16 no GELU
#ifdef XS_DBG
  constexpr int dbg = XS_DBG;
#else
  constexpr int dbg = 0;
#endif

  // ---- work item: row block rb (256 rows), column part cp of a.csplit.  The parts of one row block get block ids that are
  // equal mod 8: they run on one XCD (round-robin placement), so the second and later reads of the A block hit its L2.
  int rb, cp;
  if (a.csplit == 1) { rb = blockIdx.x; cp = 0; }
  else {
    const int per = 8 * a.csplit, g = blockIdx.x / per, rem = blockIdx.x - g * per;
    rb = g * 8 + (rem & 7); cp = rem >> 3;
  }
  if (rb * 256 >= a.M) return;
  const int upp = a.units / a.csplit;                   // units per part (host: divisible, even for GEGLU)
  const int u0 = cp * upp, nit = upp;
  const int row_w = rb * 256 + wave * (RT * 32);        // first token of the wave

  const __amdgpu_buffer_rsrc_t rs_w = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_t*>(a.w), 0, a.units * UNIT, 0x00020000);
  auto issue_unit = [&](int it, int stage) {
    unsigned char* dst = smem + stage * UNIT;
    const unsigned so = (unsigned)(u0 + it) * (unsigned)UNIT;
    if (dbg & 4) return;
#pragma unroll
    for (int i = 0; i < P; ++i) xs_dma16(rs_w, dst + (wave * P + i) * 1024, (unsigned)lane * 16u, so + (unsigned)((wave * P + i) * 1024));
    xs_dma4(rs_w, dst + KS * 1024 + wave * 256, (unsigned)lane * 4u, so + (unsigned)(KS * 1024 + wave * 256));
  };
  issue_unit(0, 0);
  if (NS == 3 && nit > 1) issue_unit(1, 1);

  // ---- the wave's 64 tokens -> B-operand fragments (rows >= M lie beyond num_records and read as zeros)
  const __amdgpu_buffer_rsrc_t rs_x = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_t*>(a.x), 0, (int)((size_t)a.M * a.ldx * 2), 0x00020000);
  u32x4 x[RT][KS];
#pragma unroll
  for (int t = 0; t < RT; ++t) {
    const int vo = (row_w + t * 32 + r) * a.ldx * 2 + h * 16;
#pragma unroll
    for (int s = 0; s < KS; ++s) x[t][s] = (dbg & 8) ? u32x4{0x3f803f80u, 0x3f803f80u, 0x3f803f80u, 0x3f803f80u} : __builtin_amdgcn_raw_buffer_load_b128(rs_x, vo + s * 32, 0, 0);
  }
  if constexpr (LN) {
    // LayerNorm of the resident rows.  Lane (r, h) holds half of row r's values (the h-halves of every k-step); the
    // other half is lane r + 32.  sum x^2 - mean^2 in fp32 over one row of K <= 640 values (as the ping-pong fold).
    const float invk = 1.f / (float)(KS * 16);
    const bf16x2 ones2 = __builtin_bit_cast(bf16x2, 0x3f803f80u);
#pragma unroll
    for (int t = 0; t < RT; ++t) {
      float sm = 0.f, sq = 0.f;
#pragma unroll
      for (int s = 0; s < KS; ++s) {
        const unsigned d[4] = {x[t][s].x, x[t][s].y, x[t][s].z, x[t][s].w};
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const bf16x2 v = __builtin_bit_cast(bf16x2, d[e]);
          sq = __builtin_amdgcn_fdot2_f32_bf16(v, v, sq, false);
          sm = __builtin_amdgcn_fdot2_f32_bf16(v, ones2, sm, false);
        }
      }
      sm += __shfl_xor(sm, 32); sq += __shfl_xor(sq, 32);
      const float mean = sm * invk;
      const float rstd = rsqrtf(fmaxf(sq * invk - mean * mean, 0.f) + a.ln_eps);
      const float nb = -mean * rstd;
#pragma unroll
      for (int s = 0; s < KS; ++s) {
        const unsigned d[4] = {x[t][s].x, x[t][s].y, x[t][s].z, x[t][s].w};
        x[t][s] = u32x4{pack2bf(fmaf(bflo(d[0]), rstd, nb), fmaf(bfhi(d[0]), rstd, nb)), pack2bf(fmaf(bflo(d[1]), rstd, nb), fmaf(bfhi(d[1]), rstd, nb)),
                        pack2bf(fmaf(bflo(d[2]), rstd, nb), fmaf(bfhi(d[2]), rstd, nb)), pack2bf(fmaf(bflo(d[3]), rstd, nb), fmaf(bfhi(d[3]), rstd, nb))};
      }
    }
  }
  // B operand of the bias k-step: 1.0 at k = 0, 1 (lanes of half 0, elements 0 and 1), zero elsewhere
  const u32x4 one_frag = {h == 0 ? 0x3f803f80u : 0u, 0u, 0u, 0u};

  const __amdgpu_buffer_rsrc_t rs_o = __builtin_amdgcn_make_buffer_rsrc(a.out, 0, (int)((size_t)a.M * a.ldo * 2), 0x00020000);
  __amdgpu_buffer_rsrc_t rs_r = rs_o;
  if (RES) rs_r = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_t*>(a.res), 0, (int)((size_t)a.M * a.ldres * 2), 0x00020000);
  int vo_r[RT];
#pragma unroll
  for (int t = 0; t < RT; ++t) {
    vo_r[t] = RES ? (row_w + t * 32 + r) * a.ldres * 2 + h * 32 : 0;
  }

#ifdef XS_STAMPS
  // probe build: wave 0 accumulates shader-clock cycles per phase (wait, barrier, DMA issue, multiply, epilogue) and leaves them
  // behind row M of the output buffer (tools/probe_xs_stamps.py allocates those rows)
  long long st_acc[6] = {0, 0, 0, 0, 0, 0};
  const long long st_begin = (long long)__builtin_amdgcn_s_memtime();
  const long long st_rbegin = (long long)__builtin_amdgcn_s_memrealtime();
  long long st_t = st_begin;
#define XS_STAMP(i) do { const long long n_ = (long long)__builtin_amdgcn_s_memtime(); st_acc[i] += n_ - st_t; st_t = n_; } while (0)
#else
#define XS_STAMP(i) do {} while (0)
#endif
  float gs[16 * RT];             // GEGLU: the gate tile (raw, then gelu'd under the value unit's MFMAs)

  // top of an iteration: this wave's pieces of unit `it` have landed (counted wait), everybody's have (barrier) and nobody
  // still reads unit it - 1, whose stage takes unit it + 2
  auto top = [&](int it, int stage) {
    XS_STAMP(4);
    // counted wait for this wave's pieces of unit `it`: everything it issued BEHIND them may still be in flight.
    //  three stages: unit it's DMAs were issued at the top of iteration it - 2 -> behind them: that iteration's epilogue, the DMAs
    //    of unit it + 1 (none in the last iteration), the epilogue of iteration it - 1; an epilogue runs every second unit;
    //  two stages (GEGLU): issued at the top of iteration it - 1 -> behind them: that iteration's epilogue only -- 8 stores behind
    //    every fourth unit (gate | value | gate | value = one 128-byte store group).
    if (it == 0) XS_WAIT_VM(0);
    else if constexpr (GEGLU) { if ((it & 3) == 0) XS_WAIT_VM(8); else XS_WAIT_VM(0); }
    else if (it == nit - 1) { if (RES) XS_WAIT_VM(16); else XS_WAIT_VM(8); }                    // 2 E
    else { if (RES) XS_WAIT_VM(22); else XS_WAIT_VM(14); }                                      // 2 E + P + 1
    static_assert(P + 1 == 6 && 2 * E == (RES ? 16 : 8), "the counted waits above are written for these instantiations");
    __builtin_amdgcn_sched_barrier(0);
    XS_STAMP(0);
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
    XS_STAMP(1);
    if (it + NS - 1 < nit) issue_unit(it + NS - 1, NS == 2 ? (stage ^ 1) : (stage >= 1 ? stage - 1 : 2));
    __builtin_amdgcn_sched_barrier(0);
    XS_STAMP(2);
  };
  // the unit's 32 channels x this wave's 64 tokens: bias k-step first (C = 0), then the KS operand k-steps.
  // SIDE (GEGLU value unit): gelu_erf of the gate tile `side` runs in the shadow of these MFMAs -- an MFMA holds the vector
  // issue port for 8 of its 32 cycles, the ~15 vector instructions of one GELU (two of them transcendental) per lane and
  // element fit the rest: 12 per MFMA, placed by sched_group_barrier.  Behind a unit of its own the GELU was 2 x ~1400 cycles
  // per output tile with the matrix pipe idle (phase stamps).
  auto multiply = [&](int stage, f32x16 (&acc)[RT], float (&side)[16 * RT], bool with_side) {
    const unsigned char* sp = smem + stage * UNIT + lane * 16;
    if (!with_side) {
      {
        const bf16x8 wb = *reinterpret_cast<const bf16x8*>(sp + KS * 1024);
#pragma unroll
        for (int t = 0; t < RT; ++t) {
          f32x16 z;
#pragma unroll
          for (int q = 0; q < 16; ++q) z[q] = 0.f;
          acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wb, __builtin_bit_cast(bf16x8, one_frag), z, 0, 0, 0);
        }
      }
#pragma unroll
      for (int s = 0; s < KS; ++s) {
        const bf16x8 wf = *reinterpret_cast<const bf16x8*>(sp + s * 1024);
#pragma unroll
        for (int t = 0; t < RT; ++t)
          if (!(dbg & 2)) acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf, __builtin_bit_cast(bf16x8, x[t][s]), acc[t], 0, 0, 0);
          else acc[t][s & 15] += __builtin_bit_cast(float, __builtin_bit_cast(u32x4, wf).x);
      }
    } else {
      // hand-placed, one step at a time (step 0 = the bias k-step, then k-step st - 1): read the step's fragment, run the GELU
      // of gate elements 2 st and 2 st + 1 while it arrives (32 elements over the first 16 of 21 steps), then the two MFMAs;
      // nothing moves between steps (one fragment and one GELU's temporaries live: the kernel is at 256 registers)
#pragma unroll
      for (int st = 0; st <= KS; ++st) {
        const bf16x8 wc = *reinterpret_cast<const bf16x8*>(sp + (st == 0 ? KS : st - 1) * 1024);
#ifndef XS_GELU_GROUP
#define XS_GELU_GROUP 2
#endif
        // XS_GELU_GROUP elements per group, every (XS_GELU_GROUP / 2)-th step: their dependent chains (~12 operations each) interleave
        if (st % (XS_GELU_GROUP / 2) == 0 && !(dbg & 16)) {
#pragma unroll
          for (int e = 0; e < XS_GELU_GROUP; ++e) {
            const int v = (st / (XS_GELU_GROUP / 2)) * XS_GELU_GROUP + e;
            if (v < 16 * RT) side[v] = gelu_erf_f(side[v]);
          }
#pragma unroll
          for (int e = 0; e < XS_GELU_GROUP; ++e) {
            const int v = (st / (XS_GELU_GROUP / 2)) * XS_GELU_GROUP + e;
            if (v < 16 * RT) asm volatile("" : "+v"(side[v]));      // finished HERE (hipcc otherwise sinks the polynomial half of every GELU behind the loop: 4 live temporaries per element)
          }
        }
#pragma unroll
        for (int t = 0; t < RT; ++t) {
          if (st == 0) {
            f32x16 z;
#pragma unroll
            for (int q = 0; q < 16; ++q) z[q] = 0.f;
            acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wc, __builtin_bit_cast(bf16x8, one_frag), z, 0, 0, 0);
          } else {
            acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wc, __builtin_bit_cast(bf16x8, x[t][st - 1]), acc[t], 0, 0, 0);
          }
        }
        __builtin_amdgcn_sched_barrier(0);
      }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
    XS_STAMP(3);
  };
  // ---- epilogue stores.  Lane (r, h) holds channels 16 h .. 16 h + 15 of a unit for token r: stored as they stand, one
  // instruction would touch 32 rows x 2 pieces of 16 bytes -- and a store's ISSUE cost grows with the lines it touches (phase
  // stamps: ~460 cycles per such store with eight waves storing, which also holds up every LDS-DMA queued behind it).  So a
  // store group (two units = 128 bytes per token; GEGLU: one output tile = 64 bytes) goes through a 4 KB per-wave staging area:
  // each lane writes its 16-byte chunks at [token][chunk ^ swizzle(token)], reads back [8 (16) tokens x 8 (4) chunks] per
  // instruction and stores WHOLE 128-byte (64-byte) row segments: 8 (16) lines per store instead of 32.  LDS operations of
  // one wave execute in order, so the area needs no barrier; both the ds_write_b128 and the ds_read_b128 are conflict free.
  constexpr int LB = 128, CPR = 8, RPI = 8, NJ = 4;                    // bytes per staged row, chunks per row, rows per instruction, instructions per row tile
  unsigned char* const stg = smem + NS * UNIT + wave * STG;           // plain: one row tile at a time (4 KB); GEGLU: both (8 KB)
  const int sw_w = r & 7;                                              // swizzle of the lane's own token row
  const int rd_row = lane / CPR, rd_c = lane % CPR;                   // the (row, chunk) this lane reads back (+ RPI rows per j)
  int vo_l[RT];
#pragma unroll
  for (int t = 0; t < RT; ++t) vo_l[t] = (row_w + t * 32 + rd_row) * a.ldo * 2 + rd_c * 16;
  const int so_j = RPI * a.ldo * 2;                                    // (scalar) byte step between a lane's rows
  auto stage_chunk = [&](int t, int chunk, u32x4 v) {                  // t: which 4 KB of the staging area (plain: always 0)
    *reinterpret_cast<u32x4*>(stg + t * 4096 + r * LB + ((chunk ^ sw_w) << 4)) = v;
  };
  auto flush_rows = [&](int t, int tbuf, int so) {
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    u32x4 v[NJ];
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
      const int row = j * RPI + rd_row;
      v[j] = *reinterpret_cast<const u32x4*>(stg + tbuf * 4096 + row * LB + ((rd_c ^ (row & 7)) << 4));
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
    for (int j = 0; j < NJ; ++j)
      if (!(dbg & 1) || v[j].x == 0x12345678u) xs_store16(v[j], rs_o, vo_l[t], so + j * so_j);
  };
  auto pack8 = [&](const f32x16& o, int b) -> u32x4 {
    return u32x4{pack2bf(o[8 * b], o[8 * b + 1]), pack2bf(o[8 * b + 2], o[8 * b + 3]), pack2bf(o[8 * b + 4], o[8 * b + 5]), pack2bf(o[8 * b + 6], o[8 * b + 7])};
  };
  auto add_res = [&](f32x16& o, int t, int so) {
    const u32x4 r0 = __builtin_amdgcn_raw_buffer_load_b128(rs_r, vo_r[t], so, 0);
    const u32x4 r1 = __builtin_amdgcn_raw_buffer_load_b128(rs_r, vo_r[t] + 16, so, 0);
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      o[2 * e] += bflo(r0[e]); o[2 * e + 1] += bfhi(r0[e]);
      o[8 + 2 * e] += bflo(r1[e]); o[8 + 2 * e + 1] += bfhi(r1[e]);
    }
  };
  // plain: units 2 g and 2 g + 1 (channels 64 g .. 64 g + 63)
  auto store_pair = [&](int g, f32x16 (&v0)[RT], f32x16 (&v1)[RT]) {
#pragma unroll
    for (int t = 0; t < RT; ++t) {
      if (RES) { add_res(v0[t], t, g * 128); add_res(v1[t], t, g * 128 + 64); }
      stage_chunk(0, 2 * h, pack8(v0[t], 0)); stage_chunk(0, 2 * h + 1, pack8(v0[t], 1));
      stage_chunk(0, 4 + 2 * h, pack8(v1[t], 0)); stage_chunk(0, 4 + 2 * h + 1, pack8(v1[t], 1));
      flush_rows(t, 0, g * 128);
    }
    __builtin_amdgcn_sched_barrier(0);
  };
  // GEGLU: the 32-channel output tile `half` (0 / 1) of a 128-byte group goes to the staging area; the second one flushes both
  // row tiles as whole lines.  (Stored per tile as 64-byte row pieces the L2 evicted half-written lines: 553 MB of traffic per
  // launch for 419 algorithmic.)
  auto stage_tile = [&](int half, const f32x16 (&v)[RT]) {
#pragma unroll
    for (int t = 0; t < RT; ++t) { stage_chunk(t, 4 * half + 2 * h, pack8(v[t], 0)); stage_chunk(t, 4 * half + 2 * h + 1, pack8(v[t], 1)); }
    __builtin_amdgcn_sched_barrier(0);
  };
  auto flush_group = [&](int g) {
#pragma unroll
    for (int t = 0; t < RT; ++t) flush_rows(t, t, g * 128);
    __builtin_amdgcn_sched_barrier(0);
  };

#ifdef XS_STAMPS
  const long long st_prolog = (long long)__builtin_amdgcn_s_memtime() - st_begin;
  st_t = (long long)__builtin_amdgcn_s_memtime();
#endif
  int stage = 0;
  if constexpr (!GEGLU) {
    // two units per store group (host: an even number of units per part)
    for (int it = 0; it < nit; it += 2) {
      f32x16 acc0[RT], acc1[RT];
      top(it, stage);
      multiply(stage, acc0, gs, false);
      stage = stage == NS - 1 ? 0 : stage + 1;
      top(it + 1, stage);
      multiply(stage, acc1, gs, false);
      stage = stage == NS - 1 ? 0 : stage + 1;
      store_pair((u0 + it) >> 1, acc0, acc1);
    }
  } else {
    // four units = one 128-byte store group: gate | value of output tile 2 g, gate | value of tile 2 g + 1 (host: units per part % 4 == 0)
    for (int it = 0; it < nit; it += 2) {
      {   // gate unit: its raw accumulators wait in gs
        top(it, stage);
        f32x16 gg[RT];
        multiply(stage, gg, gs, false);
#pragma unroll
        for (int t = 0; t < RT; ++t)
#pragma unroll
          for (int q = 0; q < 16; ++q) gs[t * 16 + q] = gg[t][q];
        stage = stage == NS - 1 ? 0 : stage + 1;
      }
      {   // value unit (gelu(gate) under its MFMAs), then value * gelu(gate) -> one 32-channel output tile
        top(it + 1, stage);
        f32x16 acc[RT];
        multiply(stage, acc, gs, true);
#pragma unroll
        for (int t = 0; t < RT; ++t)
#pragma unroll
          for (int q = 0; q < 16; ++q) acc[t][q] *= gs[t * 16 + q];
        stage_tile((it >> 1) & 1, acc);
        if (it & 2) flush_group((u0 + it) >> 2);
        stage = stage == NS - 1 ? 0 : stage + 1;
      }
    }
  }
#ifdef XS_STAMPS
  XS_STAMP(4);
  if (wave == 0 && lane == 0) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    long long* dst = reinterpret_cast<long long*>(a.out + (size_t)a.M * a.ldo) + (size_t)blockIdx.x * 8;   // (behind row M: the probe allocates them)
    for (int i = 0; i < 5; ++i) dst[i] = st_acc[i];
    dst[5] = (long long)__builtin_amdgcn_s_memrealtime() - st_rbegin; dst[6] = (long long)__builtin_amdgcn_s_memtime() - st_begin; dst[7] = st_rbegin; (void)st_prolog;
  }
#endif
}

template <int KS, bool GEGLU, bool RES, bool LN>
int launch_xs(const MvdXsArgs& a, hipStream_t s) {
  constexpr int LDS_BYTES = (GEGLU ? 2 : 3) * (KS + 1) * 1024 + 4 * (GEGLU ? 8192 : 4096);     // ring + the per-wave store staging
  static bool init[16] = {};                            // per device
  int dev = 0;
  (void)hipGetDevice(&dev);
  if (!init[dev & 15]) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_xs_kernel<KS, GEGLU, RES, LN>), hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES);
    if (e != hipSuccess) { mvd_set_error("gemm_xs: hipFuncSetAttribute: %s", hipGetErrorString(e)); return -2; }
    init[dev & 15] = true;
    if (MVD_ENV_INT("MVD_XS_TRACE", 0)) {     // (probe builds only)
      int occ = -1;
      (void)hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, reinterpret_cast<const void*>(&gemm_xs_kernel<KS, GEGLU, RES, LN>), 256, LDS_BYTES);
      fprintf(stderr, "gemm_xs<%d,%d,%d,%d>: %d workgroups per CU by the occupancy query (LDS %d B)\n", KS, (int)GEGLU, (int)RES, (int)LN, occ, LDS_BYTES);
    }
  }
  const int nrb = (a.M + 255) / 256;
  const int grid = a.csplit == 1 ? nrb : ((nrb + 7) / 8) * 8 * a.csplit;
  g_mvd_last_gemm.cfg = 9; g_mvd_last_gemm.splitk = 1; g_mvd_last_gemm.tiles = nrb * a.csplit; g_mvd_last_gemm.grid = grid; g_mvd_last_gemm.per_cu = 2;
  hipLaunchKernelGGL((gemm_xs_kernel<KS, GEGLU, RES, LN>), dim3(grid), dim3(256), LDS_BYTES, s, a);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) { mvd_set_error("gemm_xs launch: %s", hipGetErrorString(e)); return -3; }
  return 0;
}

}  // namespace

bool mvd_gemm_xs_applicable(const MvdXsArgs& a) {
  const size_t lim = (size_t)1 << 31;
  if (a.K != 320 || a.units <= 0 || a.M <= 0) return false;
  if (a.units % (a.geglu ? 4 : 2)) return false;        // store groups: two units = 128 bytes per token (GEGLU: gate | value | gate | value)
  if ((a.geglu || a.ln) && a.res) return false;
  if ((size_t)(a.M + 256) * a.ldx * 2 >= lim || (size_t)(a.M + 256) * a.ldo * 2 >= lim) return false;
  if (a.res && (size_t)(a.M + 256) * a.ldres * 2 >= lim) return false;
  if ((size_t)a.units * 21 * 1024 >= lim) return false;
  if ((a.ldx & 7) || (a.ldo & 7) || (a.res && (a.ldres & 7))) return false;     // 16-byte accesses
  return true;
}

// column split: enough work items for two rounds of 512 resident workgroups, parts of whole (GEGLU: pairs of) units
int mvd_gemm_xs_pick_csplit(const MvdXsArgs& a) {
  const int nrb = (a.M + 255) / 256;
  const int groups = a.units / (a.geglu ? 4 : 2);
  int cs = 1;
  while (nrb * cs < 1024 && groups % (cs * 2) == 0 && groups / (cs * 2) >= 3) cs *= 2;
  return cs;
}

int mvd_launch_gemm_xs(const MvdXsArgs& a_in, hipStream_t s) {
  MvdXsArgs a = a_in;
  if (a.csplit <= 0) a.csplit = mvd_gemm_xs_pick_csplit(a);
  const int groups = a.units / (a.geglu ? 4 : 2);
  if (!mvd_gemm_xs_applicable(a) || groups % a.csplit) {
    mvd_set_error("gemm_xs: M=%d K=%d units=%d geglu=%d csplit=%d is not a shape of the X-stationary kernels", a.M, a.K, a.units, a.geglu, a.csplit);
    return -1;
  }
  if (a.geglu) return a.ln ? launch_xs<20, true, false, true>(a, s) : launch_xs<20, true, false, false>(a, s);
  if (a.res) return launch_xs<20, false, true, false>(a, s);
  return a.ln ? launch_xs<20, false, false, true>(a, s) : launch_xs<20, false, false, false>(a, s);
}

extern "C" int mvd_op_linear_xs(const void* x, int ldx, const void* w_packed, int m, int k, int units, int geglu, int ln, float ln_eps,
                                const void* res, int ldres, void* out, int ldo, int csplit, void* stream) {
  MvdXsArgs a; memset(&a, 0, sizeof(a));
  a.x = (const bf16_t*)x; a.ldx = ldx; a.w = (const bf16_t*)w_packed; a.M = m; a.K = k; a.units = units; a.geglu = geglu;
  a.ln = ln; a.ln_eps = ln_eps; a.res = (const bf16_t*)res; a.ldres = ldres; a.out = (bf16_t*)out; a.ldo = ldo; a.csplit = csplit;
  if (!x || !w_packed || !out) { mvd_set_error("mvd_op_linear_xs: null operand"); return -1; }
  return mvd_launch_gemm_xs(a, (hipStream_t)stream);
}
