// Flash-style attention for head_dim 64, bf16 in/out, fp32 softmax+accumulate (gfx950).
//
// One launch covers up to two independent problems (e.g. a BasicTransformerBlock's own
// attention and the MVD adapter's cross-view attention on the same query tokens).
// Layout: token-major with heads interleaved in the channel dim, q[b][n][head*64 + d],
// arbitrary row/batch strides so fused QKV GEMM outputs are consumed in place.
//
// Structure (per wave: 32 query rows; per workgroup: NW waves sharing K/V tiles in LDS):
//   S^T = K . Q^T   v_mfma_f32_32x32x16_bf16, A = K rows from LDS (ds_read_b128, XOR
//                   swizzle), B = Q fragments held in registers for the whole kernel.
//                   The result has the QUERY on the lane and half of the keys of the KV
//                   tile in the lane's registers -> row max is an in-register reduction
//                   plus one v_permlane32_swap with the partner half-wave.
//   O^T += V^T . P^T  the S^T accumulator (converted to bf16) is directly the B operand
//                   of the second MFMA (k-index permutation of the accumulator layout is
//                   matched on the V side); V^T fragments come from the row-major V tile
//                   through ds_read_b64_tr_b16 (hardware transpose read).
//   softmax         exp2 domain, raw v_exp_f32; generic form: scale folded into one v_fma; engine form (PRE): the scale
//                   lives in the packed to_q weights and -running_max is the C operand of the first MFMA.  The
//                   rescale of O is DEFERRED until some row's max grew by > 2^6 -- and in the engine form not looked for at
//                   all: the first tile's maximum stays, a non-finite denominator triggers a checked re-run (LAZY, below).
//                   Denominators: a V^T tile whose
//                   row 0 is all ones (matrix pipe), or -- engine form (VSUM) -- one 16x16x32 MFMA per P fragment against a 0/1 selector (4 registers).
// KV tiles hold NSUB x 32 keys: 64 by default.  The 128-key variant halves the per-tile fixed costs
// (barrier, max reduction tree, rescale test, loader address math) but loses a wave per SIMD and
// measured slower (profiles/r01_probe_attention_kv128.log); it is kept behind MVD_ATTN_KV128=1.  K/V tiles are double
// buffered in LDS; tile t+1 is fetched under the MFMAs of tile t: by LDS-DMA in the engine's form (DMA), through
// registers (global_load before the MFMAs, ds_write after them) in the others.
// The engine form is bound by the vector-issue port (per 64-key tile and wave: 32 v_exp_f32 at 8 issue cycles, 16 converts,
// 18 max, 20 MFMAs at 8 -- against 576 matrix cycles), so everything else was taken off that port: the score tiles start from
// the live -max tile without a copy (mfma_from), the DMA destinations are scalar, the two halves of a row exchange their
// maxima only inside the rare rescale branch, and the denominators are summed by the matrix pipe (113 -> 76 non-MFMA vector
// instructions per tile; 17.3 -> 16.0 ms per cfg4 step).
// The engine's kernel is attn_kernel<4, 2, PRE, DMA, VSUM>: 4 waves x 32 queries, 128 VGPRs, four workgroups per CU,
// workgroups dealt to the XCDs so that all query blocks of a (batch, head) pair share one L2 (attn_block).
#include <stdlib.h>
#include <type_traits>
#include "kernels.h"

namespace {

constexpr float NEG_BIG = -1.0e30f;
constexpr float RESCALE_LOG2 = 6.0f;   // defer the online-softmax rescale while P stays below 2^6

MVD_DEVINL int k_off(int key, int chunk) { return key * 128 + ((chunk ^ ((key >> 1) & 7)) << 4); }
// V swizzle keeps 64-byte halves intact for the 4x16 transposed reads
MVD_DEVINL int v_off(int key, int chunk) { return key * 128 + ((chunk ^ (((key >> 1) & 1) << 2)) << 4); }

// Exchange with the partner lane (lane ^ 32).  NOTE: __builtin_bit_cast applied directly to a
// vector element expression (r[1]) silently reads element 0 with hipcc/ROCm 7.2, so the
// elements are copied to scalars first.
MVD_DEVINL float pair_other(float x, float& own) {
  const unsigned a = __builtin_bit_cast(unsigned, x);
  auto r = __builtin_amdgcn_permlane32_swap(a, a, false, false);
  // lanes 0-31: r[0] = own, r[1] = partner ; lanes 32-63: r[0] = partner, r[1] = own
  const unsigned r0 = r[0], r1 = r[1];
  own = __builtin_bit_cast(float, r0);
  return __builtin_bit_cast(float, r1);
}
// D = A.B + C with C a register tile that STAYS LIVE (the -running_max tile that starts every score tile).  Through the builtin
// hipcc ties D to C and first copies C (8 v_mov_b64 per 32x32 tile, on the vector-issue port that bounds this kernel); here D
// is its own early-clobber tile.  The hazard recognizer does not see an MFMA in an asm statement, so the statement carries
// its own wait states for "VALU wrote C" (s_nop 1); operands that come from LDS are still covered by the compiler's
// s_waitcnt (register operands of an asm are tracked), and the next accumulate into D is the same opcode on exactly the
// same vDst (back-to-back dependent MFMAs need no software wait states).  The LAST MFMA into D is always a builtin, so the
// MFMA -> VALU read distance of the softmax is the compiler's.
MVD_DEVINL f32x16 mfma_from(bf16x8 a, bf16x8 b, const f32x16& c) {
  f32x16 d;
  asm("s_nop 1\n\tv_mfma_f32_32x32x16_bf16 %0, %1, %2, %3" : "=&v"(d) : "v"(a), "v"(b), "v"(c));
  return d;
}
MVD_DEVINL float pair_max(float x) { float o; const float p = pair_other(x, o); return fmaxf(o, p); }
MVD_DEVINL float pair_sum(float x) { float o; const float p = pair_other(x, o); return o + p; }

// NW waves x 32 queries per workgroup, KV tile = NSUB * 32 keys.
// 2nd launch-bound = waves per SIMD (register budget for the intended residency).
// PRE: Q arrives pre-multiplied by softmax_scale * log2(e) (folded into the to_q weights by the host packing), so
// the QK^T accumulator is already the exp2-domain score; it is STARTED at -running_max (C operand of the first
// MFMA), so the exponent argument leaves the matrix pipe ready-made and the per-score v_fma disappears.
// Workgroup -> (query block, head, batch * problem) on a 1-D grid, XCD aware: hardware deals consecutive workgroup ids
// round-robin to the 8 XCDs, so with the plain (query block fastest) order the 32 query blocks of one (batch, head) -- which
// all stream the SAME K/V -- land on 8 different L2s and every L2 fetches that K/V for itself (PMC: 0.8 GB of L2 fills per
// launch on average).  Here XCD x works through the (batch, head) pairs x, x + 8, ...: all query blocks of a pair share one L2.
MVD_DEVINL void attn_block(const MvdAttnArgs& a, int& qb, int& head, int& bz, int nsplit = 1, int* split = nullptr, int* pair_out = nullptr) {
  // (split-KV: a (pair, key range) takes the place of the pair -- the query blocks that stream the same K/V RANGE share an L2)
  const int npair = a.heads * a.batch * a.nprob * nsplit;
  const int nqb = gridDim.x / npair;
  const int lid = blockIdx.x;
  int pair;
#ifdef MVD_ATTN_PLAIN_ORDER            // (A/B builds: tools/build_variant.py plain -DMVD_ATTN_PLAIN_ORDER)
  if (false) {}
#else
  if ((npair & 7) == 0) { const int j = lid >> 3; qb = j % nqb; pair = (j / nqb) * 8 + (lid & 7); }
#endif
  else { qb = lid % nqb; pair = lid / nqb; }
  if (split) { *split = pair % nsplit; pair /= nsplit; }
  head = pair % a.heads;
  bz = pair / a.heads;
  if (pair_out) *pair_out = pair;
}

// DMA: K/V tiles go global -> LDS by buffer-addressed LDS-DMA (no VGPR round trip, no ds_write, 16 registers fewer; the XOR
// swizzles move to the source side; keys >= nk lie beyond num_records and arrive as zeros) instead of load + ds_write.
// VSUM (with DMA): the softmax denominators are summed by 16x16x32 selector MFMAs (4 registers) instead of the "ones" V^T tile --
// 12 accumulator registers less, which keeps the kernel at 128 registers: FOUR waves per SIMD.
// SPLIT (engine form only): split-KV, see MvdAttnArgs::nsplit.
template <int NW, int NSUB, bool PRE, bool DMA = false, bool VSUM = false, bool SPLIT = false>
// (SPLIT runs on a nearly empty chip -- that is why it exists -- so it is compiled for three waves per SIMD: at four, its merge
//  epilogue spilled two registers to scratch)
__global__ __launch_bounds__(64 * NW, SPLIT ? 3 : (VSUM ? 4 : ((NW == 4 && NSUB == 2) ? 3 : 2))) void attn_kernel(const MvdAttnArgs a) {
  static_assert(!SPLIT || (PRE && DMA && VSUM), "split-KV exists for the engine form");
  constexpr int NT = 64 * NW;
  constexpr int QB = 32 * NW;
  constexpr int KV_TILE = 32 * NSUB;
  constexpr int TILE_BYTES = KV_TILE * 128;         // keys x 64 dims x 2 B
  constexpr int LD_IT = (KV_TILE * 8) / NT;         // 16-byte chunks per thread per tile
  static_assert((KV_TILE * 8) % NT == 0, "tile must divide over threads");
  static_assert(!DMA || (NT / 8) % 16 == 0, "DMA rows of a lane share the swizzle pattern");
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];         // 4 * TILE_BYTES: K0 K1 V0 V1

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int lq = lane & 31, lh = lane >> 5;
  int qb_, head, bz, sp = 0, pair = 0;
  const int nsplit = SPLIT ? a.nsplit : 1;
  if constexpr (SPLIT) attn_block(a, qb_, head, bz, nsplit, &sp, &pair);
  else attn_block(a, qb_, head, bz);
  const int pi = bz / a.batch;
  bz -= pi * a.batch;
  const MvdAttnProblem& P = a.p[pi];
  const int nq = P.nq, nk = P.nk;
  const int qblk0 = qb_ * QB;
  if (qblk0 >= nq) return;  // whole workgroup exits together (uniform)

  const bf16_t* qp = P.q + (size_t)bz * P.bsq + head * 64;
  const bf16_t* kp = P.k + (size_t)bz * P.bsk + head * 64;
  const bf16_t* vp = P.v + (size_t)bz * P.bsv + head * 64;
  bf16_t* op = P.o + (size_t)bz * P.bso + head * 64;

  // Q fragments: B operand of S^T = K.Q^T : lane (query lq, half lh) holds Q[q][16*ks + 8*lh .. +7]
  const int qrow = qblk0 + wave * 32 + lq;
  const int qrow_c = qrow < nq ? qrow : nq - 1;
  bf16x8 qf[4];
  // (Loading Q coalesced -- [8 rows x 128 B] per instruction through the still empty stage-1 buffers, read back as fragments --
  //  was measured in round 4: 516.8 vs 517.5 fwd/s, attention class 15.68 vs 15.72 ms: nothing; four loads per workgroup pass
  //  are not what the O stores were.)
#pragma unroll
  for (int ks = 0; ks < 4; ++ks)
    qf[ks] = *reinterpret_cast<const bf16x8*>(qp + (size_t)qrow_c * P.ldq + ks * 16 + lh * 8);

  // loader mapping: chunk id -> (key row, 16-byte chunk); one thread's chunks are NT/8 rows apart; when that is a
  // multiple of 16 they share the swizzle pattern and their LDS offsets differ by constants
  const int ld_kc = tid & 7;
  const int ld_row = tid >> 3;
  const int ld_koff = k_off(ld_row, ld_kc), ld_voff = v_off(ld_row, ld_kc);
  // Loads use a UNIFORM 64-bit base (scalar registers, advanced per tile) plus a loop-invariant 32-bit per-lane byte
  // offset, so a full tile costs no vector address arithmetic; the ragged last tile recomputes the offsets with the
  // row index clamped to nk-1 (those keys' scores are masked, so P = 0 meets a finite V row).
  const int ldk = P.ldk, ldv = P.ldv;
  const unsigned ld_ko = ((unsigned)ld_row * (unsigned)ldk + ld_kc * 8) * 2u;
  const unsigned ld_vo = ((unsigned)ld_row * (unsigned)ldv + ld_kc * 8) * 2u;
  const char* kp_b = reinterpret_cast<const char*>(kp);
  const char* vp_b = reinterpret_cast<const char*>(vp);
  typedef __attribute__((address_space(3))) void lds_void_t;
  __amdgpu_buffer_rsrc_t rs_k = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_t*>(kp), 0, (nk - 1) * ldk * 2 + 128, 0x00020000);
  __amdgpu_buffer_rsrc_t rs_v = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_t*>(vp), 0, (nk - 1) * ldv * 2 + 128, 0x00020000);
  const unsigned dma_ko = (unsigned)ld_row * (unsigned)ldk * 2u + ((ld_kc ^ ((ld_row >> 1) & 7)) << 4);
  const unsigned dma_vo = (unsigned)ld_row * (unsigned)ldv * 2u + ((ld_kc ^ (((ld_row >> 1) & 1) << 2)) << 4);
  auto dma_tile = [&](int kb, int st) {            // tile kb -> stage st (lane-linear LDS image, swizzled source chunk)
    // (the wave index as a SCALAR: the LDS destination of a DMA goes through M0, and a vector-derived address costs a v_add +
    //  v_readfirstlane per instruction on the vector-issue port this kernel is bound by)
    const int wave_s = __builtin_amdgcn_readfirstlane(wave);
    unsigned char* dk = smem + st * TILE_BYTES + wave_s * 1024;
    unsigned char* dv = smem + (2 + st) * TILE_BYTES + wave_s * 1024;
#pragma unroll
    for (int i = 0; i < LD_IT; ++i) {
      const int row0 = kb * KV_TILE + i * (NT / 8);
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_k, (lds_void_t*)(dk + i * (NT / 8) * 128), 16, (int)dma_ko, row0 * ldk * 2, 0, 0);
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_v, (lds_void_t*)(dv + i * (NT / 8) * 128), 16, (int)dma_vo, row0 * ldv * 2, 0, 0);
    }
  };
  u32x4 rk[DMA ? 1 : LD_IT], rv[DMA ? 1 : LD_IT];
  auto load_tile = [&](int kb) {
    if constexpr (DMA) return;
    const int k0 = kb * KV_TILE;
    const bool full = k0 + KV_TILE <= nk;          // uniform
#pragma unroll
    for (int i = 0; i < LD_IT; ++i) {
      const char* kb_s = kp_b;
      const char* vb_s = vp_b;
      unsigned ko, vo;
      if (full) {
        kb_s += (size_t)(k0 + i * (NT / 8)) * ldk * 2;
        vb_s += (size_t)(k0 + i * (NT / 8)) * ldv * 2;
        ko = ld_ko; vo = ld_vo;
      } else {
        int key = k0 + ld_row + i * (NT / 8);
        key = key < nk ? key : nk - 1;
        asm volatile("" : "+v"(key));              // keep this arithmetic inside the ragged branch
        ko = ((unsigned)key * (unsigned)ldk + ld_kc * 8) * 2u;
        vo = ((unsigned)key * (unsigned)ldv + ld_kc * 8) * 2u;
      }
      rk[i] = *reinterpret_cast<const u32x4*>(kb_s + ko);
      rv[i] = *reinterpret_cast<const u32x4*>(vb_s + vo);
    }
  };
  auto store_tile = [&](int st) {
    if constexpr (DMA) return;
    unsigned char* sk = smem + st * TILE_BYTES;
    unsigned char* sv = smem + (2 + st) * TILE_BYTES;
#pragma unroll
    for (int i = 0; i < LD_IT; ++i) {
      if constexpr ((NT / 8) % 16 == 0) {
        *reinterpret_cast<u32x4*>(sk + ld_koff + i * (NT / 8) * 128) = rk[i];
        *reinterpret_cast<u32x4*>(sv + ld_voff + i * (NT / 8) * 128) = rv[i];
      } else {   // single-wave workgroups: rows 8 apart change the K swizzle
        *reinterpret_cast<u32x4*>(sk + k_off(ld_row + i * (NT / 8), ld_kc)) = rk[i];
        *reinterpret_cast<u32x4*>(sv + v_off(ld_row + i * (NT / 8), ld_kc)) = rv[i];
      }
    }
  };

  f32x16 o0 = {}, o1 = {};       // O^T tiles: d 0..31 and 32..63 (rows) x query (lane)
  f32x16 ol = {};                // "ones" tile: row 0 (reg 0 of lanes 0..31) = running softmax denominators
#ifdef MVD_ATTN_SUM_ADDS
  float lsum = 0.f;              // VSUM: this lane's share (its half of the keys) of the denominator of query lq
  float lsum2 = 0.f;
#endif
  // VSUM: the softmax denominators on the matrix pipe at FOUR registers: one 16x16x32 MFMA per packed P fragment whose A
  // operand is a 0/1 selector.  As the B operand of that shape, lane l supplies k-group l >> 4 of column l & 15, so the
  // fragments of query n (lanes n, n + 32: its two key halves) are k-groups 0 and 2 of column n and those of query n + 16
  // (lanes n + 16, n + 48) are k-groups 1 and 3: selector row 0 = ones over k-groups {0, 2}, row 1 = ones over {1, 3}, other
  // rows zero.  Lanes 0..15 then hold the COMPLETE denominators (both key halves) of query n in lacc[0] and of query
  // n + 16 in lacc[1] -- of the bf16-rounded P the numerators use.  4 MFMAs (32 issue cycles, 64 matrix cycles) per tile
  // instead of 32 v_add_f32 (128 issue cycles) on the vector-issue port this kernel is bound by.
  f32x4 lacc = {0.f, 0.f, 0.f, 0.f};
  bf16x8 sel;
  {
    const int sm = lane & 15, sg = (lane >> 4) & 1;
    const __bf16 one = (sm == sg && sm < 2) ? (__bf16)1.0f : (__bf16)0.0f;
#pragma unroll
    for (int j = 0; j < 8; ++j) sel[j] = one;
  }
  bf16x8 ones_frag;
#pragma unroll
  for (int j = 0; j < 8; ++j) ones_frag[j] = (lq == 0) ? (__bf16)1.0f : (__bf16)0.0f;
  float m_run = PRE ? 0.f : NEG_BIG;
  const float c = a.scale * 1.4426950408889634f;  // scale * log2(e)   (unused when PRE)
  f32x16 negm = {};              // PRE: -m_run in every register (C operand that starts each score tile)

  const int nkb_all = (nk + KV_TILE - 1) / KV_TILE;
  const int kb0 = SPLIT ? (sp * nkb_all) / nsplit : 0;                 // this workgroup's key tiles [kb0, nkb)
  const int nkb = SPLIT ? ((sp + 1) * nkb_all) / nsplit : nkb_all;

  // transposed-read lane geometry (see header): 16-lane group g reads a 4-key x 16-dim block
  const int tr_i = lane & 15;
  const int tr_q = tr_i >> 2, tr_p = tr_i & 3;
  const int tr_dcol = ((lane >> 4) & 1) * 16 + tr_p * 4;   // dim offset inside a 32-dim tile
  // Loop-invariant LDS byte offsets.  The V swizzle bit depends on (key>>1)&1 only, which is unchanged by
  // +16*st and +8, so the two d tiles differ by an XOR of 64 bytes and everything else is an immediate.
  const int tr_base0 = v_off(4 * lh + tr_q, tr_dcol >> 3) + (tr_dcol & 7) * 2;
  const int tr_base1 = tr_base0 ^ 64;
  int k_base[4];   // K row lq, chunk (2*ks + lh) ^ swizzle(lq); rows lq+32*t share the swizzle
#pragma unroll
  for (int ks = 0; ks < 4; ++ks) k_base[ks] = k_off(lq, ks * 2 + lh);

  // LAZY (engine form): the running max is set by the FIRST tile and then left alone -- no per-tile row maximum (16 v_max3 per
  // 32 x 32 scores), no vote, no rescale branch.  Nothing in online softmax needs m to be the true maximum: O and l are both
  // scaled by 2^(m_true - m), which cancels in O / l; what it needs is that 2^(s - m) neither overflows (s - m < 128: P is bf16,
  // the sums fp32) nor that every term underflows (the first tile's maximum contributes exactly 1).  A later score more than
  // ~100 above the first tile's maximum -- 69 nats; never seen -- makes a denominator non-finite: detected behind the loop,
  // and the workgroup then re-runs its tiles with the checked (deferred-rescale) loop.  CHK = checked.
#ifdef MVD_ATTN_NO_LAZY                 // (A/B builds)
  constexpr bool LAZY = false;
#else
  constexpr bool LAZY = PRE && DMA && VSUM;
#endif
  auto run_tiles = [&](auto CHK) {
  constexpr bool chk = decltype(CHK)::value;
  o0 = f32x16{}; o1 = f32x16{}; ol = f32x16{}; lacc = f32x4{0.f, 0.f, 0.f, 0.f};
  m_run = PRE ? 0.f : NEG_BIG; negm = f32x16{};
  if constexpr (DMA) { dma_tile(kb0, 0); asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
  load_tile(kb0);
  store_tile(0);
  __syncthreads();
  for (int kb = kb0; kb < nkb; ++kb) {
    const int cur = (kb - kb0) & 1;
    const bool more = kb + 1 < nkb;
    if (more) { if constexpr (DMA) dma_tile(kb + 1, cur ^ 1); else load_tile(kb + 1); }   // (stage cur^1: last read before the previous barrier)
    const unsigned char* sk = smem + cur * TILE_BYTES;
    const unsigned char* sv = smem + (2 + cur) * TILE_BYTES;

    // ---- S^T = K.Q^T for the NSUB 32-key sub tiles
    f32x16 s[NSUB];
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
#pragma unroll
      for (int t = 0; t < NSUB; ++t) {
        const bf16x8 kf = *reinterpret_cast<const bf16x8*>(sk + k_base[ks] + t * 32 * 128);
        if (ks == 0) {
          if constexpr (PRE && VSUM) s[t] = mfma_from(kf, qf[0], negm);
          else s[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf, qf[0], PRE ? negm : f32x16{}, 0, 0, 0);
        } else {
          s[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf, qf[ks], s[t], 0, 0, 0);
        }
      }
    }

    // ---- mask keys beyond nk (only the last tile can be ragged)
    if (kb * KV_TILE + KV_TILE > nk) {
#pragma unroll
      for (int t = 0; t < NSUB; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int key = kb * KV_TILE + t * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
          if (key >= nk) s[t][r] = NEG_BIG;
        }
    }
    // ---- online softmax (this lane: one query, half of the tile's keys; partner lane^32 has the rest)
    float mx = 0.f;
    if (chk || kb == kb0) {
      mx = s[0][0];
#pragma unroll
      for (int t = 0; t < NSUB; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) mx = fmaxf(mx, s[t][r]);
    }
    // (the vote below runs over all 64 lanes, so it needs no exchange with the partner half-wave: the engine form combines
    //  the two halves of a row only inside the rare rescale branch)
    if constexpr (!(PRE && VSUM)) mx = pair_max(mx);
    // Deferred rescale: the running max is only raised (and O / the row sums rescaled) when some row's max
    // grew by more than RESCALE_LOG2 in the exp2 domain; otherwise P is taken against the old max and is bounded
    // by 2^RESCALE_LOG2 (bf16 keeps its relative precision, the accumulators are fp32).  Wave-uniform branch.
    if constexpr (PRE) {
      // s holds score - m_run (exp2 domain).  The first tile always sets the running max (m_run starts at 0, so a row
      // whose scores are all far below zero would otherwise underflow every P).
      if (kb == kb0 || (chk && !__all(mx <= RESCALE_LOG2))) {
        if constexpr (VSUM) mx = pair_max(mx);
        const float delta = kb == kb0 ? mx : fmaxf(mx, 0.f);     // m_new - m_run
        if (kb != kb0) {
          const float alpha = __builtin_amdgcn_exp2f(-delta);
#pragma unroll
          for (int r = 0; r < 16; ++r) { o0[r] *= alpha; o1[r] *= alpha; }
          ol[0] *= alpha;
#ifdef MVD_ATTN_SUM_ADDS
          lsum *= alpha;
          lsum2 *= alpha;
#else
          if constexpr (VSUM) {          // holder lane n: query n's factor is its own, query n + 16's is lane n + 16's
            lacc[0] *= alpha;
            lacc[1] *= __shfl(alpha, (lane + 16) & 63, 64);
          }
#endif
        }
        m_run += delta;
#pragma unroll
        for (int r = 0; r < 16; ++r) negm[r] = -m_run;
#pragma unroll
        for (int t = 0; t < NSUB; ++t)
#pragma unroll
          for (int r = 0; r < 16; ++r) s[t][r] -= delta;
      }
#pragma unroll
      for (int t = 0; t < NSUB; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) s[t][r] = __builtin_amdgcn_exp2f(s[t][r]);
    } else {
      if (!__all((mx - m_run) * c <= RESCALE_LOG2)) {
        const float m_new = fmaxf(m_run, mx);
        const float alpha = __builtin_amdgcn_exp2f((m_run - m_new) * c);
        m_run = m_new;
#pragma unroll
        for (int r = 0; r < 16; ++r) { o0[r] *= alpha; o1[r] *= alpha; }
        ol[0] *= alpha;                       // row sums live in row 0 of the "ones" tile
      }
      const float mc = m_run * c;
#pragma unroll
      for (int t = 0; t < NSUB; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) s[t][r] = __builtin_amdgcn_exp2f(fmaf(s[t][r], c, -mc));   // raw v_exp_f32
    }

    // ---- O^T += V^T . P^T, 16 keys (one k-step) at a time.  P^T as bf16 B operand: k-step st of sub tile t =
    //      accumulator regs 8*(st&1).. of s[st>>1]; A operand element j of lane (d, h) = V[16*st + 8*(j>>2) + 4h + (j&3)][d]
#pragma unroll
    for (int st = 0; st < 2 * NSUB; ++st) {
      bf16x8 pb;
#pragma unroll
      for (int j = 0; j < 8; ++j) pb[j] = (__bf16)s[st >> 1][8 * (st & 1) + j];
#pragma unroll
      for (int dt = 0; dt < 2; ++dt) {
        const int offA = (dt == 0 ? tr_base0 : tr_base1) + st * 16 * 128;   // keys 16*st + 4*lh + tr_q (+0..3)
        const int offB = offA + 8 * 128;                                     // keys + 8
        const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
            (__attribute__((address_space(3))) s16x4*)(sv + offA));
        const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
            (__attribute__((address_space(3))) s16x4*)(sv + offB));
        typedef __attribute__((ext_vector_type(8))) short s16x8;
        const s16x8 both = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
        const bf16x8 vf = __builtin_bit_cast(bf16x8, both);
        if (dt == 0) o0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf, pb, o0, 0, 0, 0);
        else         o1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf, pb, o1, 0, 0, 0);
      }
      // row sums: engine form (VSUM) -- the selector MFMA (see lacc above; measured against the alternatives on the vector
      // pipe: 16 v_dot2c_f32_bf16 on the packed P 16.44 ms/step, 32 plain v_add_f32 on two chains 16.30 (MVD_ATTN_SUM_ADDS
      // builds), this 16.00); otherwise a V^T tile whose row 0 is all ones accumulates sum_k P[k][q] into ol[0] (16 registers)
      if constexpr (VSUM) {
#ifdef MVD_ATTN_SUM_ADDS
#pragma unroll
        for (int j = 0; j < 8; j += 2) {
          asm("v_add_f32 %0, %0, %1" : "+v"(lsum) : "v"(s[st >> 1][8 * (st & 1) + j]));
          asm("v_add_f32 %0, %0, %1" : "+v"(lsum2) : "v"(s[st >> 1][8 * (st & 1) + j + 1]));
        }
#else
        lacc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(sel, pb, lacc, 0, 0, 0);
#endif
      } else {
        ol = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ones_frag, pb, ol, 0, 0, 0);
      }
    }
    if (more) store_tile(cur ^ 1);
    if constexpr (DMA) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
  }
  };   // run_tiles
  if constexpr (LAZY) {
    run_tiles(std::false_type{});
    // lanes 0..15 hold the complete denominators of queries n and n + 16 (see lacc): anything beyond 2^64 -> redo checked.  (The
    // bound also keeps the UNCHECKED accumulators finite: |O| <= l max|v| < 2^64 max|v|, so O overflows fp32 only for |v| > 2^63;
    // with the former bound of 1e30 a value of 1e9 behind a score 2^90 above the first tile's maximum gave l = 2^90 -- accepted --
    // and O = inf.  ADVICE r3.)
    const int bad = (lane < 16 && !(lacc[0] < 1.8446744e19f && lacc[1] < 1.8446744e19f)) ? 1 : 0;
    if (__syncthreads_or(bad)) run_tiles(std::true_type{});
  } else {
    run_tiles(std::true_type{});
  }

  // ---- epilogue: O[q][d] = O^T / l ; lane holds d = 32*dt + (r&3) + 8*(r>>2) + 4*lh
#ifdef MVD_ATTN_SUM_ADDS
  const float inv = 1.0f / pair_sum(VSUM ? lsum + lsum2 : ol[0]);   // lanes 32..63 hold row 4 of the ones tile (= 0) in ol[0]
#else
  float inv;
  if constexpr (VSUM) {
    const float d0 = __shfl(lacc[0], lq & 15, 64), d1 = __shfl(lacc[1], lq & 15, 64);
    inv = 1.0f / (lq < 16 ? d0 : d1);
  } else {
    inv = 1.0f / pair_sum(ol[0]);        // lanes 32..63 hold row 4 of the ones tile (= 0) in ol[0]
  }
#endif
  if constexpr (SPLIT) {
    if (nsplit > 1) {
      // ---- partial result of this key range -> workspace (write-through): [pair][query block][split][QB queries] rows of 64 bf16
      // (normalised by THIS range's denominator) + (running max, denominator) per query; then one ticket per workgroup on the
      // (pair, query block) counter -- the workgroup that draws the last one merges:
      //   out = sum_s w_s O_s / sum_s w_s,  w_s = l_s 2^(m_s - max m)      (exp2 domain: the scores carry log2 e)
      const int nqb = gridDim.x / (a.heads * a.batch * a.nprob * nsplit);
      const size_t grp = (size_t)pair * nqb + qb_;                         // (pair, query block)
      bf16_t* wo = reinterpret_cast<bf16_t*>(a.split_ws) + (grp * nsplit + sp) * (size_t)(QB * 64);
      float2* wml = reinterpret_cast<float2*>(reinterpret_cast<bf16_t*>(a.split_ws) + (size_t)gridDim.x * (QB * 64)) + (grp * nsplit + sp) * QB;
      __amdgpu_buffer_rsrc_t rs_o = __builtin_amdgcn_make_buffer_rsrc(wo, 0, QB * 128, 0x00020000);
      const int ql = wave * 32 + lq;                                       // query inside the block
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const u32x2 w0 = {pack2bf(o0[4 * g] * inv, o0[4 * g + 1] * inv), pack2bf(o0[4 * g + 2] * inv, o0[4 * g + 3] * inv)};
        const u32x2 w1 = {pack2bf(o1[4 * g] * inv, o1[4 * g + 1] * inv), pack2bf(o1[4 * g + 2] * inv, o1[4 * g + 3] * inv)};
        __builtin_amdgcn_raw_buffer_store_b64(w0, rs_o, ql * 128 + (8 * g + 4 * lh) * 2, 0, 16);          // aux 16 = sc1
        __builtin_amdgcn_raw_buffer_store_b64(w1, rs_o, ql * 128 + (32 + 8 * g + 4 * lh) * 2, 0, 16);
      }
      if (lh == 0) {
        typedef __attribute__((address_space(1))) unsigned long long gu64;
        const float l = 1.0f / inv;
        const unsigned long long pk = (unsigned long long)__builtin_bit_cast(unsigned, m_run) | ((unsigned long long)__builtin_bit_cast(unsigned, l) << 32);
        __hip_atomic_store((gu64*)(wml + ql), pk, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();
      int* flag = reinterpret_cast<int*>(smem);
      if (tid == 0) {
        typedef __attribute__((address_space(1))) unsigned int gu32;
        *flag = (int)__hip_atomic_fetch_add((gu32*)(a.split_cnt + grp), 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
      __syncthreads();
      if (*flag != nsplit - 1) return;
      // ---- merge (the last workgroup to arrive): thread t -> query t >> 1, dims 32 (t & 1) .. + 31
      const int mq = tid >> 1, mh = tid & 1;
      const int mrow = qblk0 + mq;
      const bf16_t* wo0 = reinterpret_cast<const bf16_t*>(a.split_ws) + grp * nsplit * (size_t)(QB * 64);
      const float2* wml0 = reinterpret_cast<const float2*>(reinterpret_cast<const bf16_t*>(a.split_ws) + (size_t)gridDim.x * (QB * 64)) + grp * nsplit * QB;
      __amdgpu_buffer_rsrc_t rs_i = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_t*>(wo0), 0, nsplit * QB * 128, 0x00020000);
      typedef __attribute__((address_space(1))) unsigned long long gu64c;
      float mmax = -3.0e38f;
      float ms[8], ls[8];
#pragma unroll
      for (int s2 = 0; s2 < 8; ++s2) {
        ms[s2] = -3.0e38f; ls[s2] = 0.f;
        if (s2 < nsplit) {
          const unsigned long long pk = __hip_atomic_load((gu64c*)(wml0 + s2 * QB + mq), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          ms[s2] = __builtin_bit_cast(float, (unsigned)pk); ls[s2] = __builtin_bit_cast(float, (unsigned)(pk >> 32));
          mmax = fmaxf(mmax, ms[s2]);
        }
      }
      float accv[32];
#pragma unroll
      for (int j = 0; j < 32; ++j) accv[j] = 0.f;
      float lt = 0.f;
#pragma unroll
      for (int s2 = 0; s2 < 8; ++s2) {
        if (s2 < nsplit) {
          const float wgt = ls[s2] * __builtin_amdgcn_exp2f(ms[s2] - mmax);
          lt += wgt;
#pragma unroll
          for (int c4 = 0; c4 < 4; ++c4) {
            const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(rs_i, (s2 * QB + mq) * 128 + mh * 64 + c4 * 16, 0, 16);   // sc1
            const unsigned dw[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
            for (int e = 0; e < 4; ++e) {
              accv[c4 * 8 + 2 * e] = fmaf(wgt, bflo(dw[e]), accv[c4 * 8 + 2 * e]);
              accv[c4 * 8 + 2 * e + 1] = fmaf(wgt, bfhi(dw[e]), accv[c4 * 8 + 2 * e + 1]);
            }
          }
        }
      }
      if (mrow < nq) {
        const float il = 1.0f / lt;
        bf16_t* orow = op + (size_t)mrow * P.ldo + mh * 32;
#pragma unroll
        for (int c4 = 0; c4 < 4; ++c4) {
          const u32x4 o = {pack2bf(accv[c4 * 8] * il, accv[c4 * 8 + 1] * il), pack2bf(accv[c4 * 8 + 2] * il, accv[c4 * 8 + 3] * il),
                           pack2bf(accv[c4 * 8 + 4] * il, accv[c4 * 8 + 5] * il), pack2bf(accv[c4 * 8 + 6] * il, accv[c4 * 8 + 7] * il)};
          *reinterpret_cast<u32x2*>(orow + c4 * 8) = u32x2{o.x, o.y};
          *reinterpret_cast<u32x2*>(orow + c4 * 8 + 4) = u32x2{o.z, o.w};
        }
      }
      return;
    }
  }
#ifndef MVD_ATTN_ROW_STORES      // (A/B builds: the round-3 epilogue)
  if constexpr (DMA && TILE_BYTES * 4 >= NW * 4096) {
    // A lane holds 16 pieces of 4 dims of ITS query's row: stored as they stand, each of 8 store instructions touches 32 rows x 2
    // pieces of 8 bytes -- and a vector-memory instruction's issue cost grows with the lines it touches (~460 cycles per such
    // store with eight waves storing, which also holds up the K/V LDS-DMAs of the workgroups still in their loops; measured on
    // the X-stationary GEMM).  An output row of one head is exactly one 128-byte line: the wave writes its
    // 32 x 128 bytes into the (now dead: the loop ends behind a barrier) K/V stages, 16-byte chunk c of row r at c ^ (r & 7), reads
    // [8 rows x 128 B] per instruction and stores WHOLE lines: 4 instructions of 8 lines instead of 8 of 32.
    // (the kernel sits at 128 registers: every address of this epilogue is derived HERE from an opaque copy of the lane id, so
    //  that nothing of it is hoisted above the tile loop and spilled)
    int el = (int)__builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u));
    asm volatile("" : "+v"(el));
    const int elq = el & 31;
    int etid = tid;
    asm volatile("" : "+v"(etid));
    const int ew = __builtin_amdgcn_readfirstlane(etid >> 6);          // the wave index again, as a scalar, derived here
    const int wbase = ew * 4096 + elq * 128 + 8 * (el >> 5), wswz = (elq & 7) << 4;
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const u32x2 w0 = {pack2bf(o0[4 * g] * inv, o0[4 * g + 1] * inv), pack2bf(o0[4 * g + 2] * inv, o0[4 * g + 3] * inv)};
      const u32x2 w1 = {pack2bf(o1[4 * g] * inv, o1[4 * g + 1] * inv), pack2bf(o1[4 * g + 2] * inv, o1[4 * g + 3] * inv)};
      *reinterpret_cast<u32x2*>(smem + wbase + ((g << 4) ^ wswz)) = w0;              // dims 8 g + 4 lh .. + 3
      *reinterpret_cast<u32x2*>(smem + wbase + (((4 + g) << 4) ^ wswz)) = w1;        // dims 32 + 8 g + 4 lh ..
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");     // (LDS operations of one wave execute in order; nobody else touches this 4 KB)
    const int rr = el >> 3, rc = el & 7;
    const int rbase = ew * 4096 + rr * 128 + ((rc ^ rr) << 4);                        // (row & 7 == rr for every j)
    const int q0 = qblk0 + ew * 32 + rr;
    bf16_t* orow = op + (size_t)q0 * P.ldo + rc * 8;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const u32x4 v = *reinterpret_cast<const u32x4*>(smem + rbase + j * 1024);
      if (q0 + 8 * j < nq) *reinterpret_cast<u32x4*>(orow + (size_t)(8 * j) * P.ldo) = v;
    }
    return;
  }
#endif
  if (qrow < nq) {
    bf16_t* orow = op + (size_t)qrow * P.ldo;
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      u32x2 w0 = {pack2bf(o0[4 * g] * inv, o0[4 * g + 1] * inv), pack2bf(o0[4 * g + 2] * inv, o0[4 * g + 3] * inv)};
      u32x2 w1 = {pack2bf(o1[4 * g] * inv, o1[4 * g + 1] * inv), pack2bf(o1[4 * g + 2] * inv, o1[4 * g + 3] * inv)};
      *reinterpret_cast<u32x2*>(orow + 8 * g + 4 * lh) = w0;
      *reinterpret_cast<u32x2*>(orow + 32 + 8 * g + 4 * lh) = w1;
    }
  }
}

#ifdef MVD_PROBE   // measurement-only kernels: probe builds only (tools/build_variant.py <tag> -DMVD_PROBE)
#include "probe/attention_probe.inc"
#endif

thread_local int g_last_attn[2] = {0, 0};
static int g_attn_pipe_override = 0;
static int g_attn_q64_override = 0;     // measurement hook (probe builds): 64 queries per wave, 4 (2) or 2 (1) waves per workgroup

#ifdef MVD_PROBE
#include "probe/attention_probe_launch.inc"
#endif

template <int NW, int NSUB>
int launch_nw(const MvdAttnArgs& a, int maxq, hipStream_t s) {
  const int qb = 32 * NW;
  if (a.nsplit > 1) {   // split-KV: the engine form at 4 waves only (mvd_launch_attention has checked the arguments)
    if constexpr (NW == 4 && NSUB == 2) {
      dim3 grid(((maxq + qb - 1) / qb) * a.heads * a.batch * a.nprob * a.nsplit);
      g_last_attn[0] = NW; g_last_attn[1] = (int)grid.x;
      hipLaunchKernelGGL((attn_kernel<4, 2, true, true, true, true>), grid, dim3(256), 4 * 32 * 2 * 128, s, a);
      hipError_t e = hipGetLastError();
      if (e != hipSuccess) { mvd_set_error("attention (split-KV) launch: %s", hipGetErrorString(e)); return -3; }
      return 0;
    } else { mvd_set_error("attention: split-KV needs the 4-wave form"); return -1; }
  }
  dim3 grid(((maxq + qb - 1) / qb) * a.heads * a.batch * a.nprob);      // 1-D: attn_block() decodes it
  // (the software-pipelined kernel is an experiment switch: at two waves per SIMD it measured 13 % SLOWER than the
  //  three-wave kernel above -- inter-wave overlap beats the intra-wave pipeline hipcc schedules; MVD_ATTN_PIPE=1)
  g_last_attn[0] = NW; g_last_attn[1] = (int)grid.x;
#ifdef MVD_PROBE
  static const int pipe_env = MVD_ENV_INT("MVD_ATTN_PIPE", 0);
  if (a.prescaled && NW == 4 && NSUB == 2 && (pipe_env || g_attn_pipe_override)) {
    hipLaunchKernelGGL((attn_pipe_kernel<4>), grid, dim3(256), 0, s, a);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) { mvd_set_error("attention launch: %s", hipGetErrorString(e)); return -3; }
    return 0;
  }
#endif
  if (a.prescaled) {
#ifdef MVD_ATTN_NO_DMA          // (A/B builds)
    constexpr bool dma = false;
#else
    constexpr bool dma = NW == 4 && NSUB == 2;
#endif
#ifdef MVD_ATTN_NO_VSUM         // (A/B builds)
    constexpr bool vsum = false;
#else
    constexpr bool vsum = dma;
#endif
    hipLaunchKernelGGL((attn_kernel<NW, NSUB, true, dma, vsum>), grid, dim3(64 * NW), 4 * 32 * NSUB * 128, s, a);
  } else hipLaunchKernelGGL((attn_kernel<NW, NSUB, false>), grid, dim3(64 * NW), 4 * 32 * NSUB * 128, s, a);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) { mvd_set_error("attention launch: %s", hipGetErrorString(e)); return -3; }
  return 0;
}

}  // namespace

int mvd_attention_pick_nw(const MvdAttnArgs& a);
static int g_attn_nw_override = -1;
// measurement hook (tools/, bench.py --attn-nw): log2 of the waves per workgroup for every later launch, -1 = heuristic
// (nw_log2 + 16: additionally the software-pipelined 4-wave variant)
extern "C" int mvd_debug_set_attention_nw(int nw_log2) {
  g_attn_q64_override = nw_log2 >= 32 ? nw_log2 - 32 : 0;
  if (nw_log2 >= 32) nw_log2 = 2;
  g_attn_pipe_override = nw_log2 >= 16;
  g_attn_nw_override = nw_log2 >= 16 ? nw_log2 - 16 : nw_log2;
  return 0;
}

// out[2] = {waves per workgroup, workgroups} of the calling thread's last attention launch
extern "C" int mvd_debug_last_attention_plan(int* out) {
  if (!out) { mvd_set_error("last_attention_plan: null argument"); return -1; }
  out[0] = g_last_attn[0]; out[1] = g_last_attn[1];
  return 0;
}

int mvd_launch_attention(const MvdAttnArgs& a, hipStream_t s) {
  if (a.nprob < 1 || a.nprob > 2 || a.batch <= 0 || a.heads <= 0) { mvd_set_error("attention: bad problem count/batch/heads"); return -1; }
  int maxq = 0, mink = 1 << 30;
  for (int i = 0; i < a.nprob; ++i) {
    const MvdAttnProblem& p = a.p[i];
    if (!p.q || !p.k || !p.v || !p.o || p.nq <= 0 || p.nk <= 0) { mvd_set_error("attention: null pointer or empty problem %d", i); return -1; }
    if ((p.ldq % 8) || (p.ldk % 8) || (p.ldv % 8) || (p.ldo % 4) || p.ldq < a.heads * 64 || p.ldk < a.heads * 64 || p.ldv < a.heads * 64 || p.ldo < a.heads * 64) { mvd_set_error("attention: bad strides in problem %d", i); return -1; }
    if (((uintptr_t)p.q | (uintptr_t)p.k | (uintptr_t)p.v) & 15 || ((uintptr_t)p.o & 7)) { mvd_set_error("attention: misaligned pointer in problem %d", i); return -1; }
    if ((p.bsq % 8) || (p.bsk % 8) || (p.bsv % 8) || (p.bso % 4)) { mvd_set_error("attention: bad batch strides in problem %d", i); return -1; }
    maxq = p.nq > maxq ? p.nq : maxq;
    mink = p.nk < mink ? p.nk : mink;
  }
  // 128-key tiles are an experiment switch only (env MVD_ATTN_KV128=1): at 197 VGPRs / 64 KB LDS they run two
  // waves per SIMD instead of three and measured 3-4 % SLOWER than 64-key tiles on every UNet shape
  // (profiles/r01_probe_attention_kv128.log)
#ifdef MVD_PROBE
  static const int kv128 = MVD_ENV_INT("MVD_ATTN_KV128", 0);
  const bool big = kv128 != 0 && mink >= 256 && !a.prescaled;
  // EXPERIMENT (probe builds, MVD_ATTN_PP=1): the ping-pong kernel -- measured 5-15 % slower than the free-running kernels
  // (profiles/r02_probe_attention_pingpong.log: its vector phase is twice as long as its matrix phase)
  static const int use_pp = MVD_ENV_INT("MVD_ATTN_PP", 0);
  const_cast<MvdAttnArgs&>(a).dbg = MVD_ENV_INT("MVD_ATTN_DBG", 0);
  {
    bool ok = use_pp && a.prescaled && maxq >= 256;
    for (int i = 0; i < a.nprob && ok; ++i) ok = (size_t)a.p[i].nk * (a.p[i].ldk > a.p[i].ldv ? a.p[i].ldk : a.p[i].ldv) * 2 < ((size_t)1 << 31);
    const long wgs = (long)((maxq + 255) / 256) * a.heads * a.batch * a.nprob;
    if (ok && wgs >= 1024) {
      dim3 grid((maxq + 255) / 256, a.heads, a.batch * a.nprob);
      g_last_attn[0] = 8; g_last_attn[1] = (int)wgs;
      hipLaunchKernelGGL(attn_pp_kernel, grid, dim3(512), 0, s, a);
      hipError_t e = hipGetLastError();
      if (e != hipSuccess) { mvd_set_error("attention (pp) launch: %s", hipGetErrorString(e)); return -3; }
      return 0;
    }
  }
#endif
  if (a.nsplit > 1) {
    if (!a.prescaled || a.nsplit > 8 || !a.split_ws || !a.split_cnt || mink < 64 * a.nsplit) { mvd_set_error("attention: bad split-KV request (nsplit=%d, fewest keys %d)", a.nsplit, mink); return -1; }
    for (int i = 0; i < a.nprob; ++i)
      if ((size_t)a.p[i].nk * (a.p[i].ldk > a.p[i].ldv ? a.p[i].ldk : a.p[i].ldv) * 2 >= ((size_t)1 << 31)) { mvd_set_error("attention: split-KV operand too large for buffer addressing"); return -1; }
    return launch_nw<4, 2>(a, maxq, s);
  }
#ifdef MVD_PROBE
  if (g_attn_q64_override && a.prescaled && maxq >= 256 && a.nsplit <= 1) {     // experiment: 64 queries per wave (see attn_q64_kernel)
    for (int i = 0; i < a.nprob; ++i)
      if ((size_t)a.p[i].nk * (a.p[i].ldk > a.p[i].ldv ? a.p[i].ldk : a.p[i].ldv) * 2 >= ((size_t)1 << 31)) { mvd_set_error("attention: operand too large for buffer addressing"); return -1; }
    return g_attn_q64_override == 1 ? launch_q64<2>(a, maxq, s) : launch_q64<4>(a, maxq, s);
  }
#endif
  switch (mvd_attention_pick_nw(a)) {
#ifdef MVD_PROBE   // (8-wave workgroups and 128-key tiles: measured slower everywhere, probe builds only)
    case 3: return launch_nw<8, 2>(a, maxq, s);
    case 2: return big ? launch_nw<4, 4>(a, maxq, s) : launch_nw<4, 2>(a, maxq, s);
#else
    case 3:
    case 2: return launch_nw<4, 2>(a, maxq, s);
#endif
    case 1: return launch_nw<2, 2>(a, maxq, s);
    default: return launch_nw<1, 2>(a, maxq, s);
  }
}

// Split-KV (batch 1): with about one wave per SIMD the kernel runs at a fraction of its rate (one wave's softmax and MFMA
// phases do not overlap); cutting the keys over several workgroups restores some occupancy.  All problems of a launch get
// the same split.
int mvd_attention_pick_split(const MvdAttnArgs& a) {
  if (!a.prescaled) return 1;
  int maxq = 0, mink = 1 << 30;
  for (int i = 0; i < a.nprob; ++i) { maxq = a.p[i].nq > maxq ? a.p[i].nq : maxq; mink = a.p[i].nk < mink ? a.p[i].nk : mink; }
  const long waves = (long)a.heads * a.batch * a.nprob * ((maxq + 127) / 128) * 4;
  // measured (profiles/r03_probe_attention_batch1.log): worth it only for long key sequences on a nearly empty chip --
  // 4096 keys x 5 heads 51 -> 38 us at 3 ranges (4: 43, 8: 56: the partials and the merge cost more than the occupancy
  // gains); 1024 keys x 10 heads 13.8 us unsplit against 16+ split
  if (maxq < 128 || mink < 2048 || waves >= 2048) return 1;
  return waves < 1024 ? 3 : 2;
}
static long attn_split_groups(const MvdAttnArgs& a) {
  int maxq = 0;
  for (int i = 0; i < a.nprob; ++i) maxq = a.p[i].nq > maxq ? a.p[i].nq : maxq;
  return (long)a.heads * a.batch * a.nprob * ((maxq + 127) / 128);
}
size_t mvd_attention_split_ws_bytes(const MvdAttnArgs& a, int nsplit) {
  return (size_t)attn_split_groups(a) * nsplit * 128 * (128 + 8);     // per (group, split): 128 queries x (64 bf16 + float2)
}
int mvd_attention_split_counters(const MvdAttnArgs& a) { return (int)attn_split_groups(a); }

// log2 of the waves per workgroup: enough workgroups to fill 256 CUs; 4-wave workgroups (128 queries) fit
// three per CU at ~165 VGPRs, the 8-wave shape only one
int mvd_attention_pick_nw(const MvdAttnArgs& a) {
  int maxq = 0;
  for (int i = 0; i < a.nprob; ++i) maxq = a.p[i].nq > maxq ? a.p[i].nq : maxq;
  const long heads_total = (long)a.heads * a.batch * a.nprob;
  static const int force = MVD_ENV_INT("MVD_ATTN_NW", -1);
  if (force >= 0) return force;
  if (g_attn_nw_override >= 0) return g_attn_nw_override;
  if (maxq >= 128 && heads_total * ((maxq + 127) / 128) >= 512) return 2;
  // few workgroups (batch 1): the 4-wave engine form (LDS-DMA staging, denominators on the matrix pipe) still wins over the
  // register-staged 1- and 2-wave forms -- 4096^2 x 5 heads: 51 us against 83 / 93 (profiles/r03_probe_attention_batch1.log)
  if (maxq >= 128 && a.prescaled) return 2;
  if (maxq >= 64) return 1;
  return 0;
}
