// Ring-pipelined bf16 MFMA GEMM / implicit 3x3 convolution for gfx950: the 256x320 tile / 128x80 wave tile of gemm.hip
// (eight waves, two per SIMD) with a deeper, drain-free pipeline.
//
//   * K advances in 32-wide steps through a ring of FOUR 36 KB LDS stages filled by LDS-DMA (global_load_lds_dwordx4,
//     issued from inline asm so that hipcc neither counts nor drains it); the loads of step g+3 are issued at the top of
//     step g and retired by a COUNTED s_waitcnt vmcnt(N) at the end of step g+1: two stages stay in flight across the
//     per-step barrier and no wait in the loop is a vmcnt(0);
//   * W fragments of step g+1 replace those of step g in place during the last row tile of step g, A fragments stream
//     through a three-entry register ring (the first two of step g+1 are fetched before step g ends): the first MFMAs of
//     a step do not wait for LDS;
//   * the ring runs straight across tile boundaries (persistent workgroups, XCD-contiguous tile chunks) and the output
//     stores of a finished tile are not waited for: the counted waits of the next two steps let them drain under the
//     next tile's MFMAs (vmcnt is in-order, see the arithmetic at `step`).
//
// Same operands, layouts and epilogue algebra as gemm_kernel (gemm.hip); GEGLU and fp32 output stay there.
// (A four-wave / 512-register variant with 128x160 wave tiles was tried first: hipcc cannot allocate it -- the
//  accumulators bounce between AGPRs, VGPRs and scratch.)
#include <stdlib.h>
#include "../kernels.h"

namespace {

constexpr int RBM = 256, RBN = 320, RS = 4;
constexpr int RA_ST = RBM * 64, RW_ST = RBN * 64, RST = RA_ST + RW_ST;   // bytes per stage (one 32-wide k-step)
constexpr int RLDS = RS * RST;                                            // 147456
constexpr int RTM = 8, RTN = 5;                                           // 16x16 MFMA tiles per wave: 128 x 80
// LDS-DMA instructions per wave per stage: 2 A pieces + 3 (waves 0-3) or 2 (waves 4-7) W pieces of 16 rows x 64 B

__device__ __attribute__((aligned(16))) unsigned int g_zero16_ring[4] = {0u, 0u, 0u, 0u};

// LDS-DMA of 16 bytes per lane to lds_dst + 16*lane (lds_dst wave-uniform).  M0 is compiler-reserved: saved/restored here.
MVD_DEVINL void glds16_asm(const void* gsrc, unsigned lds_dst) {
  unsigned keep;
  lds_dst = __builtin_amdgcn_readfirstlane(lds_dst);
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
               : "=&s"(keep) : "v"(gsrc), "s"(lds_dst) : "memory");
}
template <int N> MVD_DEVINL void wait_vmcnt() { asm volatile("s_waitcnt vmcnt(%0)" :: "n"(N) : "memory"); }

// 16-byte slot permutation inside a 64-byte row: slot' = slot ^ g((row >> 2) & 3), g = {0, 2, 3, 1}.  With 64-byte rows a
// ds_read_b128 lane group spans rows {0-3, 12-15} at one k-chunk and rows {4-11} at the next; this g makes the four
// slots met by each row residue distinct (bank-conflict free), and the DMA image stays lane-linear (the permutation is
// applied to the SOURCE column of each lane).
MVD_DEVINL int slot_perm(int row) { return (0x78 >> (2 * ((row >> 2) & 3))) & 3; }

template <int AMODE, bool SPLITK>
__global__ __launch_bounds__(512, 2) void gemm_ring_kernel(const MvdGemmArgs a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 2, wn = wave & 3;
  const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) unsigned char*)smem;

  const int ntn = a.N / RBN;
  const int ntm = (a.M + RBM - 1) / RBM;
  const int S = SPLITK ? a.splitk : 1;
  const int nwork = ntn * ntm * S;
  const int xcd = blockIdx.x & 7, xj = blockIdx.x >> 3;
  const int gx = (gridDim.x >> 3) + ((int)(gridDim.x & 7) > xcd ? 1 : 0);
  const int tq = nwork >> 3, tr = nwork & 7;
  const int tstart = xcd < tr ? xcd * (tq + 1) : tr * (tq + 1) + (xcd - tr) * tq;
  const int tend = tstart + tq + (xcd < tr ? 1 : 0);
  if (tstart + xj >= tend) return;

  constexpr bool HAS_CONV = AMODE != 0;
  const MvdASeg& cs = a.seg[0];
  const MvdASeg& ds = a.seg[AMODE == 2 ? 1 : 0];
  const int conv_c = cs.c0, conv_inW = cs.inW, conv_ups = cs.ups;
  const int limH = conv_ups ? 2 * cs.inH : cs.inH, limW = conv_ups ? 2 * cs.inW : cs.inW;
  const bf16_t* conv_p = cs.p0;
  const bf16_t* dp0 = ds.p0;
  const bf16_t* dp1 = ds.p1;
  const int dc0 = ds.c0, dc1 = ds.c1;
  const int nks_conv = HAS_CONV ? (9 * conv_c) / 32 : 0;   // k-steps of the conv segment
  const int nkt = a.Ktot / 64;                              // 64-wide slabs (the split-K unit, as in gemm.hip)

  // ---- per-lane constants
  const int r4 = lane >> 2, sl = lane & 3;
  const int kc8 = (sl ^ slot_perm(r4)) * 8;                 // this lane's source column inside a k-step (elements)
  const int fr = lane & 15, fq = lane >> 4;
  const int frag_off = fr * 64 + ((fq ^ slot_perm(fr)) << 4);

  auto step_range = [&](int work, int& k0, int& k1) __attribute__((always_inline)) {        // k-step range of a work item
    if (S == 1) { k0 = 0; k1 = 2 * nkt; return; }
    const int ks = work % S;
    k0 = 2 * ((ks * nkt) / S);
    k1 = 2 * (((ks + 1) * nkt) / S);
  };

  // ---- loader (runs four k-steps ahead of the MFMAs, possibly in the next work item)
  int L_work = tstart + xj, L_ks = 0, L_k1 = 0, L_n0 = 0;
  bool L_active = true;
  int a_m[2], a_pb[2], a_yx[2];
  auto setup_loader = [&](int work) __attribute__((always_inline)) {
    const int t = S == 1 ? work : work / S;
    const int m0 = (t / ntn) * RBM;
    L_n0 = (t % ntn) * RBN;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      int m = m0 + wave * 32 + i * 16 + r4;
      m = m < a.M ? m : a.M - 1;
      a_m[i] = m; a_pb[i] = 0; a_yx[i] = 0;
      if (HAS_CONV) {
        const int b = m / a.rows_per_batch;
        const int rem = m - b * a.rows_per_batch;
        const int oy = rem / a.outW, ox = rem - oy * a.outW;
        a_pb[i] = b * cs.inH * cs.inW;
        a_yx[i] = (oy * cs.stride) | ((ox * cs.stride) << 16);
      }
    }
  };
  auto issue_loads = [&](int slot) __attribute__((always_inline)) {
    const int ks = L_ks;
    const unsigned dA = lds0 + slot * RST + wave * (32 * 64);
    const unsigned dW = lds0 + slot * RST + RA_ST;
    if (HAS_CONV && (AMODE == 1 || ks < nks_conv)) {
      // conv K order [64-channel slice][tap][64]: two k-steps per (slice, tap)
      const int slab = ks >> 1;
      const int ld_cs = slab / 9;
      const int ld_tap = slab - ld_cs * 9;
      const int dy = ld_tap / 3, dx = ld_tap - dy * 3;
      const int col = (ld_cs << 6) + ((ks & 1) << 5) + kc8;
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        const int iy = (a_yx[i] & 0xffff) - 1 + dy, ix = (a_yx[i] >> 16) - 1 + dx;
        const bool ok = (unsigned)iy < (unsigned)limH && (unsigned)ix < (unsigned)limW;
        const int sy = conv_ups ? (iy >> 1) : iy, sx = conv_ups ? (ix >> 1) : ix;
        const bf16_t* p = conv_p + (size_t)(a_pb[i] + sy * conv_inW + sx) * conv_c + col;
        glds16_asm(ok ? (const void*)p : (const void*)g_zero16_ring, dA + i * 1024);
      }
    } else {
      const int kk = (ks - nks_conv) << 5;
      const bool first = kk < dc0;
      const bf16_t* base = first ? dp0 : dp1;
      const int ld = first ? dc0 : dc1;
      const int col = (first ? kk : kk - dc0) + kc8;
#pragma unroll
      for (int i = 0; i < 2; ++i) glds16_asm(base + (size_t)a_m[i] * ld + col, dA + i * 1024);
    }
#pragma unroll
    for (int i = 0; i < 3; ++i) {             // W pieces wave, wave+8, wave+16 (the last one only for waves 0-3)
      const int piece = wave + 8 * i;
      if (i < 2 || wave < 4)
        glds16_asm(a.W + (size_t)(L_n0 + piece * 16 + r4) * a.ldw + ks * 32 + kc8, dW + piece * 1024);
    }
  };
  auto advance_loader = [&]() __attribute__((always_inline)) {
    if (++L_ks < L_k1) return;
    L_work += gx;
    if (L_work < tend) { setup_loader(L_work); step_range(L_work, L_ks, L_k1); }
    else L_active = false;
  };

  // ---- fragments and accumulators: 160 + 20 + 12 registers
  bf16x8 fw[RTN], fa[3];
  f32x4 acc[RTM][RTN];
  auto a_frag = [&](int slot, int i) __attribute__((always_inline)) -> bf16x8 {
    return *reinterpret_cast<const bf16x8*>(smem + slot * RST + wm * (128 * 64) + frag_off + i * 1024);
  };
  auto w_frag = [&](int slot, int j) __attribute__((always_inline)) -> bf16x8 {
    return *reinterpret_cast<const bf16x8*>(smem + slot * RST + RA_ST + wn * (80 * 64) + frag_off + j * 1024);
  };
  auto zero_acc = [&]() __attribute__((always_inline)) {
#pragma unroll
    for (int i = 0; i < RTM; ++i)
#pragma unroll
      for (int j = 0; j < RTN; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
  };

  // ---- epilogue: lane holds out[m = row tile i + fr][n = column tile j + 4*fq .. +3]
  const float alpha = a.alpha;
  auto epilogue = [&](int work) __attribute__((always_inline)) -> bool {   // returns true when every wave issued exactly RTM*RTN stores
    const int t = S == 1 ? work : work / S;
    int m0 = __builtin_amdgcn_readfirstlane((t / ntn) * RBM), n0 = __builtin_amdgcn_readfirstlane((t % ntn) * RBN);
    asm volatile("" : "+s"(m0), "+s"(n0));   // keep the address arithmetic here (not hoisted into live registers)
    const int nb = n0 + wn * 80 + fq * 4;
    const bool full = m0 + RBM <= a.M && !(a.dbg & 1);
    if (SPLITK) {
      float* pp = a.part + (size_t)(work - t * S) * a.M * a.N;
#pragma unroll
      for (int i = 0; i < RTM; ++i) {
        const int m = m0 + wm * 128 + i * 16 + fr;
#pragma unroll
        for (int j = 0; j < RTN; ++j)
          if (m < a.M && !(a.dbg & 1)) *reinterpret_cast<f32x4*>(pp + (size_t)m * a.N + nb + j * 16) = acc[i][j];
      }
      return full;
    }
    // lean on registers (the accumulators are modified in place): a spill here would put scratch reloads -- and the
    // vmcnt(0) hipcc waits for them with -- into the main loop
    bf16_t* outp = reinterpret_cast<bf16_t*>(a.out) + nb;
#pragma unroll
    for (int i = 0; i < RTM; ++i) {
      const int m = m0 + wm * 128 + i * 16 + fr;
      const int mc = m < a.M ? m : a.M - 1;
      u32x2 r[RTN];
      if (a.res) {
        const bf16_t* rp = a.res + (size_t)mc * a.ldres + nb;
#pragma unroll
        for (int j = 0; j < RTN; ++j) r[j] = *reinterpret_cast<const u32x2*>(rp + j * 16);
      }
      const float* rv = a.rowvec ? a.rowvec + (size_t)(mc / a.rows_per_batch) * a.ld_rowvec + nb : nullptr;
      bf16_t* orow = outp + (size_t)m * a.ldo;
      const bool st = full || (m < a.M && !(a.dbg & 1));
#pragma unroll
      for (int j = 0; j < RTN; ++j) {
        f32x4 v = acc[i][j];
        if (a.bias) v += *reinterpret_cast<const f32x4*>(a.bias + nb + j * 16);
        if (rv) v += *reinterpret_cast<const f32x4*>(rv + j * 16);
        v *= alpha;
        if (a.res) { v[0] += bflo(r[j][0]); v[1] += bfhi(r[j][0]); v[2] += bflo(r[j][1]); v[3] += bfhi(r[j][1]); }
        if (st) {
          u32x2 o = {pack2bf(v[0], v[1]), pack2bf(v[2], v[3])};
          *reinterpret_cast<u32x2*>(orow + j * 16) = o;
        }
      }
    }
    return full;
  };

  // ---- prologue: load k-steps 0..2, publish steps 0 and 1, fetch the first fragments of step 0
  int C_work = tstart + xj, C_ks, C_k1;
  step_range(C_work, C_ks, C_k1);
  setup_loader(L_work);
  step_range(L_work, L_ks, L_k1);
  int n_pro = 0;
#pragma unroll
  for (int q = 0; q < 3; ++q)
    if (L_active) { issue_loads(q); advance_loader(); ++n_pro; }
  // counted waits: a wave's stage is 5 (waves 0-3) or 4 (waves 4-7) LDS-DMA instructions
  if (n_pro == 3) { if (wave < 4) wait_vmcnt<5>(); else wait_vmcnt<4>(); } else wait_vmcnt<0>();
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");
#pragma unroll
  for (int j = 0; j < RTN; ++j) fw[j] = w_frag(0, j);
  fa[0] = a_frag(0, 0);
  fa[1] = a_frag(0, 1);
  zero_acc();

  int stores_pending = 0;                 // > 0: a full tile's stores were issued in this or the previous step
  int slot = 0;

  // One k-step g (stage `slot` = g % 4).  At its top the loads of step g+3 go to stage (g+3) % 4, whose last readers
  // finished before the barrier that ended step g-1.  At its end the stage of step g+2 is retired (it is first read at
  // the end of step g+1).  vmcnt is in-order: younger than the g+2 group are the g+3 group (NL = 5 or 4 instructions of
  // this wave) and, if a tile finished in step g-1 or g, its RTM*RTN = 40 stores -- so vmcnt(NL + 40) then still retires
  // the g+2 group while the stores drain under the next tile's MFMAs, and vmcnt(NL) otherwise.
  for (;;) {
    const bool issued = L_active;
    if (issued) { issue_loads((slot + 3) & 3); advance_loader(); }
    const bool last = C_ks + 1 == C_k1;
    const bool more = !last || C_work + gx < tend;   // is there a step g+1?
    const int nslot = (slot + 1) & 3;
    fa[2] = a_frag(slot, 2);
#pragma unroll
    for (int i = 0; i < RTM; ++i) {
#pragma unroll
      for (int j = 0; j < RTN; ++j) {
        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fw[j], fa[i % 3], acc[i][j], 0, 0, 0);
        // last row tile: the W fragment just consumed is replaced by the one of step g+1 (read unconditionally:
        // after the final step the values are simply unused, and the MFMA stream stays one basic block)
        if (i == RTM - 1) fw[j] = w_frag(nslot, j);
      }
      if (i + 3 < RTM) fa[i % 3] = a_frag(slot, i + 3);
      else if (i == RTM - 2) fa[0] = a_frag(nslot, 0);
      else if (i == RTM - 1) fa[1] = a_frag(nslot, 1);
    }
    if (last) {
      if (epilogue(C_work)) stores_pending = 2;
      C_work += gx;
      if (more) { step_range(C_work, C_ks, C_k1); zero_acc(); }
    } else {
      ++C_ks;
    }
    if (!more) break;
    if (!issued) wait_vmcnt<0>();
    else if (wave < 4) { if (stores_pending) wait_vmcnt<45>(); else wait_vmcnt<5>(); }
    else { if (stores_pending) wait_vmcnt<44>(); else wait_vmcnt<4>(); }
    if (stores_pending) --stores_pending;
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // this wave's reads of stages g and g+1 are done
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    slot = nslot;
  }
}

template <int AMODE, bool SPLITK>
int launch_ring(const MvdGemmArgs& a, hipStream_t s) {
  static bool attr_set = false;
  if (!attr_set) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_ring_kernel<AMODE, SPLITK>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, RLDS);
    if (e != hipSuccess) { mvd_set_error("gemm_ring: hipFuncSetAttribute: %s", hipGetErrorString(e)); return -2; }
    attr_set = true;
  }
  const int ntm = (a.M + RBM - 1) / RBM, ntn = a.N / RBN;
  const int nwork = ntm * ntn * (a.splitk > 1 ? a.splitk : 1);
  int grid = 256;                                   // one 144 KB workgroup per CU
  if (nwork < grid) grid = ((nwork + 7) / 8) * 8;
  hipLaunchKernelGGL((gemm_ring_kernel<AMODE, SPLITK>), dim3(grid), dim3(512), RLDS, s, a);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) { mvd_set_error("gemm_ring launch: %s", hipGetErrorString(e)); return -3; }
  return 0;
}

}  // namespace

// Shape validation is done by mvd_launch_gemm (gemm.hip) before it dispatches here.
int mvd_launch_gemm_ring(const MvdGemmArgs& a, hipStream_t s) {
  if (a.N % RBN || a.geglu || a.out_f32 || a.Ktot % 64) { mvd_set_error("gemm_ring: needs N %% 320 == 0, K %% 64 == 0, bf16 output, no GEGLU"); return -1; }
  const bool sk = a.splitk > 1;
  if (a.seg[0].mode == MVD_A_DENSE) return sk ? launch_ring<0, true>(a, s) : launch_ring<0, false>(a, s);
  if (a.nseg == 1) return sk ? launch_ring<1, true>(a, s) : launch_ring<1, false>(a, s);
  return sk ? launch_ring<2, true>(a, s) : launch_ring<2, false>(a, s);
}
