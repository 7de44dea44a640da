// Round-5 pricing of Winograd F(2x2, 3x3) for the stride-1 3x3 resnet convolutions, part 2: the BEST CASE of
// the 16-way batched MFMA core on gfx950 -- an UPPER BOUND on any fused Winograd kernel, measured before building one.
//
//   M_p[tile][n] = sum_c V_p[tile][c] * U_p[n][c]        p = 0..15 (the 4x4 positions of F(2x2, 3x3)), tile = 2x2 output pixels
//
// What the probe is given for free (everything a real kernel would have to pay for on top):
//   * V (the transformed activations, 4x the activation bytes) and U (the transformed weights) arrive PRE-TRANSFORMED and
//     pre-packed in MFMA fragment order: no input transform (32 adds per tile-channel), no GroupNorm / SiLU, no LDS staging;
//   * no inverse transform, no bias / time-embedding row / residual / shortcut: the accumulators of a wave are summed and stored;
//   * the operands go global -> registers directly (`global_load_dwordx4` of a 1 KB fragment per wave-instruction): the
//     fragments of one position are used by ONE wave only, so LDS would only add traffic.
// Geometry: a workgroup = 8 waves = 64 tiles (256 output pixels) x 64 output channels x ALL 16 positions (the inverse transform
// needs the 16 positions of a tile in one place); wave w owns positions 2w, 2w + 1: 2 x (64 x 64) fp32 accumulators = 128
// registers per lane -- the register file (512 per SIMD lane, two waves per SIMD) caps the per-CU tile at tiles x channels = 4096.
// Per k-step of 16 input channels a wave loads 8 fragments (2 positions x (2 V + 2 U)) = 8 KB for 8 MFMAs of 32 cycles: at the
// full MFMA rate a CU would have to take in 128 B/clk from L2.
//
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/probe_wino_core tools/probe_wino_core.hip && /tmp/probe_wino_core
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) unsigned u32x4;

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

// V: [tile block][k-step][position][half][64 lanes][8]   U: [channel block][k-step][position][half][64 lanes][8]
// XCD-aware order: hardware deals consecutive workgroup ids round-robin to the 8 XCDs; XCD x works through a contiguous range of
// tile blocks with the channel blocks fastest, so the workgroups resident on one XCD share a few V blocks and all U blocks in L2.
__global__ __launch_bounds__(512, 2) void wino_core_kernel(const u32x4* __restrict__ V, const u32x4* __restrict__ U,
                                                           __bf16* __restrict__ out, int ksteps, int ntb, int nnb, int same_block) {
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int per_xcd = (ntb * nnb + 7) / 8;
  const int id = (blockIdx.x & 7) * per_xcd + (blockIdx.x >> 3);
  if (id >= ntb * nnb) return;
  // same_block: EVERY workgroup reads tile block 0 and channel block 0 (1.3 MB at C = 320: always in L2) -- the ceiling of this
  // tile geometry with perfect L2 locality, whatever order a real kernel could walk its tiles in
  const int tb = same_block ? 0 : id / nnb, nb = same_block ? 0 : id % nnb;
  // fragment index of (k-step ks, position p, half h): ((ks * 16 + p) * 2 + h) * 64 + lane
  const u32x4* vp = V + ((size_t)tb * ksteps * 32 + wave * 4) * 64 + lane;
  const u32x4* up = U + ((size_t)nb * ksteps * 32 + wave * 4) * 64 + lane;
  f32x16 acc[2][2][2];
#pragma unroll
  for (int p = 0; p < 2; ++p)
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[p][i][j][r] = 0.f;
  u32x4 vb[3][4], ub[3][4];                                 // three k-steps of fragments in registers: two in flight
  auto load = [&](int buf, int ks) {
#pragma unroll
    for (int f = 0; f < 4; ++f) {
      vb[buf][f] = vp[((size_t)ks * 32 + f) * 64];
      ub[buf][f] = up[((size_t)ks * 32 + f) * 64];
    }
  };
  auto mul = [&](int buf) {
#pragma unroll
    for (int p = 0; p < 2; ++p)
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
          acc[p][i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, ub[buf][2 * p + j]),
                                                                 __builtin_bit_cast(bf16x8, vb[buf][2 * p + i]), acc[p][i][j], 0, 0, 0);
  };
  load(0, 0);
  if (ksteps > 1) load(1, 1);
  int ks = 0;
  for (; ks + 3 <= ksteps; ks += 3) {                       // ksteps % 3 handled below
    if (ks + 2 < ksteps) load(2, ks + 2);
    mul(0);
    if (ks + 3 < ksteps) load(0, ks + 3);
    mul(1);
    if (ks + 4 < ksteps) load(1, ks + 4);
    mul(2);
  }
  if (ks < ksteps) { if (ks + 2 < ksteps) load(2, ks + 2); mul(0); ++ks; }
  if (ks < ksteps) { mul(1); ++ks; }
  // stand-in for the inverse transform + epilogue: one bf16 per (lane, register) -- 16 KB per wave, 1/16 of what the wave holds
  f32x16 s;
#pragma unroll
  for (int r = 0; r < 16; ++r) s[r] = acc[0][0][0][r] + acc[0][0][1][r] + acc[0][1][0][r] + acc[0][1][1][r] +
                                      acc[1][0][0][r] + acc[1][0][1][r] + acc[1][1][0][r] + acc[1][1][1][r];
  __bf16* o = out + (((size_t)id * 8 + wave) * 64 + lane) * 16;
#pragma unroll
  for (int r = 0; r < 16; ++r) o[r] = (__bf16)s[r];
}

static void run(const char* name, long M, int C, int N, double direct_us, int same_block = 0) {
  const int ntb = (int)(M / 4 / 64), nnb = N / 64, ksteps = C / 16;
  const size_t vel = (size_t)ntb * ksteps * 32 * 64 * 8, uel = (size_t)nnb * ksteps * 32 * 64 * 8, oel = (size_t)ntb * nnb * 8 * 64 * 16;
  unsigned short *V, *U; __bf16* O;
  CHECK(hipMalloc(&V, vel * 2)); CHECK(hipMalloc(&U, uel * 2)); CHECK(hipMalloc(&O, oel * 2));
  std::vector<unsigned short> h(vel > uel ? vel : uel);
  unsigned s = 12345u;
  for (auto& x : h) { s = s * 1664525u + 1013904223u; x = (unsigned short)(0x3c00u + ((s >> 16) & 0x3ffu) + ((s >> 31) << 15)); }   // +-[0.0078, 0.0156): finite bf16
  CHECK(hipMemcpy(V, h.data(), vel * 2, hipMemcpyHostToDevice));
  CHECK(hipMemcpy(U, h.data(), uel * 2, hipMemcpyHostToDevice));
  const int grid = ((ntb * nnb + 7) / 8) * 8;
  hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
  for (int i = 0; i < 3; ++i) hipLaunchKernelGGL(wino_core_kernel, dim3(grid), dim3(512), 0, 0, (const u32x4*)V, (const u32x4*)U, O, ksteps, ntb, nnb, same_block);
  CHECK(hipDeviceSynchronize());
  const int it = 20;
  CHECK(hipEventRecord(e0));
  for (int i = 0; i < it; ++i) hipLaunchKernelGGL(wino_core_kernel, dim3(grid), dim3(512), 0, 0, (const u32x4*)V, (const u32x4*)U, O, ksteps, ntb, nnb, same_block);
  CHECK(hipEventRecord(e1)); CHECK(hipDeviceSynchronize());
  float ms = 0; CHECK(hipEventElapsedTime(&ms, e0, e1));
  const double us = ms * 1e3 / it;
  const double wino_flop = 2.0 * 16 * (M / 4) * (double)C * N, direct_flop = 2.0 * M * (double)N * 9 * C;
  printf("%s%-34s M %6ld C %4d N %4d  %4d workgroups  core %7.1f us = %6.0f TFLOP/s on the matrix pipe (%4.1f %% of 2.5 PF), "
         "%6.0f direct-equivalent TFLOP/s;  V+U operand bytes %6.1f MB;  direct kernel today %6.1f us -> best-case speed-up %4.2fx\n",
         same_block ? "[all workgroups read ONE V block and ONE U block: every load an L2 hit] " : "", name, M, C, N, ntb * nnb, us, wino_flop / us * 1e-6, wino_flop / us * 1e-6 / 25.0, direct_flop / us * 1e-6, (vel + uel) * 2e-6,
         direct_us, direct_us / us);
  CHECK(hipFree(V)); CHECK(hipFree(U)); CHECK(hipFree(O));
}

int main() {
  // direct_us: the engine's implicit-GEMM kernel on the same shape, per launch (profiles/r04_shapes_cfg4.txt, 3 profiled steps)
  run("64x64 level, 320 -> 320 (cls 13)", 131072, 320, 320, 4688.0 / 24);
  run("64x64 level, 640 -> 320", 131072, 640, 320, 4097.0 / 12);
  run("64x64 level, 960 -> 320", 131072, 960, 320, 2973.0 / 6);
  run("32x32 level, 640 -> 640", 32768, 640, 640, 2146.0 / 12);
  run("32x32 level, 1280 -> 640", 32768, 1280, 640, 1980.0 / 6);
  run("16x16 level, 1280 -> 1280", 8192, 1280, 1280, 3282.0 / 18);
  run("16x16 level, 2560 -> 1280", 8192, 2560, 1280, 4023.0 / 12);
  run("8x8 level, 1280 -> 1280", 2048, 1280, 1280, 3662.0 / 54);
  // the same launches with perfect L2 locality (an upper bound no tile order can beat): what the L2 -> register path and the
  // matrix pipe allow this geometry
  run("64x64 level, 320 -> 320 (cls 13)", 131072, 320, 320, 4688.0 / 24, 1);
  run("64x64 level, 960 -> 320", 131072, 960, 320, 2973.0 / 6, 1);
  run("32x32 level, 640 -> 640", 32768, 640, 640, 2146.0 / 12, 1);
  run("16x16 level, 1280 -> 1280", 8192, 1280, 1280, 3282.0 / 18, 1);
  return 0;
}
