// Shared device/host helpers for the MVD hot-path kernels (gfx950 / CDNA4 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef unsigned short bf16_t;  // storage type for bf16 in global memory / LDS

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;
typedef __attribute__((ext_vector_type(2))) unsigned int u32x2;
typedef __attribute__((ext_vector_type(4))) short s16x4;

#define MVD_DEVINL __device__ __forceinline__

// Measurement switches exist in probe builds only (tools/build_variant.py <tag> -DMVD_PROBE): the product library
// reads no environment variable on a launch path.
#ifdef MVD_PROBE
#include <stdlib.h>
#define MVD_ENV_INT(name, dflt) ([] { const char* e_ = getenv(name); return e_ ? atoi(e_) : (dflt); }())
#else
#define MVD_ENV_INT(name, dflt) (dflt)
#endif

MVD_DEVINL float bf2f(bf16_t v) { return __builtin_bit_cast(float, (unsigned int)v << 16); }
// plain cast: hipcc emits v_cvt_pk_bf16_f32 (RNE, NaN stays NaN)
MVD_DEVINL bf16_t f2bf(float f) { return __builtin_bit_cast(bf16_t, (__bf16)f); }
MVD_DEVINL unsigned int pack2bf(float lo, float hi) {
  bf16x2 v = {(__bf16)lo, (__bf16)hi};
  return __builtin_bit_cast(unsigned int, v);
}
MVD_DEVINL float bflo(unsigned int u) { return __builtin_bit_cast(float, u << 16); }
MVD_DEVINL float bfhi(unsigned int u) { return __builtin_bit_cast(float, u & 0xffff0000u); }

MVD_DEVINL float silu_f(float x) { return x * __builtin_amdgcn_rcpf(1.0f + __expf(-x)); }
// exact-erf GELU with erf from Abramowitz-Stegun 7.1.26 (|error| <= 1.5e-7, far below bf16 resolution), written as
//   gelu(x) = max(x, 0) - 0.5 |x| P(t) e^{-x^2/2},  t = 1 / (1 + p |x| / sqrt 2),  P = t (a1 + t (a2 + t (a3 + t (a4 + t a5))))
// (x >= 0: 0.5 x (2 - P e); x < 0: 0.5 x P e): 13 VALU ops + rcp + exp2, no sign transfer -- the GEGLU epilogue is ~30 % of a
// K = 320 tile of the FF1 GEMM.
// (-DMVD_GELU_POLY, A/B builds: erf as an odd polynomial -- z = clamp(x, +-4.25), erf(z / sqrt 2) ~ z R(z^2), R of degree 8, no
//  transcendentals, every operation a v_pk_*_f32 -- half the vector work, |gelu error| <= 5.3e-5 absolute.  Measured: cfg4 cold
//  63.76 -> 63.53 ms (+0.4 %): the epilogue phase is not bound by its arithmetic alone.  Not worth giving up 1.5e-7; not the default.)
#ifndef MVD_GELU_POLY
// (round 4: the 0.5 lives in the coefficients and max(x, 0) is one v_max_f32 by inline asm -- fmaxf costs a canonicalising v_max of its own:
//  11 VALU + 2 transcendental instructions instead of 16 + 2; the GEGLU kernels are bound by exactly this issue time)
MVD_DEVINL float gelu_erf_f(float x) {
  const float ax = fabsf(x);
  const float t = __builtin_amdgcn_rcpf(fmaf(0.3275911f * 0.70710678118654752f, ax, 1.0f));
  float p = fmaf(0.5f * 1.061405429f, t, 0.5f * -1.453152027f);
  p = fmaf(p, t, 0.5f * 1.421413741f);
  p = fmaf(p, t, 0.5f * -0.284496736f);
  p = fmaf(p, t, 0.5f * 0.254829592f);
  const float zl = ax * 0.84932180028801907f;                 // |x| sqrt(log2(e) / 2):  e^{-x^2/2} = 2^{-zl^2}
  const float e = __builtin_amdgcn_exp2f(-zl * zl);
  const float h = (ax * (p * t)) * e;
  float mx;                                   // max(x, 0) as ONE instruction (fmaxf / fmed3 come with a canonicalising v_max x, x in front)
  asm("v_max_f32 %0, 0, %1" : "=v"(mx) : "v"(x));
  return mx - h;
}
#else
MVD_DEVINL float gelu_erf_f(float x) {
  const float z = __builtin_amdgcn_fmed3f(x, -4.25f, 4.25f);
  const float t = z * z;
  float r = fmaf(1.112984843e-10f, t, -1.065562572e-08f);
  r = fmaf(r, t, 4.510895621e-07f);
  r = fmaf(r, t, -1.125293875e-05f);
  r = fmaf(r, t, 1.868385298e-04f);
  r = fmaf(r, t, -2.217133064e-03f);
  r = fmaf(r, t, 1.963203214e-02f);
  r = fmaf(r, t, -1.326895654e-01f);
  r = fmaf(r, t, 7.978081107e-01f);
  return x * fmaf(0.5f * z, r, 0.5f);
}
#endif

MVD_DEVINL float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}

// XCD-aware bijective block remap (blocks b, b+8, ... share an XCD): gives each XCD a
// contiguous chunk of the logical tile order so neighbouring tiles hit the same L2.
MVD_DEVINL int xcd_remap(int bid, int nwg) {
  const int q = nwg >> 3, r = nwg & 7, x = bid & 7;
  const int base = (x < r) ? x * (q + 1) : r * (q + 1) + (x - r) * q;
  return base + (bid >> 3);
}
