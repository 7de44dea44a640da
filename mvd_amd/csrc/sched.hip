// Denoising-loop helpers around the UNet: the DDPM ancestral step and the
// classifier-free-guidance combine as single fused elementwise kernels (fp32 latents, 16-byte vectors).
// Follows /root/reference/src/models/pipeline.py:156-161 and the diffusers-0.32.2 DDPMScheduler.step algebra;
// the per-step scalar coefficients are computed on the host (mvd_amd/scheduler.py) so the loop never syncs.
#include "kernels.h"

namespace {

// x0 = c0*model_out + c1*sample ; prev = c2*x0 + c3*sample + sigma*noise
__global__ void ddpm_step_kernel(const float* __restrict__ mo, const float* __restrict__ x, const float* __restrict__ nz,
                                 float c0, float c1, float c2, float c3, float sigma, float* __restrict__ y, long n4) {
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (long)gridDim.x * blockDim.x) {
    const f32x4 m = reinterpret_cast<const f32x4*>(mo)[i], s = reinterpret_cast<const f32x4*>(x)[i];
    f32x4 x0 = m * c0 + s * c1;
    f32x4 p = x0 * c2 + s * c3;
    if (nz) p += reinterpret_cast<const f32x4*>(nz)[i] * sigma;
    reinterpret_cast<f32x4*>(y)[i] = p;
  }
}

// noise_pred = uncond + g*(cond - uncond) with [uncond | cond] stacked on the batch dim
__global__ void cfg_combine_kernel(const float* __restrict__ both, float g, float* __restrict__ y, long n4) {
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (long)gridDim.x * blockDim.x) {
    const f32x4 u = reinterpret_cast<const f32x4*>(both)[i], c = reinterpret_cast<const f32x4*>(both)[i + n4];
    reinterpret_cast<f32x4*>(y)[i] = u + (c - u) * g;
  }
}

int grid_for(long n4) { long g = (n4 + 255) / 256; return (int)(g > 2048 ? 2048 : (g < 1 ? 1 : g)); }

}  // namespace

extern "C" int mvd_op_ddpm_step(const float* model_out, const float* sample, const float* noise, float c0, float c1, float c2,
                                float c3, float sigma, float* out, int64_t n, void* stream) {
  if (!model_out || !sample || !out || n <= 0 || (n & 3)) { mvd_set_error("ddpm_step: bad arguments (n must be a multiple of 4)"); return -1; }
  if (!noise && sigma != 0.f) { mvd_set_error("ddpm_step: noise required when sigma != 0"); return -1; }
  hipLaunchKernelGGL(ddpm_step_kernel, dim3(grid_for(n / 4)), dim3(256), 0, (hipStream_t)stream, model_out, sample, noise, c0, c1, c2,
                     c3, sigma, out, (long)(n / 4));
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) { mvd_set_error("ddpm_step launch: %s", hipGetErrorString(e)); return -3; }
  return 0;
}

extern "C" int mvd_op_cfg_combine(const float* uncond_cond, float guidance_scale, float* out, int64_t n_half, void* stream) {
  if (!uncond_cond || !out || n_half <= 0 || (n_half & 3)) { mvd_set_error("cfg_combine: bad arguments"); return -1; }
  hipLaunchKernelGGL(cfg_combine_kernel, dim3(grid_for(n_half / 4)), dim3(256), 0, (hipStream_t)stream, uncond_cond, guidance_scale, out,
                     (long)(n_half / 4));
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) { mvd_set_error("cfg_combine launch: %s", hipGetErrorString(e)); return -3; }
  return 0;
}
