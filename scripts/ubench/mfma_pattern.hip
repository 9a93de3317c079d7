// Micro-benchmark: how the fp32 MFMA rate of ONE CU depends on the wave-tile shape and the waves per SIMD, with the conv
// kernels' operand traffic (ds_read of A / B fragments two taps ahead of the MFMAs) but no global memory in the loop.
// build: hipcc --offload-arch=gfx950 -O3 -o mfma_pattern mfma_pattern.hip ; run: ./mfma_pattern
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int MT, int NT, int NWAVES, int MINW, bool READS>
__global__ __launch_bounds__(NWAVES * 64, MINW) void pat(float* out, int ntaps, int dil) {
  extern __shared__ float smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  for (int i = tid; i < 8192; i += NWAVES * 64) smem[i] = (float)(i & 15) * 0.001f;
  __syncthreads();
  f32x16 acc[MT][NT];
#pragma unroll
  for (int i = 0; i < MT; ++i)
#pragma unroll
    for (int j = 0; j < NT; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
  const float* wl = smem + (wave & 3) * 32 + (lane & 31) + (lane >> 5) * 1408;
  const float* xl = smem + 4096 + (lane & 31) + (lane >> 5) * 280 + (wave >> 2) * 128;
  float av[3][MT], bv[3][NT];
  auto ld = [&](int s, int t) {
    const float* wa = wl + (t & 7) * 128;
    const float* xa = xl + (t & 7) * dil;
#pragma unroll
    for (int i = 0; i < MT; ++i) av[s][i] = wa[i * 64];
#pragma unroll
    for (int j = 0; j < NT; ++j) bv[s][j] = xa[j * 32];
  };
  if (READS) { ld(0, 0); ld(1, 1); }
  else {
#pragma unroll
    for (int s = 0; s < 3; ++s) {
#pragma unroll
      for (int i = 0; i < MT; ++i) av[s][i] = wl[i * 64 + s];
#pragma unroll
      for (int j = 0; j < NT; ++j) bv[s][j] = xl[j * 32 + s];
    }
  }
  for (int t = 0; t < ntaps; t += 3) {
#pragma unroll
    for (int s = 0; s < 3; ++s) {
      __builtin_amdgcn_sched_barrier(0);
      if (READS) ld((s + 2) % 3, t + s + 2);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[s][i], bv[s][j], acc[i][j], 0, 0, 0);
    }
  }
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < MT; ++i)
#pragma unroll
    for (int j = 0; j < NT; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) s += acc[i][j][r];
  if (s == 12345.678f) out[blockIdx.x * NWAVES * 64 + tid] = s;
}

template <int MT, int NT, int NWAVES, int MINW, bool READS>
void run(const char* name, int wg_per_cu, float* out) {
  const int ntaps = 3 * 2048, grid = 256 * wg_per_cu;
  auto k = pat<MT, NT, NWAVES, MINW, READS>;
  hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  const size_t lds = wg_per_cu == 1 ? 100 * 1024 : 64 * 1024;   // pins the residency: 1 or 2 workgroups per CU
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL(k, dim3(grid), dim3(NWAVES * 64), lds, 0, out, ntaps, 5);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  for (int r = 0; r < 5; ++r) hipLaunchKernelGGL(k, dim3(grid), dim3(NWAVES * 64), lds, 0, out, ntaps, 5);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms = 0.f;
  hipEventElapsedTime(&ms, e0, e1);
  ms /= 5;
  const double mfma_per_simd = (double)ntaps * MT * NT * (NWAVES / 4) * wg_per_cu;
  const double flops = (double)grid * NWAVES * ntaps * MT * NT * 4096.0;
  printf("%-44s %8.1f us  %6.1f TFLOP/s  %6.1f ns per MFMA per SIMD (64 cycles = %.1f ns at 2.4 GHz)\n", name, ms * 1e3,
         flops / ms / 1e9, ms * 1e6 / mfma_per_simd, 64 / 2.4);
}

int main() {
  float* out;
  hipMalloc(&out, 1 << 24);
  for (int rep = 0; rep < 2; ++rep) {
    run<1, 7, 4, 1, false>("1x7 tiles, 1 wave/SIMD, no LDS reads", 1, out);
    run<1, 7, 4, 1, true>("1x7 tiles, 1 wave/SIMD", 1, out);
    run<1, 4, 8, 2, true>("1x4 tiles, 2 waves/SIMD (one WG)", 1, out);
    run<2, 2, 4, 2, true>("2x2 tiles, 2 WGs/CU x 4 waves", 2, out);
    run<2, 2, 4, 2, false>("2x2 tiles, 2 WGs/CU x 4 waves, no LDS reads", 2, out);
    run<2, 2, 8, 2, true>("2x2 tiles, 2 waves/SIMD (one WG)", 1, out);
    run<2, 4, 4, 1, true>("2x4 tiles, 1 wave/SIMD", 1, out);
    run<2, 2, 4, 1, true>("2x2 tiles, 1 wave/SIMD", 1, out);
    run<1, 4, 4, 1, true>("1x4 tiles, 1 wave/SIMD", 1, out);
    run<2, 2, 16, 4, true>("2x2 tiles, 4 waves/SIMD (one WG)", 1, out);
    // few accumulators per wave: the register-fed short-sequence kernel (1 or 2) and the 64 x 128 tiling of the 256-channel stage (2)
    run<1, 1, 4, 1, false>("1x1 tile (ONE accumulator), 1 wave/SIMD, no reads", 1, out);
    run<1, 1, 4, 1, true>("1x1 tile (ONE accumulator), 1 wave/SIMD", 1, out);
    run<2, 1, 4, 1, false>("2x1 tiles, 1 wave/SIMD, no reads", 1, out);
    run<2, 1, 4, 1, true>("2x1 tiles, 1 wave/SIMD", 1, out);
    run<2, 1, 8, 2, true>("2x1 tiles, 2 waves/SIMD (one WG)", 1, out);
    run<1, 1, 16, 4, true>("1x1 tile, 4 waves/SIMD (one WG)", 1, out);
    run<1, 2, 4, 1, true>("1x2 tiles, 1 wave/SIMD", 1, out);
  }
  return 0;
}
