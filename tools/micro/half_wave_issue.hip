// Does a wave64 VALU instruction cost less when only lanes 0..31 are active?  (k_sweep1 runs one path on 32 lanes of a wavefront.)
// One wavefront alone on its SIMD executes N fp64 FMAs (4 independent chains) with 64, 32 or 8 active lanes; cycles by s_memtime.
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/half_wave tools/micro/half_wave_issue.hip && /tmp/half_wave
#include <hip/hip_runtime.h>
#include <cstdio>

template <typename T>
__global__ void k(int lanes, int iters, double *out, unsigned long long *cyc)
{
   if ((int)(threadIdx.x & 63) >= lanes) return;
   T a = (T)threadIdx.x * (T)1e-3, b = a + 1, c = a + 2, d = a + 3;
   const T m = (T)1.0000001, q = (T)1e-9;
   const unsigned long long t0 = __builtin_readcyclecounter();
#pragma unroll 1
   for (int i = 0; i < iters; ++i)
   {
#pragma unroll
      for (int u = 0; u < 16; ++u)
      {
         a = __builtin_fma(a, m, q); b = __builtin_fma(b, m, q); c = __builtin_fma(c, m, q); d = __builtin_fma(d, m, q);
      }
   }
   const unsigned long long t1 = __builtin_readcyclecounter();
   out[threadIdx.x] = a + b + c + d;
   if (threadIdx.x == 0) *cyc = t1 - t0;
}

int main()
{
   double *out; unsigned long long *cyc, h;
   hipMalloc(&out, 64 * sizeof(double)); hipMalloc(&cyc, 8);
   const int iters = 20000;
   for (int rep = 0; rep < 2; ++rep)
      for (int lanes : {64, 32, 16, 8})
      {
         hipLaunchKernelGGL(k<double>, dim3(1), dim3(64), 0, 0, lanes, iters, out, cyc); hipMemcpy(&h, cyc, 8, hipMemcpyDeviceToHost);
         printf("fp64 fma, %2d active lanes: %.2f cycles per instruction\n", lanes, (double)h / (iters * 64.0));
         hipLaunchKernelGGL(k<float>, dim3(1), dim3(64), 0, 0, lanes, iters, out, cyc); hipMemcpy(&h, cyc, 8, hipMemcpyDeviceToHost);
         printf("fp32 fma, %2d active lanes: %.2f cycles per instruction\n", lanes, (double)h / (iters * 64.0));
      }
   return 0;
}
