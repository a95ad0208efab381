// How long does one wave take per FP64 multiply-add / per lane-read + multiply-add group on gfx950, in s_memtime ticks and in
// nanoseconds (s_memrealtime: 100 MHz)?  Background for the pivot loop of band_cholesky_lds (profiles/r05).
//   hipcc --offload-arch=gfx950 -O3 -o fp64_issue fp64_issue.hip && ./fp64_issue
#include <hip/hip_runtime.h>
#include <cstdio>
__device__ __forceinline__ double lane_value(double v, int lane)
{
  const int lo = __builtin_amdgcn_readlane(__double2loint(v), lane), hi = __builtin_amdgcn_readlane(__double2hiint(v), lane);
  return __hiloint2double(hi, lo);
}
template <int MODE>
__global__ void probe(double *out, unsigned long long *t, int active_waves)
{
  const int wave = threadIdx.x >> 6;
  double    r[16];
  for (int i = 0; i < 16; ++i)
    r[i] = 1.0 + 1e-9 * (threadIdx.x + i);
  double x = 1.0 + 1e-12 * threadIdx.x;
  __syncthreads();
  const unsigned long long t0 = __builtin_amdgcn_s_memtime(), w0 = __builtin_amdgcn_s_memrealtime();
  if (wave < active_waves)
    for (int it = 0; it < 1000; ++it)
      {
        if (MODE == 0) // 16 independent multiply-adds
          {
#pragma unroll
            for (int i = 0; i < 16; ++i)
              r[i] = fma(r[i], x, 1e-9);
          }
        else if (MODE == 1) // 16 dependent multiply-adds
          {
#pragma unroll
            for (int i = 0; i < 16; ++i)
              x = fma(x, 1.0000001, 1e-9);
          }
        else if (MODE == 2) // 16 groups: lane read of a value + multiply-add with it
          {
#pragma unroll
            for (int i = 0; i < 16; ++i)
              r[i] = fma(-x, lane_value(x, i), r[i]);
            x += 1e-12;
          }
        else if (MODE == 4) // 16 groups, lane reads in batches of four ahead of their multiply-adds
          {
#pragma unroll
            for (int i0 = 0; i0 < 16; i0 += 4)
              {
                double l[4];
#pragma unroll
                for (int i = 0; i < 4; ++i)
                  l[i] = lane_value(x, i0 + i);
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int i = 0; i < 4; ++i)
                  r[i0 + i] = fma(-x, l[i], r[i0 + i]);
                __builtin_amdgcn_sched_barrier(0);
              }
            x += 1e-12;
          }
        else if (MODE == 5) // 16 groups through the LDS crossbar (ds_bpermute) instead of scalar lane reads
          {
#pragma unroll
            for (int i = 0; i < 16; ++i)
              {
                const int    lo = __builtin_amdgcn_ds_bpermute(4 * i, __double2loint(x)), hi = __builtin_amdgcn_ds_bpermute(4 * i, __double2hiint(x));
                r[i] = fma(-x, __hiloint2double(hi, lo), r[i]);
              }
            x += 1e-12;
          }
        else if (MODE == 6) // only the lane reads (32 v_readlane_b32), results summed on the scalar side
          {
            int acc = 0;
#pragma unroll
            for (int i = 0; i < 16; ++i)
              acc += __builtin_amdgcn_readlane(__double2loint(x), i) ^ __builtin_amdgcn_readlane(__double2hiint(x), i);
            x += 1e-12 * (acc & 1);
          }
        else if (MODE == 3) // 16 independent f32 multiply-adds (for scale)
          {
            float f[16];
#pragma unroll
            for (int i = 0; i < 16; ++i)
              f[i] = float(r[i]);
#pragma unroll
            for (int k = 0; k < 1; ++k)
#pragma unroll
              for (int i = 0; i < 16; ++i)
                f[i] = fmaf(f[i], 1.0001f, 1e-9f);
#pragma unroll
            for (int i = 0; i < 16; ++i)
              r[i] = f[i];
          }
      }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime(), w1 = __builtin_amdgcn_s_memrealtime();
  double s = x;
  for (int i = 0; i < 16; ++i)
    s += r[i];
  out[threadIdx.x] = s;
  if (threadIdx.x == 0)
    {
      t[0] = t1 - t0;
      t[1] = w1 - w0;
    }
}
int main()
{
  double             *out;
  unsigned long long *t, h[2];
  hipMalloc(&out, 1024 * 8);
  hipMalloc(&t, 16);
  const char *names[7] = {"16 independent f64 fma", "16 dependent f64 fma", "16 x (lane read + f64 fma)", "16 f64->f32 cvt + f32 fma + cvt back",
                          "16 x (lane read + f64 fma), reads 4 ahead", "16 x (ds_bpermute x 2 + f64 fma)", "16 x 2 v_readlane_b32 alone"};
  for (int mode = 0; mode < 7; ++mode)
    for (int cfg = 0; cfg < 1; ++cfg)
      {
        const int threads = cfg == 0 ? 64 : 1024, act = cfg == 0 ? 1 : (cfg == 1 ? 4 : 16);
        for (int rep = 0; rep < 2; ++rep)
          {
            if (mode == 0)
              hipLaunchKernelGGL(probe<0>, dim3(1), dim3(threads), 0, 0, out, t, act);
            else if (mode == 1)
              hipLaunchKernelGGL(probe<1>, dim3(1), dim3(threads), 0, 0, out, t, act);
            else if (mode == 2)
              hipLaunchKernelGGL(probe<2>, dim3(1), dim3(threads), 0, 0, out, t, act);
            else if (mode == 3)
              hipLaunchKernelGGL(probe<3>, dim3(1), dim3(threads), 0, 0, out, t, act);
            else if (mode == 4)
              hipLaunchKernelGGL(probe<4>, dim3(1), dim3(threads), 0, 0, out, t, act);
            else if (mode == 5)
              hipLaunchKernelGGL(probe<5>, dim3(1), dim3(threads), 0, 0, out, t, act);
            else
              hipLaunchKernelGGL(probe<6>, dim3(1), dim3(threads), 0, 0, out, t, act);
            hipDeviceSynchronize();
          }
        hipMemcpy(h, t, 16, hipMemcpyDeviceToHost);
        printf("%-40s %4d threads, %2d waves working: %7.2f ticks, %6.2f ns per instruction (group) of wave 0; %.0f ticks per us\n", names[mode], threads, act,
               double(h[0]) / 16000.0, double(h[1]) * 10.0 / 16000.0, double(h[0]) / (double(h[1]) * 0.01));
      }
  return 0;
}
