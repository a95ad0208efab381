// probe: can a workgroup use more than 64 KB of static LDS on this device?
#include <hip/hip_runtime.h>
#include <cstdio>
template <int KB>
__global__ __launch_bounds__(512) void k(double *out)
{
  __shared__ double s[KB * 128];
  for (int i = threadIdx.x; i < KB * 128; i += 512)
    s[i] = i;
  __syncthreads();
  double t = 0;
  for (int i = threadIdx.x; i < KB * 128; i += 512)
    t += s[(i * 7) % (KB * 128)];
  out[blockIdx.x * 512 + threadIdx.x] = t;
}
int main()
{
  double *d;
  hipMalloc(&d, 512 * 8 * 4);
  hipLaunchKernelGGL(k<48>, dim3(4), dim3(512), 0, 0, d);
  printf("48 KB: %s / %s\n", hipGetErrorString(hipGetLastError()), hipGetErrorString(hipDeviceSynchronize()));
  hipLaunchKernelGGL(k<83>, dim3(4), dim3(512), 0, 0, d);
  printf("83 KB: %s / %s\n", hipGetErrorString(hipGetLastError()), hipGetErrorString(hipDeviceSynchronize()));
  hipLaunchKernelGGL(k<150>, dim3(4), dim3(512), 0, 0, d);
  printf("150 KB: %s / %s\n", hipGetErrorString(hipGetLastError()), hipGetErrorString(hipDeviceSynchronize()));
  return 0;
}
