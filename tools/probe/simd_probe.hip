// simd_probe.hip -- on which SIMD of its CU does wave w of a 4-wave workgroup run?  (round 4: assemble_q2sf gives the
// serial prologue of every cell to wave 0; if wave 0 of every workgroup lands on the same SIMD, the three workgroups a CU
// holds serialise their prologues there while the other SIMDs wait.)
//   hipcc --offload-arch=gfx950 -O3 tools/probe/simd_probe.hip -o /tmp/simd_probe && /tmp/simd_probe
#include <hip/hip_runtime.h>

#include <cstdio>
#include <vector>

__global__ __launch_bounds__(256, 3) void probe(unsigned *out, int spin)
{
  __shared__ double big[6400]; // 51 kB like the element kernel: three workgroups per CU
  const int w = threadIdx.x >> 6;
  // HW_REG_HW_ID (4): wave 3:0, simd 5:4, pipe 7:6, cu 11:8, sh 12, se 15:13
  const unsigned hw = __builtin_amdgcn_s_getreg((31 << 11) | (0 << 6) | 4);
  double a = threadIdx.x;
  for (int i = 0; i < spin; ++i)
    a = a * 1.0000001 + 0.5;
  big[threadIdx.x] = a;
  __syncthreads();
  if ((threadIdx.x & 63) == 0)
    out[blockIdx.x * 4 + w] = hw | (big[(threadIdx.x + 64) & 255] == -1.0 ? 0x80000000u : 0u);
}

int main()
{
  const int nb = 25672;
  unsigned *d;
  hipMalloc(&d, nb * 4 * sizeof(unsigned));
  hipLaunchKernelGGL(probe, dim3(nb), dim3(256), 0, 0, d, 2000);
  hipDeviceSynchronize();
  std::vector<unsigned> h(nb * 4);
  hipMemcpy(h.data(), d, h.size() * sizeof(unsigned), hipMemcpyDeviceToHost);
  long cnt[4][4] = {};
  long same_simd_blocks = 0;
  for (int b = 0; b < nb; ++b)
    {
      for (int w = 0; w < 4; ++w)
        ++cnt[w][(h[b * 4 + w] >> 4) & 3];
      const unsigned s0 = (h[b * 4] >> 4) & 3;
      bool all_diff = true;
      for (int w = 1; w < 4; ++w)
        for (int v = 0; v < w; ++v)
          all_diff = all_diff && (((h[b * 4 + w] >> 4) & 3) != ((h[b * 4 + v] >> 4) & 3));
      same_simd_blocks += all_diff ? 0 : 1;
      (void)s0;
    }
  printf("wave -> simd histogram over %d workgroups of 4 waves\n", nb);
  for (int w = 0; w < 4; ++w)
    printf("  wave %d: simd0 %ld  simd1 %ld  simd2 %ld  simd3 %ld\n", w, cnt[w][0], cnt[w][1], cnt[w][2], cnt[w][3]);
  printf("workgroups with two waves on one SIMD: %ld\n", same_simd_blocks);
  for (int b = 0; b < 12; ++b)
    printf("  block %d: simd %u %u %u %u  cu %u se %u\n", b, (h[b * 4] >> 4) & 3, (h[b * 4 + 1] >> 4) & 3, (h[b * 4 + 2] >> 4) & 3,
           (h[b * 4 + 3] >> 4) & 3, (h[b * 4] >> 8) & 15, (h[b * 4] >> 13) & 7);
  return 0;
}
