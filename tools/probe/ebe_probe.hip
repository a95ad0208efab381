// probe: product with UNASSEMBLED symmetric element matrices (3D Q2: 27 nodes, 378 lower-triangle node-pair blocks
// of 3x3 per cell, stored [cell][e = 0..8][block = 0..377] so that a wave reads 512 contiguous bytes per load).
// One workgroup (384 threads) per cell of one colour: thread = block (a >= b): y_a += K_ab x_b, y_b += K_ab^T x_a,
// partial results through LDS, reduced in a fixed order, added to y (colouring => race free).
//   ./ebe_probe [cells per side = 59] [reps = 10]
#include <hip/hip_runtime.h>

#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CHK(x)                                                                                      \
  do                                                                                                \
    {                                                                                               \
      hipError_t e_ = (x);                                                                          \
      if (e_ != hipSuccess)                                                                         \
        {                                                                                           \
          printf("%s failed: %s (line %d)\n", #x, hipGetErrorString(e_), __LINE__);                 \
          exit(1);                                                                                  \
        }                                                                                           \
    }                                                                                               \
  while (0)

constexpr int NPC = 27, NBLK = NPC * (NPC + 1) / 2, NT = 384, ESTRIDE = NBLK; // 378 blocks

template <bool NTL, int MODE = 0>
__global__ __launch_bounds__(NT) void ebe_product(const double *__restrict__ ke, const int *__restrict__ conn,
                                                  const double *__restrict__ x, double *y, long cell0,
                                                  const int *__restrict__ lex = nullptr, double *ye = nullptr)
{
  __shared__ double s_x[NPC * 3];
  __shared__ double s_p[NBLK * 6 + 6];
  __shared__ int    s_conn[NPC];
  const int  tid  = threadIdx.x;
  const long cell = cell0 + blockIdx.x;
  const double *__restrict__ kp = ke + cell * (9L * ESTRIDE) + tid;
  double k[9];
  const bool act = tid < NBLK;
  if (act)
    {
#pragma unroll
      for (int e = 0; e < 9; ++e)
        k[e] = NTL ? __builtin_nontemporal_load(&kp[e * ESTRIDE]) : kp[e * ESTRIDE];
    }
  if (tid < NPC)
    s_conn[tid] = conn[cell * NPC + tid];
  __syncthreads();
  if (tid < NPC * 3)
    s_x[tid] = x[long(s_conn[tid / 3]) * 3 + tid % 3];
  __syncthreads();
  if (act)
    {
      // block index -> (a, b), a >= b
      int a = int((sqrtf(8.0f * float(tid) + 1.0f) - 1.0f) * 0.5f);
      while ((a + 1) * (a + 2) / 2 <= tid)
        ++a;
      while (a * (a + 1) / 2 > tid)
        --a;
      const int    b = tid - a * (a + 1) / 2;
      const double xa0 = s_x[a * 3], xa1 = s_x[a * 3 + 1], xa2 = s_x[a * 3 + 2];
      const double xb0 = s_x[b * 3], xb1 = s_x[b * 3 + 1], xb2 = s_x[b * 3 + 2];
      double *p = &s_p[tid * 6];
      p[0]      = k[0] * xb0 + k[1] * xb1 + k[2] * xb2; // (K_ab x_b)
      p[1]      = k[3] * xb0 + k[4] * xb1 + k[5] * xb2;
      p[2]      = k[6] * xb0 + k[7] * xb1 + k[8] * xb2;
      const double s = (a == b) ? 0.0 : 1.0;            // the diagonal block counts once
      p[3]      = s * (k[0] * xa0 + k[3] * xa1 + k[6] * xa2); // (K_ab^T x_a)
      p[4]      = s * (k[1] * xa0 + k[4] * xa1 + k[7] * xa2);
      p[5]      = s * (k[2] * xa0 + k[5] * xa1 + k[8] * xa2);
    }
  __syncthreads();
  if (tid < NPC * 3)
    {
      const int a = tid / 3, i = tid - a * 3;
      double    s = 0.0;
      for (int b = 0; b <= a; ++b)
        s += s_p[(a * (a + 1) / 2 + b) * 6 + i];
      for (int c = a + 1; c < NPC; ++c)
        s += s_p[(c * (c + 1) / 2 + a) * 6 + 3 + i];
      if (MODE == 0)
        y[long(s_conn[a]) * 3 + i] += s;
      else if (MODE == 1)
        ye[long(lex[cell]) * 81 + tid] = s;
      else if (MODE == 2)
        ye[cell * 81 + tid] = s;
      else
        __builtin_nontemporal_store(s, &ye[long(lex[cell]) * 81 + tid]);
    }
}

#include <hip/hip_cooperative_groups.h>
namespace cg = cooperative_groups;

// persistent variant: ONE cooperative launch per product; every workgroup walks its share of the cells of colour 0, then
// all workgroups meet at a grid-wide barrier, then colour 1, ...
struct Colours
{
  long begin[9];
};
__global__ __launch_bounds__(NT) void ebe_persistent(const double *__restrict__ ke, const int *__restrict__ conn,
                                                     const double *__restrict__ x, double *y, Colours cb)
{
  __shared__ double s_x[NPC * 3];
  __shared__ double s_p[NBLK * 6 + 6];
  __shared__ int    s_conn[NPC];
  cg::grid_group grid = cg::this_grid();
  const int  tid = threadIdx.x;
  const bool act = tid < NBLK;
  int a = 0, b = 0;
  if (act)
    {
      a = int((sqrtf(8.0f * float(tid) + 1.0f) - 1.0f) * 0.5f);
      while ((a + 1) * (a + 2) / 2 <= tid)
        ++a;
      while (a * (a + 1) / 2 > tid)
        --a;
      b = tid - a * (a + 1) / 2;
    }
  for (int col = 0; col < 8; ++col)
    {
      for (long cell = cb.begin[col] + blockIdx.x; cell < cb.begin[col + 1]; cell += gridDim.x)
        {
          const double *__restrict__ kp = ke + cell * (9L * ESTRIDE) + tid;
          double k[9];
          if (act)
            {
#pragma unroll
              for (int e = 0; e < 9; ++e)
                k[e] = __builtin_nontemporal_load(&kp[e * ESTRIDE]);
            }
          if (tid < NPC)
            s_conn[tid] = conn[cell * NPC + tid];
          __syncthreads();
          if (tid < NPC * 3)
            s_x[tid] = x[long(s_conn[tid / 3]) * 3 + tid % 3];
          __syncthreads();
          if (act)
            {
              const double xa0 = s_x[a * 3], xa1 = s_x[a * 3 + 1], xa2 = s_x[a * 3 + 2];
              const double xb0 = s_x[b * 3], xb1 = s_x[b * 3 + 1], xb2 = s_x[b * 3 + 2];
              double *p = &s_p[tid * 6];
              p[0] = k[0] * xb0 + k[1] * xb1 + k[2] * xb2;
              p[1] = k[3] * xb0 + k[4] * xb1 + k[5] * xb2;
              p[2] = k[6] * xb0 + k[7] * xb1 + k[8] * xb2;
              const double s = (a == b) ? 0.0 : 1.0;
              p[3] = s * (k[0] * xa0 + k[3] * xa1 + k[6] * xa2);
              p[4] = s * (k[1] * xa0 + k[4] * xa1 + k[7] * xa2);
              p[5] = s * (k[2] * xa0 + k[5] * xa1 + k[8] * xa2);
            }
          __syncthreads();
          if (tid < NPC * 3)
            {
              const int na = tid / 3, i = tid - na * 3;
              double    s = 0.0;
              for (int bb = 0; bb <= na; ++bb)
                s += s_p[(na * (na + 1) / 2 + bb) * 6 + i];
              for (int c = na + 1; c < NPC; ++c)
                s += s_p[(c * (c + 1) / 2 + na) * 6 + 3 + i];
              y[long(s_conn[na]) * 3 + i] += s;
            }
          __syncthreads();
        }
      if (col < 7)
        grid.sync();
    }
}

int main(int argc, char **argv)
{
  const int n = argc > 1 ? atoi(argv[1]) : 59, reps = argc > 2 ? atoi(argv[2]) : 10;
  const long nn = 2L * n + 1, nnodes = nn * nn * nn, ncells = long(n) * n * n;
  // colour-sorted cells
  std::vector<int>  conn(ncells * NPC);
  std::vector<long> cbegin(9, 0);
  long              pos = 0;
  for (int col = 0; col < 8; ++col)
    {
      cbegin[col] = pos;
      for (long c = 0; c < ncells; ++c)
        {
          const int ci[3] = {int(c % n), int((c / n) % n), int(c / (long(n) * n))};
          if (((ci[0] & 1) | ((ci[1] & 1) << 1) | ((ci[2] & 1) << 2)) != col)
            continue;
          for (int a = 0; a < NPC; ++a)
            {
              const int ai[3] = {a % 3, (a / 3) % 3, a / 9};
              conn[pos * NPC + a] = int((2 * ci[0] + ai[0]) + nn * ((2 * ci[1] + ai[1]) + nn * (2 * ci[2] + ai[2])));
            }
          ++pos;
        }
    }
  cbegin[8] = pos;
  const size_t nke = size_t(ncells) * 9 * ESTRIDE;
  printf("n = %d: %ld cells, %ld dofs, element matrices %.3f GB\n", n, ncells, nnodes * 3, nke * 8 / 1e9);
  double *d_ke, *d_x, *d_y;
  int    *d_conn;
  CHK(hipMalloc(&d_ke, nke * 8));
  CHK(hipMalloc(&d_x, nnodes * 3 * 8));
  CHK(hipMalloc(&d_y, nnodes * 3 * 8));
  CHK(hipMalloc(&d_conn, conn.size() * 4));
  CHK(hipMemcpy(d_conn, conn.data(), conn.size() * 4, hipMemcpyHostToDevice));
  {
    // fill: small mesh exactly (checked against the host), big mesh with a repeating pattern
    std::vector<double> chunk(size_t(1) << 22);
    for (size_t i = 0; i < chunk.size(); ++i)
      chunk[i] = 1e-3 * double((i * 2654435761u) % 1000) - 0.5;
    for (size_t off = 0; off < nke; off += chunk.size())
      CHK(hipMemcpy(d_ke + off, chunk.data(), std::min(chunk.size(), nke - off) * 8, hipMemcpyHostToDevice));
    std::vector<double> hx(nnodes * 3);
    for (size_t i = 0; i < hx.size(); ++i)
      hx[i] = std::sin(0.37 * double(i));
    CHK(hipMemcpy(d_x, hx.data(), hx.size() * 8, hipMemcpyHostToDevice));
    if (ncells <= 4096)
      {
        // host reference
        std::vector<double> hke(nke), ref(nnodes * 3, 0.0), got(nnodes * 3);
        CHK(hipMemcpy(hke.data(), d_ke, nke * 8, hipMemcpyDeviceToHost));
        for (long c = 0; c < ncells; ++c)
          for (int a = 0; a < NPC; ++a)
            for (int b = 0; b <= a; ++b)
              {
                const int blk = a * (a + 1) / 2 + b;
                for (int i = 0; i < 3; ++i)
                  for (int j = 0; j < 3; ++j)
                    {
                      const double v = hke[c * 9 * ESTRIDE + (i * 3 + j) * ESTRIDE + blk];
                      ref[long(conn[c * NPC + a]) * 3 + i] += v * hx[long(conn[c * NPC + b]) * 3 + j];
                      if (a != b)
                        ref[long(conn[c * NPC + b]) * 3 + j] += v * hx[long(conn[c * NPC + a]) * 3 + i];
                    }
              }
        CHK(hipMemset(d_y, 0, nnodes * 3 * 8));
        for (int col = 0; col < 8; ++col)
          if (cbegin[col + 1] > cbegin[col])
            hipLaunchKernelGGL(ebe_product<true>, dim3(cbegin[col + 1] - cbegin[col]), dim3(NT), 0, 0, d_ke, d_conn, d_x, d_y,
                               cbegin[col]);
        CHK(hipMemcpy(got.data(), d_y, got.size() * 8, hipMemcpyDeviceToHost));
        double err = 0, mx = 0;
        for (size_t i = 0; i < got.size(); ++i)
          {
            err = std::max(err, std::fabs(got[i] - ref[i]));
            mx  = std::max(mx, std::fabs(ref[i]));
          }
        printf("check vs host: max abs err %.3e (max |y| %.3e)\n", err, mx);
      }
  }
  hipEvent_t e0, e1;
  CHK(hipEventCreate(&e0));
  CHK(hipEventCreate(&e1));
  // per-cell result variants
  {
    std::vector<int> lexv(ncells);
    long p2 = 0;
    for (int col = 0; col < 8; ++col)
      for (long c = 0; c < ncells; ++c)
        {
          const int ci[3] = {int(c % n), int((c / n) % n), int(c / (long(n) * n))};
          if (((ci[0] & 1) | ((ci[1] & 1) << 1) | ((ci[2] & 1) << 2)) == col)
            lexv[p2++] = int(c);
        }
    int *d_lex;
    double *d_ye;
    CHK(hipMalloc(&d_lex, ncells * 4));
    CHK(hipMalloc(&d_ye, ncells * 81 * 8));
    CHK(hipMemcpy(d_lex, lexv.data(), ncells * 4, hipMemcpyHostToDevice));
    for (int variant = 0; variant < 6; ++variant)
      {
        float sum = 0;
        for (int r = 0; r < reps + 1; ++r)
          {
            CHK(hipEventRecord(e0, 0));
            const bool one = variant % 2 == 0;
            const int  mode = variant / 2 + 1;
            for (int col = 0; col < (one ? 1 : 8); ++col)
              {
                const long c0 = one ? 0 : cbegin[col], cn = one ? ncells : cbegin[col + 1] - cbegin[col];
                if (mode == 1)
                  hipLaunchKernelGGL((ebe_product<true, 1>), dim3(cn), dim3(NT), 0, 0, d_ke, d_conn, d_x, d_y, c0, d_lex, d_ye);
                else if (mode == 2)
                  hipLaunchKernelGGL((ebe_product<true, 2>), dim3(cn), dim3(NT), 0, 0, d_ke, d_conn, d_x, d_y, c0, d_lex, d_ye);
                else
                  hipLaunchKernelGGL((ebe_product<true, 3>), dim3(cn), dim3(NT), 0, 0, d_ke, d_conn, d_x, d_y, c0, d_lex, d_ye);
              }
            CHK(hipEventRecord(e1, 0));
            CHK(hipEventSynchronize(e1));
            float ms;
            CHK(hipEventElapsedTime(&ms, e0, e1));
            if (r > 0)
              sum += ms;
          }
        printf("per-cell results, %s, %s: avg %.3f ms\n", variant / 2 == 0 ? "lexicographic slot" : variant / 2 == 1 ? "own slot" : "lexicographic slot, nt store",
               variant % 2 == 0 ? "ONE launch" : "8 launches", sum / reps);
      }
  }
  {
    int per_cu = 0, ncu = 0;
    CHK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, ebe_persistent, NT, 0));
    hipDeviceProp_t prop;
    CHK(hipGetDeviceProperties(&prop, 0));
    ncu = prop.multiProcessorCount;
    Colours cb;
    for (int i = 0; i < 9; ++i)
      cb.begin[i] = cbegin[i];
    for (int frac = 1; frac <= 2; ++frac)
      {
        const int grid = ncu * per_cu / frac;
        float sum = 0;
        for (int r = 0; r < reps + 1; ++r)
          {
            CHK(hipMemsetAsync(d_y, 0, nnodes * 3 * 8, 0));
            CHK(hipEventRecord(e0, 0));
            void *args[] = {(void *)&d_ke, (void *)&d_conn, (void *)&d_x, (void *)&d_y, (void *)&cb};
            CHK(hipLaunchCooperativeKernel((const void *)ebe_persistent, dim3(grid), dim3(NT), args, 0, 0));
            CHK(hipEventRecord(e1, 0));
            CHK(hipEventSynchronize(e1));
            float ms;
            CHK(hipEventElapsedTime(&ms, e0, e1));
            if (r > 0)
              sum += ms;
          }
        printf("persistent cooperative launch, %d workgroups (%d per CU): avg %.3f ms per product\n", grid, per_cu / frac, sum / reps);
      }
  }
  for (int split = 1; split <= 2; split *= 2)
    {
      float sum = 0;
      for (int r = 0; r < reps + 1; ++r)
        {
          CHK(hipMemsetAsync(d_y, 0, nnodes * 3 * 8, 0));
          CHK(hipEventRecord(e0, 0));
          for (int col = 0; col < 8; ++col)
            {
              const long cn = cbegin[col + 1] - cbegin[col], per = (cn + split - 1) / split;
              for (long c0 = 0; c0 < cn; c0 += per)
                hipLaunchKernelGGL(ebe_product<true>, dim3(std::min(per, cn - c0)), dim3(NT), 0, 0, d_ke, d_conn, d_x, d_y,
                                   cbegin[col] + c0);
            }
          CHK(hipEventRecord(e1, 0));
          CHK(hipEventSynchronize(e1));
          float ms;
          CHK(hipEventElapsedTime(&ms, e0, e1));
          if (r > 0)
            sum += ms;
        }
      printf("RMW y, every colour in %d launches (%d per product): avg %.3f ms\n", split, 8 * split, sum / reps);
    }
  for (int variant = 0; variant < 2; ++variant)
    {
      float best = 1e30f, sum = 0;
      for (int r = 0; r < reps + 1; ++r)
        {
          CHK(hipMemsetAsync(d_y, 0, nnodes * 3 * 8, 0));
          CHK(hipEventRecord(e0, 0));
          for (int col = 0; col < 8; ++col)
            if (cbegin[col + 1] > cbegin[col])
              {
                if (variant == 0)
                  hipLaunchKernelGGL(ebe_product<false>, dim3(cbegin[col + 1] - cbegin[col]), dim3(NT), 0, 0, d_ke, d_conn,
                                     d_x, d_y, cbegin[col]);
                else
                  hipLaunchKernelGGL(ebe_product<true>, dim3(cbegin[col + 1] - cbegin[col]), dim3(NT), 0, 0, d_ke, d_conn,
                                     d_x, d_y, cbegin[col]);
              }
          CHK(hipEventRecord(e1, 0));
          CHK(hipEventSynchronize(e1));
          float ms;
          CHK(hipEventElapsedTime(&ms, e0, e1));
          if (r > 0)
            {
              best = std::min(best, ms);
              sum += ms;
            }
        }
      printf("%s loads: product (8 colour launches) avg %.3f ms, best %.3f ms -> %.0f GB/s of element-matrix bytes\n",
             variant ? "non-temporal" : "plain", sum / reps, best, nke * 8 / 1e6 / (sum / reps));
    }
  return 0;
}
