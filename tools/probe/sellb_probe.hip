// probe: ONE matrix layout for assembly and SpMV.  The sliced-ELL product reads vals[((off+k)*9 + e)*64 + lane] (every
// wave load one contiguous 512-byte segment) -- a layout the element scatter cannot write (a 3x3 block would be nine
// 8-byte pieces 512 bytes apart), hence the block-CSR -> sliced-ELL copy of rounds 1-2.  Candidate single layout
// ("slice-interleaved block rows"): vals[((off+k)*64 + lane)*9 + e] -- a block stays 72 contiguous bytes for the scatter,
// and the 64 blocks a wave needs for one k are one contiguous 4608-byte chunk.  This probe times ways of reading it:
//   0  reference: the transposed layout, 9 coalesced 512-byte loads per k (the production kernel of round 2)
//   1  direct: every lane loads its own 72 bytes (stride 72 between lanes)
//   2  coalesced 16-byte loads -> registers -> LDS (linear image) -> ds_read_b64 at stride 72 bytes
//   3  LDS-DMA (global_load_lds_dwordx4, 1 KiB per wave instruction) -> ds_read_b64 at stride 72 bytes
//   ./sellb_probe [cells per side = 59] [reps = 20]
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "../../dealii-adapter_amd/csrc/mi_mesh.hpp"

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

struct P
{
  const int32_t *perm, *len, *rowbox;
  const int64_t *off;
  const double  *vals;
  const double  *x;
  double        *y;
  int32_t        nslices, nn0, nn1;
};

__device__ __forceinline__ double hash_val(int64_t i)
{
  uint64_t z = uint64_t(i) * 0x9E3779B97F4A7C15ull + 0x632BE59BD9B4E019ull;
  z ^= z >> 29;
  z *= 0xBF58476D1CE4E5B9ull;
  z ^= z >> 32;
  return double(int64_t(z >> 12) - (int64_t(1) << 51)) * (1.0 / double(int64_t(1) << 51));
}

// x-line-interleaved order (mi_mesh.hpp, val_layout 1): block (g, kx) of lane l of a slice with x-width wx at
// off*64 + g*64*wx + l*wx + kx; one thread per (slice, k, lane)
__global__ void fill_c(double *vc, const int64_t *off, const int32_t *len, const int32_t *wxs, int nslices)
{
  const int sl = blockIdx.x;
  if (sl >= nslices)
    return;
  const int     L = len[sl], wx = wxs[sl];
  const int64_t o = off[sl];
  for (int i = threadIdx.x; i < L * 64; i += blockDim.x)
    {
      const int     k = i / 64, lane = i - 64 * k, g = k / wx, kx = k - g * wx;
      const int64_t blk = o * 64 + int64_t(g) * 64 * wx + lane * wx + kx;
      for (int e = 0; e < 9; ++e)
        vc[blk * 9 + e] = hash_val(((o + k) * 64 + lane) * 9 + e); // the value the other layouts hold for (slice, k, lane, e)
    }
}

// logical entry (slice-block index sb = off+k, lane, e) -> both layouts
__global__ void fill(double *vt, double *vb, int64_t nblk64)
{
  const int64_t i = int64_t(blockIdx.x) * blockDim.x + threadIdx.x; // over nblk64*576
  if (i >= nblk64 * 576)
    return;
  const int64_t sb = i / 576;
  const int     r = int(i - sb * 576), lane = r / 9, e = r - lane * 9;
  const double  v = hash_val(i);
  vb[i]                              = v;
  vt[(sb * 9 + e) * 64 + lane]       = v;
}

struct ColGen
{
  int32_t cur, cx = 0, cy = 0, wx, wy, jy, jz;
  __device__ ColGen(const P &p, int64_t slot)
  {
    const int32_t b0 = p.rowbox[slot * 2], b1 = p.rowbox[slot * 2 + 1];
    cur              = b0;
    wx               = b1 & 255;
    wy               = (b1 >> 8) & 255;
    jy               = p.nn0 - wx;
    jz               = p.nn0 * (p.nn1 - wy);
  }
  __device__ __forceinline__ int32_t next()
  {
    const int32_t c = cur;
    ++cur;
    if (++cx == wx)
      {
        cx = 0;
        cur += jy;
        if (++cy == wy)
          {
            cy = 0;
            cur += jz;
          }
      }
    return c;
  }
};

#define FMA9(acc, v, xx)                                  \
  _Pragma("unroll") for (int i_ = 0; i_ < 3; ++i_)        \
    _Pragma("unroll") for (int j_ = 0; j_ < 3; ++j_)      \
      acc[i_] += v[i_ * 3 + j_] * xx[j_];

// ---------------------------------------------------------------- 0: transposed layout (reference)
template <int U>
__global__ __launch_bounds__(256) void k_ref(P p)
{
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int per  = (p.nslices + gridDim.x - 1) / gridDim.x;
  const int s0 = blockIdx.x * per, s1 = min(p.nslices, s0 + per);
  for (int sl0 = s0 + wave; sl0 < s1; sl0 += 4)
    {
      const int     sl  = __builtin_amdgcn_readfirstlane(sl0);
      const int     len = p.len[sl];
      const int64_t off = p.off[sl];
      const int     node = p.perm[int64_t(sl) * 64 + lane];
      const double *__restrict__ vp = p.vals + off * 576 + lane;
      ColGen g(p, int64_t(sl) * 64 + lane);
      double acc[3] = {0, 0, 0};
      int    k = 0;
      for (; k + U <= len; k += U)
        {
          int32_t c[U];
          double  v[U][9], xx[U][3];
#pragma unroll
          for (int u = 0; u < U; ++u)
            c[u] = g.next();
#pragma unroll
          for (int u = 0; u < U; ++u)
#pragma unroll
            for (int e = 0; e < 9; ++e)
              v[u][e] = __builtin_nontemporal_load(&vp[(int64_t(k + u) * 9 + e) * 64]);
#pragma unroll
          for (int u = 0; u < U; ++u)
#pragma unroll
            for (int j = 0; j < 3; ++j)
              xx[u][j] = p.x[int64_t(c[u]) * 3 + j];
#pragma unroll
          for (int u = 0; u < U; ++u)
            {
              FMA9(acc, v[u], xx[u]);
            }
        }
      for (; k < len; ++k)
        {
          const int32_t c = g.next();
          double        v[9], xx[3];
#pragma unroll
          for (int e = 0; e < 9; ++e)
            v[e] = vp[(int64_t(k) * 9 + e) * 64];
#pragma unroll
          for (int j = 0; j < 3; ++j)
            xx[j] = p.x[int64_t(c) * 3 + j];
          FMA9(acc, v, xx);
        }
      if (node >= 0)
#pragma unroll
        for (int i = 0; i < 3; ++i)
          p.y[int64_t(node) * 3 + i] = acc[i];
    }
}

// ---------------------------------------------------------------- 1: block layout, every lane reads its own 72 bytes
template <int U, bool NT>
__global__ __launch_bounds__(256) void k_direct(P p)
{
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int per  = (p.nslices + gridDim.x - 1) / gridDim.x;
  const int s0 = blockIdx.x * per, s1 = min(p.nslices, s0 + per);
  for (int sl0 = s0 + wave; sl0 < s1; sl0 += 4)
    {
      const int     sl  = __builtin_amdgcn_readfirstlane(sl0);
      const int     len = p.len[sl];
      const int64_t off = p.off[sl];
      const int     node = p.perm[int64_t(sl) * 64 + lane];
      const double *__restrict__ vp = p.vals + off * 576 + lane * 9;
      ColGen g(p, int64_t(sl) * 64 + lane);
      double acc[3] = {0, 0, 0};
      int    k = 0;
      for (; k + U <= len; k += U)
        {
          int32_t c[U];
          double  v[U][9], xx[U][3];
#pragma unroll
          for (int u = 0; u < U; ++u)
            c[u] = g.next();
#pragma unroll
          for (int u = 0; u < U; ++u)
#pragma unroll
            for (int e = 0; e < 9; ++e)
              v[u][e] = NT ? __builtin_nontemporal_load(&vp[int64_t(k + u) * 576 + e]) : vp[int64_t(k + u) * 576 + e];
#pragma unroll
          for (int u = 0; u < U; ++u)
#pragma unroll
            for (int j = 0; j < 3; ++j)
              xx[u][j] = p.x[int64_t(c[u]) * 3 + j];
#pragma unroll
          for (int u = 0; u < U; ++u)
            {
              FMA9(acc, v[u], xx[u]);
            }
        }
      for (; k < len; ++k)
        {
          const int32_t c = g.next();
          double        v[9], xx[3];
#pragma unroll
          for (int e = 0; e < 9; ++e)
            v[e] = vp[int64_t(k) * 576 + e];
#pragma unroll
          for (int j = 0; j < 3; ++j)
            xx[j] = p.x[int64_t(c) * 3 + j];
          FMA9(acc, v, xx);
        }
      if (node >= 0)
#pragma unroll
        for (int i = 0; i < 3; ++i)
          p.y[int64_t(node) * 3 + i] = acc[i];
    }
}

#define WAVE_SYNC()                                               \
  do                                                              \
    {                                                             \
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");      \
      __builtin_amdgcn_wave_barrier();                            \
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");      \
    }                                                             \
  while (0)

// ---------------------------------------------------------------- 2: coalesced loads, transposed through LDS (register staging)
// a group of 2 k-steps = 9216 contiguous bytes = 9 x (64 lanes x 16 bytes)
template <int WPB>
__global__ __launch_bounds__(WPB * 64) void k_lds(P p)
{
  __shared__ __attribute__((aligned(16))) double s_buf[WPB][1152];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int per  = (p.nslices + gridDim.x - 1) / gridDim.x;
  const int s0 = blockIdx.x * per, s1 = min(p.nslices, s0 + per);
  double   *sb = s_buf[wave];
  typedef const volatile __attribute__((address_space(3))) double *lds_cvp;
  for (int sl0 = s0 + wave; sl0 < s1; sl0 += WPB)
    {
      const int     sl  = __builtin_amdgcn_readfirstlane(sl0);
      const int     len = p.len[sl];
      const int64_t off = p.off[sl];
      const int     node = p.perm[int64_t(sl) * 64 + lane];
      typedef double v2d __attribute__((ext_vector_type(2)));
      const v2d *__restrict__ gp = reinterpret_cast<const v2d *>(p.vals + off * 576) + lane;
      ColGen g(p, int64_t(sl) * 64 + lane);
      double acc[3] = {0, 0, 0};
      int    k = 0;
      for (; k + 2 <= len; k += 2)
        {
          v2d t[9];
#pragma unroll
          for (int j = 0; j < 9; ++j)
            t[j] = __builtin_nontemporal_load(&gp[int64_t(k) * 288 + j * 64]);
          int32_t c[2];
          double  xx[2][3];
#pragma unroll
          for (int u = 0; u < 2; ++u)
            c[u] = g.next();
#pragma unroll
          for (int u = 0; u < 2; ++u)
#pragma unroll
            for (int j = 0; j < 3; ++j)
              xx[u][j] = p.x[int64_t(c[u]) * 3 + j];
          WAVE_SYNC();
#pragma unroll
          for (int j = 0; j < 9; ++j)
            reinterpret_cast<v2d *>(sb)[j * 64 + lane] = t[j];
          WAVE_SYNC();
#pragma unroll
          for (int u = 0; u < 2; ++u)
            {
              double v[9];
#pragma unroll
              for (int e = 0; e < 9; ++e)
                v[e] = ((lds_cvp)sb)[u * 576 + lane * 9 + e];
              FMA9(acc, v, xx[u]);
            }
        }
      for (; k < len; ++k) // odd tail: direct
        {
          const int32_t c = g.next();
          double        v[9], xx[3];
#pragma unroll
          for (int e = 0; e < 9; ++e)
            v[e] = p.vals[(off + k) * 576 + lane * 9 + e];
#pragma unroll
          for (int j = 0; j < 3; ++j)
            xx[j] = p.x[int64_t(c) * 3 + j];
          FMA9(acc, v, xx);
        }
      if (node >= 0)
#pragma unroll
        for (int i = 0; i < 3; ++i)
          p.y[int64_t(node) * 3 + i] = acc[i];
    }
}

// ---------------------------------------------------------------- 3: LDS-DMA (global_load_lds_dwordx4), G k-steps per group
// group of G = 2 k-steps: 9 KiB = 9 DMA instructions; the odd tail k-step: 4 full + 1 half-masked DMA
template <int WPB, int NBUF>
__global__ __launch_bounds__(WPB * 64) void k_glds(P p)
{
  __shared__ __attribute__((aligned(16))) double s_buf[WPB][NBUF][1152];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int per  = (p.nslices + gridDim.x - 1) / gridDim.x;
  const int s0 = blockIdx.x * per, s1 = min(p.nslices, s0 + per);
  typedef const volatile __attribute__((address_space(3))) double *lds_cvp;
  typedef __attribute__((address_space(3))) void                  *lds_vp;
  typedef const __attribute__((address_space(1))) void            *glb_vp;
  for (int sl0 = s0 + wave; sl0 < s1; sl0 += WPB)
    {
      const int     sl  = __builtin_amdgcn_readfirstlane(sl0);
      const int     len = p.len[sl];
      const int64_t off = p.off[sl];
      const int     node = p.perm[int64_t(sl) * 64 + lane];
      const char *__restrict__ gbase = reinterpret_cast<const char *>(p.vals + off * 576) + lane * 16;
      ColGen g(p, int64_t(sl) * 64 + lane);
      double acc[3] = {0, 0, 0};
      const int ngroups = (len + 1) / 2;
      auto issue = [&](int grp, int buf) {
        const int  kk   = grp * 2;
        const bool full = kk + 2 <= len;
        double    *dst  = s_buf[wave][buf];
        const char *src = gbase + int64_t(kk) * 4608;
        if (full)
          {
#pragma unroll
            for (int j = 0; j < 9; ++j)
              __builtin_amdgcn_global_load_lds((glb_vp)(src + j * 1024), (lds_vp)(dst + j * 128), 16, 0, 2);
          }
        else
          {
#pragma unroll
            for (int j = 0; j < 4; ++j)
              __builtin_amdgcn_global_load_lds((glb_vp)(src + j * 1024), (lds_vp)(dst + j * 128), 16, 0, 2);
            if (lane < 32)
              __builtin_amdgcn_global_load_lds((glb_vp)(src + 4 * 1024), (lds_vp)(dst + 4 * 128), 16, 0, 2);
          }
      };
      issue(0, 0);
      for (int grp = 0; grp < ngroups; ++grp)
        {
          const int buf = NBUF == 1 ? 0 : (grp % NBUF);
          const int nk  = min(2, len - grp * 2);
          int32_t   c[2];
          double    xx[2][3];
#pragma unroll
          for (int u = 0; u < 2; ++u)
            if (u < nk)
              {
                c[u] = g.next();
#pragma unroll
                for (int j = 0; j < 3; ++j)
                  xx[u][j] = p.x[int64_t(c[u]) * 3 + j];
              }
          if (NBUF > 1 && grp + 1 < ngroups)
            issue(grp + 1, (grp + 1) % NBUF);
          // everything this wave has in flight lands before the reads (the compiler waits vmcnt(0) for x anyway)
          asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
          WAVE_SYNC();
          const double *sb = s_buf[wave][buf];
#pragma unroll
          for (int u = 0; u < 2; ++u)
            if (u < nk)
              {
                double v[9];
#pragma unroll
                for (int e = 0; e < 9; ++e)
                  v[e] = ((lds_cvp)sb)[u * 576 + lane * 9 + e];
                FMA9(acc, v, xx[u]);
              }
          WAVE_SYNC();
          if (NBUF == 1 && grp + 1 < ngroups)
            issue(grp + 1, 0);
        }
      if (node >= 0)
#pragma unroll
        for (int i = 0; i < 3; ++i)
          p.y[int64_t(node) * 3 + i] = acc[i];
    }
}

// ---------------------------------------------------------------- 4: x-line-interleaved order, LDS-DMA of a whole group
// group g of a slice = 64 lanes x wx blocks = one contiguous chunk of 64*wx*72 bytes (13.5 / 22.5 KiB); the columns of a
// group are consecutive nodes, so x is one contiguous run of 3*wx doubles per lane
template <int WX>
__device__ __forceinline__ void sellc_slice(const P &p, const int32_t *wxs, int sl, int lane, double *stage)
{
  typedef const volatile __attribute__((address_space(3))) double *lds_cvp;
  typedef __attribute__((address_space(3))) void                  *lds_vp;
  typedef const __attribute__((address_space(1))) void            *glb_vp;
  constexpr int CH = 64 * WX * 72, NFULL = CH / 1024, REM = (CH % 1024) / 16; // REM lanes in the last, partial DMA
  const int     len = p.len[sl], ng = len / WX;
  const int64_t off = p.off[sl];
  const int     node = p.perm[int64_t(sl) * 64 + lane];
  const char *__restrict__ gbase = reinterpret_cast<const char *>(p.vals + off * 576) + lane * 16;
  // column box of the row: first column, widths
  const int32_t b0 = p.rowbox[(int64_t(sl) * 64 + lane) * 2], b1 = p.rowbox[(int64_t(sl) * 64 + lane) * 2 + 1];
  const int     wy = (b1 >> 8) & 255;
  int32_t       c0 = b0;
  int           gy = 0;
  double        acc[3] = {0, 0, 0};
  const lds_cvp rd = (lds_cvp)(stage + lane * (WX * 9));
  for (int g = 0; g < ng; ++g)
    {
      const char *src = gbase + int64_t(g) * CH;
#pragma unroll
      for (int j = 0; j < NFULL; ++j)
        __builtin_amdgcn_global_load_lds((glb_vp)(src + j * 1024), (lds_vp)(stage + j * 128), 16, 0, 2);
      if (REM > 0 && lane < REM)
        __builtin_amdgcn_global_load_lds((glb_vp)(src + NFULL * 1024), (lds_vp)(stage + NFULL * 128), 16, 0, 2);
      double xx[WX * 3];
#pragma unroll
      for (int j = 0; j < WX * 3; ++j)
        xx[j] = p.x[int64_t(c0) * 3 + j];
      // next group's first column: next x-line of the box
      c0 += p.nn0;
      if (++gy == wy)
        {
          gy = 0;
          c0 += p.nn0 * (p.nn1 - wy);
        }
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      WAVE_SYNC();
#pragma unroll
      for (int kx = 0; kx < WX; ++kx)
        {
          double v[9];
#pragma unroll
          for (int e = 0; e < 9; ++e)
            v[e] = rd[kx * 9 + e];
          FMA9(acc, v, (&xx[kx * 3]));
        }
      WAVE_SYNC();
    }
  if (node >= 0)
#pragma unroll
    for (int i = 0; i < 3; ++i)
      p.y[int64_t(node) * 3 + i] = acc[i];
}

template <int WPB>
__global__ __launch_bounds__(WPB * 64) void k_sellc(P p, const int32_t *wxs)
{
  __shared__ __attribute__((aligned(16))) double s_buf[WPB][2880]; // 23,040 bytes per wave
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int per  = (p.nslices + gridDim.x - 1) / gridDim.x;
  const int s0 = blockIdx.x * per, s1 = min(p.nslices, s0 + per);
  for (int sl0 = s0 + wave; sl0 < s1; sl0 += WPB)
    {
      const int sl = __builtin_amdgcn_readfirstlane(sl0);
      if (wxs[sl] == 3)
        sellc_slice<3>(p, wxs, sl, lane, s_buf[wave]);
      else
        sellc_slice<5>(p, wxs, sl, lane, s_buf[wave]);
    }
}

int main(int argc, char **argv)
{
  const int n    = argc > 1 ? atoi(argv[1]) : 59;
  const int reps = argc > 2 ? atoi(argv[2]) : 20;
  mi::HostMesh m;
  const int    rp[3] = {n, n, n}, role[6] = {1, 7, 7, 7, 7, 7};
  const double lo[3] = {0, 0, 0}, hi[3] = {1, 1, 1};
  m.build(3, 2, rp, lo, hi, role, nullptr);
  printf("mesh %d^3 Q2: %ld nodes, %ld blocks, %ld slices, %ld slice-blocks x 64 (%.2f GB of values)\n", n, (long)m.nnodes,
         (long)m.nnzb, (long)m.sell_nslices, (long)m.sell_nblk64, double(m.sell_nblk64) * 576 * 8 / 1e9);
  // free what the probe does not need
  std::vector<uint16_t>().swap(m.off);
  int32_t *d_perm, *d_len, *d_box;
  int64_t *d_off;
  double  *d_vt, *d_vb, *d_x, *d_y0, *d_y;
  const size_t nv = size_t(m.sell_nblk64) * 576;
  CHK(hipMalloc(&d_perm, m.sell_perm.size() * 4));
  CHK(hipMalloc(&d_len, m.sell_len.size() * 4));
  CHK(hipMalloc(&d_box, m.sell_box.size() * 4));
  CHK(hipMalloc(&d_off, m.sell_off.size() * 8));
  double  *d_vc;
  int32_t *d_wx;
  CHK(hipMalloc(&d_vc, nv * 8));
  CHK(hipMemset(d_vc, 0, nv * 8));
  CHK(hipMalloc(&d_wx, m.sell_wx.size() * 4));
  CHK(hipMemcpy(d_wx, m.sell_wx.data(), m.sell_wx.size() * 4, hipMemcpyHostToDevice));
  CHK(hipMalloc(&d_vt, nv * 8));
  CHK(hipMalloc(&d_vb, nv * 8));
  CHK(hipMalloc(&d_x, size_t(m.ndofs) * 8));
  CHK(hipMalloc(&d_y0, size_t(m.ndofs) * 8));
  CHK(hipMalloc(&d_y, size_t(m.ndofs) * 8));
  CHK(hipMemcpy(d_perm, m.sell_perm.data(), m.sell_perm.size() * 4, hipMemcpyHostToDevice));
  CHK(hipMemcpy(d_len, m.sell_len.data(), m.sell_len.size() * 4, hipMemcpyHostToDevice));
  CHK(hipMemcpy(d_box, m.sell_box.data(), m.sell_box.size() * 4, hipMemcpyHostToDevice));
  CHK(hipMemcpy(d_off, m.sell_off.data(), m.sell_off.size() * 8, hipMemcpyHostToDevice));
  std::vector<double> hx(size_t(m.ndofs));
  for (size_t i = 0; i < hx.size(); ++i)
    hx[i] = std::sin(0.001 * double(i)) + 0.5;
  CHK(hipMemcpy(d_x, hx.data(), hx.size() * 8, hipMemcpyHostToDevice));
  hipLaunchKernelGGL(fill, dim3(unsigned((nv + 255) / 256)), dim3(256), 0, 0, d_vt, d_vb, int64_t(m.sell_nblk64));
  hipLaunchKernelGGL(fill_c, dim3(unsigned(m.sell_nslices)), dim3(256), 0, 0, d_vc, d_off, d_len, d_wx, int(m.sell_nslices));
  CHK(hipDeviceSynchronize());

  P p{d_perm, d_len, d_box, d_off, d_vt, d_x, d_y0, int32_t(m.sell_nslices), m.nn[0], m.nn[1]};
  const double gbytes = (double(nv) * 8 + double(m.nnodes) * (12 + 48)) / 1e9;
  hipEvent_t   e0, e1;
  CHK(hipEventCreate(&e0));
  CHK(hipEventCreate(&e1));
  std::vector<double> y0(size_t(m.ndofs)), y1(size_t(m.ndofs));

  const double *which_vals = nullptr;
  auto run = [&](const char *name, auto launch, bool reference) {
    P q = p;
    q.vals = which_vals ? which_vals : (reference ? d_vt : d_vb);
    q.y    = reference ? d_y0 : d_y;
    CHK(hipMemset(q.y, 0, size_t(m.ndofs) * 8));
    launch(q);
    CHK(hipDeviceSynchronize());
    float best = 1e30f, sum = 0;
    for (int r = 0; r < reps; ++r)
      {
        CHK(hipEventRecord(e0, 0));
        launch(q);
        CHK(hipEventRecord(e1, 0));
        CHK(hipEventSynchronize(e1));
        float ms;
        CHK(hipEventElapsedTime(&ms, e0, e1));
        best = std::min(best, ms);
        sum += ms;
      }
    double err = 0;
    if (reference)
      CHK(hipMemcpy(y0.data(), d_y0, y0.size() * 8, hipMemcpyDeviceToHost));
    else
      {
        CHK(hipMemcpy(y1.data(), d_y, y1.size() * 8, hipMemcpyDeviceToHost));
        for (size_t i = 0; i < y0.size(); ++i)
          err = std::max(err, std::fabs(y0[i] - y1[i]));
      }
    printf("%-44s avg %.3f ms  best %.3f ms  %.2f TB/s   max|dy| %.1e\n", name, sum / reps, best, gbytes / (sum / reps), err);
  };
  const int ns = int(m.sell_nslices);
  for (int grid : {(ns + 3) / 4, (ns + 7) / 8})
    {
      printf("grid %d\n", grid);
      run("0 transposed layout U=2 (round 2)", [&](const P &q) { hipLaunchKernelGGL((k_ref<2>), dim3(grid), dim3(256), 0, 0, q); }, true);
      run("1 direct 72 B per lane U=2", [&](const P &q) { hipLaunchKernelGGL((k_direct<2, false>), dim3(grid), dim3(256), 0, 0, q); }, false);
      run("1 direct 72 B per lane U=2 nt", [&](const P &q) { hipLaunchKernelGGL((k_direct<2, true>), dim3(grid), dim3(256), 0, 0, q); }, false);
      run("1 direct 72 B per lane U=4 nt", [&](const P &q) { hipLaunchKernelGGL((k_direct<4, true>), dim3(grid), dim3(256), 0, 0, q); }, false);
      run("2 registers -> LDS, 4 waves", [&](const P &q) { hipLaunchKernelGGL((k_lds<4>), dim3(grid), dim3(256), 0, 0, q); }, false);
      run("3 LDS-DMA 1 buffer, 4 waves", [&](const P &q) { hipLaunchKernelGGL((k_glds<4, 1>), dim3(grid), dim3(256), 0, 0, q); }, false);
      run("3 LDS-DMA 2 buffers, 4 waves", [&](const P &q) { hipLaunchKernelGGL((k_glds<4, 2>), dim3(grid), dim3(256), 0, 0, q); }, false);
      which_vals = d_vc;
      run("4 x-line order, LDS-DMA per group, 1 wave/WG", [&](const P &q) { hipLaunchKernelGGL((k_sellc<1>), dim3(grid * 4), dim3(64), 0, 0, q, d_wx); }, false);
      run("4 x-line order, LDS-DMA per group, 2 waves/WG", [&](const P &q) { hipLaunchKernelGGL((k_sellc<2>), dim3(grid * 2), dim3(128), 0, 0, q, d_wx); }, false);
      run("4 x-line order, LDS-DMA per group, 3 waves/WG", [&](const P &q) { hipLaunchKernelGGL((k_sellc<3>), dim3((grid * 4 + 2) / 3), dim3(192), 0, 0, q, d_wx); }, false);
      which_vals = nullptr;
    }
  return 0;
}
