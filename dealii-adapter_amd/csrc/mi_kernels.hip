// mi_kernels.hip -- hand-written gfx950 kernels of the hot path (fp64, HBM/LDS/VALU; no MFMA by design).
//
//   assemble_cells   : per-cell tangent + residual of the neo-Hookean Newmark problem, one workgroup per
//                      cell, scattered by graph colouring into the block rows of the global tangent (stored in the
//                      slice-interleaved order the SpMV reads: one array for assembly and product)
//                      (nonlinear_elasticity.cc:872-1036 + :760-774; maths restated in DESIGN.md section 4)
//   neumann_faces    : interface traction with area pull-back incl. the reference's cell-QP quirk (:791-859)
//   sell_spmv        : y = K x on the sliced-ELL rows of the tangent (lane = row), fused dot product
//   blockrow_spmv_check : y = K x one wavefront per block row through the block pattern (cross-check)
//   cg_* / vec_*     : fused CG vector updates with deterministic two-level reductions (:1153-1191)
//   newmark_*        : Newmark predictor/corrector vector updates (:592-622)
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <stdint.h>

#include <type_traits>

#include "mi_kernels.h"

namespace mi
{
  __device__ __forceinline__ int64_t imin64(int64_t a, int64_t b)
  {
    return a < b ? a : b;
  }

  // ------------------------------------------------------------------ reductions
  __device__ __forceinline__ double wave_sum(double v)
  {
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1)
      v += __shfl_xor(v, o, 64);
    return v;
  }

  // The same sum through the DPP network (row shifts inside rows of 16 lanes, then the two row broadcasts): about 90 clocks
  // against ~900 for the six dependent trips through the LDS crossbar above.  The total arrives in lane 63 ONLY.
  template <int CTRL, int ROW_MASK, int BANK_MASK>
  __device__ __forceinline__ double dpp_add(double v)
  {
    const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(v), CTRL, ROW_MASK, BANK_MASK, false);
    const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(v), CTRL, ROW_MASK, BANK_MASK, false);
    return v + __hiloint2double(hi, lo); // (lanes without a source or masked out add 0)
  }
  __device__ __forceinline__ double wave_sum_lane63(double v)
  {
    v = dpp_add<0x111, 0xf, 0xf>(v); // row_shr:1
    v = dpp_add<0x112, 0xf, 0xf>(v); // row_shr:2
    v = dpp_add<0x114, 0xf, 0xe>(v); // row_shr:4
    v = dpp_add<0x118, 0xf, 0xc>(v); // row_shr:8   -> lane 15 of a row: the row's sum
    v = dpp_add<0x142, 0xa, 0xf>(v); // row_bcast:15 -> rows 1 and 3 take the row before
    v = dpp_add<0x143, 0xc, 0xf>(v); // row_bcast:31 -> rows 2 and 3 take lane 31
    return v;
  }

  // sum over the workgroup, result valid in every thread; s_red needs blockDim/64 doubles
  template <int NT>
  __device__ __forceinline__ double block_sum(double v, double *s_red)
  {
    constexpr int NW = NT / 64;
    v                = wave_sum(v);
    if constexpr (NW == 1)
      return v;
    __syncthreads();
    if ((threadIdx.x & 63) == 0)
      s_red[threadIdx.x >> 6] = v;
    __syncthreads();
    double t = 0;
#pragma unroll
    for (int w = 0; w < NW; ++w)
      t += s_red[w];
    return t;
  }

  // every workgroup reduces the same partials in the same order -> identical totals everywhere
  template <int NT>
  __device__ __forceinline__ double reduce_partials(const double *__restrict__ part, int n, double *s_red)
  {
    double s = 0;
    for (int i = threadIdx.x; i < n; i += NT)
      s += part[i];
    return block_sum<NT>(s, s_red);
  }

  // ------------------------------------------------------------------ element traits
  template <int DIM, int P>
  struct Elem
  {
    static constexpr int NP1  = P + 1;
    static constexpr int NPC  = (DIM == 2) ? NP1 * NP1 : NP1 * NP1 * NP1;
    static constexpr int NQ1  = P + 2; // qf_cell(p+2), nonlinear_elasticity.cc:74
    static constexpr int NQ   = (DIM == 2) ? NQ1 * NQ1 : NQ1 * NQ1 * NQ1;
    static constexpr int NQF  = (DIM == 2) ? NQ1 : NQ1 * NQ1;
    static constexpr int NV   = 1 << DIM;
    static constexpr int DD   = DIM * DIM;
    static constexpr int NT2  = (NPC + 1) / 2;        // node pairs along one side of the 2x2-block tile grid
    static constexpr int NPCP = 2 * NT2;              // node count padded to even
    static constexpr int NTILES = NT2 * (NT2 + 1) / 2; // lower triangle of the tile grid
  };

  constexpr int RQ = 32; // doubles per quadrature-point record in LDS
  constexpr int RN = 10; // doubles per (qp,node) record: g[3], m[3], t[3] = tau g, n
#ifndef MI_NDPAD
#define MI_NDPAD 2
#endif
  constexpr int NDPAD = MI_NDPAD; // see assemble_cells
  // quadrature-point record layout
  constexpr int Q_M    = 0;  // 9: Jinv * Finv  (unit gradient -> spatial gradient)
  constexpr int Q_TAU  = 9;  // 6: tau      xx yy zz xy xz yz
  constexpr int Q_TISO = 15; // 6: tau_iso
  constexpr int Q_W    = 21; // JxW
  constexpr int Q_WCII = 22; // JxW * c_II
  constexpr int Q_CS2  = 23; // c_S / 2
  constexpr int Q_SQN  = 24; // sqrt(alpha1 rho JxW)
  constexpr int Q_NINV = 25; // 1 / Q_SQN (0 if rho == 0)
  constexpr int Q_FACC = 26; // 3: rho JxW (acc - b)

  // (templated on the scalar type: double everywhere except the opt-in fp32 form of the matrix-free smoother product)
  template <typename T>
  __device__ __forceinline__ T det3x3(const T *A)
  {
    return A[0] * (A[4] * A[8] - A[5] * A[7]) - A[1] * (A[3] * A[8] - A[5] * A[6]) + A[2] * (A[3] * A[7] - A[4] * A[6]);
  }
  template <typename T>
  __device__ __forceinline__ void inv3x3r(const T *A, T r, T *B) // r = 1 / det A
  {
    B[0]           = (A[4] * A[8] - A[5] * A[7]) * r;
    B[1]           = (A[2] * A[7] - A[1] * A[8]) * r;
    B[2]           = (A[1] * A[5] - A[2] * A[4]) * r;
    B[3]           = (A[5] * A[6] - A[3] * A[8]) * r;
    B[4]           = (A[0] * A[8] - A[2] * A[6]) * r;
    B[5]           = (A[2] * A[3] - A[0] * A[5]) * r;
    B[6]           = (A[3] * A[7] - A[4] * A[6]) * r;
    B[7]           = (A[1] * A[6] - A[0] * A[7]) * r;
    B[8]           = (A[0] * A[4] - A[1] * A[3]) * r;
  }
  __device__ __forceinline__ void inv3x3(const double *A, double det, double *B)
  {
    inv3x3r(A, 1.0 / det, B);
  }

  // 3x3 (row-major, embedded: for DIM==2 the [2][2] entry is 1 and off entries 0) geometry Jacobian of the
  // d-linear cell map at unit point xi:  Jm[i][j] = dX_i / dxi_j
  template <int DIM>
  __device__ __forceinline__ void q1_jacobian(const double *__restrict__ verts, const double *xi, double *Jm)
  {
#pragma unroll
    for (int k = 0; k < 9; ++k)
      Jm[k] = 0.0;
    if constexpr (DIM == 2)
      Jm[8] = 1.0;
#pragma unroll
    for (int v = 0; v < (1 << DIM); ++v)
      {
        double f[3], s[3];
#pragma unroll
        for (int d = 0; d < DIM; ++d)
          {
            const bool hi = (v >> d) & 1;
            f[d]          = hi ? xi[d] : 1.0 - xi[d];
            s[d]          = hi ? 1.0 : -1.0;
          }
#pragma unroll
        for (int j = 0; j < DIM; ++j)
          {
            double g = s[j];
#pragma unroll
            for (int d = 0; d < DIM; ++d)
              if (d != j)
                g *= f[d];
#pragma unroll
            for (int i = 0; i < DIM; ++i)
              Jm[i * 3 + j] += verts[v * DIM + i] * g;
          }
      }
  }

  // unit-cell value and gradient of cell shape function a at cell quadrature point q from the 1D tables
  template <int DIM, int P>
  __device__ __forceinline__ void shape_at_qp(const double *__restrict__ sN1, const double *__restrict__ sdN1, int q,
                                              int a, double &N, double *dN)
  {
    using E = Elem<DIM, P>;
    int qi[3], ai[3];
    qi[0] = q % E::NQ1;
    qi[1] = (q / E::NQ1) % E::NQ1;
    qi[2] = (DIM == 3) ? q / (E::NQ1 * E::NQ1) : 0;
    ai[0] = a % E::NP1;
    ai[1] = (a / E::NP1) % E::NP1;
    ai[2] = (DIM == 3) ? a / (E::NP1 * E::NP1) : 0;
    double n[3], d[3];
#pragma unroll
    for (int k = 0; k < DIM; ++k)
      {
        n[k] = sN1[qi[k] * E::NP1 + ai[k]];
        d[k] = sdN1[qi[k] * E::NP1 + ai[k]];
      }
    if constexpr (DIM == 2)
      {
        N     = n[0] * n[1];
        dN[0] = d[0] * n[1];
        dN[1] = n[0] * d[1];
        dN[2] = 0.0;
      }
    else
      {
        N     = n[0] * n[1] * n[2];
        dN[0] = d[0] * n[1] * n[2];
        dN[1] = n[0] * d[1] * n[2];
        dN[2] = n[0] * n[1] * d[2];
      }
  }

  // Kinematics + compressible neo-Hookean response at one quadrature point
  // (nonlinear_elasticity.cc:927-934, compressible_neo_hook_material.h:17-138 in closed form):
  //   F = I + Grad u, J = det F, b_bar = J^(-2/d) F F^T, tau_bar = mu b_bar,
  //   tau_iso = dev(tau_bar), tau = tau_iso + kappa/2 (J^2-1) I,
  //   c_II = kappa J^2 - 2/d^2 tr(tau_bar),  c_S = -kappa (J^2-1) + 2/d tr(tau_bar)
  // gu is the 3x3-embedded displacement gradient w.r.t. reference coordinates.
  // the response from F, J = det F, Jm = J^(-2/d) and rJ = 1/J (what the matrix-free product keeps per point, see mf_spmv)
  template <int DIM, typename T>
  __device__ __forceinline__ void neo_hooke_from_F(const T *F, T J, T Jm, T rJ, T mu, T kappa, T *Finv, T *tau, T *tiso, T &cII, T &cS)
  {
    inv3x3r(F, rJ, Finv);
    T       b[6];                                                  // xx yy zz xy xz yz
    b[0]            = F[0] * F[0] + F[1] * F[1] + (DIM == 3 ? F[2] * F[2] : T(0.0));
    b[1]            = F[3] * F[3] + F[4] * F[4] + (DIM == 3 ? F[5] * F[5] : T(0.0));
    b[2]            = (DIM == 3) ? F[6] * F[6] + F[7] * F[7] + F[8] * F[8] : T(0.0);
    b[3]            = F[0] * F[3] + F[1] * F[4] + (DIM == 3 ? F[2] * F[5] : T(0.0));
    b[4]            = (DIM == 3) ? F[0] * F[6] + F[1] * F[7] + F[2] * F[8] : T(0.0);
    b[5]            = (DIM == 3) ? F[3] * F[6] + F[4] * F[7] + F[5] * F[8] : T(0.0);
    const T s  = mu * Jm;
    const T tr = s * (b[0] + b[1] + b[2]);
#pragma unroll
    for (int k = 0; k < 6; ++k)
      tiso[k] = s * b[k];
    const T trd = tr * T(1.0 / DIM); // (a product, not tr / DIM: an IEEE division costs the matrix-free product ten
                                     // instructions per point for the last bit of a number the oracle agrees with to 1e-12)
    tiso[0] -= trd;
    tiso[1] -= trd;
    if constexpr (DIM == 3)
      tiso[2] -= trd;
    const T pv = T(0.5) * kappa * (J * J - T(1.0));
#pragma unroll
    for (int k = 0; k < 6; ++k)
      tau[k] = tiso[k];
    tau[0] += pv;
    tau[1] += pv;
    if constexpr (DIM == 3)
      tau[2] += pv;
    cII = kappa * J * J - T(2.0 / (DIM * DIM)) * tr;
    cS  = -kappa * (J * J - T(1.0)) + T(2.0 / DIM) * tr;
  }
  template <int DIM>
  __device__ __forceinline__ void neo_hooke_qp(const double *gu, double mu, double kappa, double *Finv, double &J,
                                               double *tau, double *tiso, double &cII, double &cS, double *F, double &Jm,
                                               double &rJ)
  {
#pragma unroll
    for (int k = 0; k < 9; ++k)
      F[k] = gu[k];
    F[0] += 1.0;
    F[4] += 1.0;
    F[8] += 1.0; // DIM==2: gu[8]==0 -> F33 = 1 (embedding keeps det and inverse of the 2x2 part)
    J  = det3x3(F);
    rJ = 1.0 / J;
    Jm = (DIM == 3) ? 1.0 / (cbrt(J) * cbrt(J)) : rJ; // J^(-2/d)
    neo_hooke_from_F<DIM>(F, J, Jm, rJ, mu, kappa, Finv, tau, tiso, cII, cS);
  }
  template <int DIM>
  __device__ __forceinline__ void neo_hooke_qp(const double *gu, double mu, double kappa, double *Finv, double &J,
                                               double *tau, double *tiso, double &cII, double &cS)
  {
    double F[9], Jm, rJ;
    neo_hooke_qp<DIM>(gu, mu, kappa, Finv, J, tau, tiso, cII, cS, F, Jm, rJ);
  }

  __device__ __forceinline__ void sym_mul(const double *S, const double *g, double *out)
  {
    out[0] = S[0] * g[0] + S[3] * g[1] + S[4] * g[2];
    out[1] = S[3] * g[0] + S[1] * g[1] + S[5] * g[2];
    out[2] = S[4] * g[0] + S[5] * g[1] + S[2] * g[2];
  }

  // ------------------------------------------------------------------ cell assembly
  // One workgroup per cell of the current colour.  Threads are (tile, qslot): a tile is a 2x2 group of
  // node-pair blocks in the lower triangle of the element tangent, kept in registers; the QSPLIT lanes of a
  // tile split the quadrature points and are summed by wave shuffles at the end.
  //   phase A: per-QP kinematics/material records -> LDS (4 lanes per QP)
  //   per chunk of QC points: phase B (qp,node) records g, m, v, n -> LDS; main loop; residual
  //   epilogue: read-modify-write of the cell's blocks into the global block-CSR (colouring => race free)
  // ABL (timing only, results wrong): 1 = no tangent scatter, 2 = no main loop and no scatter, 3 = scatter only
  // ABL = 4 (production): RESIDUAL ONLY -- phases A/B and the residual, bit-identical with the full kernel's residual;
  //          the tangent is neither formed nor touched.  Serves the Newton convergence check (:446-469), whose
  //          assembly is never multiplied when the check passes.
  template <int DIM, int P, int QSPLIT, int NT, int QC, int MINW = 1, int ABL = 0>
  __global__ __launch_bounds__(NT, MINW) void assemble_cells(AsmParams prm)
  {
    using E = Elem<DIM, P>;
    constexpr int NPC = E::NPC, NPCP = E::NPCP, NQ = E::NQ, NQ1 = E::NQ1, NP1 = E::NP1, DD = E::DD, NV = E::NV;
    static_assert(NQ % QC == 0, "chunk must divide the number of quadrature points");
    static_assert(NT % QSPLIT == 0, "the lanes of a tile must not straddle the workgroup");
    static_assert(NPC * DIM <= NT, "residual needs one thread per local dof");
    // the tile grid is covered in NPASS passes of TPP tiles (one pass for every element up to 3D Q2; 3D Q3/Q4 have
    // 528 / 2016 tiles: each pass repeats the cheap phase B and accumulates its own tiles)
    constexpr int TPP = NT / QSPLIT, NPASS = (ABL == 4) ? 1 : (E::NTILES + TPP - 1) / TPP;

    __shared__ double s_N1[NQ1 * NP1], s_dN1[NQ1 * NP1], s_qw[NQ1], s_qx[NQ1];
    __shared__ double s_u[NPC * 3], s_a[NPC * 3], s_verts[NV * DIM];
    __shared__ int    s_conn[NPC];
    // per quadrature point NPCP records of RN doubles, padded by NDPAD doubles: the two lanes of a tile read the same
    // records of neighbouring points, and an unpadded point stride (a multiple of 32 B) put them on the same banks
    constexpr int NDS = NPCP * RN + NDPAD;
    // one buffer for the point records and the (point, node) records: after the main loop it stages the cell's element
    // tangent for a coalesced store (3D Q2: 3402 of its 4304 doubles)
    constexpr int NSTAGE = (DIM == 3 && P == 2 && ABL == 0) ? 9 * EBE_NBLK : 0;
    constexpr int NBIG   = (NQ * RQ + QC * NDS > NSTAGE) ? NQ * RQ + QC * NDS : NSTAGE;
    __shared__ __attribute__((aligned(16))) double s_big[NBIG];
    double *const s_qp = s_big, *const s_nd = s_big + NQ * RQ;

    const int     tid  = threadIdx.x;
    const int64_t cell = prm.cell_begin + blockIdx.x;

    // ---- stage tables, connectivity, vertices, gathered u_total and acceleration
    for (int i = tid; i < NQ1 * NP1; i += NT)
      {
        s_N1[i]  = prm.tab1d[i];
        s_dN1[i] = prm.tab1d[NQ1 * NP1 + i];
      }
    if (tid < NQ1)
      {
        s_qw[tid] = prm.tab1d[2 * NQ1 * NP1 + tid];
        s_qx[tid] = prm.tab1d[2 * NQ1 * NP1 + NQ1 + tid];
      }
    if (tid < NPC)
      s_conn[tid] = prm.conn[cell * NPC + tid];
    if (tid < NV * DIM)
      s_verts[tid] = prm.cverts[cell * (NV * DIM) + tid];
    for (int i = tid; i < QC * NDS; i += NT)
      s_nd[i] = 0.0; // the padding node stays zero for the whole kernel
    __syncthreads();
    for (int i = tid; i < NPC * 3; i += NT)
      {
        const int a = i / 3, c = i - a * 3;
        double    uv = 0.0, av = 0.0;
        if (c < DIM)
          {
            const int64_t g = int64_t(s_conn[a]) * DIM + c;
            uv              = prm.u[g] + prm.du[g]; // get_total_solution, :580-588
            av              = prm.acc[g];
          }
        s_u[i] = uv;
        s_a[i] = av;
      }
    __syncthreads();

    // ---- phase A: quadrature-point records, 4 lanes per point
    for (int task = tid; task < NQ * 4; task += NT)
      {
        const int q = task >> 2, part = task & 3;
        double    gxi[9], acc[3];
#pragma unroll
        for (int k = 0; k < 9; ++k)
          gxi[k] = 0.0;
        acc[0] = acc[1] = acc[2] = 0.0;
        for (int a = part; a < NPC; a += 4)
          {
            double N, dN[3];
            shape_at_qp<DIM, P>(s_N1, s_dN1, q, a, N, dN);
#pragma unroll
            for (int i = 0; i < DIM; ++i)
              {
                const double ui = s_u[a * 3 + i];
#pragma unroll
                for (int j = 0; j < DIM; ++j)
                  gxi[i * 3 + j] += ui * dN[j];
                acc[i] += s_a[a * 3 + i] * N;
              }
          }
#pragma unroll
        for (int k = 0; k < 9; ++k)
          {
            gxi[k] += __shfl_xor(gxi[k], 1, 64);
            gxi[k] += __shfl_xor(gxi[k], 2, 64);
          }
#pragma unroll
        for (int k = 0; k < 3; ++k)
          {
            acc[k] += __shfl_xor(acc[k], 1, 64);
            acc[k] += __shfl_xor(acc[k], 2, 64);
          }
        // geometry of the d-linear map at this point
        double xi[3], wq = 1.0;
        {
          int qi[3] = {q % NQ1, (q / NQ1) % NQ1, (DIM == 3) ? q / (NQ1 * NQ1) : 0};
#pragma unroll
          for (int d = 0; d < DIM; ++d)
            {
              xi[d] = s_qx[qi[d]];
              wq *= s_qw[qi[d]];
            }
        }
        double Jm[9], Ji[9];
        q1_jacobian<DIM>(s_verts, xi, Jm);
        const double detJ = det3x3(Jm);
        inv3x3(Jm, detJ, Ji);
        // Grad_X u = grad_xi u * Jinv
        double gu[9];
#pragma unroll
        for (int i = 0; i < 3; ++i)
#pragma unroll
          for (int j = 0; j < 3; ++j)
            gu[i * 3 + j] = (i < DIM && j < DIM) ? gxi[i * 3 + 0] * Ji[0 * 3 + j] + gxi[i * 3 + 1] * Ji[1 * 3 + j] +
                                                     (DIM == 3 ? gxi[i * 3 + 2] * Ji[2 * 3 + j] : 0.0) :
                                                   0.0;
        double Finv[9], J, tau[6], tiso[6], cII, cS, Fq[9], Jmq, rJq;
        neo_hooke_qp<DIM>(gu, prm.mu, prm.kappa, Finv, J, tau, tiso, cII, cS, Fq, Jmq, rJq);
        if (!(J > 0.0)) // inverted element (nonlinear_elasticity.cc:935 asserts det F > 0)
          *prm.inverted = 1.0;
        if (part == 0)
          {
            double *r = &s_qp[q * RQ];
            // M = Jinv * Finv
#pragma unroll
            for (int i = 0; i < 3; ++i)
#pragma unroll
              for (int j = 0; j < 3; ++j)
                r[Q_M + i * 3 + j] = Ji[i * 3 + 0] * Finv[0 * 3 + j] + Ji[i * 3 + 1] * Finv[1 * 3 + j] +
                                     Ji[i * 3 + 2] * Finv[2 * 3 + j];
#pragma unroll
            for (int k = 0; k < 6; ++k)
              {
                r[Q_TAU + k]  = tau[k];
                r[Q_TISO + k] = tiso[k];
              }
            const double w   = detJ * wq; // JxW of the reference configuration
            const double sqn = sqrt(prm.alpha1 * prm.rho * w);
            r[Q_W]           = w;
            r[Q_WCII]        = w * cII;
            r[Q_CS2]         = 0.5 * cS;
            r[Q_SQN]         = sqn;
            r[Q_NINV]        = sqn > 0.0 ? 1.0 / sqn : 0.0;
#pragma unroll
            for (int i = 0; i < 3; ++i)
              r[Q_FACC + i] = prm.rho * w * (acc[i] - prm.body[i]);
            if constexpr (DIM == 3 && P == 2 && ABL == 0)
              if (prm.qrec) // the state the tangent is linearised at, for the matrix-free product (mf_spmv)
                {
                  double *__restrict__ g = prm.qrec + cell * int64_t(MF_NREC * 64) + q;
#pragma unroll
                  for (int k = 0; k < 9; ++k)
                    g[k * 64] = Fq[k];
                  g[9 * 64]  = Jmq;
                  g[10 * 64] = rJq;
                }
          }
      }
    __syncthreads();

    double rres = 0.0; // residual entry of local dof tid (threads tid < NPC*DIM), accumulated in pass 0
#pragma unroll 1
    for (int pass = 0; pass < NPASS; ++pass)
      {
    // ---- tile of this thread
    const int  tile   = pass * TPP + tid / QSPLIT, qslot = tid % QSPLIT;
    const bool active = tile < E::NTILES;
    int        ta = 0, tb = 0;
    if (active)
      {
        ta = int((sqrtf(8.0f * float(tile) + 1.0f) - 1.0f) * 0.5f);
        while ((ta + 1) * (ta + 2) / 2 <= tile)
          ++ta;
        while (ta * (ta + 1) / 2 > tile)
          --ta;
        tb = tile - ta * (ta + 1) / 2;
      }
    double K[4][DD];
#pragma unroll
    for (int b = 0; b < 4; ++b)
#pragma unroll
      for (int k = 0; k < DD; ++k)
        K[b][k] = 0.0;

    for (int chunk = 0; chunk < NQ / QC; ++chunk)
      {
        // ---- phase B: (qp,node) records for this chunk
        for (int task = tid; task < QC * NPC; task += NT)
          {
            const int qq = task / NPC, a = task - qq * NPC;
            const int q  = chunk * QC + qq;
            double    N, dN[3];
            shape_at_qp<DIM, P>(s_N1, s_dN1, q, a, N, dN);
            const double *r = &s_qp[q * RQ];
            double        g[3], t[3], v[3];
#pragma unroll
            for (int j = 0; j < 3; ++j)
              g[j] = dN[0] * r[Q_M + 0 * 3 + j] + dN[1] * r[Q_M + 1 * 3 + j] + dN[2] * r[Q_M + 2 * 3 + j];
            double *o = &s_nd[qq * NDS + a * RN];
            sym_mul(&r[Q_TISO], g, t);
            sym_mul(&r[Q_TAU], g, v);
#pragma unroll
            for (int j = 0; j < 3; ++j)
              {
                o[j]     = g[j];
                o[3 + j] = (-2.0 / DIM) * t[j];
                o[6 + j] = v[j]; // tau g_a itself: the residual then is exactly zero for a stress-free state
              }
            o[9] = r[Q_SQN] * N;
          }
        __syncthreads();

        // ---- main loop: accumulate the 2x2 tile over this lane's share of the chunk
        if (active && ABL != 2 && ABL != 3 && ABL != 4)
          {
            for (int qq = qslot; qq < QC; qq += QSPLIT)
              {
                const double *r    = &s_qp[(chunk * QC + qq) * RQ];
                const double  w    = r[Q_W], wcII = r[Q_WCII], cs2 = r[Q_CS2], wcs2 = w * cs2;
                const double *nd   = &s_nd[qq * NDS];
                double        ha[2][3], gw[2][3], gc[2][3], na[2];
#pragma unroll
                for (int x = 0; x < 2; ++x)
                  {
                    const double *pa = &nd[(2 * ta + x) * RN];
#pragma unroll
                    for (int i = 0; i < DIM; ++i)
                      {
                        const double g = pa[i];
                        ha[x][i]       = wcII * g + w * pa[3 + i];
                        gw[x][i]       = w * g;
                        gc[x][i]       = wcs2 * g;
                      }
                    na[x] = pa[9];
                  }
#pragma unroll
                for (int y = 0; y < 2; ++y)
                  {
                    const double *pb = &nd[(2 * tb + y) * RN];
                    double        gb[3], mb[3], vb[3];
#pragma unroll
                    for (int j = 0; j < DIM; ++j)
                      {
                        gb[j] = pb[j];
                        mb[j] = pb[3 + j];
                        vb[j] = cs2 * pb[j] + pb[6 + j]; // v_b = (c_S/2) g_b + tau g_b
                      }
                    const double nb = pb[9];
#pragma unroll
                    for (int x = 0; x < 2; ++x)
                      {
                        double dg = na[x] * nb;
#pragma unroll
                        for (int i = 0; i < DIM; ++i)
                          dg += gw[x][i] * vb[i];
#pragma unroll
                        for (int i = 0; i < DIM; ++i)
                          {
#pragma unroll
                            for (int j = 0; j < DIM; ++j)
                              {
                                // three FMAs straight into the accumulator (a sum of products first costs a 4th op)
                                double kk = K[x * 2 + y][i * DIM + j];
                                kk        = fma(ha[x][i], gb[j], kk);
                                kk        = fma(gw[x][i], mb[j], kk);
                                kk        = fma(gb[i], gc[x][j], kk);
                                K[x * 2 + y][i * DIM + j] = kk;
                              }
                            K[x * 2 + y][i * DIM + i] += dg;
                          }
                      }
                  }
              }
          }
        // ---- residual (:984-995 collapsed by partition of unity): r_a -= w (tau g_a) + N_a rho w (acc - b)
        if (tid < NPC * DIM && pass == 0)
          {
            const int a = tid / DIM, i = tid - a * DIM;
            for (int qq = 0; qq < QC; ++qq)
              {
                const double *r  = &s_qp[(chunk * QC + qq) * RQ];
                const double *pa = &s_nd[qq * NDS + a * RN];
                rres -= r[Q_W] * pa[6 + i] + (pa[9] * r[Q_NINV]) * r[Q_FACC + i]; // w (tau g_a)_i + N_a rho w (acc - b)_i
              }
          }
        __syncthreads();
      }

    // ---- reduce the QSPLIT partial tiles
    if constexpr (ABL != 4)
      {
#pragma unroll
        for (int o = 1; o < QSPLIT; o <<= 1)
#pragma unroll
          for (int b = 0; b < 4; ++b)
#pragma unroll
            for (int k = 0; k < DD; ++k)
              K[b][k] += __shfl_xor(K[b][k], o, 64);
      }

    // ---- residual scatter (:769-773; constrained rows get no rhs)
    if (tid < NPC * DIM && pass == 0)
      {
        const int     a = tid / DIM, i = tid - a * DIM;
        const int32_t A = s_conn[a];
        if (!((prm.cmask[A] >> i) & 1))
          prm.rhs[int64_t(A) * DIM + i] += rres;
      }

    // ---- tangent scatter: lane `qslot` of a tile writes the blocks bl with bl % QSPLIT == qslot.
    // [DEAL.II distribute_local_to_global] constrained rows/cols are dropped, the diagonal of a constrained
    // dof receives |K_e(i,i)|.
    if (ABL == 1 || ABL == 2 || ABL == 3)
      {
#pragma unroll
        for (int b = 0; b < 4; ++b)
#pragma unroll
          for (int k = 0; k < DD; ++k)
            asm volatile("" ::"v"(K[b][k]));
      }
    if (active && ABL != 1 && ABL != 2 && ABL != 4)
      {
        const uint16_t *__restrict__ offc = prm.off + cell * (NPC * NPC);
#pragma unroll
        for (int bl = 0; bl < 4; ++bl)
          {
            if (bl % QSPLIT != qslot)
              continue;
            const int a = 2 * ta + (bl >> 1), b = 2 * tb + (bl & 1);
            if (a >= NPC || b >= NPC || a < b)
              continue;
            const int32_t A = s_conn[a], B = s_conn[b];
            const int     ma = prm.cmask[A], mb = prm.cmask[B];
            const uint16_t oab = offc[a * NPC + b], oba = offc[b * NPC + a];
            // block (g, kx) of a row sits at base + g * gstride + kx (slice-interleaved block rows, mi_mesh.hpp); a node
            // without a row here (ghost node of a slab) stores nothing
            const int2    ia = prm.rowinfo[A], ib = prm.rowinfo[B];
            const int32_t ra = ia.x, rb = ib.x;
            double *__restrict__ pab = prm.vals + (int64_t(ra) + ((oab >> 4) & 0x7ff) * ia.y + (oab & 15)) * DD;
            double *__restrict__ pba = prm.vals + (int64_t(rb) + ((oba >> 4) & 0x7ff) * ib.y + (oba & 15)) * DD;
            // bit 15: first touch of the block in processing order -> plain store instead of read-modify-write
            const bool first_ab = oab >> 15, first_ba = oba >> 15;
            double     vab[DD], vba[DD];
#pragma unroll
            for (int i = 0; i < DIM; ++i)
#pragma unroll
              for (int j = 0; j < DIM; ++j)
                {
                  double v = K[bl][i * DIM + j];
                  if (((ma >> i) | (mb >> j)) & 1)
                    v = (a == b && i == j) ? fabs(v) : 0.0;
                  vab[i * DIM + j] = v;
                  vba[j * DIM + i] = v;
                }
            if constexpr (DIM == 3 && P == 2)
              if (prm.ke) // the cell's own (masked) block, before it is summed into the global matrix: staged in LDS
                {
                  double *kq = s_big + (a * (a + 1) / 2 + b);
#pragma unroll
                  for (int k = 0; k < DD; ++k)
                    kq[k * EBE_NBLK] = vab[k];
                }
            if (ra >= 0)
              {
                if (!first_ab)
#pragma unroll
                  for (int k = 0; k < DD; ++k)
                    vab[k] += pab[k];
#pragma unroll
                for (int k = 0; k < DD; ++k)
                  pab[k] = vab[k];
              }
            if (a != b && rb >= 0)
              {
                if (!first_ba)
#pragma unroll
                  for (int k = 0; k < DD; ++k)
                    vba[k] += pba[k];
#pragma unroll
                for (int k = 0; k < DD; ++k)
                  pba[k] = vba[k];
              }
          }
      }
    if constexpr (DIM == 3 && P == 2 && ABL == 0)
      if (prm.ke) // the staged element tangent: 3402 contiguous doubles per cell, coalesced
        {
          __syncthreads();
          double *__restrict__ dst = prm.ke + cell * (int64_t(DD) * EBE_NBLK);
          for (int i = tid; i < DD * EBE_NBLK; i += NT)
            dst[i] = s_big[i];
        }
      } // pass
  }


  // first node of the cell at colour-sorted position pos, from the lattice description alone (mi::CellLattice): uniform
  // operands, i.e. scalar instructions and one scalar load -- no vector-memory round trip between a workgroup's start
  // and its first gather
  __device__ __forceinline__ int32_t lattice_node0(const CellLattice &L, int64_t pos, int32_t *cz = nullptr)
  {
    const int32_t p   = int32_t(pos);
    int           col = 0;
#pragma unroll
    for (int c = 1; c < 8; ++c)
      col += (p >= L.begin[c]) ? 1 : 0; // empty trailing colours begin at ncells
    int32_t begin = L.begin[0];
#pragma unroll
    for (int c = 1; c < 8; ++c)
      begin = (p >= L.begin[c]) ? L.begin[c] : begin;
    const CellLatticeRow R = L.rows[col];
    const uint32_t r   = uint32_t(p - begin);
    const uint32_t rz  = uint32_t((uint64_t(r) * R.magic_mxy) >> 42);
    const uint32_t rem = r - rz * uint32_t(R.mxy);
    const uint32_t ry  = uint32_t((uint64_t(rem) * R.magic_mx) >> 42);
    const uint32_t rx  = rem - ry * uint32_t(R.mx);
    if (cz)
      *cz = 2 * int32_t(rz) + R.pz; // the cell's layer along the last lattice direction
    return R.base + int32_t(rx) * L.sx + int32_t(ry) * L.sy + int32_t(rz) * L.sz;
  }

  // ------------------------------------------------------------------ 3D Q2 cell assembly, sum factorised (default)
  // Same element tangent, residual and scatter as assemble_cells, a third of its arithmetic.  With g_a = M^T grad_xi N_a,
  // M = Jinv Finv, every term of the tangent is a bilinear form in the UNIT-CELL gradients:
  //   K_ab^{ij} = sum_q sum_{kl} d_k N_a(q) C^{ij}_{kl}(q) d_l N_b(q)  +  delta_ij sum_q mu(q) N_a(q) N_b(q),
  //   C^{ij}_{kl} = A_ki M_lj + B_ki Tm_lj + E_kj M_li + delta_ij S_kl,   Tm = -(2/3) M tau_iso,
  //   A = w (c_II M + Tm),  B = w M,  E = w (c_S/2) M,  S = w (c_S/2 M M^T + M tau M^T),  mu = alpha_1 rho w,
  // and N_a(q) = N_a1(qx) N_a2(qy) N_a3(qz), so the sum over the 64 points is contracted one direction at a time
  // (tools/proto/sf_assembly.py checks algebra and decomposition against the independent mirror).
  // One workgroup of 4 waves per cell:
  //   wave 0, lane = quadrature point: u and the acceleration interpolated to the points by sum factorisation,
  //     kinematics + material (neo_hooke_qp), the 81 coefficient fields + mu -> LDS, the point records for mf_spmv,
  //     then the residual  r_a = -sum_q (w tau M^T grad_xi N_a + N_a rho w (acc - b))  integrated by sum factorisation;
  //   waves 1-3, lane = (ij, (a1 >= b1), a2) [162 items]: for every qz: x-contraction of C along a line of 4 points
  //     (per-lane products N_a1 N_b1), y-contraction into 12 accumulators (a2 per lane, b2 unrolled; the (k==z, l==z)
  //     type of (kl) selects the accumulator, so the sum over kl happens here), then the z-contraction into the lane's
  //     27 tangent entries K^{ij}[(a1 a2 a3),(b1 b2 b3)], b2 a3 b3 unrolled.  1D tables are scalar operands.
  // Only node pairs with (a1,a2,a3) >= (b1,b2,b3) (x most significant) are formed; they are filed under the lower
  // triangle of the usual node order, transposed where the two orders disagree, in an LDS image of the element tangent
  // [e = i*3+j][block], from which the scatter of assemble_cells runs (constraint masking, first-touch store / RMW).
  // RES_ONLY (64 threads): wave 0 alone = the residual-only pass of the Newton convergence check.
  #define MI_WAVE_SYNC()                                                                                              \
    do                                                                                                                \
      {                                                                                                               \
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");                                                        \
        __builtin_amdgcn_wave_barrier();                                                                              \
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");                                                        \
      }                                                                                                               \
    while (0)
  // XV (round 5; "asm_variant" 3-8 select the combinations for A/B, profiles/r05/asm_ab_*.txt).  The default is 4 | 128.
  //   bit 2 (adopted): 45 coefficient fields instead of 81 by the major symmetry C^{ij}_{kl} = C^{ji}_{lk} (the tangent
  //     changes at rounding level); the block table is built by wave 3 while it would otherwise wait for the prologue,
  //     in LDS of its own (the 45 fields leave room); the acceleration is interpolated behind barrier (1), beside the
  //     contractions (only the residual needs it).
  //   bit 7 (adopted): branch-free scatter (see there): 381 instead of 836 vector and 50 instead of 810 scalar
  //     instructions per wave, the scatter phase 5.0 k instead of 9.9 k clocks of a workgroup's life.
  //   bit 0 (measured, not adopted): the contraction waves run a software pipeline over their 160 (qz, kl, qy) steps --
  //     the LDS reads of step s + 4 are issued as soon as step s has consumed its ring slot -- with the (kl) loop ordered
  //     by accumulator group (k == z, l == z), each group contracted along z as soon as it is complete (3 live
  //     accumulators instead of 12); same bits.  The contraction phase shrinks (18.9 k -> 14.4 k clocks) and the other
  //     phases of the three workgroups of a CU grow by as much: 7.68 against 7.67 ms per assembly.
  //   bit 1 (measured, not adopted): wave 0 runs its prologue at raised priority: prologue 19.6 k -> 14.3 k clocks, the
  //     contractions 18.9 k -> 22.2 k, 8.2 against 7.7 ms.
  // assemble_q2sf, pipelined contraction: (kl) in the order of the accumulator groups (k == z, l == z) = (0,0): 0 1 3 4 and the
  // mass field 9, (0,1): 2 5, (1,0): 6 7, (1,1): 8 -- within a group ascending, the groups in the order the unpipelined
  // loop adds them into the tangent entries
  __device__ __forceinline__ constexpr int q2sf_kl_order(const int pos)
  {
    return pos == 0 ? 0 : pos == 1 ? 1 : pos == 2 ? 3 : pos == 3 ? 4 : pos == 4 ? 9 : pos == 5 ? 2 : pos == 6 ? 5 : pos == 7 ? 6 : pos == 8 ? 7 : 8;
  }

  // 45-field storage (major symmetry): pairs (i <= j) in the order 00 01 02 11 12 22; a diagonal pair holds the 6 fields
  // k <= l, an off-diagonal one all 9
  __device__ __forceinline__ constexpr int q2sf_sym6(const int k, const int l) // k <= l
  {
    return k == 0 ? l : k == 1 ? 2 + l : 5;
  }
  __device__ __forceinline__ constexpr int q2sf_pair_base(const int i, const int j) // i <= j
  {
    return i == 0 ? (j == 0 ? 0 : j == 1 ? 6 : 15) : i == 1 ? (j == 1 ? 24 : 30) : 39;
  }
  __device__ __forceinline__ constexpr int q2sf_field(const int i, const int j, const int k, const int l) // i <= j; i == j: k <= l
  {
    return q2sf_pair_base(i, j) + (i == j ? q2sf_sym6(k, l) : k * 3 + l);
  }

  template <bool RES_ONLY, int XV = 0>
  __global__ __launch_bounds__(RES_ONLY ? 64 : 256, RES_ONLY ? 4 : 3) void assemble_q2sf(AsmParams prm)
  {
    constexpr bool V2 = !RES_ONLY && (XV & 4) != 0, SLIM = !RES_ONLY && (XV & 128) != 0;
    // bit 8 (round 6, the default where point records exist): the tangent FROM THE RECORDS.  The residual pass (RES_ONLY,
    // one wave per cell at mf_spmv's occupancy) has gathered, differentiated and written F, J^(-2/3), 1/J of every point; this
    // kernel starts there: ALL FOUR waves recompute the material response of the cell's 64 points from the records (lane =
    // point, the assembly's own function, ~350 instructions), each writes a quarter of the 45 fields, and the contractions
    // begin after ~5 k clocks instead of behind one wave's 18.8 k-clock chain of gather, gradients, kinematics and fields;
    // wave 0 then builds the block table beside the contractions.  No residual here.
    constexpr bool REC = !RES_ONLY && (XV & 256) != 0;
    constexpr bool RECW0 = !RES_ONLY && (XV & 512) != 0; // (see wave 0)
    static_assert(!REC || (V2 && SLIM), "the record form builds on the 45-field kernel with the branch-free scatter");
    constexpr int NPC = 27, FS = 66, NF = V2 ? 46 : 82; // field stride (padded: fields of different ij on different banks), fields
    constexpr int PS = 20, PW = 9 * PS, AO = 552;
    constexpr int MASSF = NF - 1;                       // the mass field
    __shared__ __attribute__((aligned(16))) double s_C[RES_ONLY ? 2 : (NF * FS > 9 * EBE_NBLK ? NF * FS : 9 * EBE_NBLK)]; // later the element tangent [9][378]
    __shared__ __attribute__((aligned(16))) double s_w[REC ? 2 : 768 + 216];    // wave 0's scratch (as in mf_spmv)
    __shared__ uint64_t s_tab[V2 ? NPC * NPC : 1];                              // V2: the block table (wave 3 builds it during the prologue)
    __shared__ int  s_conn[NPC];
    __shared__ int2 s_ri[RES_ONLY ? 1 : NPC]; // rowinfo of the cell's nodes (where their rows are in the global matrix)
    __shared__ int  s_cm[RES_ONLY ? 1 : NPC]; // their constraint bits
    __shared__ int  s_plain;                  // 1: no node of the cell is constrained and every node has a row here (the
                                              // scatter then skips the per-entry masking: all but the boundary cells)
    typedef const volatile __attribute__((address_space(3))) double *lds_cvp;
    const int     tid  = int(threadIdx.x);
    // RES_ONLY bit 10 (round 6, the matrix-free fine level's point pass): ALL cells in one launch -- the 81 residual
    // entries of a cell go to the cell's own slots (AsmParams::res_slots at the slots of MfParams::dst, as the matrix-free
    // product's results) instead of being subtracted from system_rhs colour by colour; residual_gather sums them per node in
    // processing order.  Workgroups of one XCD take a contiguous run of cells (as mf_spmv).
    constexpr bool RSLOTS = RES_ONLY && (XV & 1024) != 0;
    int64_t        cell   = prm.cell_begin + blockIdx.x;
    if constexpr (RSLOTS)
      {
        const int64_t local = int64_t(blockIdx.x & 7) * prm.xcd_chunk + (blockIdx.x >> 3);
        if (local >= prm.cell_count)
          return;
        cell = prm.cell_begin + local;
      }
    // 1D tables (uniform)
    double S[4][3], D[4][3];
#pragma unroll
    for (int q = 0; q < 4; ++q)
#pragma unroll
      for (int a = 0; a < 3; ++a)
        {
          S[q][a] = prm.tab1d[q * 3 + a];
          D[q][a] = prm.tab1d[12 + q * 3 + a];
        }

    // diagnostic phase stamps (thread 64 = a lane of the first tangent wave; thread 0 in the residual-only form)
#define MI_STAMP(i_)                                                                                   \
  do                                                                                                   \
    {                                                                                                  \
      if (prm.stamps && tid == (RES_ONLY ? 0 : 64))                                                    \
        prm.stamps[int64_t(blockIdx.x) * 16 + (i_)] = __builtin_amdgcn_s_memtime();                   \
    }                                                                                                  \
  while (0)
    // stamps 8-15: inside the prologue, by the wave that does the work (tid_ = 0: wave 0, 192: wave 3)
#define MI_STAMPW(i_, tid_)                                                                            \
  do                                                                                                   \
    {                                                                                                  \
      if (!RES_ONLY && prm.stamps && tid == (tid_))                                                    \
        prm.stamps[int64_t(blockIdx.x) * 16 + (i_)] = __builtin_amdgcn_s_memtime();                   \
    }                                                                                                  \
  while (0)
    MI_STAMP(0);
    // state of wave 0 (lane = quadrature point) that lives across barrier (1); unused in waves 1-3
    const int     lane = tid & 63;
    double *const s0 = s_w, *const sE = s_w + 768;
    const lds_cvp v0 = (lds_cvp)s0, vE = (lds_cvp)sE;
    const int     q16 = lane & 15, pck = lane >> 2, pqx = lane & 3;
    double        M[9], tau[6], w = 0.0, accq[3] = {0.0, 0.0, 0.0};
    double        Sz[3] = {0.0, 0.0, 0.0}, Dz[3] = {0.0, 0.0, 0.0}, accn[3] = {0.0, 0.0, 0.0}, gxi[3][3];
    // state of waves 1-3 (lane = (ij, a1 >= b1, a2)) across barrier (1); unused in wave 0
    const int  it     = tid - 64;
    const bool active = tid >= 64 && it < 162;
    const int  ij = active ? it / 18 : 0, r18 = active ? it - 18 * ij : 0, pr = r18 / 3, a2 = r18 - 3 * pr;
    const int  a1 = pr >= 3 ? 2 : (pr >= 1 ? 1 : 0), b1 = pr - a1 * (a1 + 1) / 2; // pairs (0,0) (1,0) (1,1) (2,0) (2,1) (2,2)
    const int  ci = ij / 3, cj = ij - 3 * ci;
    double     P1[4][4], phi2[2][4], mflag = 0.0;

    // wave 0: interpolation of the 81 nodal values at s0[c * 27 + a] to the points by sum factorisation
    auto interp = [&](const int pass) __attribute__((always_inline)) {
        MI_WAVE_SYNC();
        if (lane < 27) // contract i
          {
            const double x0 = v0[lane * 3], x1 = v0[lane * 3 + 1], x2 = v0[lane * 3 + 2];
#pragma unroll
            for (int qx = 0; qx < 4; ++qx)
              {
                s0[AO + qx * 27 + lane] = S[qx][0] * x0 + S[qx][1] * x1 + S[qx][2] * x2;
                if (pass == 0)
                  s0[AO + 108 + qx * 27 + lane] = D[qx][0] * x0 + D[qx][1] * x1 + D[qx][2] * x2;
              }
          }
        MI_WAVE_SYNC();
        if (lane < 36) // contract j
          {
            const int    ia  = AO + pqx * 27 + pck * 3;
            const double as0 = v0[ia], as1 = v0[ia + 1], as2 = v0[ia + 2];
            double       ad0 = 0.0, ad1 = 0.0, ad2 = 0.0;
            if (pass == 0)
              {
                ad0 = v0[108 + ia];
                ad1 = v0[108 + ia + 1];
                ad2 = v0[108 + ia + 2];
              }
#pragma unroll
            for (int qy = 0; qy < 4; ++qy)
              {
                const int o = pck * PS + qy * 4 + pqx;
                if (pass == 0)
                  {
                    s0[o]      = S[qy][0] * ad0 + S[qy][1] * ad1 + S[qy][2] * ad2;
                    s0[o + PW] = D[qy][0] * as0 + D[qy][1] * as1 + D[qy][2] * as2;
                  }
                s0[o + 2 * PW] = S[qy][0] * as0 + S[qy][1] * as1 + S[qy][2] * as2;
              }
          }
        MI_WAVE_SYNC();
#pragma unroll
        for (int c = 0; c < 3; ++c) // contract k
          {
            double h0 = 0.0, h1 = 0.0, h2 = 0.0, vv = 0.0;
#pragma unroll
            for (int k = 0; k < 3; ++k)
              {
                const int    o   = (c * 3 + k) * PS + q16;
                const double bss = v0[o + 2 * PW];
                if (pass == 0)
                  {
                    h0 = fma(Sz[k], v0[o], h0);
                    h1 = fma(Sz[k], v0[o + PW], h1);
                    h2 = fma(Dz[k], bss, h2);
                  }
                else
                  vv = fma(Sz[k], bss, vv);
              }
            if (pass == 0)
              {
                gxi[c][0] = h0;
                gxi[c][1] = h1;
                gxi[c][2] = h2;
              }
            else
              accq[c] = vv;
          }
    };
    auto stage_acc = [&]() __attribute__((always_inline)) {
      MI_WAVE_SYNC();
      if (lane < NPC)
#pragma unroll
        for (int c = 0; c < 3; ++c)
          s0[c * NPC + lane] = accn[c];
    };

    // ---- while the tangent waves contract: where the 729 node-pair blocks of this cell go.  One 64-bit word per
    // block (a, b) in wave 0's own scratch, which the residual no longer needs: bits 0-31 position of the block in
    // the global matrix (base + g * gstride + kx, mi_mesh.hpp; 0xffffffff: the node has no row here), 32-40 its place
    // in the lower-triangle image of the element tangent, 41 a >= b (else: the transposed block of (b, a)),
    // 42 first touch in processing order (plain store), 43 a == b, 44-46 / 47-49 constraint bits of A / B.
    auto build_table = [&](uint64_t *const tab) __attribute__((always_inline)) {
    {
      const uint16_t *__restrict__ offc = prm.off + cell * (NPC * NPC);
      {
        const bool special = lane < NPC && (s_cm[lane] != 0 || s_ri[lane].x < 0);
        const bool any     = __builtin_amdgcn_ballot_w64(special) != 0;
        if (lane == 0)
          s_plain = any ? 0 : 1;
      }
      uint16_t        o[12];
#pragma unroll
      for (int r = 0; r < 12; ++r) // all twelve (coalesced) loads in flight at once
        o[r] = (r * 64 + lane < NPC * NPC) ? offc[r * 64 + lane] : uint16_t(0);
#pragma unroll
      for (int r = 0; r < 12; ++r)
        {
          const int blk = r * 64 + lane;
          if (blk < NPC * NPC)
            {
              const int      a = blk / NPC, b = blk - NPC * a;
              const int2     ri = s_ri[a];
              const uint32_t pos = ri.x >= 0 ? uint32_t(ri.x + int32_t((o[r] >> 4) & 0x7ff) * ri.y + int32_t(o[r] & 15)) :
                                               (SLIM ? prm.trash_blk : 0xffffffffu);
              const bool     low = a >= b;
              const int      hi = low ? a : b, lo = low ? b : a;
              tab[blk] = uint64_t(pos) | (uint64_t(hi * (hi + 1) / 2 + lo) << 32) | (uint64_t(low) << 41) |
                         (uint64_t(o[r] >> 15) << 42) | (uint64_t(a == b) << 43) | (uint64_t(s_cm[a]) << 44) |
                         (uint64_t(s_cm[b]) << 47);
            }
        }
    }
    };

    if constexpr (REC)
      {
        // ================================================================= every wave: the fields from the point records
        const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
        double    rec[MF_NREC];
        {
          const double *__restrict__ rp = prm.qrec + cell * int64_t(MF_NREC * 64) + lane;
#pragma unroll
          for (int f = 0; f < MF_NREC; ++f)
            rec[f] = rp[f * 64];
        }
        int2 ri_l = make_int2(0, 0);
        int  cm_l = 0;
        if (tid < NPC) // wave 0: where the cell's rows are (for the block table it builds beside the contractions); kept in
          {            // registers until the fields are out, so that the loads travel beside the arithmetic
            int32_t node;
            if (prm.lat.ncol > 0)
              {
                const int32_t node0 = lattice_node0(prm.lat, cell);
                const int     k9 = lane / 9, r9 = lane - 9 * k9, j3 = r9 / 3, i3 = r9 - 3 * j3;
                node               = node0 + i3 + j3 * prm.lat.nn0 + k9 * prm.lat.nn01;
              }
            else
              node = prm.conn[cell * NPC + lane];
            ri_l = prm.rowinfo[node];
            cm_l = prm.cmask[node] & 7;
          }
        const int    qz = lane >> 4;
        const double wq = prm.tab1d[24 + (lane & 3)] * prm.tab1d[24 + ((lane >> 2) & 3)] * prm.tab1d[24 + qz];
        double       Mr[9], detJ;
        double       Finv[9], tauq[6], tisoq[6], cII, cS;
        neo_hooke_from_F<3>(rec, det3x3(rec), rec[9], rec[10], prm.mu, prm.kappa, Finv, tauq, tisoq, cII, cS);
        if (prm.cellbox) // every local cell an axis-parallel box: 1/h and the volume
          {
            const double *__restrict__ cb = prm.cellbox + cell * 4;
            const double rx = cb[0], ry = cb[1], rz = cb[2];
            detJ            = cb[3];
#pragma unroll
            for (int k = 0; k < 3; ++k)
              {
                Mr[k]     = rx * Finv[k];
                Mr[3 + k] = ry * Finv[3 + k];
                Mr[6 + k] = rz * Finv[6 + k];
              }
          }
        else
          {
            const double *__restrict__ cv = prm.cverts + cell * 24;
            const double xiq[3] = {prm.tab1d[28 + (lane & 3)], prm.tab1d[28 + ((lane >> 2) & 3)], prm.tab1d[28 + qz]};
            double       verts[24], Jm[9], Ji[9];
#pragma unroll
            for (int k = 0; k < 24; ++k)
              verts[k] = cv[k];
            q1_jacobian<3>(verts, xiq, Jm);
            detJ = det3x3(Jm);
            inv3x3(Jm, detJ, Ji);
#pragma unroll
            for (int i = 0; i < 3; ++i)
#pragma unroll
              for (int j = 0; j < 3; ++j)
                Mr[i * 3 + j] = Ji[i * 3 + 0] * Finv[0 * 3 + j] + Ji[i * 3 + 1] * Finv[1 * 3 + j] + Ji[i * 3 + 2] * Finv[2 * 3 + j];
          }
        const double wr = detJ * wq;
        const double Tq[3][3]  = {{tauq[0], tauq[3], tauq[4]}, {tauq[3], tauq[1], tauq[5]}, {tauq[4], tauq[5], tauq[2]}};
        const double Tiq[3][3] = {{tisoq[0], tisoq[3], tisoq[4]}, {tisoq[3], tisoq[1], tisoq[5]}, {tisoq[4], tisoq[5], tisoq[2]}};
        const double cs2       = 0.5 * cS;
        double       Tm[3][3], A[3][3], B[3][3], E[3][3], MT[3][3], Sk[3][3] = {};
#pragma unroll
        for (int k = 0; k < 3; ++k)
#pragma unroll
          for (int i = 0; i < 3; ++i)
            {
              Tm[k][i] = (-2.0 / 3.0) * (Mr[k * 3] * Tiq[0][i] + Mr[k * 3 + 1] * Tiq[1][i] + Mr[k * 3 + 2] * Tiq[2][i]);
              MT[k][i] = Mr[k * 3] * Tq[0][i] + Mr[k * 3 + 1] * Tq[1][i] + Mr[k * 3 + 2] * Tq[2][i]; // (M tau)_ki
            }
#pragma unroll
        for (int k = 0; k < 3; ++k)
#pragma unroll
          for (int i = 0; i < 3; ++i)
            {
              A[k][i] = wr * (cII * Mr[k * 3 + i] + Tm[k][i]);
              B[k][i] = wr * Mr[k * 3 + i];
              E[k][i] = (wr * cs2) * Mr[k * 3 + i];
            }
        // the fields of the pair (i, j), i <= j (for i == j: k <= l, with S_kl on top) -- the expressions of the fused kernel
        auto emit_pair = [&](auto I_, auto J_) __attribute__((always_inline)) {
          constexpr int i = decltype(I_)::value, j = decltype(J_)::value;
#pragma unroll
          for (int k = 0; k < 3; ++k)
#pragma unroll
            for (int l = 0; l < 3; ++l)
              {
                if (i == j && k > l)
                  continue;
                double c = A[k][i] * Mr[l * 3 + j] + B[k][i] * Tm[l][j] + E[k][j] * Mr[l * 3 + i];
                if (i == j)
                  c += Sk[k][l];
                s_C[q2sf_field(i, j, k, l) * FS + lane] = c;
              }
        };
        using I0 = std::integral_constant<int, 0>;
        using I1 = std::integral_constant<int, 1>;
        using I2 = std::integral_constant<int, 2>;
        if (wv == 0) // 18 fields (this wave has no contraction tables to set up)
          {
#pragma unroll
            for (int k = 0; k < 3; ++k)
#pragma unroll
              for (int l = k; l < 3; ++l)
                Sk[k][l] = E[k][0] * Mr[l * 3] + E[k][1] * Mr[l * 3 + 1] + E[k][2] * Mr[l * 3 + 2] +
                           wr * (MT[k][0] * Mr[l * 3] + MT[k][1] * Mr[l * 3 + 1] + MT[k][2] * Mr[l * 3 + 2]);
            emit_pair(I0{}, I0{});
            emit_pair(I1{}, I1{});
            emit_pair(I2{}, I2{});
          }
        else if (wv == 1)
          {
            emit_pair(I0{}, I1{});
            s_C[MASSF * FS + lane] = prm.alpha1 * prm.rho * wr;
          }
        else if (wv == 2)
          emit_pair(I0{}, I2{});
        else
          emit_pair(I1{}, I2{});
        if (tid < NPC)
          {
            s_ri[lane] = ri_l;
            s_cm[lane] = cm_l;
          }
      }
    if (tid < 64)
      {
       if constexpr (!REC)
       {
        // ================================================================= wave 0: quadrature points
        if constexpr (!RES_ONLY && (XV & 2) != 0)
          __builtin_amdgcn_s_setprio(3);
        double tiso[6], cII, cS;
        if constexpr (RECW0)
          {
            // bit 9: wave 0 ALONE starts from the point records (no gather, no gradients, no kinematics: the point pass has
            // done them) and forms all 45 fields as in the fused kernel; waves 1-3 carry no extra arithmetic
            double rec[MF_NREC], Finv[9], detJ;
            {
              const double *__restrict__ rp = prm.qrec + cell * int64_t(MF_NREC * 64) + lane;
#pragma unroll
              for (int f = 0; f < MF_NREC; ++f)
                rec[f] = rp[f * 64];
            }
            const int    qz = lane >> 4;
            const double wq = prm.tab1d[24 + (lane & 3)] * prm.tab1d[24 + ((lane >> 2) & 3)] * prm.tab1d[24 + qz];
            neo_hooke_from_F<3>(rec, det3x3(rec), rec[9], rec[10], prm.mu, prm.kappa, Finv, tau, tiso, cII, cS);
            if (prm.cellbox)
              {
                const double *__restrict__ cb = prm.cellbox + cell * 4;
                const double rx = cb[0], ry = cb[1], rz = cb[2];
                detJ            = cb[3];
#pragma unroll
                for (int k = 0; k < 3; ++k)
                  {
                    M[k]     = rx * Finv[k];
                    M[3 + k] = ry * Finv[3 + k];
                    M[6 + k] = rz * Finv[6 + k];
                  }
              }
            else
              {
                const double *__restrict__ cv = prm.cverts + cell * 24;
                const double xiq[3] = {prm.tab1d[28 + (lane & 3)], prm.tab1d[28 + ((lane >> 2) & 3)], prm.tab1d[28 + qz]};
                double       verts[24], Jm[9], Ji[9];
#pragma unroll
                for (int k = 0; k < 24; ++k)
                  verts[k] = cv[k];
                q1_jacobian<3>(verts, xiq, Jm);
                detJ = det3x3(Jm);
                inv3x3(Jm, detJ, Ji);
#pragma unroll
                for (int i = 0; i < 3; ++i)
#pragma unroll
                  for (int j = 0; j < 3; ++j)
                    M[i * 3 + j] = Ji[i * 3 + 0] * Finv[0 * 3 + j] + Ji[i * 3 + 1] * Finv[1 * 3 + j] + Ji[i * 3 + 2] * Finv[2 * 3 + j];
              }
            w = detJ * wq;
          }
        else
          {
        const int     qz = lane >> 4;
#pragma unroll
        for (int k = 0; k < 3; ++k)
          {
            Sz[k] = prm.tab1d[qz * 3 + k];
            Dz[k] = prm.tab1d[12 + qz * 3 + k];
          }
        const double wq = prm.tab1d[24 + (lane & 3)] * prm.tab1d[24 + ((lane >> 2) & 3)] * prm.tab1d[24 + qz];
        double       xiq[3] = {prm.tab1d[28 + (lane & 3)], prm.tab1d[28 + ((lane >> 2) & 3)], prm.tab1d[28 + qz]};
        int32_t      node = 0;
        // the cell's nodes: by arithmetic on a lattice (mi::CellLattice; the gathers below are then the wave's first
        // memory accesses) or from the connectivity
        const int32_t node0 = prm.lat.ncol > 0 ? lattice_node0(prm.lat, cell) : 0;
        if (lane < NPC)
          {
            if (prm.lat.ncol > 0)
              {
                const int k9 = lane / 9, r9 = lane - 9 * k9, j3 = r9 / 3, i3 = r9 - 3 * j3;
                node         = node0 + i3 + j3 * prm.lat.nn0 + k9 * prm.lat.nn01;
              }
            else
              node = prm.conn[cell * NPC + lane];
            s_conn[lane] = node;
            if constexpr (!RES_ONLY && !V2)
              {
                s_ri[lane] = prm.rowinfo[node];
                s_cm[lane] = prm.cmask[node] & 7;
              }
#pragma unroll
            for (int c = 0; c < 3; ++c)
              {
                const int64_t g    = int64_t(node) * 3 + c;
                s0[c * NPC + lane] = prm.u[g] + prm.du[g]; // get_total_solution, :580-588
                accn[c]            = prm.acc[g];
              }
          }
        // ---- two interpolations to the points: pass 0 = u (gradients), pass 1 = acceleration (values; V2: after barrier
        // (1), beside the contractions -- only the residual needs it)
        interp(0);
        if constexpr (!V2)
          {
            stage_acc();
            interp(1);
          }
        MI_STAMPW(8, 0); // gradients at the points
        // ---- geometry, kinematics, material at this point (nonlinear_elasticity.cc:927-934)
        {
          double Ji[9], gu[9], Finv[9], J, Fq[9], Jmq, rJq, detJ;
          if (prm.cellbox && prm.box_geometry) // every local cell an axis-parallel box (the reference's grids): 1/h and the volume,
            {                                  // as mf_spmv takes them (round 6)
              const double *__restrict__ cb = prm.cellbox + cell * 4;
#pragma unroll
              for (int k = 0; k < 9; ++k)
                Ji[k] = 0.0;
              Ji[0] = cb[0], Ji[4] = cb[1], Ji[8] = cb[2];
              detJ  = cb[3];
#pragma unroll
              for (int i = 0; i < 3; ++i)
#pragma unroll
                for (int j = 0; j < 3; ++j)
                  gu[i * 3 + j] = gxi[i][j] * Ji[j * 4];
            }
          else
            {
              const double *__restrict__ cv = prm.cverts + cell * 24;
              double verts[24], Jm[9];
#pragma unroll
              for (int k = 0; k < 24; ++k)
                verts[k] = cv[k];
              q1_jacobian<3>(verts, xiq, Jm);
              detJ = det3x3(Jm);
              inv3x3(Jm, detJ, Ji);
#pragma unroll
              for (int i = 0; i < 3; ++i)
#pragma unroll
                for (int j = 0; j < 3; ++j)
                  gu[i * 3 + j] = gxi[i][0] * Ji[0 * 3 + j] + gxi[i][1] * Ji[1 * 3 + j] + gxi[i][2] * Ji[2 * 3 + j];
            }
          neo_hooke_qp<3>(gu, prm.mu, prm.kappa, Finv, J, tau, tiso, cII, cS, Fq, Jmq, rJq);
          if (!(J > 0.0)) // inverted element (nonlinear_elasticity.cc:935 asserts det F > 0)
            *prm.inverted = 1.0;
          if (prm.cellbox && prm.box_geometry)
            {
#pragma unroll
              for (int i = 0; i < 3; ++i)
#pragma unroll
                for (int j = 0; j < 3; ++j)
                  M[i * 3 + j] = Ji[i * 4] * Finv[i * 3 + j];
            }
          else
            {
#pragma unroll
          for (int i = 0; i < 3; ++i)
#pragma unroll
            for (int j = 0; j < 3; ++j)
              M[i * 3 + j] = Ji[i * 3 + 0] * Finv[0 * 3 + j] + Ji[i * 3 + 1] * Finv[1 * 3 + j] + Ji[i * 3 + 2] * Finv[2 * 3 + j];
            }
          w = detJ * wq; // JxW of the reference configuration
          // (RES_ONLY with records: the tangent pass of the matrix-free fine level, round 6 -- the residual-only pass of
          // the Newton convergence check hands in no record pointer)
            if (prm.qrec) // the state the tangent is linearised at, for the matrix-free product (mf_spmv)
              {
                double *__restrict__ g = prm.qrec + cell * int64_t(MF_NREC * 64) + lane;
#pragma unroll
                for (int k = 0; k < 9; ++k)
                  g[k * 64] = Fq[k];
                g[9 * 64]  = Jmq;
                g[10 * 64] = rJq;
                if (prm.qrec32) // (opt-in fp32 smoother product)
                  {
                    float *__restrict__ g32 = prm.qrec32 + cell * int64_t(MF_NREC * 64) + lane;
#pragma unroll
                    for (int k = 0; k < 9; ++k)
                      g32[k * 64] = float(Fq[k]);
                    g32[9 * 64]  = float(Jmq);
                    g32[10 * 64] = float(rJq);
                  }
              }
        }
          } // !RECW0
        MI_STAMPW(9, 0); // material
        const double T[3][3]  = {{tau[0], tau[3], tau[4]}, {tau[3], tau[1], tau[5]}, {tau[4], tau[5], tau[2]}};
        // ---- coefficient fields for the tangent waves
        if constexpr (!RES_ONLY)
          {
            const double Ti[3][3] = {{tiso[0], tiso[3], tiso[4]}, {tiso[3], tiso[1], tiso[5]}, {tiso[4], tiso[5], tiso[2]}};
            const double cs2      = 0.5 * cS;
            double       Tm[3][3], A[3][3], B[3][3], E[3][3], Sk[3][3], MT[3][3];
#pragma unroll
            for (int k = 0; k < 3; ++k)
#pragma unroll
              for (int i = 0; i < 3; ++i)
                {
                  Tm[k][i] = (-2.0 / 3.0) * (M[k * 3] * Ti[0][i] + M[k * 3 + 1] * Ti[1][i] + M[k * 3 + 2] * Ti[2][i]);
                  MT[k][i] = M[k * 3] * T[0][i] + M[k * 3 + 1] * T[1][i] + M[k * 3 + 2] * T[2][i]; // (M tau)_ki
                }
#pragma unroll
            for (int k = 0; k < 3; ++k)
#pragma unroll
              for (int i = 0; i < 3; ++i)
                {
                  A[k][i] = w * (cII * M[k * 3 + i] + Tm[k][i]);
                  B[k][i] = w * M[k * 3 + i];
                  E[k][i] = (w * cs2) * M[k * 3 + i];
                }
#pragma unroll
            for (int k = 0; k < 3; ++k)
#pragma unroll
              for (int l = 0; l < 3; ++l)
                Sk[k][l] = E[k][0] * M[l * 3] + E[k][1] * M[l * 3 + 1] + E[k][2] * M[l * 3 + 2] +
                           w * (MT[k][0] * M[l * 3] + MT[k][1] * M[l * 3 + 1] + MT[k][2] * M[l * 3 + 2]);
#pragma unroll
            for (int i = 0; i < 3; ++i)
#pragma unroll
              for (int j = 0; j < 3; ++j)
#pragma unroll
                for (int k = 0; k < 3; ++k)
#pragma unroll
                  for (int l = 0; l < 3; ++l)
                    {
                      // V2: C^{ij}_{kl} = C^{ji}_{lk}, so only i <= j is stored, and for i == j only k <= l (45 fields)
                      if (V2 && (i > j || (i == j && k > l)))
                        continue;
                      double c = A[k][i] * M[l * 3 + j] + B[k][i] * Tm[l][j] + E[k][j] * M[l * 3 + i];
                      if (i == j)
                        c += Sk[k][l];
                      s_C[(V2 ? q2sf_field(i, j, k, l) : (i * 3 + j) * 9 + k * 3 + l) * FS + lane] = c;
                    }
            s_C[MASSF * FS + lane] = prm.alpha1 * prm.rho * w;
          }
        MI_STAMPW(10, 0); // fields stored
       } // !REC
      }
    else if constexpr (!RES_ONLY)
      {
        // ================================================================= waves 1-3: per-lane tables
        // P1[kx*2+lx][qx] = phi^kx_a1(qx) phi^lx_b1(qx), phi2[ky][qy] = phi^ky_a2(qy); phi^1 = N', phi^0 = N
#pragma unroll
        for (int q = 0; q < 4; ++q)
          {
            const double sa = prm.tab1d[q * 3 + a1], da = prm.tab1d[12 + q * 3 + a1];
            const double sb = prm.tab1d[q * 3 + b1], db = prm.tab1d[12 + q * 3 + b1];
            P1[0][q]        = sa * sb;
            P1[1][q]        = sa * db;
            P1[2][q]        = da * sb;
            P1[3][q]        = da * db;
            phi2[0][q]      = prm.tab1d[q * 3 + a2];
            phi2[1][q]      = prm.tab1d[12 + q * 3 + a2];
          }
        mflag = (ci == cj) ? 1.0 : 0.0;
        if constexpr (V2 && !REC)
          if (tid >= 192) // wave 3, idle until barrier (1) otherwise: where the 729 node-pair blocks of this cell go
            {
              int32_t node;
              if (prm.lat.ncol > 0)
                {
                  const int32_t node0 = lattice_node0(prm.lat, cell);
                  const int     k9 = lane / 9, r9 = lane - 9 * k9, j3 = r9 / 3, i3 = r9 - 3 * j3;
                  node               = node0 + i3 + j3 * prm.lat.nn0 + k9 * prm.lat.nn01;
                }
              else
                node = lane < NPC ? prm.conn[cell * NPC + lane] : 0;
              if (lane < NPC)
                {
                  s_ri[lane] = prm.rowinfo[node];
                  s_cm[lane] = prm.cmask[node] & 7;
                }
              MI_WAVE_SYNC();
              MI_STAMPW(11, 192); // wave 3: rowinfo there
              build_table(s_tab);
              MI_STAMPW(12, 192); // wave 3: table built
            }
      }
    if constexpr (!RES_ONLY && (XV & 2) != 0)
      __builtin_amdgcn_s_setprio(0);
    if constexpr (!RES_ONLY)
      __syncthreads(); // (1) fields complete -- the one barrier every wave of the workgroup passes, outside the role branches
    MI_STAMP(1);
    if constexpr (REC)
      {
        if (tid < 64) // wave 0, beside the contractions: the block table (its row info arrived with the records)
          build_table(s_tab);
      }
    else if constexpr (RECW0)
      {
      }
    else
    if (tid < 64)
      {
        const double T[3][3] = {{tau[0], tau[3], tau[4]}, {tau[3], tau[1], tau[5]}, {tau[4], tau[5], tau[2]}};
        if constexpr (V2)
          {
            stage_acc();
            interp(1);
          }
        // ---- residual: Q[i][k] = w sum_j tau_ij M_kj, V[i] = rho w (acc - b)_i, integrated against grad N_a / N_a
        MI_WAVE_SYNC();
#pragma unroll
        for (int i = 0; i < 3; ++i)
          {
            const int ql = lane ^ ((i & 1) << 4);
#pragma unroll
            for (int k = 0; k < 3; ++k)
              // explicit fma chain: the same arithmetic in the full and in the residual-only instantiation, whatever
              // else shares these products
              s0[(i * 4 + k) * 64 + ql] = w * fma(T[i][2], M[k * 3 + 2], fma(T[i][1], M[k * 3 + 1], T[i][0] * M[k * 3]));
            s0[(i * 4 + 3) * 64 + ql] = prm.rho * w * (accq[i] - prm.body[i]);
          }
        MI_WAVE_SYNC();
        if (lane < 48) // contract qz
          {
            const int c = lane >> 4;
            double    v[4][4];
#pragma unroll
            for (int d = 0; d < 4; ++d)
#pragma unroll
              for (int z = 0; z < 4; ++z)
                v[d][z] = v0[(c * 4 + d) * 64 + ((z * 16 + q16) ^ ((c & 1) << 4))];
#pragma unroll
            for (int k = 0; k < 3; ++k)
              {
                double cds = 0.0, csd = 0.0, css = 0.0;
#pragma unroll
                for (int z = 0; z < 4; ++z)
                  {
                    cds = fma(S[z][k], v[0][z], cds);
                    csd = fma(S[z][k], v[1][z], csd);
                    css = fma(D[z][k], v[2][z], css);
                    css = fma(S[z][k], v[3][z], css);
                  }
                const int o = c * 256 + ((k * 16 + q16) ^ ((c & 1) << 4));
                s0[o]       = cds;
                s0[o + 64]  = csd;
                s0[o + 128] = css;
              }
          }
        MI_WAVE_SYNC();
        if (lane < 36) // contract qy
          {
            double cds[4], csd[4], css[4];
#pragma unroll
            for (int qy = 0; qy < 4; ++qy)
              {
                const int c = pck / 3, kk = pck - 3 * c;
                const int o = c * 256 + ((kk * 16 + qy * 4 + pqx) ^ ((c & 1) << 4));
                cds[qy]     = v0[o];
                csd[qy]     = v0[o + 64];
                css[qy]     = v0[o + 128];
              }
#pragma unroll
            for (int j = 0; j < 3; ++j)
              {
                double ed = 0.0, es = 0.0;
#pragma unroll
                for (int qy = 0; qy < 4; ++qy)
                  {
                    ed = fma(S[qy][j], cds[qy], ed);
                    es = fma(D[qy][j], csd[qy], es);
                    es = fma(S[qy][j], css[qy], es);
                  }
                const int o = pqx * 27 + pck * 3 + j;
                sE[o]       = ed;
                sE[108 + o] = es;
              }
          }
        MI_WAVE_SYNC();
        if (lane < 27) // contract qx; residual scatter (:769-773; constrained rows get no rhs)
          {
            double ed[4], es[4];
#pragma unroll
            for (int qx = 0; qx < 4; ++qx)
              {
                ed[qx] = vE[qx * 27 + lane];
                es[qx] = vE[108 + qx * 27 + lane];
              }
            const int lc = lane / 9, lkj = lane - 9 * lc;
#pragma unroll
            for (int i = 0; i < 3; ++i)
              {
                double rv = 0.0;
#pragma unroll
                for (int qx = 0; qx < 4; ++qx)
                  {
                    rv = fma(D[qx][i], ed[qx], rv);
                    rv = fma(S[qx][i], es[qx], rv);
                  }
                if constexpr (RSLOTS)
                  {
                    prm.res_slots[int64_t(prm.slot_dst[cell * NPC + lkj * 3 + i]) * 3 + lc] = rv;
                    continue;
                  }
                const int32_t A = s_conn[lkj * 3 + i];
                if (!((prm.cmask[A] >> lc) & 1))
                  prm.rhs[int64_t(A) * 3 + lc] -= rv;
              }
          }
        if constexpr (RES_ONLY)
          return;
        if constexpr (!V2)
          {
            MI_WAVE_SYNC();
            build_table(reinterpret_cast<uint64_t *>(s_w));
          }
      }
    if constexpr (!RES_ONLY)
      {
        // ================================================================= waves 1-3: the element tangent
        double     Kacc[3][3][3]; // [a3][b3][b2]
#pragma unroll
        for (int a3 = 0; a3 < 3; ++a3)
#pragma unroll
          for (int b3 = 0; b3 < 3; ++b3)
#pragma unroll
            for (int b2 = 0; b2 < 3; ++b2)
              Kacc[a3][b3][b2] = 0.0;
        if (tid >= 64)
          {
            const double *__restrict__ cb = s_C + ij * 9 * FS;
            // V2 (45 fields): field of (kl) for this lane's (ci, cj): (ci, cj, k, l) for ci < cj, (cj, ci, l, k) for ci > cj,
            // (ci, ci, min, max) on the diagonal
            const int  pbase = q2sf_pair_base(ci < cj ? ci : cj, ci < cj ? cj : ci);
            const bool fdiag = ci == cj, fswap = ci > cj;
            auto       fld   = [&](const int kl) __attribute__((always_inline)) -> const double * {
              if (kl == 9)
                return s_C + MASSF * FS;
              if constexpr (V2)
                {
                  const int k = kl / 3, l = kl - 3 * k;
                  return s_C + (pbase + (fdiag ? q2sf_sym6(k < l ? k : l, k < l ? l : k) : fswap ? l * 3 + k : kl)) * FS;
                }
              else
                return cb + kl * FS;
            };
            if constexpr ((XV & 1) != 0)
              {
                // step (qz, pos = position of kl in the group order, qy); ring slot qy: the values of step (qz, pos, qy) are
                // requested by step (qz, pos - 1, qy)
                double2 ring[4][2];
                double  acc[3] = {0.0, 0.0, 0.0};
#pragma unroll
                for (int qy = 0; qy < 4; ++qy)
                  {
                    const double *__restrict__ f_ = fld(q2sf_kl_order(0)) + qy * 4;
                    ring[qy][0]                   = *reinterpret_cast<const double2 *>(f_);
                    ring[qy][1]                   = *reinterpret_cast<const double2 *>(f_ + 2);
                  }
#pragma unroll
                for (int qz = 0; qz < 4; ++qz)
#pragma unroll
                  for (int pos = 0; pos < 10; ++pos)
                    {
                      const int  kl = q2sf_kl_order(pos), k = kl / 3, l = kl - 3 * k;
                      const bool mass = kl == 9;
                      const int  kx = (!mass && k == 0), lx = (!mass && l == 0), ky = (!mass && k == 1), ly = (!mass && l == 1),
                                kz = (!mass && k == 2), lz = (!mass && l == 2);
                      // the step after this one in the same ring slot
                      const int  npos = pos == 9 ? 0 : pos + 1, nqz = pos == 9 ? qz + 1 : qz, nkl = q2sf_kl_order(npos);
                      const double *__restrict__ fn = fld(nkl) + nqz * 16;
#pragma unroll
                      for (int qy = 0; qy < 4; ++qy)
                        {
                          const double2 c01 = ring[qy][0], c23 = ring[qy][1];
                          double        t   = P1[kx * 2 + lx][0] * c01.x;
                          t                 = fma(P1[kx * 2 + lx][1], c01.y, t);
                          t                 = fma(P1[kx * 2 + lx][2], c23.x, t);
                          t                 = fma(P1[kx * 2 + lx][3], c23.y, t);
                          if (nqz < 4) // the slot is free again: its next occupant has three steps to arrive
                            {
                              ring[qy][0] = *reinterpret_cast<const double2 *>(fn + qy * 4);
                              ring[qy][1] = *reinterpret_cast<const double2 *>(fn + qy * 4 + 2);
                            }
                          if (mass)
                            t *= mflag;
                          const double ta = t * phi2[ky][qy];
#pragma unroll
                          for (int b2 = 0; b2 < 3; ++b2)
                            acc[b2] = fma(ta, ly ? D[qy][b2] : S[qy][b2], acc[b2]);
                          if ((qy & 1) == 1)
                            __builtin_amdgcn_sched_barrier(0);
                        }
                      // last field of an accumulator group (kz, lz): positions 4 (kl = 9), 6 (5), 8 (7), 9 (8)
                      if (pos == 4 || pos == 6 || pos == 8 || pos == 9)
                        {
#pragma unroll
                          for (int b2 = 0; b2 < 3; ++b2)
                            {
                              const double v = acc[b2];
                              acc[b2]        = 0.0;
#pragma unroll
                              for (int a3 = 0; a3 < 3; ++a3)
                                {
                                  const double va = v * (kz ? D[qz][a3] : S[qz][a3]);
#pragma unroll
                                  for (int b3 = 0; b3 < 3; ++b3)
                                    Kacc[a3][b3][b2] = fma(va, lz ? D[qz][b3] : S[qz][b3], Kacc[a3][b3][b2]);
                                }
                            }
                        }
                    }
              }
            else
              {
#pragma unroll
            for (int qz = 0; qz < 4; ++qz)
              {
                double acc[2][2][3]; // [k==z][l==z][b2]
#pragma unroll
                for (int x = 0; x < 12; ++x)
                  (&acc[0][0][0])[x] = 0.0;
#pragma unroll
                for (int kl = 0; kl < 10; ++kl)
                  {
                    const int  k = kl / 3, l = kl - 3 * k;
                    const bool mass = kl == 9;
                    const int  kx = (!mass && k == 0), lx = (!mass && l == 0), ky = (!mass && k == 1),
                              ly = (!mass && l == 1), kz = (!mass && k == 2), lz = (!mass && l == 2);
                    const double *__restrict__ f = fld(kl);
#pragma unroll
                    for (int qy = 0; qy < 4; ++qy)
                      {
                        const double2 c01 = *reinterpret_cast<const double2 *>(f + qz * 16 + qy * 4);
                        const double2 c23 = *reinterpret_cast<const double2 *>(f + qz * 16 + qy * 4 + 2);
                        double t = P1[kx * 2 + lx][0] * c01.x;
                        t        = fma(P1[kx * 2 + lx][1], c01.y, t);
                        t        = fma(P1[kx * 2 + lx][2], c23.x, t);
                        t        = fma(P1[kx * 2 + lx][3], c23.y, t);
                        if (mass)
                          t *= mflag;
                        const double ta = t * phi2[ky][qy];
#pragma unroll
                        for (int b2 = 0; b2 < 3; ++b2)
                          acc[kz][lz][b2] = fma(ta, ly ? D[qy][b2] : S[qy][b2], acc[kz][lz][b2]);
                      }
                  }
#pragma unroll
                for (int kz = 0; kz < 2; ++kz)
#pragma unroll
                  for (int lz = 0; lz < 2; ++lz)
#pragma unroll
                    for (int b2 = 0; b2 < 3; ++b2)
                      {
                        const double v = acc[kz][lz][b2];
#pragma unroll
                        for (int a3 = 0; a3 < 3; ++a3)
                          {
                            const double va = v * (kz ? D[qz][a3] : S[qz][a3]);
#pragma unroll
                            for (int b3 = 0; b3 < 3; ++b3)
                              Kacc[a3][b3][b2] = fma(va, lz ? D[qz][b3] : S[qz][b3], Kacc[a3][b3][b2]);
                          }
                      }
              }
              }
          }
        MI_STAMP(2);
        __syncthreads(); // (2) fields consumed: the element tangent image goes on top of them
        MI_STAMP(3);
        if (active)
          {
#pragma unroll
            for (int a3 = 0; a3 < 3; ++a3)
#pragma unroll
              for (int b3 = 0; b3 < 3; ++b3)
#pragma unroll
                for (int b2 = 0; b2 < 3; ++b2)
                  {
                    // wanted: (a1,a2,a3) >= (b1,b2,b3), x most significant (a1 >= b1 by construction)
                    const bool want = a1 > b1 || a2 > b2 || (a2 == b2 && a3 >= b3);
                    const int  a = a1 + 3 * a2 + 9 * a3, b = b1 + 3 * b2 + 9 * b3;
                    if (want)
                      {
                        const bool low = a >= b; // lower triangle of the node order; otherwise the transposed block of (b,a)
                        const int  hi = low ? a : b, lo = low ? b : a, e = low ? ci * 3 + cj : cj * 3 + ci;
                        s_C[e * EBE_NBLK + hi * (hi + 1) / 2 + lo] = Kacc[a3][b3][b2];
                      }
                  }
          }
        __syncthreads(); // (3) image complete
        MI_STAMP(4);
        MI_STAMP(5);
        // ---- tangent scatter.  [DEAL.II distribute_local_to_global] constrained rows/cols are dropped, the diagonal of a
        // constrained dof receives |K_e(i,i)|.  Lane = ENTRY of a block: thread t owns entry e = t % 9 of block t / 9 of
        // every group of 28 blocks, so 9 consecutive lanes store one 72-byte block and a wave instruction covers 7 whole
        // blocks (round 2: one thread per block and nine 8-byte accesses each, i.e. 64 pieces of 64 different blocks per
        // instruction).  Everything that depends on the block comes out of wave 0's table in one LDS read; what depends
        // on the entry is constant per thread.
        {
          const uint64_t *const tab = V2 ? s_tab : reinterpret_cast<const uint64_t *>(s_w);
          const int      tq = tid / 9, e = tid - 9 * tq, ei = e / 3, ej = e - 3 * ei;
          const int      src_low = e * EBE_NBLK, src_tr = (ej * 3 + ei) * EBE_NBLK; // image rows of the entry / its transpose
          const uint32_t cm      = (1u << ei) | (8u << ej);                          // constraint bits that kill this entry
          const bool     on_diag = ei == ej;
          double *const  vbase   = prm.vals + e;
          const bool     plain   = s_plain != 0; // uniform: the table is complete since barrier (2)
          if constexpr (SLIM)
            {
              // Branch-free form (round 5): a first touch loads the ZERO block behind the matrix instead of skipping its load, a
              // node without a row here (ghost node of a slab) has the TRASH block as its position, so every lane runs the
              // same straight-line code: per entry one table read with an immediate offset, a select, an address, a load; then
              // the image value, an add, a store.  Threads 252-255 own no entry; of round 26 only block 728 exists (thread
              // group 0).  The masking of constrained entries is a second instantiation for the few cells that need it.
              if (tq < 28)
                {
                  const uint64_t *const tb = tab + tq;
                  auto rounds = [&](auto plain_c) __attribute__((always_inline)) {
                    constexpr bool PLAIN = decltype(plain_c)::value;
                    double         old[27];
#pragma unroll
                    for (int r = 0; r < 27; ++r)
                      {
                        old[r] = 0.0;
                        if (r < 26 || tq == 0)
                          {
                            const uint64_t t  = tb[r * 28];
                            const uint32_t ld = ((t >> 42) & 1) ? prm.zero_blk : uint32_t(t);
                            old[r]            = vbase[int64_t(ld) * 9];
                          }
                      }
#pragma unroll
                    for (int r = 0; r < 27; ++r)
                      if (r < 26 || tq == 0)
                        {
                          const uint64_t t  = tb[r * 28];
                          const uint32_t fl = uint32_t(t >> 32);
                          double         w_ = s_C[((fl >> 9) & 1 ? src_low : src_tr) + int(fl & 511)];
                          if constexpr (!PLAIN)
                            if ((fl >> 12) & cm)
                              w_ = (((fl >> 11) & 1) && on_diag) ? fabs(w_) : 0.0;
                          vbase[int64_t(uint32_t(t)) * 9] = w_ + old[r];
                        }
                  };
                  if (plain)
                    rounds(std::true_type{});
                  else
                    rounds(std::false_type{});
                }
            }
          else
            {
          // phase A: the old values of every entry that is not a first touch, all requested before anything is stored
          // (loads and stores share the wave's memory counter: a load issued after a store waits for that store).
          // Batches of nine: the table words of a batch in one LDS round trip, then its nine loads.
          double old[27];
#pragma unroll
          for (int bb = 0; bb < 27; bb += 9)
            {
              uint64_t t[9];
#pragma unroll
              for (int u = 0; u < 9; ++u)
                {
                  const int blk = (bb + u) * 28 + tq;
                  t[u]          = tab[(tq < 28 && blk < NPC * NPC) ? blk : 0];
                }
#pragma unroll
              for (int u = 0; u < 9; ++u)
                {
                  const int  blk = (bb + u) * 28 + tq;
                  const bool rd  = tq < 28 && blk < NPC * NPC && (plain || uint32_t(t[u]) != 0xffffffffu) && !((t[u] >> 42) & 1);
                  old[bb + u]    = rd ? vbase[int64_t(uint32_t(t[u])) * 9] : 0.0;
                }
            }
          // phase B: masked entries out of the image, summed, stored; per batch two LDS round trips (table words, image)
#pragma unroll
          for (int bb = 0; bb < 27; bb += 9)
            {
              uint64_t t[9];
              double   v[9];
#pragma unroll
              for (int u = 0; u < 9; ++u)
                {
                  const int blk = (bb + u) * 28 + tq;
                  t[u]          = tab[(tq < 28 && blk < NPC * NPC) ? blk : 0];
                }
#pragma unroll
              for (int u = 0; u < 9; ++u)
                {
                  const uint32_t fl = uint32_t(t[u] >> 32);
                  v[u]              = s_C[((fl >> 9) & 1 ? src_low : src_tr) + int(fl & 511)];
                }
#pragma unroll
              for (int u = 0; u < 9; ++u)
                {
                  const int      blk = (bb + u) * 28 + tq;
                  const uint32_t fl  = uint32_t(t[u] >> 32);
                  double         w_  = v[u];
                  if (!plain && ((fl >> 12) & cm))
                    w_ = (((fl >> 11) & 1) && on_diag) ? fabs(w_) : 0.0;
                  if (tq < 28 && blk < NPC * NPC && (plain || uint32_t(t[u]) != 0xffffffffu))
                    {
                      vbase[int64_t(uint32_t(t[u])) * 9] = w_ + old[bb + u];
                    }
                }
            }
            }
          if (prm.ke) // the cell's own masked blocks (what entered the global matrix) for the element-tangent product: the
            {         // image is masked in place, in a pass of its own so that nothing above waits for LDS stores
              __syncthreads();
#pragma unroll 1
              for (int u = 0; u < 27; ++u)
                {
                  const int blk = u * 28 + tq;
                  if (tq < 28 && blk < NPC * NPC)
                    {
                      const uint32_t fl = uint32_t(tab[blk] >> 32);
                      if (((fl >> 9) & 1) && ((fl >> 12) & cm))
                        {
                          const int li = src_low + int(fl & 511);
                          s_C[li]      = (((fl >> 11) & 1) && on_diag) ? fabs(s_C[li]) : 0.0;
                        }
                    }
                }
            }
        }
        MI_STAMP(6);
#undef MI_STAMP
#undef MI_STAMPW
        if (prm.ke)
          {
            __syncthreads();
            double *__restrict__ dst = prm.ke + cell * (int64_t(9) * EBE_NBLK);
            for (int i = tid; i < 9 * EBE_NBLK; i += 256)
              dst[i] = s_C[i];
          }
      }
  }

  // ------------------------------------------------------------------ linear model: K, M, stepping matrix, body force
  // linear_elasticity.cc:248-374.  The operators are constant: assembled once per set-up, so this kernel is written for
  // generality (any dim / degree the mesh tables allow, runtime loops), not for speed.  One workgroup per cell of the
  // colour; the points' inverse Jacobians and weights go to LDS, then every thread walks over node pairs (a,b):
  //   K_ab[ci][cj] = sum_q (lambda g_a[ci] g_b[cj] + mu g_a[cj] g_b[ci] + delta_cicj mu g_a.g_b) JxW      (:301-320)
  //   M_ab         = sum_q rho N_a N_b JxW  (on the diagonal of the block)                                  (:341-345)
  //   A_ab         = M_ab + theta^2 dt^2 K_ab with zero boundary values applied cell by cell: entries of constrained
  //                  rows / columns dropped, the diagonal kept -- on the summed matrix that is
  //                  MatrixTools::apply_boundary_values (:348-353, :426-451)
  // and added to the blocks of the three operators (colouring: no other cell of the launch holds the same block).
  template <int DIM>
  __global__ __launch_bounds__(256) void assemble_linear_cells(LinAsmParams prm)
  {
    constexpr int MAXQ = 125, DD = DIM * DIM;
    __shared__ double s_ji[MAXQ * 9], s_w[MAXQ], s_verts[8 * 3];
    const int     tid = threadIdx.x, np1 = prm.np1, nq1 = prm.nq1;
    const int     npc = (DIM == 2) ? np1 * np1 : np1 * np1 * np1, nq = (DIM == 2) ? nq1 * nq1 : nq1 * nq1 * nq1;
    const int64_t cell = prm.cell_begin + blockIdx.x;
    const double *__restrict__ N1 = prm.tab, *__restrict__ dN1 = prm.tab + nq1 * np1, *__restrict__ qw = prm.tab + 2 * nq1 * np1,
                               *__restrict__ qx = qw + nq1;
    if (tid < (1 << DIM) * DIM)
      s_verts[tid] = prm.cverts[cell * ((1 << DIM) * DIM) + tid];
    __syncthreads();
    if (tid < nq)
      {
        const int qi[3] = {tid % nq1, (tid / nq1) % nq1, DIM == 3 ? tid / (nq1 * nq1) : 0};
        double    xi[3] = {0.0, 0.0, 0.0}, w = 1.0, Jm[9], Ji[9];
#pragma unroll
        for (int d = 0; d < DIM; ++d)
          {
            xi[d] = qx[qi[d]];
            w *= qw[qi[d]];
          }
        q1_jacobian<DIM>(s_verts, xi, Jm);
        const double detJ = det3x3(Jm);
        inv3x3(Jm, detJ, Ji);
#pragma unroll
        for (int k = 0; k < 9; ++k)
          s_ji[tid * 9 + k] = Ji[k];
        s_w[tid] = detJ * w;
      }
    __syncthreads();
    // value and real-space gradient of shape function (ai) at point (qi)
    auto shape = [&](const int *qi, const int *ai, const double *Ji, double &n, double *g) {
      double v[3], d[3];
#pragma unroll
      for (int k = 0; k < DIM; ++k)
        {
          v[k] = N1[qi[k] * np1 + ai[k]];
          d[k] = dN1[qi[k] * np1 + ai[k]];
        }
      double dn[3];
      if constexpr (DIM == 2)
        {
          n     = v[0] * v[1];
          dn[0] = d[0] * v[1];
          dn[1] = v[0] * d[1];
          dn[2] = 0.0;
        }
      else
        {
          n     = v[0] * v[1] * v[2];
          dn[0] = d[0] * v[1] * v[2];
          dn[1] = v[0] * d[1] * v[2];
          dn[2] = v[0] * v[1] * d[2];
        }
#pragma unroll
      for (int i = 0; i < DIM; ++i)
        g[i] = dn[0] * Ji[0 * 3 + i] + dn[1] * Ji[1 * 3 + i] + (DIM == 3 ? dn[2] * Ji[2 * 3 + i] : 0.0);
    };
    for (int pair = tid; pair < npc * npc; pair += 256)
      {
        const int a = pair / npc, b = pair - a * npc;
        const int ai[3] = {a % np1, (a / np1) % np1, DIM == 3 ? a / (np1 * np1) : 0};
        const int bi[3] = {b % np1, (b / np1) % np1, DIM == 3 ? b / (np1 * np1) : 0};
        double    Kb[DD], Mb = 0.0;
#pragma unroll
        for (int k = 0; k < DD; ++k)
          Kb[k] = 0.0;
        for (int q = 0; q < nq; ++q)
          {
            const int     qi[3] = {q % nq1, (q / nq1) % nq1, DIM == 3 ? q / (nq1 * nq1) : 0};
            const double *Ji    = &s_ji[q * 9];
            double        na, nb, ga[3], gb[3];
            shape(qi, ai, Ji, na, ga);
            shape(qi, bi, Ji, nb, gb);
            const double w  = s_w[q];
            double       gg = 0.0;
#pragma unroll
            for (int k = 0; k < DIM; ++k)
              gg += ga[k] * gb[k];
#pragma unroll
            for (int ci = 0; ci < DIM; ++ci)
#pragma unroll
              for (int cj = 0; cj < DIM; ++cj)
                Kb[ci * DIM + cj] += (ga[ci] * gb[cj] * prm.lambda + ga[cj] * gb[ci] * prm.mu + (ci == cj ? gg * prm.mu : 0.0)) * w;
            Mb += prm.rho * na * nb * w;
          }
        const int32_t A  = prm.conn[cell * npc + a], B = prm.conn[cell * npc + b];
        const int2    ia = prm.rowinfo[A];
        if (ia.x < 0) // ghost row of a slab
          continue;
        const uint32_t o   = prm.off[(cell * npc + a) * npc + b];
        const int64_t  pos = (int64_t(ia.x) + int64_t((o >> 4) & 0x7ff) * ia.y + (o & 15)) * DD;
        const int      cma = prm.cmask[A], cmb = prm.cmask[B];
#pragma unroll
        for (int ci = 0; ci < DIM; ++ci)
#pragma unroll
          for (int cj = 0; cj < DIM; ++cj)
            {
              const double kv = Kb[ci * DIM + cj], mv = (ci == cj) ? Mb : 0.0;
              prm.K[pos + ci * DIM + cj] += kv;
              if (ci == cj)
                prm.M[pos + ci * DIM + cj] += mv;
              const bool drop = (((cma >> ci) | (cmb >> cj)) & 1) && !(a == b && ci == cj);
              if (!drop)
                prm.A[pos + ci * DIM + cj] += kv * prm.ctheta + mv;
            }
      }
    if (prm.bodyvec && tid < npc) // create_right_hand_side with rho b (:358-373)
      {
        const int a     = tid;
        const int ai[3] = {a % np1, (a / np1) % np1, DIM == 3 ? a / (np1 * np1) : 0};
        double    s     = 0.0;
        for (int q = 0; q < nq; ++q)
          {
            const int qi[3] = {q % nq1, (q / nq1) % nq1, DIM == 3 ? q / (nq1 * nq1) : 0};
            double    n     = 1.0;
#pragma unroll
            for (int k = 0; k < DIM; ++k)
              n *= N1[qi[k] * np1 + ai[k]];
            s += n * s_w[q];
          }
        const int64_t g = int64_t(prm.conn[cell * npc + a]) * DIM;
#pragma unroll
        for (int ci = 0; ci < DIM; ++ci)
          prm.bodyvec[g + ci] += prm.rho * prm.body[ci] * s;
      }
  }

  // ------------------------------------------------------------------ Neumann faces (:791-859)
  // one 64-thread workgroup per cell of the current colour that owns interface faces; the cell's faces are
  // processed one after the other (faces of one cell share edge/corner nodes, cells of one colour do not)
  template <int DIM, int P>
  __global__ __launch_bounds__(64) void neumann_faces(AsmParams prm, const int32_t *__restrict__ faces, int face_begin)
  {
    using E = Elem<DIM, P>;
    constexpr int NPC = E::NPC, NQ1 = E::NQ1, NP1 = E::NP1, NQF = E::NQF, NV = E::NV;
    __shared__ double s_N1[NQ1 * NP1], s_dN1[NQ1 * NP1], s_qw[NQ1], s_qx[NQ1];
    __shared__ double s_u[NPC * 3], s_t[NPC * 3], s_verts[NV * DIM];
    __shared__ int    s_conn[NPC];
    __shared__ double s_fq[NQF * 4]; // per face QP: referential traction (3) and unused
    // AsmParams::face_slots (round 6): ALL interface faces in one launch -- a cell's contributions (its faces summed in the
    // order below) go to the cell's slot [entry][NPC * DIM] instead of being added to system_rhs colour by colour;
    // neumann_gather adds the slots of a node in entry order (deterministic, two launches instead of eight per assembly)
    __shared__ double s_acc[NPC * DIM];
    const int     tid  = threadIdx.x;
    const int32_t cell = faces[2 * (face_begin + blockIdx.x)], fmask = faces[2 * (face_begin + blockIdx.x) + 1];
    for (int i = tid; i < NPC * DIM; i += 64)
      s_acc[i] = 0.0;
    for (int i = tid; i < NQ1 * NP1; i += 64)
      {
        s_N1[i]  = prm.tab1d[i];
        s_dN1[i] = prm.tab1d[NQ1 * NP1 + i];
      }
    if (tid < NQ1)
      {
        s_qw[tid] = prm.tab1d[2 * NQ1 * NP1 + tid];
        s_qx[tid] = prm.tab1d[2 * NQ1 * NP1 + NQ1 + tid];
      }
    for (int i = tid; i < NPC; i += 64) // 3D Q4 has 125 nodes per cell
      s_conn[i] = prm.conn[int64_t(cell) * NPC + i];
    if (tid < NV * DIM)
      s_verts[tid] = prm.cverts[int64_t(cell) * (NV * DIM) + tid];
    __syncthreads();
    for (int i = tid; i < NPC * 3; i += 64)
      {
        const int a = i / 3, c = i - a * 3;
        double    uv = 0.0, tv = 0.0;
        if (c < DIM)
          {
            const int64_t g = int64_t(s_conn[a]) * DIM + c;
            uv              = prm.u[g] + prm.du[g];
            tv              = prm.stress[g];
          }
        s_u[i] = uv;
        s_t[i] = tv;
      }
    __syncthreads();

    // The lattice may lie rotated over the box (decomposition along a direction other than the last, mi::AxisMap): lattice
    // direction d of the cell runs along physical coordinate ext[d], backwards if rev[d].  The ORDER of the quadrature
    // points matters here -- the reference pairs face point fq with CELL point fq (:825-827), both counted in physical
    // directions, x fastest -- so points are counted physically and translated to the cell's own directions.
    int ext[3] = {0, 1, 2}, rev[3] = {0, 0, 0}, loc[3] = {0, 1, 2};
    for (int d = 0; d < DIM; ++d)
      {
        ext[d]      = (prm.axmap >> (2 * d)) & 3;
        rev[d]      = (prm.axmap >> (6 + d)) & 1;
        loc[ext[d]] = d;
      }
    auto q1d = [&](int d, int q) { return rev[d] ? NQ1 - 1 - q : q; }; // 1D point q of a physical direction, on lattice direction d
    for (int f = 0; f < 2 * DIM; ++f)
      {
    if (!((fmask >> f) & 1))
      continue;
    const int nd = f >> 1, side = f & 1;
    // face-local axes [DEAL.II, recalled], in PHYSICAL directions: x-normal (y,z); y-normal (z,x) in 3D, (x) in 2D;
    // z-normal (x,y); ax0 / ax1 are the lattice directions they correspond to
    const int nde = ext[nd];
    int       ax0, ax1;
    if (DIM == 2)
      {
        ax0 = loc[nde == 0 ? 1 : 0];
        ax1 = 2;
      }
    else
      {
        ax0 = loc[nde == 0 ? 1 : (nde == 1 ? 2 : 0)];
        ax1 = loc[nde == 0 ? 2 : (nde == 1 ? 0 : 1)];
      }
    // four lanes per face point: they share the two loops over the cell's nodes (displacement gradient, traction) and
    // sum their parts by shuffles; everything else of a point they compute alike (16 points x 27 nodes on 16 lanes was
    // 40 us per launch: 4 % of a step at 1 M dofs)
    for (int fl = tid; fl < NQF * 4; fl += 64)
      {
        const int fq = fl >> 2, part = fl & 3;
        const int f1 = q1d(ax0, fq % NQ1), f2 = (DIM == 3) ? q1d(ax1, fq / NQ1) : 0; // face point, on the lattice directions
        // the cell point with the same PHYSICAL index, as this cell counts it
        int cq;
        {
          const int qe[3] = {fq % NQ1, (fq / NQ1) % NQ1, (DIM == 3) ? fq / (NQ1 * NQ1) : 0};
          int       ql[3] = {0, 0, 0};
          for (int d = 0; d < DIM; ++d)
            ql[d] = q1d(d, qe[ext[d]]);
          cq = ql[0] + NQ1 * (ql[1] + NQ1 * ql[2]);
        }
        // (1) F at the CELL quadrature point with index fq  -- the reference's quirk (:825-827 vs :902-903) -- or, with
        // "correct_face_F", at the face quadrature point itself: the 1D bases at the face's end of the normal direction
        // (values 0 / 1, derivatives from the table behind qx) and at the face point's abscissae along the face
        double gxi[9];
#pragma unroll
        for (int k = 0; k < 9; ++k)
          gxi[k] = 0.0;
        const double *dN_end = prm.tab1d + 2 * NQ1 * NP1 + 2 * NQ1 + side * NP1;
        for (int a = part; a < NPC; a += 4)
          {
            double N, dN[3];
            if (prm.correct_face_F)
              {
                const int ai[3] = {a % NP1, (a / NP1) % NP1, (DIM == 3) ? a / (NP1 * NP1) : 0};
                double    n1[3] = {1.0, 1.0, 1.0}, d1[3] = {0.0, 0.0, 0.0};
                n1[nd]  = (ai[nd] == (side ? P : 0)) ? 1.0 : 0.0;
                d1[nd]  = dN_end[ai[nd]];
                n1[ax0] = s_N1[f1 * NP1 + ai[ax0]];
                d1[ax0] = s_dN1[f1 * NP1 + ai[ax0]];
                if (DIM == 3)
                  {
                    n1[ax1] = s_N1[f2 * NP1 + ai[ax1]];
                    d1[ax1] = s_dN1[f2 * NP1 + ai[ax1]];
                  }
                N     = n1[0] * n1[1] * n1[2];
                dN[0] = d1[0] * n1[1] * n1[2];
                dN[1] = n1[0] * d1[1] * n1[2];
                dN[2] = (DIM == 3) ? n1[0] * n1[1] * d1[2] : 0.0;
              }
            else
              shape_at_qp<DIM, P>(s_N1, s_dN1, cq, a, N, dN);
#pragma unroll
            for (int i = 0; i < DIM; ++i)
#pragma unroll
              for (int j = 0; j < DIM; ++j)
                gxi[i * 3 + j] += s_u[a * 3 + i] * dN[j];
          }
#pragma unroll
        for (int k = 0; k < 9; ++k)
          {
            gxi[k] += __shfl_xor(gxi[k], 1);
            gxi[k] += __shfl_xor(gxi[k], 2);
          }
        double xi[3] = {0, 0, 0};
        {
          int qi[3] = {cq % NQ1, (cq / NQ1) % NQ1, (DIM == 3) ? cq / (NQ1 * NQ1) : 0};
#pragma unroll
          for (int d = 0; d < DIM; ++d)
            xi[d] = s_qx[qi[d]];
        }
        if (prm.correct_face_F) // the Jacobian at the face point as well
          {
            xi[nd]  = side ? 1.0 : 0.0;
            xi[ax0] = s_qx[f1];
            if (DIM == 3)
              xi[ax1] = s_qx[f2];
          }
        double Jm[9], Ji[9], F[9], Fi[9];
        q1_jacobian<DIM>(s_verts, xi, Jm);
        inv3x3(Jm, det3x3(Jm), Ji);
#pragma unroll
        for (int i = 0; i < 3; ++i)
#pragma unroll
          for (int j = 0; j < 3; ++j)
            F[i * 3 + j] = (i == j ? 1.0 : 0.0) + gxi[i * 3 + 0] * Ji[0 * 3 + j] + gxi[i * 3 + 1] * Ji[1 * 3 + j] +
                           gxi[i * 3 + 2] * Ji[2 * 3 + j];
        if (DIM == 2)
          F[8] = 1.0;
        const double J = det3x3(F);
        inv3x3(F, J, Fi);
        // (2) face geometry at the face quadrature point
        double xf[3] = {0, 0, 0};
        xf[nd]       = side ? 1.0 : 0.0;
        xf[ax0]      = s_qx[f1];
        double wf    = s_qw[f1];
        if (DIM == 3)
          {
            xf[ax1] = s_qx[f2];
            wf *= s_qw[f2];
          }
        q1_jacobian<DIM>(s_verts, xf, Jm);
        double cr[3];
        if (DIM == 2)
          {
            const int t = nd == 0 ? 1 : 0;
            cr[0]       = Jm[1 * 3 + t];
            cr[1]       = -Jm[0 * 3 + t];
            cr[2]       = 0.0;
          }
        else
          {
            const int    t1 = (nd + 1) % 3, t2 = (nd + 2) % 3;
            const double a0 = Jm[0 * 3 + t1], a1 = Jm[1 * 3 + t1], a2 = Jm[2 * 3 + t1];
            const double b0 = Jm[0 * 3 + t2], b1 = Jm[1 * 3 + t2], b2 = Jm[2 * 3 + t2];
            cr[0]           = a1 * b2 - a2 * b1;
            cr[1]           = a2 * b0 - a0 * b2;
            cr[2]           = a0 * b1 - a1 * b0;
          }
        double dotn = cr[0] * Jm[0 * 3 + nd] + cr[1] * Jm[1 * 3 + nd] + cr[2] * Jm[2 * 3 + nd];
        double sg   = (dotn < 0 ? -1.0 : 1.0) * (side ? 1.0 : -1.0); // outward
        const double len = sqrt(cr[0] * cr[0] + cr[1] * cr[1] + cr[2] * cr[2]);
        double       nrm[3] = {sg * cr[0] / len, sg * cr[1] / len, sg * cr[2] / len};
        // n* = det F F^{-T} N  (:831-833)
        double ns[3];
#pragma unroll
        for (int i = 0; i < 3; ++i)
          ns[i] = J * (Fi[0 * 3 + i] * nrm[0] + Fi[1 * 3 + i] * nrm[1] + Fi[2 * 3 + i] * nrm[2]);
        const double nn = sqrt(ns[0] * ns[0] + ns[1] * ns[1] + (DIM == 3 ? ns[2] * ns[2] : 0.0));
        // (3) traction interpolated on the face (:815-816): only nodes on the face have non-zero shape values
        double ts[3] = {0, 0, 0};
        for (int a = part; a < NPC; a += 4)
          {
            int ai[3] = {a % NP1, (a / NP1) % NP1, (DIM == 3) ? a / (NP1 * NP1) : 0};
            if (ai[nd] != (side ? P : 0))
              continue;
            double N = s_N1[f1 * NP1 + ai[ax0]];
            if (DIM == 3)
              N *= s_N1[f2 * NP1 + ai[ax1]];
#pragma unroll
            for (int c = 0; c < 3; ++c)
              ts[c] += N * s_t[a * 3 + c];
          }
#pragma unroll
        for (int c = 0; c < 3; ++c)
          {
            ts[c] += __shfl_xor(ts[c], 1);
            ts[c] += __shfl_xor(ts[c], 2);
          }
        const double sc = nn * len * wf; // ||n*|| JxW_face  (:836-837, :850-851)
        if (part == 0)
          {
            s_fq[fq * 4 + 0] = ts[0] * sc;
            s_fq[fq * 4 + 1] = ts[1] * sc;
            s_fq[fq * 4 + 2] = ts[2] * sc;
          }
      }
    __syncthreads();
    // (4) rhs_i += N_i * referential_stress[c_i] * JxW  (:839-856)
    for (int i = tid; i < NPC * DIM; i += 64)
      {
        const int a = i / DIM, c = i - a * DIM;
        int       ai[3] = {a % NP1, (a / NP1) % NP1, (DIM == 3) ? a / (NP1 * NP1) : 0};
        if (ai[nd] != (side ? P : 0))
          continue;
        double s = 0.0;
        for (int fq = 0; fq < NQF; ++fq)
          {
            const int f1 = q1d(ax0, fq % NQ1), f2 = (DIM == 3) ? q1d(ax1, fq / NQ1) : 0;
            double    N  = s_N1[f1 * NP1 + ai[ax0]];
            if (DIM == 3)
              N *= s_N1[f2 * NP1 + ai[ax1]];
            s += N * s_fq[fq * 4 + c];
          }
        if (prm.face_slots)
          {
            s_acc[i] += s; // (this thread alone touches entry i)
            continue;
          }
        const int32_t A = s_conn[a];
        if (!((prm.cmask[A] >> c) & 1))
          prm.rhs[int64_t(A) * DIM + c] += s;
      }
    __syncthreads(); // s_fq is reused by the next face
      }
    if (prm.face_slots)
      for (int i = tid; i < NPC * DIM; i += 64)
        prm.face_slots[int64_t(face_begin + blockIdx.x) * (NPC * DIM) + i] = s_acc[i];
  }

  // system_rhs += the interface faces' contributions, node by node in entry order (see neumann_faces, face_slots): one
  // thread per (interface node, component); src[j] = entry * NPC + local node of the j-th contribution of the node
  template <int DIM>
  __global__ __launch_bounds__(256) void neumann_gather(const double *__restrict__ slots, const int32_t *__restrict__ node_ids,
                                                        const int32_t *__restrict__ start, const int32_t *__restrict__ src,
                                                        const uint8_t *__restrict__ cmask, double *rhs, int nnodes_if)
  {
    const int t = blockIdx.x * 256 + threadIdx.x;
    if (t >= nnodes_if * DIM)
      return;
    const int k = t / DIM, c = t - k * DIM;
    const int32_t A = node_ids[k];
    if ((cmask[A] >> c) & 1)
      return;
    double s = 0.0;
    for (int32_t j = start[k]; j < start[k + 1]; ++j)
      s += slots[int64_t(src[j]) * DIM + c];
    rhs[int64_t(A) * DIM + c] += s;
  }

  // ------------------------------------------------------------------ row-per-wave product (cross-check variant)
  // y = K x read the other way round: one wavefront per block row, lane = (block of the row, entry), through the block
  // pattern (rowptr / col) and the row's position in the value array -- shares neither the slice bookkeeping nor the
  // column generator of the production kernel, which is what makes it a cross-check (tests; "spmv_variant" 1).
  template <int D>
  __global__ __launch_bounds__(256) void blockrow_spmv_check(SpmvParams prm)
  {
    if (prm.done && *prm.done)
      return;
    constexpr int DD = D * D, BPW = 64 / DD;
    __shared__ double s_red[4];
    const int     lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int     nwg = gridDim.x, b = blockIdx.x;
    const int64_t per = (prm.nrows + nwg - 1) / nwg;
    const int64_t r0 = prm.row0 + b * per, r1 = imin64(prm.row0 + prm.nrows, r0 + per);
    const int     kb = lane / DD, e = lane - kb * DD, i = e / D, j = e - i * D;
    double        dsum = 0.0;
    for (int64_t row = r0 + wave; row < r1; row += 4)
      {
        const int     rr   = __builtin_amdgcn_readfirstlane(int(row));
        const int     s    = prm.rowptr[rr], nb = prm.rowptr[rr + 1] - s;
        const int2    ri   = prm.rowinfo[rr];
        const int64_t base = ri.x;
        const int     wx   = prm.rowwx[rr];
        double        sacc = 0.0;
        if (base >= 0 && kb < BPW)
          for (int k = kb; k < nb; k += BPW)
            sacc += prm.vals[(base + int64_t(k / wx) * ri.y + k % wx) * DD + e] * prm.x[int64_t(prm.col[s + k]) * D + j];
        // sum over the lanes with the same i: entries (kb, i, *)
        double tot[D];
#pragma unroll
        for (int ii = 0; ii < D; ++ii)
          tot[ii] = wave_sum((kb < BPW && i == ii) ? sacc : 0.0);
        if (lane < D && base >= 0)
          {
            double v = tot[0];
#pragma unroll
            for (int ii = 1; ii < D; ++ii)
              v = lane == ii ? tot[ii] : v;
            prm.y[row * D + lane] = v;
            if (prm.dotv)
              dsum += v * prm.dotv[row * D + lane];
          }
      }
    if (prm.partials)
      {
        const double tot = block_sum<256>(dsum, s_red);
        if (threadIdx.x == 0)
          prm.partials[b] = tot;
      }
  }

  // ------------------------------------------------------------------ sliced-ELL SpMV (production variant)
  // The block pattern of a box mesh has only a handful of distinct row shapes, so rows are grouped by (length, x-width
  // of the column box) into slices of 64 rows without padding inside a slice, lane = row: no cross-lane reduction,
  // neighbouring rows gather neighbouring x.  Since round 3 the values are stored the way the element scatter writes
  // them, ONE array for assembly and product (rounds 1-2 copied the block-CSR tangent into a [(off+k)*DD + e][64]
  // layout before every solve: 3.3 ms and 15 GB of traffic per tangent at 5 M dofs): x-line-interleaved block rows
  // (mi_mesh.hpp) -- a block is DD contiguous doubles, the wx blocks of one x-line of a row's column box stay
  // together (the runs of 3 blocks = 216 bytes a cell writes), and for one x-line g the 64 rows of a slice form ONE
  // contiguous chunk of 64*wx*DD doubles (13.5 / 22.5 KiB for 3D Q2).  A wave brings a chunk into its own LDS buffer by
  // LDS-DMA (global_load_lds_dwordx4: 1 KiB per wave instruction, fully coalesced, no VGPRs, non-temporal: the matrix
  // is streamed once, the L2 is kept for the gathered x) and reads its row's wx blocks back with ds_read_b64 at a
  // stride of wx*DD doubles (216 / 360 bytes: conflict-free, the bank is (a/4) mod 64 and 54 l resp. 90 l mod 64 takes
  // 32 different even values over a 32-lane group).  The columns of an x-line are consecutive nodes, so a lane's x
  // values are one contiguous run of wx*D doubles.  Measured with tools/probe/sellb_probe.hip at 5 M dofs, same
  // process: 1.255 ms (6.15 TB/s) against 1.358 ms for the transposed layout of round 2 and 1.32 ms for the
  // slot-interleaved order [(off+k)*64 + lane][DD] (whose scatter, 72-byte blocks 4.6 kB apart, was 40 % slower).
  // Chunks beyond the buffer (3D Q3 / Q4) are staged in two halves of 32 rows.
  // One wavefront per slice; SELL_WPB waves per workgroup (their buffers: 3 x 22.5 KiB, two workgroups per CU).
  // DOT: the CG's q = K p with the fused partials of p.q -- a separate instantiation so that profilers list the
  //      product the roofline figure is quoted on apart from the preconditioner's products
  // F32: the matrix values come from the fp32-rounded copy (smoother only, opt-in); all arithmetic stays fp64
  // CHEB: Chebyshev-Jacobi update fused into the epilogue (see SellParams), y is not written
  // ICOL: column indices generated from the row's column box (SellParams::rowbox) instead of read from memory
  // NTL: matrix values loaded with the non-temporal hint
  // what a row does with its finished product (acc = (K x)_row): plain store, fused dot partial, or the fused
  // Chebyshev-Jacobi step / residual of the multigrid smoother (see SellParams)
  template <int D, bool DOT, bool CHEB>
  __device__ __forceinline__ void sell_row_epilogue(const SellParams &prm, const int node, const double *acc, double &dsum)
  {
    constexpr int DD = D * D;
    if (node < 0)
      return;
    if constexpr (CHEB)
      {
        // every operand first, then the arithmetic, then the stores: interleaved with the stores (d, x' may alias the
        // operands as far as the compiler knows) the loads of component i+1 waited for the stores of component i --
        // three memory round trips instead of one on levels where a launch is nothing but round trips
        double res[D], dinv[DD], dold[D], xold[D];
#pragma unroll
        for (int i = 0; i < D; ++i)
          res[i] = prm.cheb_b[int64_t(node) * D + i];
        if (prm.cheb_d)
          {
            if (prm.cheb_blk) // block-Jacobi: D^-1 is a DxD block per node
              {
#pragma unroll
                for (int k = 0; k < DD; ++k)
                  dinv[k] = prm.cheb_dinv[int64_t(node) * DD + k];
              }
            else
              {
#pragma unroll
                for (int i = 0; i < D; ++i)
                  dinv[i] = prm.cheb_dinv[int64_t(node) * D + i];
              }
            // c1 == 0 on the first step: the old d is not read (it may hold anything, e.g. the NaNs of a solve that
            // broke down)
#pragma unroll
            for (int i = 0; i < D; ++i)
              {
                dold[i] = prm.cheb_c1 != 0.0 ? prm.cheb_d[int64_t(node) * D + i] : 0.0;
                xold[i] = prm.x[int64_t(node) * D + i];
              }
          }
#pragma unroll
        for (int i = 0; i < D; ++i)
          res[i] -= acc[i];
        double out0[D], out1[D];
#pragma unroll
        for (int i = 0; i < D; ++i)
          {
            if (prm.cheb_d)
              {
                double s = dinv[i] * res[i];
                if (prm.cheb_blk)
                  {
                    s = 0.0;
#pragma unroll
                    for (int j = 0; j < D; ++j)
                      s += dinv[i * D + j] * res[j];
                  }
                const double dn = (prm.cheb_c1 != 0.0 ? prm.cheb_c1 * dold[i] : 0.0) + prm.cheb_c2 * s;
                out0[i]         = dn;
                out1[i]         = xold[i] + dn;
              }
            else
              out0[i] = res[i]; // residual mode: y = b - K x
          }
#pragma unroll
        for (int i = 0; i < D; ++i)
          {
            const int64_t idx = int64_t(node) * D + i;
            if (prm.cheb_d)
              {
                prm.cheb_d[idx]    = out0[i];
                prm.cheb_xout[idx] = out1[i];
              }
            else
              prm.y[idx] = out0[i];
          }
      }
    else
      {
#pragma unroll
        for (int i = 0; i < D; ++i)
          prm.y[int64_t(node) * D + i] = acc[i];
      }
    if (DOT && node >= prm.own_begin && node < prm.own_end)
#pragma unroll
      for (int i = 0; i < D; ++i)
        dsum += acc[i] * prm.dotv[int64_t(node) * D + i];
  }

  // ------------------------------------------------------------------ the same product for SMALL launches (k-split)
  // A launch with a few dozen slices (coarse multigrid levels, the boundary rows of a slab, mid-size problems) cannot
  // fill the chip with one wavefront per slice, and that wavefront walks through its 9-25 x-lines one memory round trip
  // at a time: 14-21 us per product whatever the level (round 2 trace).  Here one WORKGROUP of SELL_SPLIT_W wavefronts
  // owns a slice: wave w takes the x-lines g = w, w + W, ... (every lane loads its own wx blocks straight from memory --
  // the matrix of such a level sits in the L2 / Infinity Cache), the partial sums of the waves meet in LDS and wave 0
  // adds them in wave order (deterministic) and runs the row epilogue.  The summation order differs from sell_spmv's
  // (x-lines interleaved over the waves), so which of the two kernels a launch takes depends on the launch's slice count
  // alone (SELL_SPLIT_MAX_SLICES), never on timing.
  constexpr int SELL_SPLIT_W = 8;
  template <int D, bool DOT, bool F32, bool CHEB, bool ICOL>
  __global__ __launch_bounds__(SELL_SPLIT_W * 64) void sell_spmv_split(SellParams prm)
  {
    if (prm.done && *prm.done)
      return;
    constexpr int DD = D * D, W = SELL_SPLIT_W;
    using VT         = typename std::conditional<F32, float, double>::type;
    __shared__ double s_part[W][D][64];
    __shared__ double s_red[W];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int sl   = prm.slice0 + blockIdx.x;
    double    dsum = 0.0;
    const int     wx  = prm.wx[sl];
    const int     len = prm.len[sl], ng = len / wx;
    const int64_t off = prm.off[sl];
    const int     node = prm.perm[int64_t(sl) * 64 + lane];
    const VT *__restrict__ vbase =
      (F32 ? reinterpret_cast<const VT *>(prm.vals32) : reinterpret_cast<const VT *>(prm.vals)) + (off * 64 + int64_t(lane) * wx) * DD;
    const int32_t *__restrict__ cp = prm.col + off * 64 + lane;
    int32_t b0 = 0, wy = 1;
    if constexpr (ICOL)
      {
        b0 = prm.rowbox[(int64_t(sl) * 64 + lane) * 2];
        wy = (prm.rowbox[(int64_t(sl) * 64 + lane) * 2 + 1] >> 8) & 255;
      }
    double acc[D];
#pragma unroll
    for (int i = 0; i < D; ++i)
      acc[i] = 0.0;
    for (int g = wave; g < ng; g += W)
      {
        const VT *__restrict__ vp = vbase + int64_t(g) * (64 * wx) * DD;
        const int32_t c0 = ICOL ? b0 + (g % wy) * prm.nn0 + (g / wy) * prm.nn0 * prm.nn1 : 0;
        for (int kx = 0; kx < wx; ++kx)
          {
            const int32_t c = ICOL ? c0 + kx : cp[int64_t(g * wx + kx) * 64];
            double        v[DD], xx[D];
#pragma unroll
            for (int e = 0; e < DD; ++e)
              v[e] = double(vp[kx * DD + e]);
#pragma unroll
            for (int j = 0; j < D; ++j)
              xx[j] = prm.x[int64_t(c) * D + j];
#pragma unroll
            for (int i = 0; i < D; ++i)
#pragma unroll
              for (int j = 0; j < D; ++j)
                acc[i] += v[i * D + j] * xx[j];
          }
      }
#pragma unroll
    for (int i = 0; i < D; ++i)
      s_part[wave][i][lane] = acc[i];
    __syncthreads();
    if (wave == 0)
      {
#pragma unroll
        for (int i = 0; i < D; ++i)
          {
            double t = s_part[0][i][lane];
            for (int w = 1; w < W; ++w)
              t += s_part[w][i][lane];
            acc[i] = t;
          }
        sell_row_epilogue<D, DOT, CHEB>(prm, node, acc, dsum);
      }
    if (DOT)
      {
        const double tot = block_sum<W * 64>(dsum, s_red);
        if (threadIdx.x == 0)
          prm.partials[prm.part0 + blockIdx.x] = tot;
      }
  }

  constexpr int SELL_STAGE_BYTES = 23040; // 64 rows x 5 blocks x 72 bytes: the largest chunk of 3D Q2
  template <int D, int WXT, bool NTL, bool DOT, bool F32, bool CHEB, bool ICOL>
  __device__ __forceinline__ void sell_slice(const SellParams &prm, const int sl, const int wx_rt, const int lane, char *stage,
                                             double &dsum)
  {
    constexpr int DD = D * D, VB = F32 ? 4 : 8;
    using VT         = typename std::conditional<F32, float, double>::type;
    typedef const volatile __attribute__((address_space(3))) VT *lds_cvp;
    typedef __attribute__((address_space(3))) void              *lds_vp;
    typedef const __attribute__((address_space(1))) void        *glb_vp;
    const int     wx    = WXT ? WXT : wx_rt;
    const int     len   = prm.len[sl], ng = len / wx;
    const int64_t off   = prm.off[sl];
    const int     node  = prm.perm[int64_t(sl) * 64 + lane];
    const int     chunk = 64 * wx * DD * VB;                    // bytes of one x-line of the slice
    const int     ns    = chunk > SELL_STAGE_BYTES ? 2 : 1;     // staged whole, or in two halves of 32 rows
    const int     rows  = 64 / ns, seg = chunk / ns, nfull = seg / 1024, rem = (seg % 1024) / 16;
    const char *__restrict__ vbytes =
      (F32 ? reinterpret_cast<const char *>(prm.vals32) : reinterpret_cast<const char *>(prm.vals)) + off * (64 * DD * VB) + lane * 16;
    const int32_t *__restrict__ cp = prm.col + off * 64 + lane;
    const lds_cvp rd = (lds_cvp)(reinterpret_cast<VT *>(stage) + (lane % rows) * (wx * DD));
    double        acc[D];
#pragma unroll
    for (int i = 0; i < D; ++i)
      acc[i] = 0.0;
    // first column of the current x-line of the row's column box; x-lines advance along y, then z
    int32_t c0 = 0, gy = 0, wy = 1;
    if constexpr (ICOL)
      {
        c0 = prm.rowbox[(int64_t(sl) * 64 + lane) * 2];
        wy = (prm.rowbox[(int64_t(sl) * 64 + lane) * 2 + 1] >> 8) & 255;
      }
    constexpr int WXM = WXT ? WXT : 9; // widest x-line (3D / 2D Q4)
    for (int g = 0; g < ng; ++g)
      {
        // x of the wx columns of this x-line
        double xx[WXM * D];
        if constexpr (ICOL)
          {
#pragma unroll
            for (int j = 0; j < WXM * D; ++j)
              if (WXT || j < wx * D)
                xx[j] = prm.x[int64_t(c0) * D + j];
            c0 += prm.nn0;
            if (++gy == wy)
              {
                gy = 0;
                c0 += prm.nn0 * (prm.nn1 - wy);
              }
          }
        else
          {
#pragma unroll
            for (int kx = 0; kx < WXM; ++kx)
              if (WXT || kx < wx)
                {
                  const int32_t c = cp[int64_t(g * wx + kx) * 64];
#pragma unroll
                  for (int j = 0; j < D; ++j)
                    xx[kx * D + j] = prm.x[int64_t(c) * D + j];
                }
          }
        for (int s = 0; s < ns; ++s)
          {
            const char *src = vbytes + int64_t(g) * chunk + s * seg;
            if constexpr (WXT != 0)
              {
#pragma unroll
                for (int j = 0; j < (64 * WXT * DD * VB) / 1024; ++j)
                  __builtin_amdgcn_global_load_lds((glb_vp)(src + j * 1024), (lds_vp)(stage + j * 1024), 16, 0, NTL ? 2 : 0);
              }
            else
              for (int j = 0; j < nfull; ++j)
                __builtin_amdgcn_global_load_lds((glb_vp)(src + j * 1024), (lds_vp)(stage + j * 1024), 16, 0, NTL ? 2 : 0);
            if (lane < rem)
              __builtin_amdgcn_global_load_lds((glb_vp)(src + nfull * 1024), (lds_vp)(stage + nfull * 1024), 16, 0, NTL ? 2 : 0);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); // the chunk has landed (and x has arrived)
            MI_WAVE_SYNC();
            if (ns == 1 || lane / rows == s)
              {
#pragma unroll
                for (int kx = 0; kx < WXM; ++kx)
                  if (WXT || kx < wx)
                    {
                      double v[DD];
#pragma unroll
                      for (int e = 0; e < DD; ++e)
                        v[e] = double(rd[kx * DD + e]);
#pragma unroll
                      for (int i = 0; i < D; ++i)
#pragma unroll
                        for (int j = 0; j < D; ++j)
                          acc[i] += v[i * D + j] * xx[kx * D + j];
                    }
              }
            MI_WAVE_SYNC(); // the buffer is free again
          }
      }
    sell_row_epilogue<D, DOT, CHEB>(prm, node, acc, dsum);
  }

  template <int D, bool NTL = true, bool DOT = false, bool F32 = false, bool CHEB = false, bool ICOL = false>
  __global__ __launch_bounds__(SELL_WPB * 64) void sell_spmv(SellParams prm)
  {
    if (prm.done && *prm.done)
      return;
    __shared__ double s_red[4];
    __shared__ __attribute__((aligned(16))) char s_stage[SELL_WPB][SELL_STAGE_BYTES];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int nwg = gridDim.x, b = blockIdx.x;
    const int lb  = (prm.xcd_remap && nwg % 8 == 0) ? (b % 8) * (nwg / 8) + b / 8 : b;
    const int per = (prm.nslices + nwg - 1) / nwg;
    const int s0 = prm.slice0 + lb * per, s1 = min(prm.slice0 + prm.nslices, s0 + per);
    double    dsum = 0.0;
    for (int sl0 = s0 + wave; sl0 < s1; sl0 += SELL_WPB)
      {
        const int sl = __builtin_amdgcn_readfirstlane(sl0);
        const int wx = prm.wx[sl];
        if (D == 3 && wx == 3) // the two x-widths of 3D Q2 rows, unrolled
          sell_slice<D, (D == 3 ? 3 : 0), NTL, DOT, F32, CHEB, ICOL>(prm, sl, wx, lane, s_stage[wave], dsum);
        else if (D == 3 && wx == 5)
          sell_slice<D, (D == 3 ? 5 : 0), NTL, DOT, F32, CHEB, ICOL>(prm, sl, wx, lane, s_stage[wave], dsum);
        else
          sell_slice<D, 0, NTL, DOT, F32, CHEB, ICOL>(prm, sl, wx, lane, s_stage[wave], dsum);
      }
    if (DOT)
      {
        const double tot = block_sum<SELL_WPB * 64>(dsum, s_red);
        if (threadIdx.x == 0)
          prm.partials[prm.part0 + b] = tot;
      }
  }

  // ------------------------------------------------------------------ product with unassembled element tangents
  // y += sum over the cells of one colour of P_e^T K_e P_e x, with K_e stored as its 378 lower-triangle node-pair
  // blocks (3D Q2), layout [cell][e][block]: a wave reads 512 contiguous bytes per load, every stored number is read
  // once and used for both K_ab x_b and K_ab^T x_a.  The symmetric half of the element tangents is 5.59 GB at 5 M dofs
  // against the 7.62 GB of the assembled matrix, which is why the multigrid smoother (86 % of the fine-level products of
  // a time step) multiplies with this form; the CG's own product stays on the assembled matrix.
  // One workgroup per cell: thread = block (a >= b); the 6 partial results per block go through LDS and are summed per
  // local dof in a fixed order (deterministic); cells of one colour share no node, so the update of y is race free; the
  // first cell (in processing order) that contains a node stores, later ones add, so y needs no zero fill.
  // Constrained rows/columns were masked when the blocks were stored (exactly the values that entered the global matrix).
  __global__ __launch_bounds__(384) void ebe_spmv(EbeParams prm, int64_t cell0)
  {
    constexpr int NPC = 27, NBLK = EBE_NBLK;
    __shared__ double s_x[NPC * 3];
    __shared__ double s_p[NBLK * 6 + 6];
    __shared__ int    s_conn[NPC];
    const int     tid  = threadIdx.x;
    const int64_t cell = cell0 + blockIdx.x;
    const double *__restrict__ kp = prm.ke + cell * (9 * int64_t(NBLK)) + tid;
    const bool act = tid < NBLK;
    double     k[9];
    if (act)
      {
#pragma unroll
        for (int e = 0; e < 9; ++e)
          k[e] = __builtin_nontemporal_load(&kp[e * NBLK]); // streamed once
      }
    if (tid < NPC)
      s_conn[tid] = prm.conn[cell * NPC + tid];
    __syncthreads();
    if (tid < NPC * 3)
      s_x[tid] = prm.x[int64_t(s_conn[tid / 3]) * 3 + tid % 3];
    __syncthreads();
    if (act)
      {
        int a = int((sqrtf(8.0f * float(tid) + 1.0f) - 1.0f) * 0.5f); // block index -> (a, b), a >= b
        while ((a + 1) * (a + 2) / 2 <= tid)
          ++a;
        while (a * (a + 1) / 2 > tid)
          --a;
        const int    b = tid - a * (a + 1) / 2;
        const double xa0 = s_x[a * 3], xa1 = s_x[a * 3 + 1], xa2 = s_x[a * 3 + 2];
        const double xb0 = s_x[b * 3], xb1 = s_x[b * 3 + 1], xb2 = s_x[b * 3 + 2];
        double      *p = &s_p[tid * 6];
        p[0]           = k[0] * xb0 + k[1] * xb1 + k[2] * xb2; // K_ab x_b
        p[1]           = k[3] * xb0 + k[4] * xb1 + k[5] * xb2;
        p[2]           = k[6] * xb0 + k[7] * xb1 + k[8] * xb2;
        const double s = (a == b) ? 0.0 : 1.0;                 // the diagonal block counts once
        p[3]           = s * (k[0] * xa0 + k[3] * xa1 + k[6] * xa2); // K_ab^T x_a
        p[4]           = s * (k[1] * xa0 + k[4] * xa1 + k[7] * xa2);
        p[5]           = s * (k[2] * xa0 + k[5] * xa1 + k[8] * xa2);
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
        double *yp = &prm.y[int64_t(s_conn[a]) * 3 + i];
        *yp        = ((prm.first[cell] >> a) & 1u) ? s : *yp + s; // first touch of the node: store
      }
  }

  // ------------------------------------------------------------------ matrix-free product from quadrature-point records
  // y += sum over the cells of one colour of P_e^T K_e P_e x WITHOUT K_e: the assembly leaves, per cell and quadrature
  // point, the state its tangent is linearised at (F, J^(-2/3), 1/J: MF_NREC x 64 doubles = 5.6 kB per 3D Q2 cell against
  // 27.2 kB for the symmetric element tangent and 37.6 kB of assembled rows); the product recomputes the material
  // response from it with the assembly's own function (neo_hooke_from_F) and evaluates  y_a = sum_q Q(q) grad_xi N_a(q):
  //   M = Jinv Finv,  H = sum_b x_b (x) grad_xi N_b,  h = H M,
  //   S = (c_II tr h - 2/3 tau_iso:h) I - 2/3 tr h tau_iso + c_S/2 (h + h^T) + h tau,
  //   Q = JxW S M^T  (+ the mass term alpha_1 rho JxW N_a N_b),
  // which is K_e x_e term by term (assemble_cells' node-pair form summed over b; tools/proto/mf_product.py checks the
  // algebra against the independent mirror).  Gradients and the integration are contracted one lattice direction at a
  // time (3 nodes <-> 4 points); in every pass a lane owns one line and produces ALL outputs along the contracted
  // direction, so the 1D tables are scalar operands and every intermediate is written to LDS once.
  // One wavefront = one cell = its 64 quadrature points.  Update of y as in ebe_spmv (colours, first touch stores).
  // Geometry: BOX (every cell an axis-parallel box, the reference's grids) takes 1/h and the volume from the cell's
  // corner vertices; otherwise the Jacobian of the trilinear map is evaluated at the point as in the assembly.
  // Constrained dofs: x is masked on the way in; their rows receive diag(K) x from the assembled tangent at the first touch
  // (|K_e(i,i)| summed over the cells is what the assembly put there), nothing otherwise.
  // SLOTS: instead of updating y colour by colour, every cell stores its 81 results in its own slots of a
  // contribution array (slot = position of the cell among the cells of the node, in processing order) and ONE launch
  // covers all cells; mf_gather then sums the slots of every node in that order -- the same additions in the same
  // order as the colour-by-colour update, i.e. the same bits, in 2 launches instead of 8.
  // DBG (diagnostic instantiations, never in production): bit 0 stage stamps; timing-only ablations: bit 1 no result
  // stores, bit 2 every cell reads the records of cell 0 (cache hits), bit 3 every cell gathers the x of cell 0's nodes
  // T (round 5, opt-in "smoother_precision" 32): the scalar type of the arithmetic and of the records.  float: the records
  // come from prm.qrec32, x is converted on the way in, the results on the way out; vectors, slots and every other kernel
  // stay fp64 -- a preconditioner-only change (the production shape only).
  template <bool BOX, bool SLOTS, bool LAT, int DBG = 0, int OCC = 4, typename T = double>
  __global__ __launch_bounds__(64, OCC) void mf_spmv(MfParams prm, int64_t cell0) // OCC = 5 (96 VGPRs) spills 8 registers: A/B MI_MF_OCC=5
  {
    static_assert(std::is_same<T, double>::value || (BOX && SLOTS && LAT && DBG == 0), "fp32 form: production shape only");
    constexpr bool STAMP = (DBG & 1) != 0;
    const T        mu_t = T(prm.mu), kappa_t = T(prm.kappa), mass_t = T(prm.mass);
    constexpr int NPC = 27;
    // LDS (per cell, 7.9 kB): s0 = x (81 at AO) and the (i,j)-contracted
    // planes B (3 x 9 x 20 at 0; plane stride 20 and the lane order (c*3+k)*4 + qx keep their stores conflict free), then
    // the point results Q (12 x 64; the 16-lane groups of component 1 are swapped so that components 0 and 1, one
    // half-wave in I3, read different banks), then the qz-contracted planes C IN PLACE of the Q entries their lane
    // consumed; sE = the qy-contracted lines E (2 x 108).
    constexpr int PS = 20, PW = 9 * PS, AO = 552;
    __shared__ T s0[768];
    __shared__ T sE[216];
    __shared__ int    s_conn[NPC], s_cm[NPC];
    // reads go through volatile pointers: single ds_read_b64 (2 LDS cycles per wave) instead of merged ds_read2_b64 (8)
    typedef const volatile __attribute__((address_space(3))) T *lds_cvp;
    const lds_cvp v0 = (lds_cvp)s0, vE = (lds_cvp)sE;
    const int     lane = threadIdx.x;
    // workgroups go round robin over the 8 XCDs: give each XCD a contiguous run of cells, so that the x / y lines shared
    // by neighbouring cells of the colour meet in ONE L2 (MI_MF_XCD=0: plain order, for A/B)
    int64_t cell = cell0 + blockIdx.x;
    if (prm.xcd_chunk > 0)
      {
        const int64_t local = int64_t(blockIdx.x & 7) * prm.xcd_chunk + (blockIdx.x >> 3);
        if (local >= prm.count)
          return;
        cell = cell0 + local;
      }
    if constexpr (LAT)
      if (prm.sel_n > 0) // a launch over some layers of a slab: `cell` counts the selected cells, colour by colour
        {
          const int32_t l    = int32_t(cell - cell0);
          int32_t       b    = prm.sel_begin[0], p0 = prm.sel_pos0[0];
#pragma unroll
          for (int c = 1; c < 8; ++c)
            if (l >= prm.sel_begin[c])
              {
                b  = prm.sel_begin[c];
                p0 = prm.sel_pos0[c];
              }
          cell = int64_t(p0) + (l - b);
        }
    // diagnostic instantiation: shader-clock stamps at the stage boundaries, kept in LDS until the end (a global store
    // in the middle of the kernel makes the compiler give up the scalar registers of the 1D tables)
    __shared__ unsigned long long s_st[STAMP ? 8 : 1];
#define MF_STAMP(i_)                                                                     \
  do                                                                                     \
    {                                                                                    \
      if constexpr (STAMP)                                                               \
        if (lane == 0)                                                                   \
          s_st[(i_)] = __builtin_amdgcn_s_memtime();                                     \
    }                                                                                    \
  while (0)
    // LAT: the cell's nodes by arithmetic (mi::CellLattice) -- the gather of x is then the FIRST memory access of the
    // wave and the records follow it (the memory counter retires loads in order: what is needed first is asked first);
    // otherwise the records go first and the gather waits for the connectivity
    int32_t node_l = 0, cm_l = 0; // lane < 27: this lane's node and its constraint bits
    T  xg[3]  = {0.0, 0.0, 0.0};
    int32_t node0  = 0;
    if constexpr (LAT)
      {
        node0 = lattice_node0(prm.lat, (DBG & 8) ? int64_t(0) : cell);
        if (lane < NPC)
          {
            const int k9 = lane / 9, r9 = lane - 9 * k9, j3 = r9 / 3, i3 = r9 - 3 * j3;
            node_l       = node0 + i3 + j3 * prm.lat.nn0 + k9 * prm.lat.nn01;
#pragma unroll
            for (int c = 0; c < 3; ++c)
              xg[c] = prm.x[int64_t(node_l) * 3 + c];
            cm_l = prm.cmask[node_l];
          }
      }
    // the cell's records: consumed after the gradient passes
    T rec[MF_NREC];
    {
      const T *__restrict__ rp = (std::is_same<T, float>::value ? reinterpret_cast<const T *>(prm.qrec32) : reinterpret_cast<const T *>(prm.qrec)) +
                                 ((DBG & 4) ? int64_t(0) : cell) * int64_t(MF_NREC * 64) + lane;
#pragma unroll
      for (int f = 0; f < MF_NREC; ++f)
        rec[f] = __builtin_nontemporal_load(&rp[f * 64]);
    }
    // 1D tables: S[q][a] = N_a(x_q), D[q][a] = N_a'(x_q) (uniform -> scalar registers)
    T S[4][3], D[4][3];
#pragma unroll
    for (int q = 0; q < 4; ++q)
#pragma unroll
      for (int a = 0; a < 3; ++a)
        {
          S[q][a] = prm.tab1d[q * 3 + a];
          D[q][a] = prm.tab1d[12 + q * 3 + a];
        }
    const int qz = lane >> 4, q16 = lane & 15;
    T    Sz[3], Dz[3]; // this lane's rows of the tables for the last gradient pass
#pragma unroll
    for (int k = 0; k < 3; ++k)
      {
        Sz[k] = prm.tab1d[qz * 3 + k];
        Dz[k] = prm.tab1d[12 + qz * 3 + k];
      }
    T Sx[3], Dx[3]; // rows qx = lane & 3 of the tables for the fused first gradient stage (lane = (c*3+k)*4 + qx)
#pragma unroll
    for (int i = 0; i < 3; ++i)
      {
        Sx[i] = prm.tab1d[(lane & 3) * 3 + i];
        Dx[i] = prm.tab1d[12 + (lane & 3) * 3 + i];
      }
    // this lane's quadrature weight and (general geometry) unit-cell point: tab1d holds qw[4] at 24 and qx[4] at 28
    const T wq = prm.tab1d[24 + (lane & 3)] * prm.tab1d[24 + ((lane >> 2) & 3)] * prm.tab1d[24 + qz];
    T       xiq[3];
    if constexpr (!BOX)
      {
        xiq[0] = prm.tab1d[28 + (lane & 3)];
        xiq[1] = prm.tab1d[28 + ((lane >> 2) & 3)];
        xiq[2] = prm.tab1d[28 + qz];
      }
    // the cell's geometry (BOX): 1/hx, 1/hy, 1/hz, hx hy hz -- uniform, scalar loads
    T cbox[4] = {0.0, 0.0, 0.0, 0.0};
    if constexpr (BOX)
      {
        const double *__restrict__ cb = prm.cellbox + cell * 4;
#pragma unroll
        for (int k = 0; k < 4; ++k)
          cbox[k] = cb[k];
      }
    MF_STAMP(0); // (after every uniform load: the clock read is a scalar-memory instruction with side effects, and loads
                 // behind it would no longer be scalar)
    // ---- gather x (constrained entries masked): X[c][a] at c*27 + a, a = (k*3 + j)*3 + i
    if (lane < NPC)
      {
        if constexpr (!LAT)
          {
            node_l = prm.conn[cell * NPC + lane];
            cm_l   = prm.cmask[node_l];
#pragma unroll
            for (int c = 0; c < 3; ++c)
              xg[c] = prm.x[int64_t(node_l) * 3 + c];
          }
        s_conn[lane] = node_l;
        s_cm[lane]   = cm_l;
#pragma unroll
        for (int c = 0; c < 3; ++c)
          s0[AO + c * NPC + lane] = ((cm_l >> c) & 1) ? T(0.0) : xg[c]; // (behind the planes B: they can land while x is read)
      }
    // the entries of y this lane will update at the very end (lane = line (c,k,j), its three nodes i): read now, the
    // colouring keeps every other cell of this launch away from them
    const int lc = lane / 9, lkj = lane - 9 * lc;
    double yold[3];
    int32_t   ydst[3];
    if (lane < 27)
      {
#pragma unroll
        for (int i = 0; i < 3; ++i)
          if constexpr (SLOTS)
            ydst[i] = prm.dst[cell * NPC + lkj * 3 + i];
          else if constexpr (LAT)
            yold[i] = prm.y[int64_t(node0 + i + (lkj % 3) * prm.lat.nn0 + (lkj / 3) * prm.lat.nn01) * 3 + lc];
          else
            yold[i] = prm.y[int64_t(prm.conn[cell * NPC + lkj * 3 + i]) * 3 + lc];
      }
    __syncthreads();
    MF_STAMP(1); // x has arrived
    // ---- E1 + E2 in one stage (round 4): contract i, then j.  lane = (c*3+k)*4 + qx reads the nine x values of its
    // plane (c,k) -- 9 LDS reads, as many as the two stages had between them --, forms A_S / A_D [qx][c,k,j] for ITS qx in
    // registers (no lane computes one twice: 36 lanes x 6 = the 216 values the old first stage wrote to LDS and the second
    // read back) and contracts j.  One stage and eight LDS stores fewer per cell; the 1D tables of the i-contraction
    // are per-lane operands here (row qx of S and D).  x sits behind the planes (at AO, where the first stage's lines
    // used to go), so they can be stored while other lanes still read.  B_DS / B_SD / B_SS [c*3+k][qy][qx] at {0,PW,2PW}
    const int pck = lane >> 2, pqx = lane & 3; // plane index c*3+k and qx of this lane in E12 / I2
    if (lane < 36)
      {
        T bds[4], bsd[4], bss[4];
        T as[3], ad[3];
#pragma unroll
        for (int j = 0; j < 3; ++j)
          {
            const T x0 = v0[AO + pck * 9 + j * 3], x1 = v0[AO + pck * 9 + j * 3 + 1], x2 = v0[AO + pck * 9 + j * 3 + 2];
            as[j] = Sx[0] * x0 + Sx[1] * x1 + Sx[2] * x2;
            ad[j] = Dx[0] * x0 + Dx[1] * x1 + Dx[2] * x2;
          }
#pragma unroll
        for (int qy = 0; qy < 4; ++qy)
          {
            bds[qy] = S[qy][0] * ad[0] + S[qy][1] * ad[1] + S[qy][2] * ad[2]; // d/dx
            bsd[qy] = D[qy][0] * as[0] + D[qy][1] * as[1] + D[qy][2] * as[2]; // d/dy
            bss[qy] = S[qy][0] * as[0] + S[qy][1] * as[1] + S[qy][2] * as[2]; // value / d/dz
          }
#pragma unroll
        for (int qy = 0; qy < 4; ++qy)
          {
            const int o    = pck * PS + qy * 4 + pqx;
            s0[o]          = bds[qy];
            s0[o + PW]     = bsd[qy];
            s0[o + 2 * PW] = bss[qy];
          }
      }
    __syncthreads();
    // ---- E3: contract k.  lane = quadrature point; H[c][l] = d x_c / d xi_l, V[c] = x_c
    T H[3][3], V[3];
#pragma unroll
    for (int c = 0; c < 3; ++c)
      {
        H[c][0] = H[c][1] = H[c][2] = V[c] = 0.0;
#pragma unroll
        for (int k = 0; k < 3; ++k)
          {
            const int    o   = (c * 3 + k) * PS + q16;
            const T bds = v0[o], bsd = v0[o + PW], bss = v0[o + 2 * PW];
            H[c][0]          = fma(Sz[k], bds, H[c][0]);
            H[c][1]          = fma(Sz[k], bsd, H[c][1]);
            H[c][2]          = fma(Dz[k], bss, H[c][2]);
            V[c]             = fma(Sz[k], bss, V[c]);
          }
      }
    __syncthreads(); // B is consumed: the point results go on top of it
    MF_STAMP(2); // gradients at the points
    if constexpr (BOX) // the mass term first: it needs only the cell's volume, and V dies before the tensor algebra
      {
        const T wm = mass_t * cbox[3] * wq;
#pragma unroll
        for (int i = 0; i < 3; ++i)
          s0[(i * 4 + 3) * 64 + (lane ^ ((i & 1) << 4))] = wm * V[i];
      }
    // ---- quadrature point: Q = JxW S M^T
    {
      T M[9], tau[6], w, wcII, cs2;
      {
        const T *F = rec;
        T        Finv[9], tiso[6], cII, cS, Ji[9], detJ;
        neo_hooke_from_F<3>(F, det3x3(F), rec[9], rec[10], mu_t, kappa_t, Finv, tau, tiso, cII, cS);
        if constexpr (BOX)
          {
            const T rx = cbox[0], ry = cbox[1], rz = cbox[2];
            detJ            = cbox[3];
#pragma unroll
            for (int k = 0; k < 3; ++k)
              {
                M[k]     = rx * Finv[k];
                M[3 + k] = ry * Finv[3 + k];
                M[6 + k] = rz * Finv[6 + k];
              }
          }
        else
          {
            const double *__restrict__ cv = prm.cverts + cell * 24; // uniform: scalar loads
            T verts[24], Jm[9];
#pragma unroll
            for (int k = 0; k < 24; ++k)
              verts[k] = cv[k];
            q1_jacobian<3>(verts, xiq, Jm);
            detJ = det3x3(Jm);
            inv3x3(Jm, detJ, Ji);
#pragma unroll
            for (int i = 0; i < 3; ++i)
#pragma unroll
              for (int j = 0; j < 3; ++j)
                M[i * 3 + j] = Ji[i * 3 + 0] * Finv[0 * 3 + j] + Ji[i * 3 + 1] * Finv[1 * 3 + j] + Ji[i * 3 + 2] * Finv[2 * 3 + j];
          }
        w    = detJ * wq;
        wcII = w * cII;
        cs2  = T(0.5) * cS;
      }
      T        h[3][3];
#pragma unroll
      for (int j = 0; j < 3; ++j)
#pragma unroll
        for (int k = 0; k < 3; ++k)
          h[j][k] = H[j][0] * M[k] + H[j][1] * M[3 + k] + H[j][2] * M[6 + k];
      const T pv  = (tau[0] + tau[1] + tau[2]) * T(1.0 / 3.0); // tau_iso = dev tau
      const T ti0 = tau[0] - pv, ti1 = tau[1] - pv, ti2 = tau[2] - pv;
      const T trh = h[0][0] + h[1][1] + h[2][2];
      const T th  = ti0 * h[0][0] + ti1 * h[1][1] + ti2 * h[2][2] + tau[3] * (h[0][1] + h[1][0]) +
                        tau[4] * (h[0][2] + h[2][0]) + tau[5] * (h[1][2] + h[2][1]);
      const T aI = wcII * trh - T(2.0 / 3.0) * w * th;
      const T m3 = T(-(2.0 / 3.0)) * trh;
      const T Tt[3][3] = {{tau[0], tau[3], tau[4]}, {tau[3], tau[1], tau[5]}, {tau[4], tau[5], tau[2]}};
      const T Ti[3][3] = {{ti0, tau[3], tau[4]}, {tau[3], ti1, tau[5]}, {tau[4], tau[5], ti2}};
      T       Sm[3][3];
#pragma unroll
      for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int j = 0; j < 3; ++j)
          {
            T v = m3 * Ti[i][j] + cs2 * (h[i][j] + h[j][i]) + h[i][0] * Tt[0][j] + h[i][1] * Tt[1][j] + h[i][2] * Tt[2][j];
            Sm[i][j] = w * v + (i == j ? aI : T(0.0));
          }
      const T wm = mass_t * w;
#pragma unroll
      for (int i = 0; i < 3; ++i)
        {
          const int ql = lane ^ ((i & 1) << 4);
#pragma unroll
          for (int l = 0; l < 3; ++l)
            s0[(i * 4 + l) * 64 + ql] = Sm[i][0] * M[l * 3] + Sm[i][1] * M[l * 3 + 1] + Sm[i][2] * M[l * 3 + 2];
          if constexpr (!BOX)
            s0[(i * 4 + 3) * 64 + ql] = wm * V[i];
        }
    }
    __syncthreads();
    MF_STAMP(3); // point stage (waits for the records)
    // ---- I3: contract qz.  lane = c*16 + (qy*4+qx); C_DS / C_SD / C_SS [c][k][qy][qx] replace Q[c][0 / 1 / 2][z = k][qy][qx],
    // entries only this lane has read
    if (lane < 48)
      {
        const int c = lane >> 4;
        T    v[4][4];
#pragma unroll
        for (int d = 0; d < 4; ++d)
#pragma unroll
          for (int z = 0; z < 4; ++z)
            v[d][z] = v0[(c * 4 + d) * 64 + ((z * 16 + q16) ^ ((c & 1) << 4))];
#pragma unroll
        for (int k = 0; k < 3; ++k)
          {
            T cds = 0.0, csd = 0.0, css = 0.0;
#pragma unroll
            for (int z = 0; z < 4; ++z)
              {
                cds = fma(S[z][k], v[0][z], cds);
                csd = fma(S[z][k], v[1][z], csd);
                css = fma(D[z][k], v[2][z], css);
                css = fma(S[z][k], v[3][z], css);
              }
            const int o = c * 256 + ((k * 16 + q16) ^ ((c & 1) << 4));
            s0[o]       = cds;
            s0[o + 64]  = csd;
            s0[o + 128] = css;
          }
      }
    __syncthreads();
    // ---- I2: contract qy.  lane = (c*3+k)*4 + qx; E_D / E_S [qx][c,k,j] in sE at {0,108} + qx*27 + (c*3+k)*3 + j
    if (lane < 36)
      {
        T cds[4], csd[4], css[4];
#pragma unroll
        for (int qy = 0; qy < 4; ++qy)
          {
            const int c = pck / 3, kk = pck - 3 * c;
            const int o = c * 256 + ((kk * 16 + qy * 4 + pqx) ^ ((c & 1) << 4));
            cds[qy]     = v0[o];
            csd[qy]     = v0[o + 64];
            css[qy]     = v0[o + 128];
          }
#pragma unroll
        for (int j = 0; j < 3; ++j)
          {
            T ed = 0.0, es = 0.0;
#pragma unroll
            for (int qy = 0; qy < 4; ++qy)
              {
                ed = fma(S[qy][j], cds[qy], ed);
                es = fma(D[qy][j], csd[qy], es);
                es = fma(S[qy][j], css[qy], es);
              }
            const int o  = pqx * 27 + pck * 3 + j;
            sE[o]       = ed;
            sE[108 + o] = es;
          }
      }
    __syncthreads();
    MF_STAMP(4); // I3 + I2
    // ---- I1: contract qx and update y.  lane = line (c,k,j)
    if (lane < 27)
      {
        T ed[4], es[4];
#pragma unroll
        for (int qx = 0; qx < 4; ++qx)
          {
            ed[qx] = vE[qx * 27 + lane];
            es[qx] = vE[108 + qx * 27 + lane];
          }
        const uint32_t fb = SLOTS ? 0u : prm.first[cell];
#pragma unroll
        for (int i = 0; i < 3; ++i)
          {
            T yv = 0.0;
#pragma unroll
            for (int qx = 0; qx < 4; ++qx)
              {
                yv = fma(D[qx][i], ed[qx], yv);
                yv = fma(S[qx][i], es[qx], yv);
              }
            if constexpr (SLOTS)
              {
                if (!(DBG & 2) || yv == T(-1.2345678e30)) // (DBG 2, timing only: the store stays in the code, never taken)
                  prm.yc[int64_t(ydst[i]) * 3 + lc] = yv;
                continue;
              }
            const int     a     = lkj * 3 + i;
            const bool    first = (fb >> a) & 1u;
            const int64_t yi    = int64_t(s_conn[a]) * 3 + lc;
            if ((s_cm[a] >> lc) & 1)
              {
                if (first)
                  {
                    const int32_t dp = prm.diagpos[s_conn[a]]; // -1: ghost node of a slab (its y is never read)
                    prm.y[yi]        = dp >= 0 ? prm.vals[int64_t(dp) * 9 + lc * 4] * prm.x[yi] : 0.0;
                  }
              }
            else
              prm.y[yi] = first ? yv : yold[i] + yv;
          }
      }
    MF_STAMP(5); // I1 and the stores issued
    if constexpr (STAMP)
      {
        __syncthreads();
        if (lane < 6)
          prm.stamps[cell * 8 + lane] = s_st[lane];
      }
#undef MF_STAMP
  }

  // Sum of the contributions of a node's cells to component c, in processing order.  Node-major slots: a contiguous run.
  // Cell-major slots (MfParams::slot_src): the positions come from an index -- all (up to eight: a vertex node of a 3D mesh)
  // index loads are issued together, then all value loads, then the additions in order: two dependent round trips per node
  // instead of two per contribution; the same additions in the same order.
  __device__ __forceinline__ double mf_slot_sum(const MfParams &prm, const int32_t b0, const int32_t b1, const int c)
  {
    if (prm.slot_src && b1 - b0 <= 8)
      {
        const int cnt = b1 - b0;
        int32_t   idx[8];
        double    v[8];
#pragma unroll
        for (int j = 0; j < 8; ++j)
          idx[j] = prm.slot_src[j < cnt ? b0 + j : b0];
#pragma unroll
        for (int j = 0; j < 8; ++j)
          v[j] = prm.yc[int64_t(idx[j]) * 3 + c];
        double s = v[0];
#pragma unroll
        for (int j = 1; j < 8; ++j)
          s = j < cnt ? s + v[j] : s;
        return s;
      }
    double s = prm.yc[int64_t(prm.slot_src ? prm.slot_src[b0] : b0) * 3 + c];
    for (int32_t k = b0 + 1; k < b1; ++k)
      s = s + prm.yc[int64_t(prm.slot_src ? prm.slot_src[k] : k) * 3 + c];
    return s;
  }

  // y = sum of the cells' contributions, node by node in slot order (= processing order of the cells: the order of the
  // colour-by-colour update); constrained rows: diag(K) x from the assembled tangent.  One thread per DOF.
  __global__ __launch_bounds__(256) void mf_gather(MfParams prm, int64_t ndofs)
  {
    const int64_t g = int64_t(blockIdx.x) * 256 + threadIdx.x;
    if (g >= ndofs)
      return;
    const int64_t n = g / 3;
    const int     c = int(g - n * 3);
    if ((prm.cmask[n] >> c) & 1)
      {
        const int32_t dp = prm.diagpos[n]; // -1: ghost node of a slab (its y is never read)
        prm.y[g]         = dp >= 0 ? prm.vals[int64_t(dp) * 9 + c * 4] * prm.x[g] : 0.0;
        return;
      }
    const int32_t b0 = prm.slot_base[n], b1 = prm.slot_base[n + 1];
    prm.y[g] = mf_slot_sum(prm, b0, b1, c);
  }

  // the same sum fused with its consumer in the multigrid smoother: q = (K x) from the slots, then the Chebyshev step
  // with the 3x3 block-Jacobi diagonal, d = c1 d + c2 D^-1 (b - q), x += d (in place: the product is complete), or,
  // with d == null, the residual y = b - q.  One thread per DOF of the nodes [node0, node0 + nnodes), 64 nodes per block.
  __global__ __launch_bounds__(192) void mf_gather_cheb(MfParams prm, const double *__restrict__ b,
                                                        const double *__restrict__ dinv, double *d, double *xio, double *yres,
                                                        double c1, double c2, int64_t node0, int64_t nnodes)
  {
    __shared__ double s_res[192];
    const int     ld = threadIdx.x;
    const int64_t nl = int64_t(blockIdx.x) * 64 + ld / 3;
    const bool    in = nl < nnodes;
    const int64_t n  = node0 + nl;
    const int     c  = ld % 3;
    const int64_t g  = n * 3 + c;
    double        res = 0.0;
    if (in)
      {
        double q;
        if ((prm.cmask[n] >> c) & 1)
          q = prm.vals[int64_t(prm.diagpos[n]) * 9 + c * 4] * prm.x[g];
        else
          {
            const int32_t b0 = prm.slot_base[n], b1 = prm.slot_base[n + 1];
            q                = mf_slot_sum(prm, b0, b1, c);
          }
        res = b[g] - q;
      }
    if (!d)
      {
        if (in)
          yres[g] = res;
        return;
      }
    s_res[ld] = res;
    __syncthreads();
    if (!in)
      return;
    const int    r0 = (ld / 3) * 3;
    const double a0 = dinv[g * 3], a1 = dinv[g * 3 + 1], a2 = dinv[g * 3 + 2];
    double       s  = 0.0;
    s += a0 * s_res[r0];
    s += a1 * s_res[r0 + 1];
    s += a2 * s_res[r0 + 2];
    const double dn = (c1 != 0.0 ? c1 * d[g] : 0.0) + c2 * s;
    d[g]            = dn;
    xio[g]          = xio[g] + dn;
  }

  // the same step in three-term form with the symmetric half of D^-1 (round 3): x'' = x' + c1 (x' - x) + c2 D^-1 (b - q)
  // reads x' (the product's operand), x (the iterate before; null: zero) and writes x'' over x -- two reads and a write
  // of 40 MB vectors instead of d and x read and written, 48 instead of 72 bytes of D^-1 per node: 380 instead of 460 MB
  // per step of the smoother at 5 M dofs.  The caller swaps the two x buffers.
  __global__ __launch_bounds__(192) void mf_gather_cheb3(MfParams prm, const double *__restrict__ b,
                                                         const double *__restrict__ dinv6, const double *xprev,
                                                         const double *__restrict__ xcur, double *xnext, double c1, double c2,
                                                         int64_t node0, int64_t nnodes)
  {
    __shared__ double s_res[192];
    const int     ld = threadIdx.x;
    const int64_t nl = int64_t(blockIdx.x) * 64 + ld / 3;
    const bool    in = nl < nnodes;
    const int64_t n  = node0 + nl;
    const int     c  = ld % 3;
    const int64_t g  = n * 3 + c;
    double        res = 0.0, xc = 0.0, xp = 0.0, a0 = 0.0, a1 = 0.0, a2 = 0.0;
    if (in)
      {
        // row c of the symmetric block (xx yy zz xy xz yz)
        const double *q6 = dinv6 + n * 6;
        a0               = q6[c == 0 ? 0 : (c == 1 ? 3 : 4)];
        a1               = q6[c == 0 ? 3 : (c == 1 ? 1 : 5)];
        a2               = q6[c == 0 ? 4 : (c == 1 ? 5 : 2)];
        xc               = xcur[g];
        xp               = (c1 != 0.0 && xprev) ? xprev[g] : 0.0;
        double q;
        if ((prm.cmask[n] >> c) & 1)
          q = prm.vals[int64_t(prm.diagpos[n]) * 9 + c * 4] * xc;
        else
          {
            const int32_t b0 = prm.slot_base[n], b1 = prm.slot_base[n + 1];
            q                = mf_slot_sum(prm, b0, b1, c);
          }
        res = b[g] - q;
      }
    s_res[ld] = res;
    __syncthreads();
    if (!in)
      return;
    const int r0 = (ld / 3) * 3;
    double    s  = 0.0;
    s += a0 * s_res[r0];
    s += a1 * s_res[r0 + 1];
    s += a2 * s_res[r0 + 2];
    const double dn = (c1 != 0.0 ? c1 * (xc - xp) : 0.0) + c2 * s;
    xnext[g]        = xc + dn;
  }

  // y = sum of the slots as mf_gather, with the partials of dotv . y over the owned dofs [own0, own0 + own_n) in the same
  // launch: the CG's q = K p and p.q on the matrix-free fine level (round 6).  Fixed grid (grid-stride over the dofs, as
  // dot_partials): the partials -- one per workgroup -- and their sum do not depend on the launch.
  __global__ __launch_bounds__(256) void mf_gather_dot(MfParams prm, int64_t ndofs, const double *__restrict__ dotv,
                                                       double *partials, int64_t own0, int64_t own_n)
  {
    __shared__ double s_red[4];
    double            acc = 0.0;
    const int64_t     per = ((ndofs + gridDim.x - 1) / gridDim.x + 255) / 256 * 256;
    const int64_t     g0 = int64_t(blockIdx.x) * per, g1 = imin64(ndofs, g0 + per);
    for (int64_t g = g0 + threadIdx.x; g < g1; g += 256)
      {
        const int64_t n = g / 3;
        const int     c = int(g - n * 3);
        double        s;
        if ((prm.cmask[n] >> c) & 1)
          {
            const int32_t dp = prm.diagpos[n];
            s                = dp >= 0 ? prm.vals[int64_t(dp) * 9 + c * 4] * prm.x[g] : 0.0;
          }
        else
          {
            const int32_t b0 = prm.slot_base[n], b1 = prm.slot_base[n + 1];
            s                = mf_slot_sum(prm, b0, b1, c);
          }
        prm.y[g] = s;
        if (g >= own0 && g < own0 + own_n)
          acc = fma(dotv[g], s, acc);
      }
    acc = block_sum<256>(acc, s_red);
    if (threadIdx.x == 0)
      partials[blockIdx.x] = acc;
  }

  // ------------------------------------------------------------------ matrix-free fine level: the diagonal blocks (round 6)
  // With the fine level matrix-free end to end (tuning "fine_level" 1) nothing multiplies the assembled fine tangent any
  // more; what the level still needs of it are the 3x3 DIAGONAL blocks K_aa of every node: the block-Jacobi smoother's
  // D, the Jacobi diagonal, and the diagonal entries of constrained dofs (what their rows of the operator hold).  They
  // come from the point records the residual pass has just written, by the node-pair formula of SURVEY section 10 with
  // a = b  [REF nonlinear_elasticity.cc:1011-1023]:
  //   g = M^T grad_xi N_a,  t = tau g,  v = (c_II + c_S/2 + 4/3 p)/2 g - 2/3 t   (p = kappa/2 (J^2-1): tau_iso g = t - p g)
  //   K_aa += w [ g (x) v + v (x) g + (c_S/2 g.g + g.t + alpha_1 rho N_a^2) I ]
  // One wave per cell.  Stage 1, lane = quadrature point: material response from the record (the assembly's own
  // function), 20 numbers per point to LDS.  Stage 2, lane = (node a, half h of the points): 32 points each, the point's
  // numbers as LDS broadcasts, the lane's 1D table rows in registers; the two halves meet in LDS and the node's six
  // numbers (xx yy zz xy xz yz) go to the cell's slot of the node (MfParams::dst, as the product's results): no two
  // cells share a slot, so ONE launch covers all colours, and mf_diag_gather sums the slots of a node in processing order.
  constexpr int DG_NF = 20; // M[9], tau[6], (c_II + c_S/2 + 4/3 p) w / 2, 2/3 w, c_S/2 w, w, alpha_1 rho w
  template <bool BOX>
  __global__ __launch_bounds__(64, 3) void mf_diag(MfParams prm, double *__restrict__ slots6)
  {
    constexpr int NPC = 27;
    // (the two halves of the points two doubles apart in bank space: a wave's broadcast read touches ONE address per half, and
    // without the pad both land on the same banks -- 34 % of the LDS cycles were conflicts, profiles/r06/pmc_counters_mf_diag_n59.json)
    constexpr int DG_HALF = 32 * DG_NF + 2;
    __shared__ __attribute__((aligned(16))) double sF[2 * DG_HALF];
    __shared__ double sR[NPC * 6];
    const int     lane = threadIdx.x;
    int64_t       cell = blockIdx.x;
    if (prm.xcd_chunk > 0)
      {
        const int64_t local = int64_t(blockIdx.x & 7) * prm.xcd_chunk + (blockIdx.x >> 3);
        if (local >= prm.count)
          return;
        cell = local;
      }
    // ---- stage 1: lane = quadrature point
    {
      double rec[MF_NREC];
      const double *__restrict__ rp = prm.qrec + cell * int64_t(MF_NREC * 64) + lane;
#pragma unroll
      for (int f = 0; f < MF_NREC; ++f)
        rec[f] = rp[f * 64];
      const int    qz = lane >> 4;
      const double wq = prm.tab1d[24 + (lane & 3)] * prm.tab1d[24 + ((lane >> 2) & 3)] * prm.tab1d[24 + qz];
      double       Finv[9], tau[6], tiso[6], cII, cS, M[9], detJ;
      neo_hooke_from_F<3>(rec, det3x3(rec), rec[9], rec[10], prm.mu, prm.kappa, Finv, tau, tiso, cII, cS);
      if constexpr (BOX)
        {
          const double *__restrict__ cb = prm.cellbox + cell * 4;
          const double rx = cb[0], ry = cb[1], rz = cb[2];
          detJ            = cb[3];
#pragma unroll
          for (int k = 0; k < 3; ++k)
            {
              M[k]     = rx * Finv[k];
              M[3 + k] = ry * Finv[3 + k];
              M[6 + k] = rz * Finv[6 + k];
            }
        }
      else
        {
          const double *__restrict__ cv = prm.cverts + cell * 24;
          double verts[24], Jm[9], Ji[9];
          const double xiq[3] = {prm.tab1d[28 + (lane & 3)], prm.tab1d[28 + ((lane >> 2) & 3)], prm.tab1d[28 + qz]};
#pragma unroll
          for (int k = 0; k < 24; ++k)
            verts[k] = cv[k];
          q1_jacobian<3>(verts, xiq, Jm);
          detJ = det3x3(Jm);
          inv3x3(Jm, detJ, Ji);
#pragma unroll
          for (int i = 0; i < 3; ++i)
#pragma unroll
            for (int j = 0; j < 3; ++j)
              M[i * 3 + j] = Ji[i * 3 + 0] * Finv[0 * 3 + j] + Ji[i * 3 + 1] * Finv[1 * 3 + j] + Ji[i * 3 + 2] * Finv[2 * 3 + j];
        }
      const double w  = detJ * wq;
      const double pv = tau[0] - tiso[0]; // kappa/2 (J^2 - 1)
      double      *o  = sF + (lane >> 5) * DG_HALF + (lane & 31) * DG_NF;
#pragma unroll
      for (int k = 0; k < 9; ++k)
        o[k] = M[k];
#pragma unroll
      for (int k = 0; k < 6; ++k)
        o[9 + k] = tau[k];
      o[15] = 0.5 * w * (cII + 0.5 * cS + (4.0 / 3.0) * pv);
      o[16] = (2.0 / 3.0) * w;
      o[17] = 0.5 * cS * w;
      o[18] = w;
      o[19] = prm.mass * w;
    }
    __syncthreads();
    // ---- stage 2: lane = (a, h)
    const int  h = lane >= NPC ? 1 : 0, a = lane - NPC * h;
    const bool act = lane < 2 * NPC;
    const int  a3 = a / 9, a2 = (a - 9 * a3) / 3, a1 = a - 9 * a3 - 3 * a2;
    double     Sx[4], Dx[4], Sy[4], Dy[4], Sz[2], Dz[2];
#pragma unroll
    for (int q = 0; q < 4; ++q)
      {
        Sx[q] = prm.tab1d[q * 3 + (act ? a1 : 0)];
        Dx[q] = prm.tab1d[12 + q * 3 + (act ? a1 : 0)];
        Sy[q] = prm.tab1d[q * 3 + (act ? a2 : 0)];
        Dy[q] = prm.tab1d[12 + q * 3 + (act ? a2 : 0)];
      }
#pragma unroll
    for (int z = 0; z < 2; ++z)
      {
        Sz[z] = prm.tab1d[(2 * h + z) * 3 + (act ? a3 : 0)];
        Dz[z] = prm.tab1d[12 + (2 * h + z) * 3 + (act ? a3 : 0)];
      }
    double K[6] = {0.0, 0.0, 0.0, 0.0, 0.0, 0.0}; // xx yy zz xy xz yz (the diagonal as sum of g_i v_i: doubled, + the isotropic part, at the end)
    double Kiso = 0.0;
    const double *__restrict__ fb = sF + h * DG_HALF;
#pragma unroll
    for (int z = 0; z < 2; ++z)
#pragma unroll
      for (int qy = 0; qy < 4; ++qy)
        {
          const double syz = Sy[qy] * Sz[z], dyz = Dy[qy] * Sz[z], sdz = Sy[qy] * Dz[z];
#pragma unroll
          for (int qx = 0; qx < 4; ++qx)
            {
              const double *__restrict__ f = fb + (z * 16 + qy * 4 + qx) * DG_NF;
              const double d0 = Dx[qx] * syz, d1 = Sx[qx] * dyz, d2 = Sx[qx] * sdz, N = Sx[qx] * syz;
              double       g[3], t[3], v[3];
#pragma unroll
              for (int j = 0; j < 3; ++j)
                g[j] = fma(d2, f[6 + j], fma(d1, f[3 + j], d0 * f[j]));
              sym_mul(f + 9, g, t);
              const double gg = fma(g[2], g[2], fma(g[1], g[1], g[0] * g[0]));
              const double gt = fma(g[2], t[2], fma(g[1], t[1], g[0] * t[0]));
              const double dd = fma(f[19] * N, N, fma(f[18], gt, f[17] * gg));
#pragma unroll
              for (int j = 0; j < 3; ++j)
                v[j] = fma(-f[16], t[j], f[15] * g[j]);
              Kiso += dd;
              K[0] = fma(g[0], v[0], K[0]);
              K[1] = fma(g[1], v[1], K[1]);
              K[2] = fma(g[2], v[2], K[2]);
              K[3] = fma(g[0], v[1], fma(v[0], g[1], K[3]));
              K[4] = fma(g[0], v[2], fma(v[0], g[2], K[4]));
              K[5] = fma(g[1], v[2], fma(v[1], g[2], K[5]));
            }
        }
#pragma unroll
    for (int e = 0; e < 3; ++e)
      K[e] = fma(2.0, K[e], Kiso);
    if (h == 1 && act)
#pragma unroll
      for (int e = 0; e < 6; ++e)
        sR[a * 6 + e] = K[e];
    __syncthreads();
    if (lane < NPC)
      {
        double *__restrict__ o = slots6 + int64_t(prm.dst[cell * NPC + lane]) * 6;
#pragma unroll
        for (int e = 0; e < 6; ++e)
          o[e] = K[e] + sR[lane * 6 + e];
      }
  }

  // the slots of every node summed in processing order under the rule of the assembled matrix ([DEAL.II]
  // distribute_local_to_global as assemble_q2sf's scatter applies it): entries in the row or column of a constrained
  // component are dropped, its diagonal receives |K_e(i,i)| of every cell.  Out: the block (what mf_spmv / the gathers
  // read at constrained dofs and mi_get_diagonal_blocks returns), its inverse in full and as symmetric half (block-Jacobi
  // smoother), 1 / diagonal (Jacobi).  Nodes without a row here (ghost planes of a slab): zeros, never read.
  __global__ __launch_bounds__(256) void mf_diag_gather(const double *__restrict__ slots6, const int32_t *__restrict__ slot_base,
                                                        const int32_t *__restrict__ slot_src,
                                                        const uint8_t *__restrict__ cmask, const int32_t *__restrict__ diagpos,
                                                        double *blk, double *dinv, double *dinv_blk, double *sym6, int64_t nnodes)
  {
    const int64_t n = int64_t(blockIdx.x) * 256 + threadIdx.x;
    if (n >= nnodes)
      return;
    double A[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0}, B[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
    if (diagpos[n] >= 0)
      {
        const int     cm = cmask[n] & 7;
        const int32_t b0 = slot_base[n], b1 = slot_base[n + 1];
        double        s[6] = {0, 0, 0, 0, 0, 0};
        for (int32_t k = b0; k < b1; ++k)
          {
            const double2 *__restrict__ p = reinterpret_cast<const double2 *>(slots6 + int64_t(slot_src ? slot_src[k] : k) * 6);
            const double2 p0 = p[0], p1 = p[1], p2 = p[2];
            double        v[6] = {p0.x, p0.y, p1.x, p1.y, p2.x, p2.y};
#pragma unroll
            for (int c = 0; c < 3; ++c)
              if ((cm >> c) & 1)
                v[c] = fabs(v[c]);
            if (cm & 3) // xy: row / column x or y constrained
              v[3] = 0.0;
            if (cm & 5)
              v[4] = 0.0;
            if (cm & 6)
              v[5] = 0.0;
#pragma unroll
            for (int e = 0; e < 6; ++e)
              s[e] = (k == b0) ? v[e] : s[e] + v[e];
          }
        A[0] = s[0], A[4] = s[1], A[8] = s[2];
        A[1] = A[3] = s[3];
        A[2] = A[6] = s[4];
        A[5] = A[7] = s[5];
        inv3x3(A, det3x3(A), B);
      }
#pragma unroll
    for (int k = 0; k < 9; ++k)
      blk[n * 9 + k] = A[k];
    if (dinv_blk)
#pragma unroll
      for (int k = 0; k < 9; ++k)
        dinv_blk[n * 9 + k] = B[k];
    if (sym6)
      {
        double *q = sym6 + n * 6;
        q[0] = B[0], q[1] = B[4], q[2] = B[8], q[3] = B[3], q[4] = B[6], q[5] = B[7];
      }
#pragma unroll
    for (int c = 0; c < 3; ++c)
      dinv[n * 3 + c] = diagpos[n] >= 0 ? 1.0 / A[c * 4] : 0.0;
  }

  // ------------------------------------------------------------------ smoother quadrature 3 x 3 x 3 (round 6)
  // The multigrid smoother's fine-level operator A' = the same tangent integrated with the 27-point Gauss rule (3 per
  // direction: the full-order rule of Q2 elements; the reference -- and the CG's own operator, every residual and the
  // assembly -- integrate with 4 per direction, qf_cell(p+2), nonlinear_elasticity.cc:74).  A preconditioner-side choice, all
  // fp64: the V-cycle smooths, forms its fine residual and estimates its eigenvalues with A'; the CG still multiplies with A.
  // With 27 points every stage of the sum-factorised product has 27 work items, so ONE wave carries TWO cells (lanes 0-26
  // and 32-58), and per cell the wave issues 156 FP64 instructions instead of 412 and reads 27 x 11 instead of 64 x 11 record
  // numbers.  Stages per half-wave (R = 324 doubles reused in place: every stage reads its inputs into registers, the wave
  // synchronises, then the outputs go on top of them; X = the 81 gathered values):
  //   E12  item (c,k,qx): x-line values of plane (c,k) -> B_DS / B_SD / B_SS [c,k][qy][qx]         27 + 18 multiply-adds
  //   E3   item = point:  H[c][l] = d x_c / d xi_l, V[c] = x_c                                      36
  //   point (mf_spmv's: neo_hooke_from_F on the 27-point record, Q = JxW S M^T, mass)               ~150
  //   I3   item (c,qy,qx): contract qz -> C_DS / C_SD / C_SS [c][k][qy][qx]                         36
  //   I2   item (c,k,qx):  contract qy -> E_D / E_S [c,k][j][qx]                                    27
  //   I1   item (c,k,j):   contract qx -> the 81 results into the cell's slots (as mf_spmv)         18
  // mf_records27 runs E12 + E3 on u + du and stores F, J^(-2/3), 1/J of the 27 points: [cell][11][27].
#ifndef MF27_OCC
#define MF27_OCC 5 // waves per SIMD of mf_spmv27 on box meshes: 96 VGPRs with one spilled double (0.196 -> 0.190 ms); 6 spills 76 bytes
#endif
#ifndef MF27_ABL
#define MF27_ABL 0 // timing-only ablations (wrong results): 2 no result stores, 4 every cell reads the records of cell 0 / 1, 8 ... gathers their x
#endif
  constexpr int Q27 = 27, H27 = 336; // points per cell; doubles of LDS per half-wave (R = 324, padded: the halves 16 banks apart; the
                                     // gathered values X live in R's last 81 entries, which E12 -- their only reader -- does not write)
  // S[q][a] = N_a(x_q), D[q][a] = N_a'(x_q) on the 3-point rule: uniform, scalar registers
#define MF27_TABLES(t_)                                 \
  double S27[3][3], D27[3][3];                          \
  _Pragma("unroll") for (int q_ = 0; q_ < 3; ++q_)      \
    _Pragma("unroll") for (int a_ = 0; a_ < 3; ++a_)    \
    {                                                   \
      S27[q_][a_] = (t_)[q_ * 3 + a_];                  \
      D27[q_][a_] = (t_)[9 + q_ * 3 + a_];              \
    }
  // E12 + E3 of one half-wave: X (81 values [c][a], a = (k*3+j)*3+i) -> H, V at point `it` (lanes it < 27)
  __device__ __forceinline__ void mf27_gradients(const double (&S27)[3][3], const double (&D27)[3][3], const double *__restrict__ tab,
                                                 double *__restrict__ R, const double *__restrict__ X, const int it, const bool act, double H[3][3],
                                                 double V[3])
  {
    const int ck = it / 3, qx = it - 3 * ck; // E12 item
    // this lane's rows of the tables (row qx for the x-contraction, row qz for the z-contraction): per-lane loads -- a select
    // over the uniform tables turns into a dynamically indexed private array, i.e. scratch
    const int qzl = it / 9;
    double    sx[3], dx[3], sz[3], dz[3];
#pragma unroll
    for (int i = 0; i < 3; ++i)
      {
        sx[i] = tab[(act ? qx : 0) * 3 + i];
        dx[i] = tab[9 + (act ? qx : 0) * 3 + i];
        sz[i] = tab[(act ? qzl : 0) * 3 + i];
        dz[i] = tab[9 + (act ? qzl : 0) * 3 + i];
      }
    if (act)
      {
        double as[3], ad[3];
#pragma unroll
        for (int j = 0; j < 3; ++j)
          {
            const double x0 = X[ck * 9 + j * 3], x1 = X[ck * 9 + j * 3 + 1], x2 = X[ck * 9 + j * 3 + 2];
            as[j] = sx[0] * x0 + sx[1] * x1 + sx[2] * x2;
            ad[j] = dx[0] * x0 + dx[1] * x1 + dx[2] * x2;
          }
#pragma unroll
        for (int qy = 0; qy < 3; ++qy)
          {
            const int o  = qy * 27 + it; // B[kind][qy][(c,k)][qx]: the lanes of a store are consecutive (conflict free)
            R[o]         = S27[qy][0] * ad[0] + S27[qy][1] * ad[1] + S27[qy][2] * ad[2]; // d/dx
            R[81 + o]    = D27[qy][0] * as[0] + D27[qy][1] * as[1] + D27[qy][2] * as[2]; // d/dy
            R[162 + o]   = S27[qy][0] * as[0] + S27[qy][1] * as[1] + S27[qy][2] * as[2]; // value / d/dz
          }
      }
    __syncthreads();
    const int qz = it / 9, q9 = it - 9 * qz, qy3 = q9 / 3, qx3 = q9 - 3 * qy3; // E3: item = point
#pragma unroll
    for (int c = 0; c < 3; ++c)
      {
        H[c][0] = H[c][1] = H[c][2] = V[c] = 0.0;
        if (act)
#pragma unroll
          for (int k = 0; k < 3; ++k)
            {
              const int    o   = qy3 * 27 + (c * 3 + k) * 3 + qx3;
              const double bds = R[o], bsd = R[81 + o], bss = R[162 + o];
              H[c][0] = fma(sz[k], bds, H[c][0]);
              H[c][1] = fma(sz[k], bsd, H[c][1]);
              H[c][2] = fma(dz[k], bss, H[c][2]);
              V[c]    = fma(sz[k], bss, V[c]);
            }
      }
    __syncthreads(); // B is consumed
  }
  // geometry of the cell at point (qx, qy, qz) of the 27-point rule: Ji = Jinv (row-major), detJ
  template <bool BOX>
  __device__ __forceinline__ void mf27_geometry(const MfParams &prm, const int64_t cell, const int it, double Ji[9], double &detJ)
  {
    if constexpr (BOX)
      {
        const double *__restrict__ cb = prm.cellbox + cell * 4;
#pragma unroll
        for (int k = 0; k < 9; ++k)
          Ji[k] = 0.0;
        Ji[0] = cb[0], Ji[4] = cb[1], Ji[8] = cb[2];
        detJ  = cb[3];
      }
    else
      {
        const double *__restrict__ cv = prm.cverts + cell * 24;
        const int    qz = it / 9, qy = (it - 9 * qz) / 3, qx = it - 9 * qz - 3 * qy;
        const double xiq[3] = {prm.tab27[21 + qx], prm.tab27[21 + qy], prm.tab27[21 + qz]};
        double       verts[24], Jm[9];
#pragma unroll
        for (int k = 0; k < 24; ++k)
          verts[k] = cv[k];
        q1_jacobian<3>(verts, xiq, Jm);
        detJ = det3x3(Jm);
        inv3x3(Jm, detJ, Ji);
      }
  }

  template <bool BOX, bool LAT>
  __global__ __launch_bounds__(64, 4) void mf_records27(MfParams prm, const double *__restrict__ u, const double *__restrict__ du,
                                                        double *__restrict__ rec27)
  {
    __shared__ double s_lds[2 * H27];
    // (idle lanes -- five per half, and the second half of the last wave of an odd cell count -- MIRROR work item 26 / the last
    // cell: they compute and store the same numbers to the same places as the lane they mirror, so no stage needs a branch)
    const int     lane = threadIdx.x, cw = lane >> 5, it = (lane & 31) < Q27 ? (lane & 31) : Q27 - 1;
    const int64_t pair = int64_t(blockIdx.x & 7) * prm.xcd_chunk + (blockIdx.x >> 3);
    if (pair * 2 >= prm.count)
      return;
    const int64_t cell = pair * 2 + cw < prm.count ? pair * 2 + cw : int64_t(prm.count) - 1;
    constexpr bool act = true;
    double *const R = s_lds + cw * H27, *const X = R + 243;
    MF27_TABLES(prm.tab27)
    if (act)
      {
        int32_t node;
        if constexpr (LAT)
          {
            const int32_t node0 = lattice_node0(prm.lat, cell);
            const int     k9 = it / 9, r9 = it - 9 * k9, j3 = r9 / 3, i3 = r9 - 3 * j3;
            node               = node0 + i3 + j3 * prm.lat.nn0 + k9 * prm.lat.nn01;
          }
        else
          node = prm.conn[cell * Q27 + it];
#pragma unroll
        for (int c = 0; c < 3; ++c)
          X[c * Q27 + it] = u[int64_t(node) * 3 + c] + du[int64_t(node) * 3 + c]; // get_total_solution, :580-588
      }
    __syncthreads();
    double H[3][3], V[3];
    mf27_gradients(S27, D27, prm.tab27, R, X, it, act, H, V);
    if (!act)
      return;
    double Ji[9], detJ, gu[9], F[9];
    mf27_geometry<BOX>(prm, cell, it, Ji, detJ);
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
      for (int j = 0; j < 3; ++j)
        gu[i * 3 + j] = H[i][0] * Ji[0 * 3 + j] + H[i][1] * Ji[1 * 3 + j] + H[i][2] * Ji[2 * 3 + j];
#pragma unroll
    for (int k = 0; k < 9; ++k)
      F[k] = gu[k];
    F[0] += 1.0, F[4] += 1.0, F[8] += 1.0;
    double J = det3x3(F);
    // (the assembly reports det F <= 0 at ITS 64 points, nonlinear_elasticity.cc:935; a point of this rule that folds where
    // none of those does takes the undeformed state: the smoother's operator stays finite and positive definite)
    const bool folded = !(J > 0.0);
#pragma unroll
    for (int k = 0; k < 9; ++k)
      F[k] = folded ? (k % 4 == 0 ? 1.0 : 0.0) : F[k];
    J = folded ? 1.0 : J;
    const double rJ = 1.0 / J, Jm = 1.0 / (cbrt(J) * cbrt(J));
    double *__restrict__ g = rec27 + cell * int64_t(MF_NREC * Q27) + it;
#pragma unroll
    for (int k = 0; k < 9; ++k)
      g[k * Q27] = F[k];
    g[9 * Q27]  = Jm;
    g[10 * Q27] = rJ;
  }

  template <bool BOX, bool LAT>
  __global__ __launch_bounds__(64, BOX ? MF27_OCC : 4) void mf_spmv27(MfParams prm)
  {
    __shared__ double s_lds[2 * H27];
    // (idle lanes -- five per half, and the second half of the last wave of an odd cell count -- MIRROR work item 26 / the last
    // cell: they compute and store the same numbers to the same places as the lane they mirror, so no stage needs a branch)
    const int     lane = threadIdx.x, cw = lane >> 5, it = (lane & 31) < Q27 ? (lane & 31) : Q27 - 1;
    const int64_t pair = int64_t(blockIdx.x & 7) * prm.xcd_chunk + (blockIdx.x >> 3);
    if (pair * 2 >= prm.count)
      return;
    int64_t cell = pair * 2 + cw < prm.count ? pair * 2 + cw : int64_t(prm.count) - 1;
    if constexpr (LAT)
      if (prm.sel_n > 0) // a launch over some layers of a slab (as mf_spmv): `cell` counts the selected cells, colour by colour
        {
          const int32_t l = int32_t(cell);
          int32_t       b = prm.sel_begin[0], p0 = prm.sel_pos0[0];
#pragma unroll
          for (int c = 1; c < 8; ++c)
            if (l >= prm.sel_begin[c])
              {
                b  = prm.sel_begin[c];
                p0 = prm.sel_pos0[c];
              }
          cell = int64_t(p0) + (l - b);
        }
    constexpr bool act = true;
    double *const R = s_lds + cw * H27, *const X = R + 243;
    // gather x (constrained entries masked) and this lane's record
    double rec[MF_NREC];
#pragma unroll
    for (int f = 0; f < MF_NREC; ++f)
      rec[f] = 0.0;
    rec[0] = rec[4] = rec[8] = rec[9] = rec[10] = 1.0; // (idle lanes: the identity, no NaN in flight)
    int32_t ydst[3] = {0, 0, 0};
    if (act)
      {
        int32_t node;
        if constexpr (LAT)
          {
            const int32_t node0 = lattice_node0(prm.lat, (MF27_ABL & 8) ? int64_t(cw) : cell);
            const int     k9 = it / 9, r9 = it - 9 * k9, j3 = r9 / 3, i3 = r9 - 3 * j3;
            node               = node0 + i3 + j3 * prm.lat.nn0 + k9 * prm.lat.nn01;
          }
        else
          node = prm.conn[cell * Q27 + it];
        const int cm = prm.cmask[node];
#pragma unroll
        for (int c = 0; c < 3; ++c)
          X[c * Q27 + it] = ((cm >> c) & 1) ? 0.0 : prm.x[int64_t(node) * 3 + c];
        const double *__restrict__ rp = prm.qrec27 + ((MF27_ABL & 4) ? int64_t(cw) : cell) * int64_t(MF_NREC * Q27) + it;
#pragma unroll
        for (int f = 0; f < MF_NREC; ++f)
          rec[f] = __builtin_nontemporal_load(&rp[f * Q27]);
        // I1's item of this lane: line (c,k,j) = it, its three nodes i (cell-major slots: no table to read)
        const int lkj = it - 9 * (it / 9);
#pragma unroll
        for (int i = 0; i < 3; ++i)
          ydst[i] = prm.slot_inline ? int32_t(cell) * Q27 + lkj * 3 + i : prm.dst[cell * Q27 + lkj * 3 + i];
      }
    MF27_TABLES(prm.tab27)
    __syncthreads();
    double H[3][3], V[3];
    mf27_gradients(S27, D27, prm.tab27, R, X, it, act, H, V);
    // ---- point stage: Q = JxW S M^T (as mf_spmv), 12 numbers per point on top of B
    {
      const int    qz = it / 9, qy = (it - 9 * qz) / 3, qx = it - 9 * qz - 3 * qy;
      const double wq = act ? prm.tab27[18 + qx] * prm.tab27[18 + qy] * prm.tab27[18 + qz] : 0.0;
      double       Finv[9], tau[6], tiso[6], cII, cS, Ji[9], detJ = 1.0, M[9];
      neo_hooke_from_F<3>(rec, det3x3(rec), rec[9], rec[10], prm.mu, prm.kappa, Finv, tau, tiso, cII, cS);
      if (act)
        mf27_geometry<BOX>(prm, cell, it, Ji, detJ);
      else
        {
#pragma unroll
          for (int k = 0; k < 9; ++k)
            Ji[k] = 0.0;
        }
#pragma unroll
      for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int j = 0; j < 3; ++j)
          M[i * 3 + j] = BOX ? Ji[i * 4] * Finv[i * 3 + j] :
                               Ji[i * 3 + 0] * Finv[0 * 3 + j] + Ji[i * 3 + 1] * Finv[1 * 3 + j] + Ji[i * 3 + 2] * Finv[2 * 3 + j];
      const double w = detJ * wq, wcII = w * cII, cs2 = 0.5 * cS;
      double       h[3][3];
#pragma unroll
      for (int j = 0; j < 3; ++j)
#pragma unroll
        for (int k = 0; k < 3; ++k)
          h[j][k] = H[j][0] * M[k] + H[j][1] * M[3 + k] + H[j][2] * M[6 + k];
      const double pv  = (tau[0] + tau[1] + tau[2]) * (1.0 / 3.0);
      const double ti0 = tau[0] - pv, ti1 = tau[1] - pv, ti2 = tau[2] - pv;
      const double trh = h[0][0] + h[1][1] + h[2][2];
      const double th  = ti0 * h[0][0] + ti1 * h[1][1] + ti2 * h[2][2] + tau[3] * (h[0][1] + h[1][0]) + tau[4] * (h[0][2] + h[2][0]) +
                        tau[5] * (h[1][2] + h[2][1]);
      const double aI = wcII * trh - (2.0 / 3.0) * w * th;
      const double m3 = -(2.0 / 3.0) * trh;
      const double Tt[3][3] = {{tau[0], tau[3], tau[4]}, {tau[3], tau[1], tau[5]}, {tau[4], tau[5], tau[2]}};
      const double Ti[3][3] = {{ti0, tau[3], tau[4]}, {tau[3], ti1, tau[5]}, {tau[4], tau[5], ti2}};
      const double wm       = prm.mass * w;
      if (act)
#pragma unroll
        for (int i = 0; i < 3; ++i)
          {
            double Sm[3];
#pragma unroll
            for (int j = 0; j < 3; ++j)
              {
                const double v = m3 * Ti[i][j] + cs2 * (h[i][j] + h[j][i]) + h[i][0] * Tt[0][j] + h[i][1] * Tt[1][j] + h[i][2] * Tt[2][j];
                Sm[j]          = w * v + (i == j ? aI : 0.0);
              }
#pragma unroll
            for (int l = 0; l < 3; ++l)
              R[(i * 4 + l) * Q27 + it] = Sm[0] * M[l * 3] + Sm[1] * M[l * 3 + 1] + Sm[2] * M[l * 3 + 2];
            R[(i * 4 + 3) * Q27 + it] = wm * V[i];
          }
    }
    __syncthreads();
    // ---- I3: contract qz.  item (c, q9 = qy*3+qx)
    {
      const int c = it / 9, q9 = it - 9 * c;
      double    v[4][3];
      if (act)
#pragma unroll
        for (int d = 0; d < 4; ++d)
#pragma unroll
          for (int z = 0; z < 3; ++z)
            v[d][z] = R[(c * 4 + d) * Q27 + z * 9 + q9];
      __syncthreads(); // Q is consumed: C goes on top of it
      if (act)
#pragma unroll
        for (int k = 0; k < 3; ++k)
          {
            double cds = 0.0, csd = 0.0, css = 0.0;
#pragma unroll
            for (int z = 0; z < 3; ++z)
              {
                cds = fma(S27[z][k], v[0][z], cds);
                csd = fma(S27[z][k], v[1][z], csd);
                css = fma(D27[z][k], v[2][z], css);
                css = fma(S27[z][k], v[3][z], css);
              }
            R[(0 * 3 + k) * Q27 + it] = cds; // C[kind][k][c][qy][qx]: consecutive lanes
            R[(1 * 3 + k) * Q27 + it] = csd;
            R[(2 * 3 + k) * Q27 + it] = css;
          }
    }
    __syncthreads();
    // ---- I2: contract qy.  item (c, k, qx)
    {
      const int ck = it / 3, qx = it - 3 * ck, c = ck / 3, k = ck - 3 * c;
      double    cds[3], csd[3], css[3];
      if (act)
#pragma unroll
        for (int qy = 0; qy < 3; ++qy)
          {
            const int o = k * Q27 + c * 9 + qy * 3 + qx;
            cds[qy]     = R[0 * 81 + o];
            csd[qy]     = R[1 * 81 + o];
            css[qy]     = R[2 * 81 + o];
          }
      __syncthreads(); // C is consumed: E goes on top of it
      if (act)
#pragma unroll
        for (int j = 0; j < 3; ++j)
          {
            double ed = 0.0, es = 0.0;
#pragma unroll
            for (int qy = 0; qy < 3; ++qy)
              {
                ed = fma(S27[qy][j], cds[qy], ed);
                es = fma(D27[qy][j], csd[qy], es);
                es = fma(S27[qy][j], css[qy], es);
              }
            R[j * Q27 + it]      = ed; // E[kind][j][(c,k)][qx]: consecutive lanes
            R[81 + j * Q27 + it] = es;
          }
    }
    __syncthreads();
    // ---- I1: contract qx, results into the cell's slots.  item = line (c,k,j) = it
    if (act)
      {
        const int lc = it / 9, ckl = it / 3, jl = it - 3 * ckl; // the line (c,k,j)
        double    ed[3], es[3];
#pragma unroll
        for (int qx = 0; qx < 3; ++qx)
          {
            ed[qx] = R[jl * Q27 + ckl * 3 + qx];
            es[qx] = R[81 + jl * Q27 + ckl * 3 + qx];
          }
#pragma unroll
        for (int i = 0; i < 3; ++i)
          {
            double yv = 0.0;
#pragma unroll
            for (int qx = 0; qx < 3; ++qx)
              {
                yv = fma(D27[qx][i], ed[qx], yv);
                yv = fma(S27[qx][i], es[qx], yv);
              }
            if (!(MF27_ABL & 2) || yv == -1.2345678e30)
              prm.yc[int64_t(ydst[i]) * 3 + lc] = yv;
          }
      }
  }

  // system_rhs from the cells' residual slots (point pass in one launch): rhs = 0 - r_1 - r_2 - ... in slot order, the
  // subtractions of the colour-by-colour update in their order; constrained rows get no rhs (:769-773).  One thread per dof.
  __global__ __launch_bounds__(256) void residual_gather(const double *__restrict__ slots3, const int32_t *__restrict__ slot_base,
                                                         const int32_t *__restrict__ slot_src,
                                                         const uint8_t *__restrict__ cmask, double *rhs, int64_t ndofs)
  {
    const int64_t g = int64_t(blockIdx.x) * 256 + threadIdx.x;
    if (g >= ndofs)
      return;
    const int64_t n = g / 3;
    const int     c = int(g - n * 3);
    double        s = 0.0;
    if (!((cmask[n] >> c) & 1))
      {
        const int32_t b0 = slot_base[n], b1 = slot_base[n + 1];
        for (int32_t k = b0; k < b1; ++k)
          s = s - slots3[int64_t(slot_src ? slot_src[k] : k) * 3 + c];
      }
    rhs[g] = s;
  }

  // fp32-rounded copy of the value array for the multigrid smoother (opt-in "precond_storage" 32; same layout)
  __global__ __launch_bounds__(256) void vals_to_f32(const double *__restrict__ v, float *__restrict__ o, int64_t n)
  {
    for (int64_t i = int64_t(blockIdx.x) * 256 + threadIdx.x; i < n; i += int64_t(gridDim.x) * 256)
      o[i] = float(v[i]);
  }

  // one-off: sliced-ELL column indices from the block-CSR pattern (padding rows point at column 0)
  __global__ __launch_bounds__(256) void sell_build_cols(SellParams prm, const int32_t *__restrict__ rowptr,
                                                         const int32_t *__restrict__ bsr_col, int32_t *sell_col)
  {
    const int sl = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (sl >= prm.nslices)
      return;
    const int     node = prm.perm[int64_t(sl) * 64 + lane];
    const int     len  = prm.len[sl];
    const int64_t off  = prm.off[sl];
    for (int k = 0; k < len; ++k)
      sell_col[(off + k) * 64 + lane] = node >= 0 ? bsr_col[rowptr[node] + k] : 0;
  }

  // timing-only: stream the value array with W-byte loads per lane (W = 8 or 16), one partial per workgroup
  template <int W>
  __global__ __launch_bounds__(256) void stream_read(const double *__restrict__ v, int64_t n, double *partials)
  {
    __shared__ double s_red[4];
    const int64_t per = ((n / gridDim.x) / 512) * 512;
    const int64_t i0  = int64_t(blockIdx.x) * per;
    double        s   = 0.0;
    if constexpr (W == 16)
      {
        const double2 *__restrict__ p = reinterpret_cast<const double2 *>(v + i0);
        for (int64_t i = threadIdx.x; i < per / 2; i += 1024)
          {
            const double2 a = p[i], b = (i + 256 < per / 2) ? p[i + 256] : make_double2(0, 0);
            const double2 c = (i + 512 < per / 2) ? p[i + 512] : make_double2(0, 0);
            const double2 d = (i + 768 < per / 2) ? p[i + 768] : make_double2(0, 0);
            s += a.x + a.y + b.x + b.y + c.x + c.y + d.x + d.y;
          }
      }
    else
      {
        const double *__restrict__ p = v + i0;
        for (int64_t i = threadIdx.x; i < per; i += 1024)
          {
            const double a = p[i], b = (i + 256 < per) ? p[i + 256] : 0.0;
            const double c = (i + 512 < per) ? p[i + 512] : 0.0, d = (i + 768 < per) ? p[i + 768] : 0.0;
            s += a + b + c + d;
          }
      }
    s = block_sum<256>(s, s_red);
    if (threadIdx.x == 0)
      partials[blockIdx.x] = s;
  }

  // ------------------------------------------------------------------ CG vector kernels (Jacobi-PCG)
  // scalars: sc[0..1] rz ping-pong, sc[2] tolerance (absolute), sc[3] final residual; flags: [0] 1 converged / 2 breakdown (non-finite residual or p.Ap <= 0), [1] its
  //
  // cg_update_p (iteration it >= 1): totals of the previous update's partials give ||r||^2 and r.z;
  // decide convergence (SolverControl: ||r|| <= tol), else p = z + beta p with z = dinv*r.
  __global__ __launch_bounds__(256) void cg_update_p(CgParams c, int it)
  {
    __shared__ double s_red[4];
    if (c.flags[0])
      return;
    const double rr = c.totals ? c.totals[0] : reduce_partials<256>(c.part_rr, c.npart, s_red);
    const double rz = c.totals ? c.totals[1] : reduce_partials<256>(c.part_rz, c.npart, s_red);
    const double res = sqrt(rr);
    if (!(rr == rr) || !(rz == rz) || rr > 1.79e308 || fabs(rz) > 1.79e308)
      {
        // breakdown: a non-finite residual never passes `res <= tol` (SolverControl fails on NaN at once)
        if (blockIdx.x == 0 && threadIdx.x == 0)
          {
            c.flags[0] = 2;
            c.flags[1] = it - 1;
            c.sc[3]    = res;
          }
        return;
      }
    if (res <= c.sc[2])
      {
        if (blockIdx.x == 0 && threadIdx.x == 0)
          {
            c.flags[0] = 1;
            c.flags[1] = it - 1;
            c.sc[3]    = res;
          }
        return;
      }
    const double beta = (it == 1) ? 0.0 : rz / c.sc[(it - 1) & 1];
    if (blockIdx.x == 0 && threadIdx.x == 0)
      {
        c.sc[it & 1] = rz;
        c.sc[3]      = res;
        c.flags[1]   = it - 1;
      }
    const int64_t per = (c.n + gridDim.x - 1) / gridDim.x;
    const int64_t i0 = blockIdx.x * per, i1 = imin64(c.n, i0 + per);
    if (c.hist && blockIdx.x == 0 && threadIdx.x == 0)
      c.hist[2 * (it - 1) + 1] = beta;
    // it == 1: the old p is not read (0 * NaN would keep the remains of a solve that broke down alive)
    if (c.z)
      for (int64_t i = i0 + threadIdx.x; i < i1; i += 256)
        c.p[i] = c.z[i] + (it == 1 ? 0.0 : beta * c.p[i]);
    else
      for (int64_t i = i0 + threadIdx.x; i < i1; i += 256)
        c.p[i] = c.dinv[i] * c.r[i] + (it == 1 ? 0.0 : beta * c.p[i]);
  }

  // cg_update_xr: alpha = rz / (p.Ap); x += alpha p; r -= alpha Ap; partials of ||r||^2 and r.dinv.r
  __global__ __launch_bounds__(256) void cg_update_xr(CgParams c, int it)
  {
    __shared__ double s_red[4];
    if (c.flags[0])
      return;
    const double pq    = c.totals ? c.totals[2] : reduce_partials<256>(c.part_pq, c.npart_pq, s_red);
    if (!(pq > 0.0) || pq > 1.79e308)
      {
        // breakdown: p.Ap <= 0 (indefinite tangent or preconditioner) or non-finite; every workgroup sees the same pq
        if (blockIdx.x == 0 && threadIdx.x == 0)
          {
            c.flags[0] = 2;
            c.flags[1] = it - 1;
          }
        return;
      }
    const double alpha = c.sc[it & 1] / pq;
    if (c.hist && blockIdx.x == 0 && threadIdx.x == 0)
      c.hist[2 * (it - 1)] = alpha;
    const int64_t per = (c.n + gridDim.x - 1) / gridDim.x;
    const int64_t i0 = blockIdx.x * per, i1 = imin64(c.n, i0 + per);
    double        srr = 0.0, srz = 0.0;
    if (c.z) // general preconditioner: r.z is formed after z = M^-1 r, the Jacobi diagonal is not read
      for (int64_t i = i0 + threadIdx.x; i < i1; i += 256)
        {
          c.x[i] += alpha * c.p[i];
          const double ri = c.r[i] - alpha * c.q[i];
          c.r[i]          = ri;
          srr += ri * ri;
        }
    else
      for (int64_t i = i0 + threadIdx.x; i < i1; i += 256)
        {
          c.x[i] += alpha * c.p[i];
          const double ri = c.r[i] - alpha * c.q[i];
          c.r[i]          = ri;
          srr += ri * ri;
          srz += ri * ri * c.dinv[i];
        }
    srr = block_sum<256>(srr, s_red);
    srz = block_sum<256>(srz, s_red);
    if (threadIdx.x == 0)
      {
        c.part_rr[blockIdx.x] = srr;
        if (!c.z) // with a general preconditioner r.z is formed after z = M^-1 r (dot_partials)
          c.part_rz[blockIdx.x] = srz;
      }
  }

  // cg_update_single (iteration it >= 1 of the single-reduction form of the preconditioned CG, Chronopoulos & Gear 1989):
  // with z = M^-1 r and w = A z at hand and ONE reduction {||r||^2, gamma = r.z, delta = z.w} behind them,
  //   beta = gamma / gamma_old,  alpha = gamma / (delta - beta gamma / alpha_old)        (it == 1: beta = 0, alpha = gamma / delta)
  //   p = z + beta p,  s = w + beta s  (= A p by recurrence),  x += alpha p,  r -= alpha s,
  // the same iterates as cg_update_p / cg_update_xr in exact arithmetic.  The convergence test of the residual the
  // iteration started from is taken first, as in cg_update_p.  Scalars: sc[0..1] gamma ping-pong, sc[5..6] alpha ping-pong.
  __global__ __launch_bounds__(256) void cg_update_single(CgParams c, int it)
  {
    __shared__ double s_red[4];
    if (c.flags[0])
      return;
    const double rr  = c.totals ? c.totals[0] : reduce_partials<256>(c.part_rr, c.npart, s_red);
    const double rz  = c.totals ? c.totals[1] : reduce_partials<256>(c.part_rz, c.npart, s_red);
    const double zw  = c.totals ? c.totals[2] : reduce_partials<256>(c.part_pq, c.npart_pq, s_red);
    const double res = sqrt(rr);
    if (!(rr == rr) || !(rz == rz) || rr > 1.79e308 || fabs(rz) > 1.79e308)
      {
        if (blockIdx.x == 0 && threadIdx.x == 0)
          {
            c.flags[0] = 2;
            c.flags[1] = it - 1;
            c.sc[3]    = res;
          }
        return;
      }
    if (res <= c.sc[2])
      {
        if (blockIdx.x == 0 && threadIdx.x == 0)
          {
            c.flags[0] = 1;
            c.flags[1] = it - 1;
            c.sc[3]    = res;
          }
        return;
      }
    const double beta  = (it == 1) ? 0.0 : rz / c.sc[(it - 1) & 1];
    const double denom = (it == 1) ? zw : zw - beta * rz / c.sc[5 + ((it - 1) & 1)]; // p.Ap of the standard recurrence
    if (!(denom > 0.0) || denom > 1.79e308)
      {
        if (blockIdx.x == 0 && threadIdx.x == 0)
          {
            c.flags[0] = 2;
            c.flags[1] = it - 1;
            c.sc[3]    = res;
          }
        return;
      }
    const double alpha = rz / denom;
    if (blockIdx.x == 0 && threadIdx.x == 0)
      {
        c.sc[it & 1]       = rz;
        c.sc[5 + (it & 1)] = alpha;
        c.sc[3]            = res;
        c.flags[1]         = it - 1;
        if (c.hist)
          {
            c.hist[2 * (it - 1)]     = alpha;
            c.hist[2 * (it - 1) + 1] = beta;
          }
      }
    const int64_t per = (c.n + gridDim.x - 1) / gridDim.x;
    const int64_t i0 = blockIdx.x * per, i1 = imin64(c.n, i0 + per);
    double        srr = 0.0;
    // (it == 1: the old p and s are not read)
    for (int64_t i = i0 + threadIdx.x; i < i1; i += 256)
      {
        const double pi = c.z[i] + (it == 1 ? 0.0 : beta * c.p[i]);
        const double si = c.q[i] + (it == 1 ? 0.0 : beta * c.s[i]);
        c.p[i]          = pi;
        c.s[i]          = si;
        c.x[i] += alpha * pi;
        const double ri = c.r[i] - alpha * si;
        c.r[i]          = ri;
        srr += ri * ri;
      }
    srr = block_sum<256>(srr, s_red);
    if (threadIdx.x == 0)
      c.part_rr[blockIdx.x] = srr;
  }

  // partials of a . b over [0,n)
  __global__ __launch_bounds__(256) void dot_partials(const double *__restrict__ a, const double *__restrict__ b,
                                                      int64_t n, double *part)
  {
    __shared__ double s_red[4];
    const int64_t per = (n + gridDim.x - 1) / gridDim.x;
    const int64_t i0 = blockIdx.x * per, i1 = imin64(n, i0 + per);
    double        s  = 0.0;
    for (int64_t i = i0 + threadIdx.x; i < i1; i += 256)
      s += a[i] * b[i];
    s = block_sum<256>(s, s_red);
    if (threadIdx.x == 0)
      part[blockIdx.x] = s;
  }

  // ------------------------------------------------------------------ multigrid pieces
  // One Chebyshev-Jacobi step for A x = b:  res = b - q (q = A x, or 0 if q == null);
  // d = c1 d + c2 D^-1 res;  x += d.   (Saad, Iterative Methods, Alg. 12.1 with D^-1 A)
  __global__ __launch_bounds__(256) void cheb_step(double *x, double *d, const double *__restrict__ b,
                                                   const double *__restrict__ q, const double *__restrict__ dinv,
                                                   double c1, double c2, int64_t n)
  {
    const int64_t i = int64_t(blockIdx.x) * 256 + threadIdx.x;
    if (i >= n)
      return;
    const double res = b[i] - (q ? q[i] : 0.0);
    const double dn  = (c1 != 0.0 ? c1 * d[i] : 0.0) + c2 * dinv[i] * res; // first step: old d not read
    d[i]             = dn;
    x[i]             = (q ? x[i] : 0.0) + dn; // first step starts from x = 0
  }
  // Chebyshev smoother of the 4th kind with optimised weights (Lottes, "Optimal polynomial smoothers for multigrid
  // V-cycles", 2022, Alg. 3) on D^-1 A; only an upper bound rho of the spectrum is needed.
  //   start: r = b - q (q = A x0, null for x0 = 0 which is then written);  d = s0 D^-1 r
  //   step : x += beta d;  then (q = A d given)  r -= q;  d = ca d + cb D^-1 r     (q == null: last step, x only)
  __global__ __launch_bounds__(256) void cheb4_start(double *x, double *d, double *r, const double *__restrict__ b,
                                                     const double *__restrict__ q, const double *__restrict__ dinv,
                                                     double s0, int64_t n)
  {
    const int64_t i = int64_t(blockIdx.x) * 256 + threadIdx.x;
    if (i >= n)
      return;
    const double res = b[i] - (q ? q[i] : 0.0);
    r[i]             = res;
    d[i]             = s0 * dinv[i] * res;
    if (!q)
      x[i] = 0.0;
  }
  __global__ __launch_bounds__(256) void cheb4_step(double *x, double *d, double *r, const double *__restrict__ q,
                                                    const double *__restrict__ dinv, double beta, double ca, double cb,
                                                    int64_t n)
  {
    const int64_t i = int64_t(blockIdx.x) * 256 + threadIdx.x;
    if (i >= n)
      return;
    const double di = d[i];
    x[i] += beta * di;
    if (q)
      {
        const double res = r[i] - q[i];
        r[i]             = res;
        d[i]             = ca * di + cb * dinv[i] * res;
      }
  }
  // dst = s * a .* b   (b == null: dst = s * a)
  __global__ __launch_bounds__(256) void vec_scale_mul(double *dst, const double *__restrict__ a,
                                                       const double *__restrict__ b, double s, int64_t n)
  {
    const int64_t i = int64_t(blockIdx.x) * 256 + threadIdx.x;
    if (i < n)
      dst[i] = s * a[i] * (b ? b[i] : 1.0);
  }
  // res = b - q
  __global__ __launch_bounds__(256) void vec_residual(double *res, const double *__restrict__ b,
                                                      const double *__restrict__ q, int64_t n)
  {
    const int64_t i = int64_t(blockIdx.x) * 256 + threadIdx.x;
    if (i < n)
      res[i] = b[i] - q[i];
  }
  // x = a h1 + b h2 (start vector of a linear solve predicted from the solutions of the previous time steps)
  __global__ __launch_bounds__(256) void vec_lincomb2(double *x, double a, const double *__restrict__ h1, double b,
                                                      const double *__restrict__ h2, int64_t n)
  {
    const int64_t i = int64_t(blockIdx.x) * 256 + threadIdx.x;
    if (i < n)
      x[i] = b != 0.0 ? a * h1[i] + b * h2[i] : a * h1[i];
  }
  // start vector h of a solve scaled to its best multiple: x = alpha h, q = alpha (A h) with alpha = h.b / h.Ah (the
  // multiple with the smallest energy-norm error; sc = {h.b, h.Ah}; alpha = 0 if h.Ah is not positive)
  __global__ __launch_bounds__(256) void scale_start(double *x, double *q, const double *__restrict__ sc, int64_t n)
  {
    const int64_t i = int64_t(blockIdx.x) * 256 + threadIdx.x;
    if (i >= n)
      return;
    const double hb = sc[0], hq = sc[1];
    const double alpha = (hq > 0.0 && hb == hb) ? hb / hq : 0.0;
    x[i] *= alpha;
    q[i] *= alpha;
  }
  // y = mask(x): copy the dofs inside [own0, own0+own_n), zero elsewhere (right-hand side of the V-cycle)
  __global__ __launch_bounds__(256) void copy_owned(double *y, const double *__restrict__ x, int64_t n, int64_t own0,
                                                    int64_t own_n)
  {
    const int64_t i = int64_t(blockIdx.x) * 256 + threadIdx.x;
    if (i < n)
      y[i] = (i >= own0 && i < own0 + own_n) ? x[i] : 0.0;
  }

  // Lattice transfers in index space.  Every level is a tensor-product lattice; along each direction the table
  // gives, for a TARGET index t, the source index i0[t] and the weight w[t] of i0[t]+1 (linear interpolation).
  //   interp  : target[t] (+)= sum over the 2^dim source corners  (prolongation: target fine, source coarse;
  //             state transfer: target coarse, source fine)
  //   restrict: coarse[I] = sum over fine nodes whose interpolation stencil touches I (transpose of interp),
  //             written as a gather over per-direction lists  start[I] .. start[I+1]  of (fine index, weight)
  template <int D, bool ADD>
  __global__ __launch_bounds__(256) void lattice_interp(LatticeParams p, double *tgt, const double *__restrict__ src,
                                                        const uint8_t *__restrict__ cmask_tgt)
  {
    const int64_t t = int64_t(blockIdx.x) * 256 + threadIdx.x;
    if (t >= p.n_tgt)
      return;
    int ti[3] = {int(t % p.nt[0]), int((t / p.nt[0]) % p.nt[1]), int(t / (int64_t(p.nt[0]) * p.nt[1]))};
    int    i0[3] = {0, 0, 0};
    double w[3]  = {0, 0, 0};
#pragma unroll
    for (int d = 0; d < D; ++d)
      {
        i0[d] = p.i0[d][ti[d]];
        w[d]  = p.w[d][ti[d]];
      }
    bool served = true; // a negative source index: this target node is served by another slab (contributes 0)
#pragma unroll
    for (int d = 0; d < D; ++d)
      served = served && (i0[d] >= 0);
    double acc[D];
#pragma unroll
    for (int c = 0; c < D; ++c)
      acc[c] = 0.0;
#pragma unroll
    for (int corner = 0; corner < (1 << D); ++corner)
      {
        double  wt = 1.0;
        int64_t s  = 0, stride = 1;
        bool    ok = served;
#pragma unroll
        for (int d = 0; d < D; ++d)
          {
            const int hi = (corner >> d) & 1;
            const double wd = hi ? w[d] : 1.0 - w[d];
            wt *= wd;
            const int idx = i0[d] + hi;
            ok            = ok && (wd != 0.0);
            s += int64_t(idx < p.ns[d] ? idx : p.ns[d] - 1) * stride;
            stride *= p.ns[d];
          }
        // (loads unconditional -- the index is clamped --, so that the 2^D corners travel together; the sum skips the
        // corners of weight zero and the nodes another slab serves, as before)
        double v[D];
#pragma unroll
        for (int c = 0; c < D; ++c)
          v[c] = src[(served ? s : 0) * D + c];
        if (ok)
#pragma unroll
          for (int c = 0; c < D; ++c)
            acc[c] += wt * v[c];
      }
    const int m = cmask_tgt ? cmask_tgt[t] : 0;
#pragma unroll
    for (int c = 0; c < D; ++c)
      {
        const double v = ((m >> c) & 1) ? 0.0 : acc[c];
        if (ADD)
          tgt[t * D + c] += v;
        else
          tgt[t * D + c] = v;
      }
  }

  template <int D>
  __global__ __launch_bounds__(256) void lattice_restrict(LatticeParams p, double *coarse, const double *__restrict__ fine,
                                                          const uint8_t *__restrict__ cmask_coarse)
  {
    const int64_t I = int64_t(blockIdx.x) * 256 + threadIdx.x;
    if (I >= p.n_tgt)
      return;
    int ci[3] = {int(I % p.nt[0]), int((I / p.nt[0]) % p.nt[1]), int(I / (int64_t(p.nt[0]) * p.nt[1]))};
    int b[3] = {0, 0, 0}, e[3] = {1, 1, 1};
#pragma unroll
    for (int d = 0; d < D; ++d)
      {
        b[d] = p.rstart[d][ci[d]];
        e[d] = p.rstart[d][ci[d] + 1];
      }
    double acc[D];
#pragma unroll
    for (int c = 0; c < D; ++c)
      acc[c] = 0.0;
    for (int kz = b[2]; kz < e[2]; ++kz)
      {
        const double  wz = (D == 3) ? p.rw[2][kz] : 1.0;
        const int64_t fz = (D == 3) ? p.ri[2][kz] : 0;
        for (int ky = b[1]; ky < e[1]; ++ky)
          {
            const double  wy = p.rw[1][ky] * wz;
            const int64_t fy = p.ri[1][ky];
            for (int kx = b[0]; kx < e[0]; ++kx)
              {
                const double  wt = p.rw[0][kx] * wy;
                const int64_t f  = p.ri[0][kx] + int64_t(p.ns[0]) * (fy + int64_t(p.ns[1]) * fz);
#pragma unroll
                for (int c = 0; c < D; ++c)
                  acc[c] += wt * fine[f * D + c];
              }
          }
      }
    const int m = cmask_coarse ? cmask_coarse[I] : 0;
#pragma unroll
    for (int c = 0; c < D; ++c)
      coarse[I * D + c] = ((m >> c) & 1) ? 0.0 : acc[c];
  }

  // the same sums (same order, same bits) with the lists in registers first: one thread per DOF, every list at most
  // MAXR long (factor-2 coarsening: 3, 4 where the lattices are not nested), so the table entries are loaded once and
  // the up to MAXR^D fine values are independent loads -- three memory round trips deep instead of one per term
  // (the small levels of a V-cycle are latency, not bandwidth: 20 -> 6 us on the 31^3 level)
  template <int D, int MAXR>
  __global__ __launch_bounds__(256) void lattice_restrict_unrolled(LatticeParams p, double *coarse, const double *__restrict__ fine,
                                                                   const uint8_t *__restrict__ cmask_coarse)
  {
    const int64_t g = int64_t(blockIdx.x) * 256 + threadIdx.x;
    if (g >= p.n_tgt * D)
      return;
    const int64_t I = g / D;
    const int     c = int(g - I * D);
    const int     ci[3] = {int(I % p.nt[0]), int((I / p.nt[0]) % p.nt[1]), int(I / (int64_t(p.nt[0]) * p.nt[1]))};
    int           cnt[3] = {1, 1, 1}, fi[3][MAXR];
    double        fw[3][MAXR];
#pragma unroll
    for (int d = 0; d < 3; ++d)
#pragma unroll
      for (int k = 0; k < MAXR; ++k)
        {
          fi[d][k] = 0;
          fw[d][k] = (d >= D && k == 0) ? 1.0 : 0.0;
        }
#pragma unroll
    for (int d = 0; d < D; ++d)
      {
        const int b = p.rstart[d][ci[d]];
        cnt[d]      = p.rstart[d][ci[d] + 1] - b;
#pragma unroll
        for (int k = 0; k < MAXR; ++k)
          if (k < cnt[d])
            {
              fi[d][k] = p.ri[d][b + k];
              fw[d][k] = p.rw[d][b + k];
            }
      }
    double acc = 0.0;
#pragma unroll
    for (int kz = 0; kz < (D == 3 ? MAXR : 1); ++kz)
#pragma unroll
      for (int ky = 0; ky < MAXR; ++ky)
#pragma unroll
        for (int kx = 0; kx < MAXR; ++kx)
          {
            // the load itself is unconditional (entries past a list point at fine node 0): guarded, the up to MAXR^D
            // loads would wait for one another; only the sum skips them
            const double  wz = (D == 3) ? fw[2][kz] : 1.0;
            const double  wt = fw[0][kx] * (fw[1][ky] * wz);
            const int64_t f  = fi[0][kx] + int64_t(p.ns[0]) * (fi[1][ky] + int64_t(p.ns[1]) * (D == 3 ? fi[2][kz] : 0));
            const double  v  = fine[f * D + c];
            if (kx < cnt[0] && ky < cnt[1] && kz < cnt[2])
              acc += wt * v;
          }
    const int m = cmask_coarse ? cmask_coarse[I] : 0;
    coarse[g]   = ((m >> c) & 1) ? 0.0 : acc;
  }

  // The same restriction for 3 components with the FIRST step of the coarse level's Chebyshev smoother from a zero start in
  // its epilogue (round 5: one launch per level and V-cycle fewer): b = R q as above, then on the nodes [node0, node0 + nnodes)
  // d = x = c2 D^-1 b with the block-Jacobi diagonal -- the arithmetic of cheb_step_blk3 with c1 = 0 and no product, bit by
  // bit.  192 threads = 64 nodes: a node's three components share their values through LDS.
  template <int MAXR>
  __global__ __launch_bounds__(192) void lattice_restrict_first_step3(LatticeParams p, double *coarse, const double *__restrict__ fine,
                                                                      const uint8_t *__restrict__ cmask_coarse, double *x, double *d,
                                                                      const double *__restrict__ dinv, double c2, int64_t node0,
                                                                      int64_t nnodes)
  {
    constexpr int     D = 3;
    __shared__ double s_b[192];
    const int         ld = threadIdx.x;
    const int64_t     g  = int64_t(blockIdx.x) * 192 + ld;
    const bool        in = g < p.n_tgt * D;
    const int64_t     I  = in ? g / D : 0;
    const int         c  = int(g - I * D);
    double            bv = 0.0;
    if (in)
      {
        const int ci[3] = {int(I % p.nt[0]), int((I / p.nt[0]) % p.nt[1]), int(I / (int64_t(p.nt[0]) * p.nt[1]))};
        int       cnt[3], fi[3][MAXR];
        double    fw[3][MAXR];
#pragma unroll
        for (int dd = 0; dd < 3; ++dd)
          {
            const int b = p.rstart[dd][ci[dd]];
            cnt[dd]     = p.rstart[dd][ci[dd] + 1] - b;
#pragma unroll
            for (int k = 0; k < MAXR; ++k)
              {
                fi[dd][k] = 0;
                fw[dd][k] = 0.0;
                if (k < cnt[dd])
                  {
                    fi[dd][k] = p.ri[dd][b + k];
                    fw[dd][k] = p.rw[dd][b + k];
                  }
              }
          }
        double acc = 0.0;
#pragma unroll
        for (int kz = 0; kz < MAXR; ++kz)
#pragma unroll
          for (int ky = 0; ky < MAXR; ++ky)
#pragma unroll
            for (int kx = 0; kx < MAXR; ++kx)
              {
                const double  wt = fw[0][kx] * (fw[1][ky] * fw[2][kz]);
                const int64_t f  = fi[0][kx] + int64_t(p.ns[0]) * (fi[1][ky] + int64_t(p.ns[1]) * fi[2][kz]);
                const double  v  = fine[f * D + c];
                if (kx < cnt[0] && ky < cnt[1] && kz < cnt[2])
                  acc += wt * v;
              }
        const int m = cmask_coarse ? cmask_coarse[I] : 0;
        bv          = ((m >> c) & 1) ? 0.0 : acc;
        coarse[g]   = bv;
      }
    s_b[ld] = bv;
    __syncthreads();
    if (!in || I < node0 || I >= node0 + nnodes)
      return;
    const int    r0 = (ld / 3) * 3;
    const double a0 = dinv[g * 3], a1 = dinv[g * 3 + 1], a2 = dinv[g * 3 + 2];
    double       s  = 0.0;
    s += a0 * s_b[r0];
    s += a1 * s_b[r0 + 1];
    s += a2 * s_b[r0 + 2];
    const double dn = 0.0 + c2 * s;
    d[g]            = dn;
    x[g]            = 0.0 + dn;
  }

  // r = b - q (q = A x0), partials of ||r||^2, r.dinv.r and ||b||^2
  __global__ __launch_bounds__(256) void cg_init_residual(CgParams c, const double *__restrict__ b, double *part_bb)
  {
    __shared__ double s_red[4];
    const int64_t per = (c.n + gridDim.x - 1) / gridDim.x;
    const int64_t i0 = blockIdx.x * per, i1 = imin64(c.n, i0 + per);
    double        srr = 0.0, srz = 0.0, sbb = 0.0;
    for (int64_t i = i0 + threadIdx.x; i < i1; i += 256)
      {
        const double bi = b[i], ri = bi - c.q[i];
        c.r[i]          = ri;
        srr += ri * ri;
        srz += ri * ri * c.dinv[i];
        sbb += bi * bi;
      }
    srr = block_sum<256>(srr, s_red);
    srz = block_sum<256>(srz, s_red);
    sbb = block_sum<256>(sbb, s_red);
    if (threadIdx.x == 0)
      {
        c.part_rr[blockIdx.x] = srr;
        c.part_rz[blockIdx.x] = srz;
        part_bb[blockIdx.x]   = sbb;
      }
  }

  // tol = rel_tol * ||b||  (:1171-1172), or the absolute value -rel_tol when negative; reset flags
  __global__ __launch_bounds__(256) void cg_set_tolerance(CgParams c, const double *part_bb, double rel_tol)
  {
    __shared__ double s_red[4];
    const double bb = c.totals ? c.totals[3] : reduce_partials<256>(part_bb, c.npart, s_red);
    if (threadIdx.x == 0)
      {
        c.sc[2]    = rel_tol >= 0.0 ? rel_tol * sqrt(bb) : -rel_tol;
        c.sc[3]    = 0.0;
        c.sc[4]    = sqrt(bb);
        c.flags[0] = 0;
        c.flags[1] = 0;
      }
  }

  // last check after the final enqueued iteration (convergence is otherwise detected by the next update_p)
  __global__ __launch_bounds__(256) void cg_final_check(CgParams c, int it)
  {
    __shared__ double s_red[4];
    if (c.flags[0])
      return;
    const double rr  = c.totals ? c.totals[0] : reduce_partials<256>(c.part_rr, c.npart, s_red);
    const double res = sqrt(rr);
    if (threadIdx.x == 0)
      {
        c.sc[3]    = res;
        c.flags[1] = it;
        if (res <= c.sc[2])
          c.flags[0] = 1;
        else if (!(rr == rr) || rr > 1.79e308)
          c.flags[0] = 2;
      }
  }

  // ------------------------------------------------------------------ whole Jacobi-PCG in one launch (small problems)
  // The reference's own geometries have 10^2..10^4 dofs (FSI3 flap: 1,100 at the shipped degree).  There a CG
  // iteration is three 4-microsecond launches plus a poll every 16 iterations -- pure latency.  This kernel runs the
  // complete solve (same recurrences, stopping rule and bookkeeping as cg_init_residual / cg_update_p / sell_spmv /
  // cg_update_xr / cg_final_check) inside ONE workgroup of 1024 threads: vectors and matrix stay in the L2, phases
  // are separated by workgroup barriers, reductions run in a fixed order (deterministic).
  __device__ __forceinline__ double block_sum_1024(double v, double *s_red)
  {
    v = wave_sum(v);
    __syncthreads();
    if ((threadIdx.x & 63) == 0)
      s_red[threadIdx.x >> 6] = v;
    __syncthreads();
    double t = 0;
#pragma unroll
    for (int w = 0; w < 16; ++w)
      t += s_red[w];
    return t;
  }

  // y = K x over all slices of the slab; returns this thread's share of dotv . y (0 if dotv == null)
  template <int D>
  __device__ __forceinline__ double small_spmv(const SellParams &prm, const double *__restrict__ x, double *y,
                                               const double *__restrict__ dotv)
  {
    constexpr int DD = D * D;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    double    dsum = 0.0;
    for (int sl = wave; sl < prm.nslices; sl += 16)
      {
        const int     len  = prm.len[sl];
        const int64_t off  = prm.off[sl];
        const int     node = prm.perm[int64_t(sl) * 64 + lane];
        const int32_t *__restrict__ cp = prm.col + off * 64 + lane;
        const int     wx   = prm.wx[sl]; // block k = (g, kx) of the lane at (off*64 + g*64*wx + lane*wx + kx) (mi_mesh.hpp)
        const double *__restrict__ vp  = prm.vals + (off * 64 + int64_t(lane) * wx) * DD; // the lane's own blocks (matrix <= 1 MiB: L2 hits)
        auto          bpos = [&](int k) { return (int64_t(k / wx) * (64 * wx) + k % wx) * DD; };
        double acc[D];
#pragma unroll
        for (int i = 0; i < D; ++i)
          acc[i] = 0.0;
        // U blocks in flight per lane: the row loop is a chain of L2 round trips otherwise (same summation order)
        constexpr int U = (D == 2) ? 8 : 4;
        int           k = 0;
        for (; k + U <= len; k += U)
          {
            int32_t c[U];
            double  v[U][DD], xx[U][D];
#pragma unroll
            for (int u = 0; u < U; ++u)
              c[u] = cp[int64_t(k + u) * 64];
#pragma unroll
            for (int u = 0; u < U; ++u)
#pragma unroll
              for (int e = 0; e < DD; ++e)
                v[u][e] = vp[bpos(k + u) + e];
#pragma unroll
            for (int u = 0; u < U; ++u)
#pragma unroll
              for (int j = 0; j < D; ++j)
                xx[u][j] = x[int64_t(c[u]) * D + j];
#pragma unroll
            for (int u = 0; u < U; ++u)
#pragma unroll
              for (int i = 0; i < D; ++i)
#pragma unroll
                for (int j = 0; j < D; ++j)
                  acc[i] += v[u][i * D + j] * xx[u][j];
          }
        for (; k < len; ++k)
          {
            const int32_t c = cp[int64_t(k) * 64];
            double        v[DD], xx[D];
#pragma unroll
            for (int e = 0; e < DD; ++e)
              v[e] = vp[bpos(k) + e];
#pragma unroll
            for (int j = 0; j < D; ++j)
              xx[j] = x[int64_t(c) * D + j];
#pragma unroll
            for (int i = 0; i < D; ++i)
#pragma unroll
              for (int j = 0; j < D; ++j)
                acc[i] += v[i * D + j] * xx[j];
          }
        if (node >= 0)
#pragma unroll
          for (int i = 0; i < D; ++i)
            {
              y[int64_t(node) * D + i] = acc[i];
              if (dotv)
                dsum += acc[i] * dotv[int64_t(node) * D + i];
            }
      }
    return dsum;
  }

  template <int D>
  __global__ __launch_bounds__(1024) void cg_small(SellParams prm, CgParams c, const double *__restrict__ b,
                                                   double rel_tol, int max_it)
  {
    __shared__ double s_red[16];
    const int tid = threadIdx.x;
    const int n   = int(c.n);
    // r0 = b - A x0, tolerance = rel_tol * ||b|| (or |rel_tol| absolute)   (:1171-1172)
    small_spmv<D>(prm, c.x, c.q, nullptr);
    __syncthreads();
    double srr = 0.0, srz = 0.0, sbb = 0.0;
    for (int i = tid; i < n; i += 1024)
      {
        const double bi = b[i], ri = bi - c.q[i];
        c.r[i]          = ri;
        srr += ri * ri;
        srz += ri * ri * c.dinv[i];
        sbb += bi * bi;
      }
    double       rr = block_sum_1024(srr, s_red), rz = block_sum_1024(srz, s_red);
    const double bb = block_sum_1024(sbb, s_red);
    const double tol = rel_tol >= 0.0 ? rel_tol * sqrt(bb) : -rel_tol;
    double       rz_prev = 1.0, res = sqrt(rr);
    int          it = 0, done = res <= tol;
    if (!(rr == rr) || !(rz == rz) || rr > 1.79e308)
      done = 2; // breakdown (non-finite start residual)
    while (!done && it < max_it)
      {
        ++it;
        const double beta = (it == 1) ? 0.0 : rz / rz_prev;
        rz_prev           = rz;
        for (int i = tid; i < n; i += 1024)
          c.p[i] = c.dinv[i] * c.r[i] + (it == 1 ? 0.0 : beta * c.p[i]);
        __syncthreads();
        const double pq    = block_sum_1024(small_spmv<D>(prm, c.p, c.q, c.p), s_red);
        if (!(pq > 0.0) || pq > 1.79e308)
          {
            done = 2; // breakdown: p.Ap <= 0 or non-finite
            --it;
            break;
          }
        const double alpha = rz_prev / pq;
        srr = srz = 0.0;
        for (int i = tid; i < n; i += 1024)
          {
            c.x[i] += alpha * c.p[i];
            const double ri = c.r[i] - alpha * c.q[i];
            c.r[i]          = ri;
            srr += ri * ri;
            srz += ri * ri * c.dinv[i];
          }
        rr   = block_sum_1024(srr, s_red);
        rz   = block_sum_1024(srz, s_red);
        res  = sqrt(rr);
        done = res <= tol;
        if (!(rr == rr) || !(rz == rz) || rr > 1.79e308)
          done = 2;
      }
    if (tid == 0)
      {
        c.sc[2]    = tol;
        c.sc[3]    = res;
        c.sc[4]    = sqrt(bb);
        c.flags[0] = done;
        c.flags[1] = it;
      }
  }

  // ------------------------------------------------------------------ exact solve on the coarsest multigrid level
  // The coarsest level (2^dim cells: 81 dofs in 3D) is small enough to invert: one workgroup scatters the level's
  // sliced-ELL matrix into a dense n x n array in LDS, inverts it in place (Gauss-Jordan without pivoting: the matrix
  // is symmetric positive definite, constrained dofs are decoupled diagonal entries) and writes the inverse out; a
  // V-cycle then applies it in ONE launch instead of a degree-12 polynomial (12 launches).
  constexpr int DENSE_MAX = 96;
  template <int D>
  __global__ __launch_bounds__(256) void dense_inverse_from_sell(SellParams prm, int n, double *out)
  {
    constexpr int DD = D * D;
    __shared__ double A[DENSE_MAX * DENSE_MAX];
    const int tid = threadIdx.x;
    for (int i = tid; i < n * n; i += 256)
      A[i] = 0.0;
    __syncthreads();
    for (int r = tid; r < prm.nslices * 64; r += 256)
      {
        const int sl = r >> 6, lane = r & 63;
        const int node = prm.perm[r];
        if (node < 0)
          continue;
        const int     len = prm.len[sl], wx = prm.wx[sl];
        const int64_t off = prm.off[sl];
        for (int k = 0; k < len; ++k)
          {
            const int32_t c = prm.col[(off + k) * 64 + lane];
#pragma unroll
            for (int i = 0; i < D; ++i)
#pragma unroll
              for (int j = 0; j < D; ++j)
                A[(node * D + i) * n + c * D + j] +=
                  prm.vals[(off * 64 + int64_t(k / wx) * (64 * wx) + lane * wx + k % wx) * DD + i * D + j];
          }
      }
    __syncthreads();
    for (int p = 0; p < n; ++p)
      {
        const double r = 1.0 / A[p * n + p];
        __syncthreads();
        for (int j = tid; j < n; j += 256)
          if (j != p)
            A[p * n + j] *= r;
        __syncthreads();
        for (int e = tid; e < n * n; e += 256)
          {
            const int i = e / n, j = e - i * n;
            if (i != p && j != p)
              A[e] -= A[i * n + p] * A[p * n + j];
          }
        __syncthreads();
        for (int i = tid; i < n; i += 256)
          A[i * n + p] = (i == p) ? r : -A[i * n + p] * r;
        __syncthreads();
      }
    for (int i = tid; i < n * n; i += 256)
      out[i] = A[i];
  }
  // x = Ainv b (n <= DENSE_MAX); the inverse of a symmetric matrix is read by columns
  __global__ __launch_bounds__(128) void dense_apply(const double *__restrict__ inv, const double *__restrict__ b, double *x,
                                                     int n)
  {
    __shared__ double s_b[DENSE_MAX];
    const int         i = threadIdx.x;
    if (i < n)
      s_b[i] = b[i];
    __syncthreads();
    if (i >= n)
      return;
    double s = 0.0;
    for (int j = 0; j < n; ++j)
      s += inv[j * n + i] * s_b[j];
    x[i] = s;
  }

  // The same for a coarsest level of up to DENSE_BIG_MAX dofs (round 5: 4^3 cells = 375 dofs in 3D, so that the hierarchy
  // ends one level earlier).  The dense array (1.1 MB) lives in device memory and stays in the L2; ONE workgroup inverts it
  // in place by BLOCKED Gauss-Jordan elimination without pivoting (the matrix is symmetric positive definite): per block
  // of GJB pivots one pass over the array -- the scaled pivot rows R = D^-1 A[P,:] and the pivot columns C = A[:,P] are
  // staged in LDS, every other entry takes a GJB-term update from them.  (An unblocked elimination makes one pass per
  // pivot: 24 ms for 300 dofs, one CU's share of the L2 bandwidth; this form 24 passes in all.)  Runs when the coarse
  // operators are rebuilt, not per V-cycle.
  constexpr int DENSE_BIG_MAX = 384, GJB = 16;
  template <int D>
  __global__ __launch_bounds__(1024) void dense_inverse_from_sell_big(SellParams prm, int n, double *A)
  {
    constexpr int DD = D * D;
    __shared__ double sR[GJB][DENSE_BIG_MAX]; // D^-1 A[P, :]
    __shared__ double sC[DENSE_BIG_MAX][GJB + 1]; // A[:, P]
    __shared__ double sD[GJB][GJB + 1], sDi[GJB][GJB + 1];
    const int tid = threadIdx.x;
    for (int i = tid; i < n * n; i += 1024)
      A[i] = 0.0;
    __syncthreads();
    for (int r = tid; r < prm.nslices * 64; r += 1024) // (a row of the level matrix belongs to one thread: no races)
      {
        const int sl = r >> 6, lane = r & 63;
        const int node = prm.perm[r];
        if (node < 0)
          continue;
        const int     len = prm.len[sl], wx = prm.wx[sl];
        const int64_t off = prm.off[sl];
        for (int k = 0; k < len; ++k)
          {
            const int32_t c = prm.col[(off + k) * 64 + lane];
#pragma unroll
            for (int i = 0; i < D; ++i)
#pragma unroll
              for (int j = 0; j < D; ++j)
                A[(node * D + i) * n + c * D + j] +=
                  prm.vals[(off * 64 + int64_t(k / wx) * (64 * wx) + lane * wx + k % wx) * DD + i * D + j];
          }
      }
    __syncthreads();
    for (int p0 = 0; p0 < n; p0 += GJB)
      {
        const int nb = min(GJB, n - p0);
        // the pivot block, its rows and its columns
        for (int e = tid; e < GJB * GJB; e += 1024)
          {
            const int i = e / GJB, j = e - i * GJB;
            sD[i][j]  = (i < nb && j < nb) ? A[(p0 + i) * n + p0 + j] : (i == j ? 1.0 : 0.0);
            sDi[i][j] = (i == j) ? 1.0 : 0.0;
          }
        for (int e = tid; e < nb * n; e += 1024)
          {
            const int i = e / n, j = e - i * n;
            sR[i][j] = A[(p0 + i) * n + j];
            sC[j][i] = A[j * n + p0 + i];
          }
        __syncthreads();
        // D^-1 by Gauss-Jordan on [D | I] in LDS: thread = (row, column of the augmented part), GJB pivots
        for (int q = 0; q < GJB; ++q)
          {
            double piv = 0.0, mine = 0.0, minei = 0.0, rowq = 0.0, rowqi = 0.0, f = 0.0;
            const int i = tid / GJB, j = tid - (tid / GJB) * GJB;
            if (tid < GJB * GJB)
              {
                piv   = 1.0 / sD[q][q];
                rowq  = sD[q][j] * piv;
                rowqi = sDi[q][j] * piv;
                f     = sD[i][q];
                mine  = sD[i][j];
                minei = sDi[i][j];
              }
            __syncthreads();
            if (tid < GJB * GJB)
              {
                sD[i][j]  = (i == q) ? rowq : mine - f * rowq;
                sDi[i][j] = (i == q) ? rowqi : minei - f * rowqi;
              }
            __syncthreads();
          }
        // R = D^-1 A[P, :]  (columns of the pivot block itself are not used below)
        double rn[6]; // this thread's entries of the new R: e = tid + k * 1024 < GJB * DENSE_BIG_MAX = 6 * 1024
#pragma unroll
        for (int k = 0; k < 6; ++k)
          {
            const int e = tid + k * 1024;
            rn[k]       = 0.0;
            if (e < nb * n)
              {
                const int i = e / n, j = e - i * n;
                for (int m = 0; m < nb; ++m)
                  rn[k] += sDi[i][m] * sR[m][j];
              }
          }
        __syncthreads();
#pragma unroll
        for (int k = 0; k < 6; ++k)
          {
            const int e = tid + k * 1024;
            if (e < nb * n)
              sR[e / n][e - (e / n) * n] = rn[k];
          }
        __syncthreads();
        // one pass: every entry outside the pivot rows and columns; then the pivot rows, columns and block
        for (int e = tid; e < n * n; e += 1024)
          {
            const int i = e / n, j = e - i * n;
            const bool ip = i >= p0 && i < p0 + nb, jp = j >= p0 && j < p0 + nb;
            double v;
            if (ip && jp)
              v = sDi[i - p0][j - p0];
            else if (ip)
              v = sR[i - p0][j];
            else if (jp)
              {
                v = 0.0;
                for (int m = 0; m < nb; ++m)
                  v -= sC[i][m] * sDi[m][j - p0];
              }
            else
              {
                v = A[e];
                if (nb == GJB) // (unrolled: the sixteen pairs of LDS reads travel together)
                  {
#pragma unroll
                    for (int m = 0; m < GJB; ++m)
                      v -= sC[i][m] * sR[m][j];
                  }
                else
                  for (int m = 0; m < nb; ++m)
                    v -= sC[i][m] * sR[m][j];
              }
            A[e] = v;
          }
        __syncthreads();
      }
  }
  // x = Ainv b for n <= DENSE_BIG_MAX: 16 rows per workgroup, thread = (row, one of 16 chunks of the sum); the chunks are
  // added in a fixed order.  The inverse of a symmetric matrix is read by columns.
  __global__ __launch_bounds__(256) void dense_apply_big(const double *__restrict__ inv, const double *__restrict__ b, double *x, int n)
  {
    __shared__ double s_b[DENSE_BIG_MAX];
    __shared__ double s_p[16][17];
    for (int j = threadIdx.x; j < n; j += 256)
      s_b[j] = b[j];
    __syncthreads();
    const int c = threadIdx.x & 15, ch = threadIdx.x >> 4, i = blockIdx.x * 16 + c;
    const int per = (n + 15) / 16, j0 = ch * per, j1 = min(n, j0 + per);
    double    s = 0.0;
    if (i < n)
      for (int j = j0; j < j1; ++j)
        s += inv[j * n + i] * s_b[j];
    s_p[ch][c] = s;
    __syncthreads();
    if (threadIdx.x < 16 && i < n)
      {
        double t = 0.0;
#pragma unroll
        for (int k = 0; k < 16; ++k)
          t += s_p[k][threadIdx.x];
        x[i] = t;
      }
  }

  // ------------------------------------------------------------------ banded Cholesky (direct solver for small systems)
  // "Solver type = Direct" (the reference's shipped default, parameters.prm:43: SparseDirectUMFPACK re-factorised in every
  // Newton iteration, nonlinear_elasticity.cc:1192-1200) for the sizes the reference's own geometries have: the tangent
  // is symmetric positive definite (constrained rows carry their diagonal only), so K = L L^T in a band.  The nodes are
  // renumbered with the SHORTEST lattice direction running fastest (HostMesh::band_perm): the FSI3 flap with 18 x 3 Q3
  // cells has a half bandwidth of 67 dofs instead of 337.  Storage: lower band, column c at band[c*(hbw+1) + (r - c)],
  // c <= r <= c + hbw.  band_extract fills it from the assembled rows; band_cholesky_solve is ONE workgroup of 1024
  // threads that factorises in block columns of 16 (diagonal block by one wavefront in LDS, panel = one triangular
  // solve per row, trailing update from the panel in LDS) and then substitutes forward and backward.  Latency bound by
  // construction (three barriers per block column): 0.3-1 ms for 1-3 k dofs against 9 ms for the 2,300 Jacobi-PCG
  // iterations the same solve took at 1e-12.
  template <int D>
  __global__ __launch_bounds__(256) void band_extract(SellParams prm, const int32_t *__restrict__ bperm, double *band, int hbw)
  {
    constexpr int DD = D * D;
    const int     sl = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (sl >= prm.nslices)
      return;
    const int node = prm.perm[int64_t(sl) * 64 + lane];
    if (node < 0)
      return;
    const int     len = prm.len[sl], wx = prm.wx[sl], ld = hbw + 1;
    const int64_t off = prm.off[sl];
    const int     pr  = bperm[node];
    // (eight entries of the row at a time: their column ids, then the columns' band positions, then the values are requested
    // together -- one entry after the other was 50 x three dependent round trips to the L2 for a 2D Q3 row: 34 us)
    constexpr int NU = 8;
    for (int k0 = 0; k0 < len; k0 += NU)
      {
        int cn[NU], pc[NU];
#pragma unroll
        for (int u = 0; u < NU; ++u)
          cn[u] = k0 + u < len ? prm.col[(off + k0 + u) * 64 + lane] : node;
#pragma unroll
        for (int u = 0; u < NU; ++u)
          pc[u] = bperm[cn[u]];
#pragma unroll
        for (int u = 0; u < NU; ++u)
          {
            const int k = k0 + u;
            if (k < len)
              {
                const double *v = prm.vals + (off * 64 + int64_t(k / wx) * (64 * wx) + lane * wx + k % wx) * DD;
                double        e[DD];
#pragma unroll
                for (int q = 0; q < DD; ++q)
                  e[q] = v[q];
#pragma unroll
                for (int i = 0; i < D; ++i)
#pragma unroll
                  for (int j = 0; j < D; ++j)
                    {
                      const int r = pr * D + i, c = pc[u] * D + j;
                      if (r >= c)
                        band[int64_t(c) * ld + (r - c)] = e[i * D + j];
                    }
              }
          }
      }
  }

  // f(integral_constant<int, I>) for I = I0 .. N - 1 (loop bodies that need the index as a constant expression)
  template <int I, int N, typename F>
  __device__ __forceinline__ void static_for(F &&f)
  {
    if constexpr (I < N)
      {
        f(std::integral_constant<int, I>{});
        static_for<I + 1, N>(f);
      }
  }
  // value of lane `lane` (uniform) in every lane: two v_readlane_b32 into scalar registers -- a few cycles, where __shfl goes
  // through the LDS crossbar (ds_bpermute) and costs a hundred
  __device__ __forceinline__ double lane_value(double v, int lane)
  {
    const int lo = __builtin_amdgcn_readlane(__double2loint(v), lane), hi = __builtin_amdgcn_readlane(__double2hiint(v), lane);
    return __hiloint2double(hi, lo);
  }
  // 1 / sqrt(d): the hardware estimate (26 bits) and two Newton steps (sqrt and a division in double precision are two long
  // dependent instruction chains, on the critical path of every pivot of a factorisation)
  __device__ __forceinline__ double rsqrt_nr(double d)
  {
    double rd = __builtin_amdgcn_rsq(d);
#pragma unroll
    for (int nr = 0; nr < 2; ++nr)
      {
        const double e = fma(-d * rd, rd, 1.0);
        rd             = fma(0.5 * rd, e, rd);
      }
    return rd;
  }
  // x = K^-1 b with K in `band` (overwritten by its factor).  b, x: vectors in the library's dof order (dof = D node + c);
  // bperm: node -> band position; work: n doubles.  flag[0] = 1 when a pivot is not positive (K not positive definite).
  // factor_only / solve_only split the two halves (the linear model factorises its constant matrix once).
  template <int D>
  __global__ __launch_bounds__(1024) void band_cholesky_solve(double *band, int n, int hbw, const int32_t *__restrict__ bperm,
                                                             int nnodes, const double *__restrict__ b, double *x, double *work,
                                                             int32_t *flag, int do_factor, int do_solve)
  {
    constexpr int NB = BAND_NB;
    __shared__ double sL[NB][NB + 1];
    __shared__ double sP[BAND_MAXH][NB + 1];
    __shared__ double sy[NB];
    const int tid = threadIdx.x, ld = hbw + 1;
    auto      A   = [&](int r, int c) -> double & { return band[int64_t(c) * ld + (r - c)]; }; // c <= r <= c + hbw
    if (do_factor)
      for (int j0 = 0; j0 < n; j0 += NB)
        {
          const int nb = min(NB, n - j0);
          // (a) diagonal block -> LDS, factorised by the lanes of wave 0 (lane = row)
          if (tid < NB * NB)
            {
              const int r = tid / NB, c = tid % NB;
              sL[r][c]    = (r < nb && c <= r && r - c <= hbw) ? A(j0 + r, j0 + c) : 0.0; // outside the band: zero
            }
          __syncthreads();
          if (tid < 64)
            {
              // lane r holds row r of the block in registers; the entry of another row comes by a lane read (the loops are
              // fully unrolled: static register indices, 136 lane reads instead of 16 x 15 dependent LDS round trips)
              double row[NB];
#pragma unroll
              for (int c = 0; c < NB; ++c)
                row[c] = sL[tid < NB ? tid : 0][c];
#pragma unroll
              for (int c = 0; c < NB; ++c)
                {
                  const double d = __shfl(row[c], c, 64);
                  if (c < nb && !(d > 0.0) && tid == 0)
                    flag[0] = 1;
                  const double rd = (c < nb) ? 1.0 / sqrt(d) : 1.0;
                  if (tid >= c)
                    row[c] *= rd; // column c of L (the diagonal becomes sqrt(d))
#pragma unroll
                  for (int c2 = c + 1; c2 < NB; ++c2)
                    {
                      const double l2 = __shfl(row[c], c2, 64); // L[c2][c]
                      if (tid >= c2)
                        row[c2] -= row[c] * l2;
                    }
                }
              if (tid < NB)
#pragma unroll
                for (int c = 0; c < NB; ++c)
                  sL[tid][c] = (tid < nb && c <= tid) ? row[c] : (tid == c ? 1.0 : 0.0);
            }
          __syncthreads();
          if (tid < NB * NB)
            {
              const int r = tid / NB, c = tid % NB;
              if (r < nb && c <= r && r - c <= hbw)
                A(j0 + r, j0 + c) = sL[r][c];
            }
          // (b) panel: the rows below the block that reach into its columns, one triangular solve per row
          const int r0 = j0 + nb, m = min(n, r0 + hbw) - r0;
          if (tid < m)
            {
              const int r = r0 + tid;
              double    xr[NB];
#pragma unroll
              for (int c = 0; c < NB; ++c)
                {
                  double a = (c < nb && r - (j0 + c) <= hbw) ? A(r, j0 + c) : 0.0;
                  for (int q = 0; q < c; ++q)
                    a -= xr[q] * sL[c][q];
                  xr[c] = c < nb ? a / sL[c][c] : 0.0;
                }
#pragma unroll
              for (int c = 0; c < NB; ++c)
                {
                  if (c < nb && r - (j0 + c) <= hbw)
                    A(r, j0 + c) = xr[c];
                  sP[tid][c] = xr[c];
                }
            }
          __syncthreads();
          // (c) trailing update of the window below / right of the block: A[s][t] -= sum_c P[s][c] P[t][c], t <= s
          // s fastest: consecutive threads touch consecutive entries of a band column; eight entries per thread in
          // flight (the old values of a batch are requested before any of it is stored).  (4 x 4 register tiles would
          // quarter the LDS reads but need 128+ VGPRs at 1024 threads per workgroup: measured 2.5-4x slower, spilled.)
          for (int p0 = tid; p0 < m * m; p0 += 1024 * 8)
            {
              double  upd[8];
              double *ptr[8];
#pragma unroll
              for (int u = 0; u < 8; ++u)
                {
                  const int p = p0 + u * 1024, t = p / m, s = p - t * m;
                  ptr[u]      = (p < m * m && t <= s) ? &A(r0 + s, r0 + t) : nullptr;
                  double acc  = 0.0;
                  if (ptr[u])
                    {
#pragma unroll
                      for (int c = 0; c < NB; ++c)
                        acc += sP[s][c] * sP[t][c];
                      acc = *ptr[u] - acc;
                    }
                  upd[u] = acc;
                }
#pragma unroll
              for (int u = 0; u < 8; ++u)
                if (ptr[u])
                  *ptr[u] = upd[u];
            }
          __syncthreads();
        }
    if (!do_solve)
      return;
    // ---- right-hand side into band order
    for (int i = tid; i < n; i += 1024)
      {
        const int node = i / D, c = i - node * D;
        work[bperm[node] * D + c] = b[i];
      }
    __syncthreads();
    // ---- forward substitution L y = b, block columns of NB
    for (int j0 = 0; j0 < n; j0 += NB)
      {
        const int nb = min(NB, n - j0);
        if (tid < NB * NB) // the diagonal block of L in LDS: the serial part below runs on LDS latency, not on L2 latency
          {
            const int r = tid / NB, c = tid % NB;
            sL[r][c]    = (r < nb && c <= r && r - c <= hbw) ? A(j0 + r, j0 + c) : 0.0; // outside the band: zero
          }
        __syncthreads();
        if (tid < 64) // lane r = row r of the block: column sweeps, the finished y_c broadcast from lane c
          {
            double       w  = tid < nb ? work[j0 + tid] : 0.0;
            const double ri = tid < nb ? 1.0 / sL[tid][tid] : 1.0; // all reciprocals at once, off the sweep's chain
            for (int c = 0; c < nb; ++c)
              {
                const double yc = lane_value(w, c) * lane_value(ri, c); // (scalar lane reads: round 4)
                if (tid == c)
                  w = yc;
                else if (tid > c && tid < nb)
                  w -= sL[tid][c] * yc;
              }
            if (tid < nb)
              {
                sy[tid]        = w;
                work[j0 + tid] = w;
              }
          }
        __syncthreads();
        const int r0 = j0 + nb, m = min(n, r0 + hbw) - r0;
        if (tid < m)
          {
            const int r = r0 + tid;
            double    v = work[r];
            for (int c = 0; c < nb; ++c)
              if (r - (j0 + c) <= hbw)
                v -= A(r, j0 + c) * sy[c];
            work[r] = v;
          }
        __syncthreads();
      }
    // ---- backward substitution L^T x = y
    for (int j0 = ((n - 1) / NB) * NB; j0 >= 0; j0 -= NB)
      {
        const int nb = min(NB, n - j0), r0 = j0 + nb, m = min(n, r0 + hbw) - r0;
        // contributions of the rows below the block: s_c = sum_r L[r][c] x[r]; 64 threads per column, then a wave sum
        {
          const int c = tid >> 6, l = tid & 63;
          double    s = 0.0;
          if (c < nb)
            for (int t = l; t < m; t += 64)
              if (r0 + t - (j0 + c) <= hbw)
                s += A(r0 + t, j0 + c) * work[r0 + t];
          s = wave_sum(s);
          if (l == 0 && c < NB)
            sy[c] = s;
          if (tid < NB * NB)
            {
              const int r = tid / NB, cc = tid % NB;
              sL[r][cc]   = (r < nb && cc <= r && r - cc <= hbw) ? A(j0 + r, j0 + cc) : 0.0;
            }
        }
        __syncthreads();
        if (tid < 64) // lane c = column c of the block, swept from the last row upwards
          {
            double       w  = tid < nb ? work[j0 + tid] - sy[tid] : 0.0;
            const double ri = tid < nb ? 1.0 / sL[tid][tid] : 1.0;
            for (int q = nb - 1; q >= 0; --q)
              {
                const double xq = lane_value(w, q) * lane_value(ri, q);
                if (tid == q)
                  w = xq;
                else if (tid < q)
                  w -= sL[q][tid] * xq;
              }
            if (tid < nb)
              work[j0 + tid] = w;
          }
        __syncthreads();
      }
    for (int i = tid; i < n; i += 1024)
      {
        const int node = i / D, c = i - node * D;
        x[i]           = work[bperm[node] * D + c];
      }
  }

  // ------------------------------------------------------------------ backward substitution of the LDS-window kernels
  // L^T x = y for the factor L in `band` (memory); y in `work` (from_work) or already in xs; x to the library's dof order.
  // xl: x lives in LDS (xs, at least n doubles), otherwise in `work`.  All 1024 threads of the one workgroup.
  constexpr int BAND_BACK_NA = 4; // rows below a block column per lane of a column's wave: half bandwidths up to 256
  template <int D>
  __device__ __forceinline__ void band_backward_lds(const double *__restrict__ band, int n, int hbw, bool xl, double *xs, double *work,
                                                    double (*sL)[BAND_NB + 1], double *sy, const int32_t *__restrict__ bperm,
                                                    double *x, bool from_work = true)
  {
    // ---- backward substitution L^T x = y, L from memory.  A block column is: s_c = sum_r L[r][c] x[r] over the rows below
    // the block (wave c = column c, a wave sum), then the block's own 16 unknowns swept by wave 0.  Nothing in it waits for
    // the L2 (band_cholesky_solve pays two round trips per block column): x lives in LDS (the window is free now; systems
    // beyond its 16.5 k entries keep x in memory), and the entries of L that a block column needs -- up to three per
    // thread for the sums, one per thread of the diagonal block -- are requested one block column ahead.
    constexpr int NB = BAND_NB;
    const int     tid = threadIdx.x, ld = hbw + 1;
    auto          A  = [&](int r, int c) -> const double & { return band[int64_t(c) * ld + (r - c)]; };
    if (xl && from_work)
      for (int i = tid; i < n; i += 1024)
        xs[i] = work[i];
    constexpr int NA = BAND_BACK_NA; // rows below a block, per lane of a column's wave
    double        acur[NA], dcur = 0.0;
    auto          request = [&](int j0, double *a, double &dd) {
      const int nb = min(NB, n - j0), r0 = j0 + nb, m = min(n, r0 + hbw) - r0, c = tid >> 6, l = tid & 63;
#pragma unroll
      for (int u = 0; u < NA; ++u)
        {
          const int t = l + 64 * u;
          a[u]        = (j0 >= 0 && c < nb && t < m && r0 + t - (j0 + c) <= hbw) ? A(r0 + t, j0 + c) : 0.0;
        }
      const int r = tid / NB, cc = tid % NB;
      dd          = (j0 >= 0 && tid < NB * NB && r < nb && cc <= r && r - cc <= hbw) ? A(j0 + r, j0 + cc) : 0.0;
    };
    const int jlast = ((n - 1) / NB) * NB;
    request(jlast, acur, dcur);
    __syncthreads();
    for (int j0 = jlast; j0 >= 0; j0 -= NB)
      {
        const int nb = min(NB, n - j0), r0 = j0 + nb, m = min(n, r0 + hbw) - r0;
        double    anext[NA], dnext;
        request(j0 - NB, anext, dnext); // (below the first block column: zeros, nothing is loaded)
        {
          const int c = tid >> 6, l = tid & 63;
          double    sacc = 0.0;
#pragma unroll
          for (int u = 0; u < NA; ++u)
            {
              const int t = l + 64 * u;
              if (t < m)
                sacc += acur[u] * (xl ? xs[r0 + t] : work[r0 + t]);
            }
          sacc = wave_sum_lane63(sacc);
          if (l == 63)
            sy[c] = sacc;
          if (tid < NB * NB)
            sL[tid / NB][tid % NB] = dcur; // (outside the band or the block: zero)
        }
        if (xl)
          asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); // (the requests for the next block stay in flight)
        else
          __syncthreads();
        if (tid < 64)
          {
            // lane t = column t of the block, swept from the last row upwards: x_q = w_q / L_qq goes into lane q of a register
            // of its own (v_writelane), every lane takes w -= L[q][t] x_q (zero above the diagonal: the lanes of finished
            // unknowns keep what nobody reads)
            const int tt = tid < NB ? tid : 0;
            double    col[NB];
#pragma unroll
            for (int q = 0; q < NB; ++q)
              col[q] = sL[q][tt];
            const double dg = sL[tt][tt];
            double       w  = tid < nb ? (xl ? xs[j0 + tid] : work[j0 + tid]) - sy[tid] : 0.0;
            const double ri = tid < nb ? 1.0 / dg : 1.0;
            int          xflo = 0, xfhi = 0;
            static_for<0, NB>([&](auto qq) {
              constexpr int q   = NB - 1 - decltype(qq)::value;
              const double  ws  = w * ri;
              const int     xlo = __builtin_amdgcn_readlane(__double2loint(ws), q), xhi = __builtin_amdgcn_readlane(__double2hiint(ws), q);
              asm("v_writelane_b32 %0, %1, %2" : "+v"(xflo) : "s"(xlo), "n"(q));
              asm("v_writelane_b32 %0, %1, %2" : "+v"(xfhi) : "s"(xhi), "n"(q));
              w = fma(-col[q], __hiloint2double(xhi, xlo), w);
            });
            if (tid < nb)
              {
                if (xl)
                  xs[j0 + tid] = __hiloint2double(xfhi, xflo);
                else
                  work[j0 + tid] = __hiloint2double(xfhi, xflo);
              }
          }
        if (xl)
          asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
        else
          __syncthreads();
#pragma unroll
        for (int u = 0; u < NA; ++u)
          acur[u] = anext[u];
        dcur = dnext;
      }
    for (int i = tid; i < n; i += 1024)
      {
        const int node = i / D, c = i - node * D, bi = bperm[node] * D + c;
        x[i]           = xl ? xs[bi] : work[bi];
      }
  }

  // ------------------------------------------------------------------ banded Cholesky, LDS-window form (rounds 4-5)
  // The same factorisation K = L L^T + substitutions for half bandwidths hbw <= BAND_LDS_W - BAND_NB + BAND_FAR_MAX.  The
  // ACTIVE part of the matrix -- the BAND_LDS_W rows below the current block column -- lives in LDS as a circular window
  // (entry (r, c) at [(r mod W) * (W + 1) + (c mod W)]); a block column never waits for the L2: the finished columns go
  // out to the band in memory and the NB rows that enter the window next come in (requested at the top of an iteration)
  // beside the arithmetic.  band_cholesky_solve above pays three dependent round trips to the L2 per block column.
  //
  // One block column (two workgroup barriers):
  //   T1  the first tile column of the trailing update -- every panel row block against the block column's first 16 rows,
  //       i.e. everything the NEXT block column's diagonal block and panel are made of; meanwhile the finished columns
  //       are read into registers and the entering rows take the slots of the block's own rows
  //   T2  waves 0-3: the next block column, diagonal block AND panel in one instruction stream (factor_block below);
  //       waves 4-15: the other tiles of the trailing update; everybody: the finished columns out to memory
  // factor_block: lane r < 16 of a wave holds row r of the diagonal block, lanes 16-63 hold 48 panel rows, all as 16
  // registers; the right-looking elimination of the diagonal block (pivot, scale, 15 multiply-adds with the pivot
  // column's entries read from the lanes of the diagonal rows) IS the panel rows' triangular solve when the same
  // instructions run on their lanes, and with the right-hand side as a 17th column the forward substitution as well.
  // Each of the four waves repeats the diagonal rows (no exchange between them), so a block column's panel costs no phase
  // of its own: the dependent chain per block column is one tile + one 16-pivot elimination.
  // The trailing update runs on the matrix cores (v_mfma_f64_16x16x4_f64, update_tile): a multiply-add loop reads
  // 2 x 16 LDS operands per entry and was LDS-bandwidth bound.
  //
  // FAR (half bandwidths 113-160; the reference's 3D plate at degree 2 has 152): a block column's panel reaches nfar =
  // hbw + NB - W rows BEYOND the window.  Their entries stay in the band in memory: far panel rows are lanes of
  // factor_block like the others (finished entries stored to memory, the rows kept in sF for the iteration's tiles); tiles
  // of the 16 entering rows are tiles of the window; tiles of the rows beyond those are read-modify-writes of the band in
  // memory, two tiles per wave at a time (update_mem_pair) -- except their first tile column, whose results are what the
  // next block column's far lanes start from: it goes to LDS (sX) and is never stored unfinished, and the far rows'
  // right-hand side lives in LDS too (yF), so factor_block's chain neither loads from nor waits for memory.  The window
  // itself never holds an entry with r - c >= W: when a row enters, its columns further left are finished.
  constexpr int BAND_LDS_W = 128, BAND_FAR_MAX = 48;
  template <int D, bool FAR = false>
  __global__ __launch_bounds__(1024) void band_cholesky_lds(double *band, int n, int hbw, const int32_t *__restrict__ bperm,
                                                            int nnodes, const double *__restrict__ b, double *x, double *work,
                                                            int32_t *flag, int do_solve, unsigned long long *dbg)
  {
    constexpr int NB = BAND_NB, W = BAND_LDS_W, LD = W + 1;
    unsigned long long tacc[3] = {0, 0, 0}, tlast = 0; // diagnostic (dbg != null): clocks per phase, thread 0
#define BAND_STAMP(i_)                                             \
  do                                                               \
    {                                                              \
      if (dbg && tid == 0)                                         \
        {                                                          \
          const unsigned long long now_ = __builtin_amdgcn_s_memtime(); \
          tacc[i_] += now_ - tlast;                                \
          tlast = now_;                                            \
        }                                                          \
    }                                                              \
  while (0)
    static_assert(W % NB == 0, "a block column must not wrap inside the window");
    __shared__ double S[W * LD];      // slot row (r mod W), slot column (c mod W); slots outside a row's band hold ZERO
    __shared__ double yv[W];          // right-hand side / y of the rows in the window (circular)
    __shared__ double sL[NB][NB + 1]; // the current diagonal block of L (for the stream of finished columns)
    __shared__ double sy[NB];         // (backward substitution)
    __shared__ double sF[FAR ? 2 : 1][FAR ? BAND_FAR_MAX : 1][NB + 1]; // FAR: panel rows beyond the window (row rF0 + f);
                                                                       // two copies: one read by tiles, one being made
    // FAR: what the NEXT block column's far panel rows start from -- their entries in its 16 columns, as the first tile column
    // of the trailing update leaves them (sX: rows beyond the window never wait for the L2 inside factor_block) -- and the
    // right-hand side of the rows beyond the window (yF, circular over 64 rows; a row's y goes to memory when it is final)
    __shared__ double sX[FAR ? BAND_FAR_MAX : 1][NB + 1];
    __shared__ double yF[FAR ? 64 : 1];
    const int tid = threadIdx.x, ld = hbw + 1;
    const unsigned long long tk0 = (dbg && tid == 0) ? __builtin_amdgcn_s_memtime() : 0;
    // ---- right-hand side into band order (memory)
    if (do_solve)
      for (int i = tid; i < n; i += 1024)
        {
          const int node = i / D, c = i - node * D;
          work[bperm[node] * D + c] = b[i];
        }
    // ---- the first window: rows [0, W); every slot gets its band entry or zero (slot cs of row r holds column
    // c = r - ((r - cs) mod W), the one column of (r - W, r] with that residue)
    for (int idx = tid; idx < W * W; idx += 1024)
      {
        const int r = idx >> 7, cs = idx & (W - 1), c = r - ((r - cs) & (W - 1)), k = r - c;
        S[r * LD + cs] = (r < n && c >= 0 && k <= hbw) ? band[int64_t(c) * ld + k] : 0.0;
      }
    __syncthreads();
    if (do_solve && tid < W)
      yv[tid] = tid < n ? work[tid] : 0.0;
    if constexpr (FAR)
      {
        if (tid < BAND_FAR_MAX * NB) // block column 0's far rows W + f, columns 0 .. 15
          {
            const int f = tid >> 4, c = tid & (NB - 1), r = W + f;
            sX[f][c]    = (r < n && r - c <= hbw) ? band[int64_t(c) * ld + (r - c)] : 0.0;
          }
        if (tid >= 960)
          yF[tid - 960] = (do_solve && tid - 960 < BAND_FAR_MAX && W + tid - 960 < n) ? work[W + tid - 960] : 0.0; // rows W .. W + 47
      }
    __syncthreads();
    // Block column jb, diagonal block and panel (waves 0-3; see the head comment).  Straight-line code on purpose (no lane
    // or block-size conditions in the elimination): entries above the diagonal are computed and never used, a short last
    // block is padded with the identity -- so the scheduler can fill the latency of one pivot's chain (lane read,
    // 1 / sqrt, two Newton steps) with the updates the previous pivot left behind.  fb: which copy of sF to fill.
    auto factor_block = [&](int jb, int fb) {
      const int nbb = min(NB, n - jb), jcb = jb & (W - 1), rb = jb + nbb;
      const int mallb = min(n, rb + hbw) - rb, mb = FAR ? min(mallb, W - nbb) : mallb;
      int       ln = tid & 63;
      if constexpr (FAR) // (registers are short there: keeps lane constants from being hoisted out of the loop and spilled)
        asm volatile("" : "+v"(ln));
      const bool isdiag = ln < NB;
      const int  q = (tid >> 6) * (64 - NB) + ln - NB; // panel row rb + q
      const int  r = isdiag ? jb + ln : rb + q;
      const bool inwin = !isdiag && q < mb, isfar = FAR && !isdiag && q >= mb && q < mallb;
      double     row[NB], y = 0.0;
      double    *rp = &S[(r & (W - 1)) * LD + jcb];
      double    *pb = band + (int64_t(jb) * ld + (r - jb)); // far rows: entry (r, jb + c) at pb[c (ld - 1)]
      if (isdiag)
        {
#pragma unroll
          for (int c = 0; c < NB; ++c)
            row[c] = (ln < nbb && c <= ln && c < nbb) ? rp[c] : (ln == c ? 1.0 : 0.0);
          if (do_solve && ln < nbb)
            y = yv[r & (W - 1)];
        }
      else if (inwin)
        {
#pragma unroll
          for (int c = 0; c < NB; ++c)
            row[c] = rp[c];
          if (do_solve)
            y = yv[r & (W - 1)];
        }
      else if (isfar)
        {
          if constexpr (FAR)
            {
#pragma unroll
              for (int c = 0; c < NB; ++c)
                row[c] = sX[q - mb][c]; // (zero outside the band)
              if (do_solve)
                y = yF[r & 63];
            }
        }
      else
        {
#pragma unroll
          for (int c = 0; c < NB; ++c)
            row[c] = 0.0;
        }
      bool bad  = false;
      int  yflo = 0, yfhi = 0; // the finished y of the diagonal rows, lane c = row c
      static_for<0, NB>([&](auto cc) {
          constexpr int c = decltype(cc)::value; // (a constant: v_writelane takes its lane as an immediate)
          const double d  = lane_value(row[c], c);
          bad             = bad || !(d > 0.0);
          const double rd = rsqrt_nr(d);
          row[c] *= rd; // column c of L (the diagonal becomes sqrt(d))
          // - L[r][c] L[c2][c]; the lane reads four ahead of their multiply-adds: a scalar register written by a lane read
          // is not ready for the next instruction, and one pair after the other costs 21 clocks per entry against 15 this
          // way (tools/probe/fp64_issue.hip)
#pragma unroll
          for (int c2 = c + 1; c2 < NB; c2 += 4)
            {
              double l4[4];
#pragma unroll
              for (int i = 0; i < 4; ++i)
                if (c2 + i < NB)
                  l4[i] = lane_value(row[c], c2 + i);
              __builtin_amdgcn_sched_barrier(0);
#pragma unroll
              for (int i = 0; i < 4; ++i)
                if (c2 + i < NB)
                  row[c2 + i] = fma(-row[c], l4[i], row[c2 + i]);
              __builtin_amdgcn_sched_barrier(0);
            }
          // y_c = y_c / L_cc is final: written into lane c of a register of its own (v_writelane: no lane masks), and the
          // rows below take it -- the multiply-add runs on every lane, like the ones above; what it leaves on the lanes of
          // finished diagonal rows is not used
          const double ys  = y * rd;
          const int    ylo = __builtin_amdgcn_readlane(__double2loint(ys), c), yhi = __builtin_amdgcn_readlane(__double2hiint(ys), c);
          asm("v_writelane_b32 %0, %1, %2" : "+v"(yflo) : "s"(ylo), "n"(c));
          asm("v_writelane_b32 %0, %1, %2" : "+v"(yfhi) : "s"(yhi), "n"(c));
          y                = fma(-row[c], __hiloint2double(yhi, ylo), y);
      });
      if (isdiag)
        y = __hiloint2double(yfhi, yflo);
      if (bad && tid == 0)
        flag[0] = 1;
      if (isdiag)
        {
          if (tid < NB) // (wave 0 keeps the diagonal block; the other waves only needed it)
            {
#pragma unroll
              for (int c = 0; c < NB; ++c)
                sL[tid][c] = (tid < nbb && c <= tid) ? row[c] : (tid == c ? 1.0 : 0.0);
              if (do_solve && tid < nbb)
                work[jb + tid] = y;
            }
        }
      else if (inwin)
        {
#pragma unroll
          for (int c = 0; c < NB; ++c)
            rp[c] = row[c];
          if (do_solve)
            yv[r & (W - 1)] = y;
        }
      else if constexpr (FAR)
        {
          const int f = q - mb; // (mb = W - NB wherever a far row exists)
          if (isfar)
            {
#pragma unroll
              for (int c = 0; c < NB; ++c)
                {
                  sF[fb][f][c] = row[c];
                  if (r - jb - c <= hbw)
                    pb[int64_t(c) * (ld - 1)] = row[c];
                }
              if (do_solve)
                yF[r & 63] = y;
            }
          else if (mallb > mb && f < BAND_FAR_MAX) // rows of sF past the panel: zero rows (whole tiles are multiplied)
            {
#pragma unroll
              for (int c = 0; c < NB; ++c)
                sF[fb][f][c] = 0.0;
            }
        }
    };
    // One 16 x 16 tile of the trailing update: rows s0 + i of row block I against rows t0 + j of row block J <= I (blocks
    // of 16 rows counted from r0), D[i][j] = sum_k P[s0+i][k] P[t0+j][k] with the panel P in the window (S, the block's
    // columns) or -- FAR -- in sF.  Operands: lane l holds A[i = l & 15][k = l >> 4] and B[k = l >> 4][j = l & 15], the
    // result D[(l >> 4) + 4 v][l & 15] in register v.  Panel rows past the band's edge or the matrix's end are zero rows, so
    // whole tiles are safe.  For tiles whose rows are in the window (the entering rows included).
    // init: the tile's entries come from there (D layout) instead of from S -- a tile of rows that are not stored yet.
    typedef double v4f64 __attribute__((ext_vector_type(4)));
    auto update_tile = [&](int I, int J, int jc, int r0, int rF0, int fb, const double *init) {
      const int           lane = tid & 63, li = lane & 15, lk = lane >> 4;
      const int           s0 = r0 + NB * I, t0 = r0 + NB * J, rs = s0 + li, rt = t0 + li;
      const double *const ps = (FAR && rs >= rF0) ? &sF[fb][rs - rF0][0] : &S[(rs & (W - 1)) * LD + jc];
      const double *const pt = (FAR && rt >= rF0) ? &sF[fb][rt - rF0][0] : &S[(rt & (W - 1)) * LD + jc];
      double              a[4], b[4];
#pragma unroll
      for (int kk = 0; kk < 4; ++kk)
        {
          a[kk] = ps[4 * kk + lk];
          b[kk] = pt[4 * kk + lk];
        }
      v4f64 d = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
      for (int kk = 0; kk < 4; ++kk)
        d = __builtin_amdgcn_mfma_f64_16x16x4f64(a[kk], b[kk], d, 0, 0, 0);
      // (all four reads, then the four writes; an entry above the diagonal of a diagonal tile is the dead slot of a
      // finished column: it is written back unchanged)
      double *o[4], cur[4];
#pragma unroll
      for (int v = 0; v < 4; ++v)
        {
          const int sr = s0 + lk + 4 * v, tc = t0 + li;
          o[v]         = &S[(sr & (W - 1)) * LD + (tc & (W - 1))];
          cur[v]       = init ? init[v] : *o[v];
        }
#pragma unroll
      for (int v = 0; v < 4; ++v)
        *o[v] = cur[v] - ((I != J || t0 + li <= s0 + lk + 4 * v) ? d[v] : 0.0);
    };
    // FAR: tiles whose rows lie beyond the entering ones live in the band in memory.  Transposed (D'[column t0 + lk + 4 v]
    // [row s0 + li]: lanes walk a column's contiguous rows), and TWO tiles at a time (J1 < 0: one): both tiles' old values are
    // requested before either is stored -- one round trip to the L2 and one drain of the stores per pair (a wave's memory
    // operations retire in order, so tile after tile would pay both per tile)
    auto update_mem_pair = [&](int I0, int J0, int I1, int J1, int jc, int r0, int rF0, int nfar, int fb) {
      const int  lane = tid & 63, li = lane & 15, lk = lane >> 4;
      const bool two = J1 >= 0;
      const int  sA = r0 + NB * I0, tA = r0 + NB * J0, sB = two ? r0 + NB * I1 : sA, tB = two ? r0 + NB * J1 : tA;
      double     oldA[4], oldB[4];
#pragma unroll
      for (int v = 0; v < 4; ++v)
        {
          const int tc = tA + lk + 4 * v, sr = sA + li;
          oldA[v]      = (sr - rF0 < nfar && tc <= sr) ? band[int64_t(tc) * ld + (sr - tc)] : 0.0;
        }
#pragma unroll
      for (int v = 0; v < 4; ++v)
        {
          const int tc = tB + lk + 4 * v, sr = sB + li;
          oldB[v]      = (two && sr - rF0 < nfar && tc <= sr) ? band[int64_t(tc) * ld + (sr - tc)] : 0.0;
        }
      auto product = [&](int s0, int t0) {
        const int           rs = s0 + li, rt = t0 + li;
        const double *const ps = &sF[fb][rs - rF0][0]; // (rows beyond the window)
        const double *const pt = rt >= rF0 ? &sF[fb][rt - rF0][0] : &S[(rt & (W - 1)) * LD + jc];
        v4f64               d  = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
        for (int kk = 0; kk < 4; ++kk)
          d = __builtin_amdgcn_mfma_f64_16x16x4f64(pt[4 * kk + lk], ps[4 * kk + lk], d, 0, 0, 0);
        return d;
      };
      const v4f64 dA = product(sA, tA), dB = product(sB, tB);
#pragma unroll
      for (int v = 0; v < 4; ++v)
        {
          const int tc = tA + lk + 4 * v, sr = sA + li;
          if (sr - rF0 < nfar && tc <= sr)
            band[int64_t(tc) * ld + (sr - tc)] = oldA[v] - dA[v];
        }
#pragma unroll
      for (int v = 0; v < 4; ++v)
        {
          const int tc = tB + lk + 4 * v, sr = sB + li;
          if (two && sr - rF0 < nfar && tc <= sr)
            band[int64_t(tc) * ld + (sr - tc)] = oldB[v] - dB[v];
        }
    };
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6); // this wave (a scalar: the tile loops branch on it)
    if (tid < 256)
      factor_block(0, 0);
    __syncthreads();
    if (dbg && tid == 0)
      {
        tlast  = __builtin_amdgcn_s_memtime();
        dbg[3] = tlast - tk0; // before the loop: right-hand side, first window, first block column
      }
    for (int j0 = 0; j0 < n; j0 += NB)
      {
        const int nb = min(NB, n - j0), jc = j0 & (W - 1), fb = (j0 / NB) & 1; // the block's columns: slots jc .. jc + NB - 1
        // the panel: mall rows below the block reach into its columns, m of them in the window, nfar beyond it (FAR)
        const int r0 = j0 + nb, mall = min(n, r0 + hbw) - r0, m = FAR ? min(mall, W - nb) : mall, nfar = mall - m, rF0 = j0 + W;
        // row blocks of 16 below the block column: nbw in the window, then (FAR) the entering rows and the rows beyond
        const int nbw = (m + NB - 1) / NB, nbf = FAR ? (nfar + NB - 1) / NB : 0, nbr = nbw + nbf;
        const int nwin = FAR ? min(nbr, W / NB) : nbr; // row blocks whose tiles are tiles of the window
        // the rows that enter the window when this block column is done: requested now (an iteration ahead is no faster: T1
        // is as long as its tile; and FAR could not -- the tiles of the iteration before still update those rows in memory)
        double pre[2], prey = 0.0;
#pragma unroll
        for (int u = 0; u < 2; ++u)
          {
            const int idx = tid + u * 1024, rr = idx >> 7, cs = idx & (W - 1), rn = j0 + W + rr;
            const int c = rn - ((rn - cs) & (W - 1)), k = rn - c;
            // (c >= 0: rn >= W.  FAR: a slot whose column belongs to this block column is dead -- the entry is finished
            // in memory by the far panel row's lane)
            pre[u] = (rn < n && k <= hbw && (!FAR || c >= r0)) ? band[int64_t(c) * ld + k] : 0.0;
          }
        if (!FAR && do_solve && tid < NB && j0 + W + tid < n)
          prey = work[j0 + W + tid];
        // FAR: the entering rows' entries in the next block column (their tile of the first tile column) come in through
        // the wave that updates them, in the tile's layout -- the other waves' stores of entering rows are not visible to it
        // before the barrier
        constexpr int WENTER = 4 + W / NB - 1; // that wave
        double        pre7[FAR ? 4 : 1];
        if constexpr (FAR)
          if (wv == WENTER)
            {
#pragma unroll
              for (int v = 0; v < 4; ++v)
                {
                  const int rn = rF0 + ((tid & 63) >> 4) + 4 * v, tc = r0 + (tid & 15);
                  pre7[v]      = (rn < n && rn - tc <= hbw) ? band[int64_t(tc) * ld + (rn - tc)] : 0.0;
                }
            }
        // ---- T1: the first tile column (one tile per wave, waves 4 ...), before anything that waits for the rows requested above
        if (wv >= 4)
          {
            const int I = wv - 4;
            if (I < nwin && I < W / NB - 1)
              update_tile(I, 0, jc, r0, rF0, fb, nullptr);
            if constexpr (FAR)
              {
                if (wv == WENTER)
                  {
                    if (nfar > 0)
                      update_tile(W / NB - 1, 0, jc, r0, rF0, fb, pre7);
                    else // (no far rows: the entries are stored as they came)
                      {
#pragma unroll
                        for (int v = 0; v < 4; ++v)
                          S[((rF0 + ((tid & 63) >> 4) + 4 * v) & (W - 1)) * LD + ((r0 + (tid & 15)) & (W - 1))] = pre7[v];
                      }
                  }
                // the next block column's far rows rF0 + 16 + f in its columns r0 .. r0 + 15, into sX: f < 32 are this block
                // column's rows beyond the entering ones -- the tiles (8, 0) and (9, 0) of its trailing update, taken from
                // memory, updated, and NOT stored back (factor_block finishes and stores them) --, f >= 32 come as they are
                if (I == W / NB || I == W / NB + 1)
                  {
                    const int lane = tid & 63, li = lane & 15, lk = lane >> 4;
                    double    old[2][4];
#pragma unroll
                    for (int h = 0; h < 2; ++h)
#pragma unroll
                      for (int v = 0; v < 4; ++v)
                        {
                          const int f = 32 * (I - W / NB) + 16 * h + li, rr = rF0 + NB + f, tc = r0 + lk + 4 * v;
                          old[h][v]   = (f < BAND_FAR_MAX && rr < n && rr - tc <= hbw) ? band[int64_t(tc) * ld + (rr - tc)] : 0.0;
                        }
#pragma unroll
                    for (int h = 0; h < 2; ++h)
                      {
                        const int f0 = 32 * (I - W / NB) + 16 * h; // rows rF0 + 16 + f0 ..: row block 8 + f0 / 16 of this block column
                        v4f64     d  = {0.0, 0.0, 0.0, 0.0};
                        if (f0 < 32 && nfar > NB + f0)
                          {
                            const double *const ps = &sF[fb][NB + f0 + li][0], *const pt = &S[((r0 + li) & (W - 1)) * LD + jc];
#pragma unroll
                            for (int kk = 0; kk < 4; ++kk)
                              d = __builtin_amdgcn_mfma_f64_16x16x4f64(pt[4 * kk + lk], ps[4 * kk + lk], d, 0, 0, 0);
                          }
                        if (f0 < BAND_FAR_MAX)
                          {
#pragma unroll
                            for (int v = 0; v < 4; ++v)
                              sX[f0 + li][lk + 4 * v] = old[h][v] - d[v];
                          }
                      }
                    if (I == W / NB + 1 && do_solve && lane < NB) // the right-hand side of the rows that become far rows now
                      yF[(rF0 + NB + 32 + lane) & 63] = rF0 + NB + 32 + lane < n ? work[rF0 + NB + 32 + lane] : 0.0;
                  }
              }
          }
        // the finished columns leave the window: into registers now, out to the band in memory as the iteration's LAST
        // memory operation -- the memory counter retires in order, and the new rows' loads must not queue behind these
        // stores.  (The block's own rows come from sL: their slots in S are dead since the block was factorised, which is
        // what lets the new rows take them right here.)
        double wb[2];
#pragma unroll
        for (int u = 0; u < 2; ++u)
          {
            // (entry k of finished column cc; rows of the window only: a FAR column's rows beyond it went out from their lanes)
            const int idx = tid + u * 1024, cc = idx >> 7, k = idx & (W - 1), r = j0 + cc + k;
            wb[u] = (cc < nb && k <= hbw && r < n && r < j0 + W) ? (r < r0 ? sL[cc + k][cc] : S[(r & (W - 1)) * LD + jc + cc]) : 0.0;
          }
        // the new rows take the slots of the block's rows (whole slot rows: band entries and zeros)
#pragma unroll
        for (int u = 0; u < 2; ++u)
          {
            const int idx = tid + u * 1024, rr = idx >> 7, cs = idx & (W - 1);
            if (!FAR || ((cs - r0) & (W - 1)) >= NB) // (FAR: the slots of the next block column: see pre7)
              S[((j0 + W + rr) & (W - 1)) * LD + cs] = pre[u];
          }
        if (do_solve && tid < NB)
          yv[(j0 + W + tid) & (W - 1)] = FAR ? yF[(j0 + W + tid) & 63] : prey;
        // (LDS only: nothing stored to memory in T1 is read in T2, and the finished columns' stores of the iteration before
        // need not be waited for)
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
        BAND_STAMP(0);
        // ---- T2
        if (wv < 4)
          {
            if (r0 < n)
              factor_block(r0, fb ^ 1);
            if (dbg && tid == 0)
              tacc[2] += __builtin_amdgcn_s_memtime() - tlast; // (of T2: until wave 0 is through)
          }
        else
          {
            // the tiles (I, J) with J >= 1, twelve waves.  FAR: those of the last nbr - 8 row blocks live in memory, at most
            // 17, in pairs on the first nine of these waves -- first thing, the window's tiles run behind their stores
            const int w12 = wv - 4;
            if constexpr (FAR)
              if (nbr > W / NB)
                {
                  const int nm0 = W / NB, nm = nm0 + (nbr > W / NB + 1 ? W / NB + 1 : 0); // tiles of row block 8 (J = 1 .. 8), of 9 (1 .. 9)
                  const int q0 = 2 * w12, q1 = q0 + 1;
                  if (q0 < nm)
                    {
                      const int I0 = q0 < nm0 ? W / NB : W / NB + 1, J0 = q0 < nm0 ? q0 + 1 : q0 - nm0 + 1;
                      const int I1 = q1 < nm0 ? W / NB : W / NB + 1, J1 = q1 < nm ? (q1 < nm0 ? q1 + 1 : q1 - nm0 + 1) : -1;
                      update_mem_pair(I0, J0, I1, J1, jc, r0, rF0, nfar, fb);
                    }
                }
            // window tiles: (I, J), 1 <= J <= I < nwin, in rows of I tiles
            int I = 1, J = 1 + w12;
            while (J > I)
              {
                J -= I;
                ++I;
              }
            while (I < nwin)
              {
                update_tile(I, J, jc, r0, rF0, fb, nullptr);
                J += 12;
                while (J > I)
                  {
                    J -= I;
                    ++I;
                  }
              }
          }
        // the finished columns go out
#pragma unroll
        for (int u = 0; u < 2; ++u)
          {
            const int idx = tid + u * 1024, cc = idx >> 7, k = idx & (W - 1), r = j0 + cc + k;
            if (cc < nb && k <= hbw && r < n && r < j0 + W)
              band[int64_t(j0 + cc) * ld + k] = wb[u];
          }
        // (the finished columns, the far rows' finished entries and y are read again after the loop only: the barrier orders
        // the LDS and those stores drain beside the next tile.  FAR: the memory tiles of this iteration ARE read by the next
        // one's requests, so the waves that stored them wait for their stores first)
        if constexpr (FAR)
          if (wv >= 4 && nbr > W / NB)
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
        BAND_STAMP(1);
      }
    if (dbg && tid == 0)
      for (int i = 0; i < 3; ++i)
        dbg[i] = tacc[i];
#undef BAND_STAMP
    if (!do_solve)
      return;
    __syncthreads();
    band_backward_lds<D>(band, n, hbw, n <= W * LD, S, work, sL, sy, bperm, x);
    if (dbg && tid == 0)
      dbg[4] = __builtin_amdgcn_s_memtime() - tlast; // the backward substitution
  }

  // ------------------------------------------------------------------ substitutions alone, x in LDS (round 5)
  // x = K^-1 b for a band that holds the factor already (the linear model factorises its constant matrix once and
  // substitutes every time step).  One workgroup, x in LDS from the first read of b to the last write of x, the entries of
  // L requested one block column ahead in both sweeps -- band_cholesky_solve's substitutions go through memory and pay five
  // dependent round trips to the L2 per block column.  Forward, per block column (two barriers): wave 0 sweeps the block's
  // 16 unknowns (lane t = row t of the diagonal block in registers, y_q into lane q by v_writelane); then waves 1-15 take
  // one 16-row tile of the panel each, y_panel -= P y_block on the matrix cores (operand A = the tile as requested,
  // B = y_block on every column).  Backward: band_backward_lds.  Systems of up to BAND_SOLVE_XMAX dofs, half bandwidths up to
  // 15 tiles.
  constexpr int BAND_SOLVE_XMAX = 18432, BAND_SOLVE_MAXH = 15 * BAND_NB;
  template <int D>
  __global__ __launch_bounds__(1024) void band_solve_lds(const double *__restrict__ band, int n, int hbw, const int32_t *__restrict__ bperm,
                                                         int nnodes, const double *__restrict__ b, double *x)
  {
    constexpr int NB = BAND_NB;
    __shared__ double xs[BAND_SOLVE_XMAX];
    __shared__ double sL[NB][NB + 1];
    __shared__ double sy[NB];
    const int tid = threadIdx.x, ld = hbw + 1, lane = tid & 63, li = lane & 15, lk = lane >> 4;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    auto      A  = [&](int r, int c) -> const double & { return band[int64_t(c) * ld + (r - c)]; };
    for (int i = tid; i < n; i += 1024)
      {
        const int node = i / D, c = i - node * D;
        xs[bperm[node] * D + c] = b[i];
      }
    // what block column j0 needs from memory: its diagonal block (one entry per thread of waves 0-3) and, waves 1-15, the
    // wave's tile of the panel in the layout of the matrix cores' A operand (lane: row s0 + li, columns j0 + 4 kk + lk)
    auto request = [&](int j0, double &dd, double *pa) {
      const int nb = min(NB, n - j0), r0 = j0 + nb, m = min(n, r0 + hbw) - r0;
      const int r = tid / NB, cc = tid % NB;
      dd          = (j0 < n && tid < NB * NB && r < nb && cc <= r && r - cc <= hbw) ? A(j0 + r, j0 + cc) : 0.0;
      const int row = r0 + NB * (wv - 1) + li;
#pragma unroll
      for (int kk = 0; kk < 4; ++kk)
        {
          const int col = j0 + 4 * kk + lk;
          pa[kk]        = (j0 < n && wv >= 1 && row - r0 < m && row - col <= hbw) ? A(row, col) : 0.0; // (m > 0: nb = NB)
        }
    };
    typedef double v4f64 __attribute__((ext_vector_type(4)));
    double dcur, pcur[4];
    request(0, dcur, pcur);
    if (tid < NB * NB)
      sL[tid / NB][tid % NB] = dcur;
    __syncthreads();
    for (int j0 = 0; j0 < n; j0 += NB)
      {
        const int nb = min(NB, n - j0), r0 = j0 + nb, m = min(n, r0 + hbw) - r0;
        double    dnext, pnext[4];
        request(j0 + NB, dnext, pnext);
        if (wv == 0)
          {
            // lane t = row t of the block: y_q = w_q / L_qq into lane q of a register of its own, the rows below take
            // w -= L[t][q] y_q (zero above the diagonal: the lanes of finished rows keep what nobody reads)
            const int tt = tid < NB ? tid : 0;
            double    rowl[NB];
#pragma unroll
            for (int q = 0; q < NB; ++q)
              rowl[q] = sL[tt][q];
            const double dg = sL[tt][tt];
            double       w  = tid < nb ? xs[j0 + tid] : 0.0;
            const double ri = tid < nb ? 1.0 / dg : 1.0;
            int          yflo = 0, yfhi = 0;
            static_for<0, NB>([&](auto qq) {
              constexpr int q   = decltype(qq)::value;
              const double  ws  = w * ri;
              const int     ylo = __builtin_amdgcn_readlane(__double2loint(ws), q), yhi = __builtin_amdgcn_readlane(__double2hiint(ws), q);
              asm("v_writelane_b32 %0, %1, %2" : "+v"(yflo) : "s"(ylo), "n"(q));
              asm("v_writelane_b32 %0, %1, %2" : "+v"(yfhi) : "s"(yhi), "n"(q));
              w = fma(-rowl[q], __hiloint2double(yhi, ylo), w);
            });
            if (tid < NB)
              {
                const double yt = tid < nb ? __hiloint2double(yfhi, yflo) : 0.0;
                sy[tid]         = yt;
                if (tid < nb)
                  xs[j0 + tid] = yt;
              }
          }
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); // (the requests for the next block stay in flight)
        if (wv >= 1 && NB * (wv - 1) < m)
          {
            v4f64 d = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
            for (int kk = 0; kk < 4; ++kk)
              d = __builtin_amdgcn_mfma_f64_16x16x4f64(pcur[kk], sy[4 * kk + lk], d, 0, 0, 0);
            if (li == 0) // (every column of the product is the same vector)
              {
#pragma unroll
                for (int v = 0; v < 4; ++v)
                  {
                    const int t = NB * (wv - 1) + lk + 4 * v;
                    if (t < m)
                      xs[r0 + t] -= d[v];
                  }
              }
          }
        if (tid < NB * NB)
          sL[tid / NB][tid % NB] = dnext; // the next diagonal block (outside the band or the block: zero)
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
#pragma unroll
        for (int kk = 0; kk < 4; ++kk)
          pcur[kk] = pnext[kk];
      }
    band_backward_lds<D>(band, n, hbw, true, xs, nullptr, sL, sy, bperm, x, false);
  }

  // ------------------------------------------------------------------ small vector kernels
  // dinv = 1 / diag(K); constraints.distribute afterwards keeps constrained entries of x at 0
  template <int D>
  __global__ __launch_bounds__(256) void extract_dinv(const double *__restrict__ vals,
                                                      const int32_t *__restrict__ diagpos, double *dinv, int64_t nnodes)
  {
    const int64_t i = int64_t(blockIdx.x) * 256 + threadIdx.x;
    if (i >= nnodes * D)
      return;
    const int64_t n = i / D;
    const int     c = int(i - n * D);
    const int32_t dp = diagpos[n]; // -1: the node has no row here (ghost node of a slab; its entries are never read)
    dinv[i]          = dp >= 0 ? 1.0 / vals[int64_t(dp) * (D * D) + c * D + c] : 0.0;
  }

  // the DxD diagonal block of every node out of the assembled tangent (mi_get_diagonal_blocks; nodes without a row: zeros)
  template <int D>
  __global__ __launch_bounds__(256) void gather_diag_blocks(const double *__restrict__ vals, const int32_t *__restrict__ diagpos,
                                                            double *out, int64_t nnodes)
  {
    const int64_t i = int64_t(blockIdx.x) * 256 + threadIdx.x;
    if (i >= nnodes * (D * D))
      return;
    const int64_t n  = i / (D * D);
    const int32_t dp = diagpos[n];
    out[i]           = dp >= 0 ? vals[int64_t(dp) * (D * D) + (i - n * (D * D))] : 0.0;
  }

  // inverse of the DxD diagonal block of every node (block-Jacobi smoother of the multigrid).  Dirichlet rows and
  // columns of a block are unit rows/columns up to the kept diagonal, so the block stays invertible.
  template <int D>
  __global__ __launch_bounds__(256) void extract_dinv_blk(const double *__restrict__ vals,
                                                          const int32_t *__restrict__ diagpos, double *dinv, double *sym6,
                                                          int64_t nnodes)
  {
    const int64_t n = int64_t(blockIdx.x) * 256 + threadIdx.x;
    if (n >= nnodes)
      return;
    double       *o = dinv + n * (D * D);
    if (diagpos[n] < 0) // no row here (ghost node of a slab; never read)
      {
#pragma unroll
        for (int k = 0; k < D * D; ++k)
          o[k] = 0.0;
        if (D == 3 && sym6)
#pragma unroll
          for (int k = 0; k < 6; ++k)
            sym6[n * 6 + k] = 0.0;
        return;
      }
    const double *a = vals + int64_t(diagpos[n]) * (D * D);
    if constexpr (D == 2)
      {
        const double r = 1.0 / (a[0] * a[3] - a[1] * a[2]);
        o[0]           = a[3] * r;
        o[1]           = -a[1] * r;
        o[2]           = -a[2] * r;
        o[3]           = a[0] * r;
      }
    else
      {
        double A[9];
#pragma unroll
        for (int k = 0; k < 9; ++k)
          A[k] = a[k];
        inv3x3(A, det3x3(A), o);
        if (sym6) // the same inverse, symmetric half (xx yy zz xy xz yz): what the matrix-free smoother step reads
          {
            double *q = sym6 + n * 6;
            q[0]      = o[0];
            q[1]      = o[4];
            q[2]      = o[8];
            q[3]      = o[3];
            q[4]      = o[6];
            q[5]      = o[7];
          }
      }
  }
  // out = Dblk^-1 a (node blocks); out may alias a
  template <int D>
  __global__ __launch_bounds__(256) void blk_apply(double *out, const double *__restrict__ a,
                                                   const double *__restrict__ dinv, int64_t nnodes)
  {
    const int64_t n = int64_t(blockIdx.x) * 256 + threadIdx.x;
    if (n >= nnodes)
      return;
    double v[D], r[D];
#pragma unroll
    for (int i = 0; i < D; ++i)
      v[i] = a[n * D + i];
#pragma unroll
    for (int i = 0; i < D; ++i)
      {
        r[i] = 0.0;
#pragma unroll
        for (int j = 0; j < D; ++j)
          r[i] += dinv[n * (D * D) + i * D + j] * v[j];
      }
#pragma unroll
    for (int i = 0; i < D; ++i)
      out[n * D + i] = r[i];
  }
  // Chebyshev step with the block-Jacobi diagonal: one thread per node of [node0, node0 + nnodes)
  template <int D>
  __global__ __launch_bounds__(256) void cheb_step_blk(double *x, double *d, const double *__restrict__ b,
                                                       const double *__restrict__ q, const double *__restrict__ dinv,
                                                       double c1, double c2, int64_t node0, int64_t nnodes)
  {
    const int64_t k = int64_t(blockIdx.x) * 256 + threadIdx.x;
    if (k >= nnodes)
      return;
    const int64_t n = node0 + k;
    double        res[D];
#pragma unroll
    for (int i = 0; i < D; ++i)
      res[i] = b[n * D + i] - (q ? q[n * D + i] : 0.0);
#pragma unroll
    for (int i = 0; i < D; ++i)
      {
        double s = 0.0;
#pragma unroll
        for (int j = 0; j < D; ++j)
          s += dinv[n * (D * D) + i * D + j] * res[j];
        const double dn = (c1 != 0.0 ? c1 * d[n * D + i] : 0.0) + c2 * s; // first step: old d not read
        d[n * D + i]    = dn;
        x[n * D + i]    = (q ? x[n * D + i] : 0.0) + dn;
      }
  }

  // the same step for 3 components with one thread per DOF (192 threads = 64 nodes): every load and store is a
  // contiguous stream (the node-per-thread form reads 24-byte records at a stride), the residual of a node's three
  // components is shared through LDS
  __global__ __launch_bounds__(192) void cheb_step_blk3(double *x, double *d, const double *__restrict__ b,
                                                        const double *__restrict__ q, const double *__restrict__ dinv,
                                                        double c1, double c2, int64_t node0, int64_t nnodes)
  {
    __shared__ double s_res[192];
    const int     ld  = threadIdx.x;
    const int64_t g   = (node0 + int64_t(blockIdx.x) * 64) * 3 + ld;
    const bool    in  = int64_t(blockIdx.x) * 64 + ld / 3 < nnodes;
    double        res = 0.0;
    if (in)
      res = b[g] - (q ? q[g] : 0.0);
    s_res[ld] = res;
    __syncthreads();
    if (!in)
      return;
    const int    r0 = (ld / 3) * 3;
    const double a0 = dinv[g * 3], a1 = dinv[g * 3 + 1], a2 = dinv[g * 3 + 2];
    double       s  = 0.0;
    s += a0 * s_res[r0];
    s += a1 * s_res[r0 + 1];
    s += a2 * s_res[r0 + 2];
    const double dn = (c1 != 0.0 ? c1 * d[g] : 0.0) + c2 * s; // first step: old d not read
    d[g]            = dn;
    x[g]            = (q ? x[g] : 0.0) + dn;
  }

  // masked l2 norm partials: sum over dofs whose constraint bit is clear (:549-576)
  template <int D>
  __global__ __launch_bounds__(256) void masked_norm_partials(const double *__restrict__ v,
                                                              const uint8_t *__restrict__ cmask, int64_t n, double *part)
  {
    __shared__ double s_red[4];
    const int64_t per = (n + gridDim.x - 1) / gridDim.x;
    const int64_t i0 = blockIdx.x * per, i1 = imin64(n, i0 + per);
    double        s  = 0.0;
    for (int64_t i = i0 + threadIdx.x; i < i1; i += 256)
      {
        const int64_t nd = i / D;
        const int     c  = int(i - nd * D);
        if (!((cmask[nd] >> c) & 1))
          s += v[i] * v[i];
      }
    s = block_sum<256>(s, s_red);
    if (threadIdx.x == 0)
      part[blockIdx.x] = s;
  }
  // total of a partial array (sum of squares for the norms; the square root is taken after the all-reduce)
  __global__ __launch_bounds__(256) void finish_sum(const double *part, int n, double *out)
  {
    __shared__ double s_red[4];
    if (!part || n <= 0)
      return;
    const double s = reduce_partials<256>(part, n, s_red);
    if (threadIdx.x == 0)
      *out = s;
  }
  // up to three partial arrays -> totals (distributed CG: local sums that are all-reduced afterwards)
  __global__ __launch_bounds__(256) void reduce_to_totals(const double *pa, int na, double *oa, const double *pb, int nb,
                                                          double *ob, const int32_t *done)
  {
    __shared__ double s_red[4];
    if (done && *done)
      return;
    if (pa)
      {
        const double s = reduce_partials<256>(pa, na, s_red);
        if (threadIdx.x == 0)
          *oa = s;
      }
    if (pb)
      {
        const double s = reduce_partials<256>(pb, nb, s_red);
        if (threadIdx.x == 0)
          *ob = s;
      }
  }
  // emulated all-reduce(sum) over the slab contexts of one process: bufs[r][off..off+cnt) := sum over r
  __global__ __launch_bounds__(64) void team_sum(double *const *bufs, int nranks, int off, int cnt)
  {
    const int k = threadIdx.x;
    if (k >= cnt)
      return;
    double s = 0.0;
    for (int r = 0; r < nranks; ++r)
      s += bufs[r][off + k];
    for (int r = 0; r < nranks; ++r)
      bufs[r][off + k] = s;
  }
  // out[slot[i]*D + c] = v[node[i]*D + c]   (owned interface nodes into the global interface buffer)
  template <int D>
  __global__ __launch_bounds__(256) void gather_to_slots(const double *__restrict__ v,
                                                         const int32_t *__restrict__ nodes,
                                                         const int32_t *__restrict__ slots, int n, double *out)
  {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i < n * D)
      out[int64_t(slots[i / D]) * D + i % D] = v[int64_t(nodes[i / D]) * D + i % D];
  }

  // constraints.distribute for homogeneous constraints (:1208): x[constrained] = 0
  template <int D>
  __global__ __launch_bounds__(256) void zero_constrained(double *x, const uint8_t *__restrict__ cmask, int64_t n)
  {
    const int64_t i = int64_t(blockIdx.x) * 256 + threadIdx.x;
    if (i >= n)
      return;
    const int64_t nd = i / D;
    if ((cmask[nd] >> int(i - nd * D)) & 1)
      x[i] = 0.0;
  }

  // update_acceleration (:592-599): a = alpha1 du - alpha2 v_old - alpha3 a_old
  __global__ __launch_bounds__(256) void newmark_acceleration(NewmarkParams p)
  {
    const int64_t i = int64_t(blockIdx.x) * 256 + threadIdx.x;
    if (i < p.n)
      {
        double a = p.alpha1 * p.du[i];
        a += -p.alpha2 * p.v_old[i] + -p.alpha3 * p.a_old[i];
        p.a[i] = a;
      }
  }
  // delta += newton_update (:487)
  __global__ __launch_bounds__(256) void vec_add(double *y, const double *__restrict__ x, int64_t n)
  {
    const int64_t i = int64_t(blockIdx.x) * 256 + threadIdx.x;
    if (i < n)
      y[i] += x[i];
  }
  // run() :139-144 fused: u += du; a = ...; v = alpha4 du + alpha5 v_old + alpha6 a_old; old := new
  __global__ __launch_bounds__(256) void newmark_finish(NewmarkParams p)
  {
    const int64_t i = int64_t(blockIdx.x) * 256 + threadIdx.x;
    if (i < p.n)
      {
        const double du = p.du[i], vo = p.v_old[i], ao = p.a_old[i];
        const double u  = p.u[i] + du;
        double       a  = p.alpha1 * du;
        a += -p.alpha2 * vo + -p.alpha3 * ao;
        double v = p.alpha4 * du;
        v += p.alpha5 * vo + p.alpha6 * ao;
        p.u[i]     = u;
        p.u_old[i] = u;
        p.a[i]     = a;
        p.a_old[i] = a;
        p.v[i]     = v;
        p.v_old[i] = v;
      }
  }

  // ---- linear model (linear_elasticity.cc:378-454, :579-586)
  // f = F + body; rhs = theta dt f + (1-theta) dt F_old; F_old = f; v_old = v; d_old = d;
  // w = theta (1-theta) dt^2 v + dt d      (so that  M v - K w  completes the right-hand side with 2 SpMVs)
  __global__ __launch_bounds__(256) void linear_rhs_prepare(LinearParams p)
  {
    const int64_t i = int64_t(blockIdx.x) * 256 + threadIdx.x;
    if (i >= p.n)
      return;
    const double f = p.load[i] + (p.body ? p.body[i] : 0.0);
    const double v = p.v[i], d = p.d[i];
    double       r = f * (p.dt * p.theta);
    r += p.dt * (1 - p.theta) * p.f_old[i];
    p.rhs[i]   = r;
    p.f_old[i] = f;
    p.v_old[i] = v;
    p.d_old[i] = d;
    p.w[i]     = p.theta * p.dt * p.dt * (1 - p.theta) * v + p.dt * d;
  }
  // rhs += M v_old - K w; boundary values (zero): rhs = 0 and v = 0 on constrained dofs (:426-451)
  template <int D>
  __global__ __launch_bounds__(256) void linear_rhs_finish(LinearParams p, const double *__restrict__ mv,
                                                           const double *__restrict__ kw,
                                                           const uint8_t *__restrict__ cmask)
  {
    const int64_t i = int64_t(blockIdx.x) * 256 + threadIdx.x;
    if (i >= p.n)
      return;
    const int64_t nd = i / D;
    if ((cmask[nd] >> int(i - nd * D)) & 1)
      {
        p.rhs[i] = 0.0;
        p.v[i]   = 0.0;
      }
    else
      p.rhs[i] += mv[i] - kw[i];
  }
  // D_n+1 = D_n + dt theta V_n+1 + dt (1-theta) V_n
  __global__ __launch_bounds__(256) void linear_update_displacement(LinearParams p)
  {
    const int64_t i = int64_t(blockIdx.x) * 256 + threadIdx.x;
    if (i >= p.n)
      return;
    double d = p.d[i];
    d += p.dt * p.theta * p.v[i];
    d += p.dt * (1 - p.theta) * p.v_old[i];
    p.d[i] = d;
  }

  // interface gather / scatter (adapter.h:389-443)
  template <int D>
  __global__ __launch_bounds__(256) void gather_nodes(const double *__restrict__ v, const int32_t *__restrict__ nodes,
                                                      int n, double *out)
  {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i < n * D)
      out[i] = v[int64_t(nodes[i / D]) * D + i % D];
  }
  template <int D>
  __global__ __launch_bounds__(256) void scatter_nodes(double *v, const int32_t *__restrict__ nodes, int n,
                                                       const double *__restrict__ in)
  {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i < n * D)
      v[int64_t(nodes[i / D]) * D + i % D] = in[i];
  }

  // ------------------------------------------------------------------ launchers
  template <int DIM, int P, int QSPLIT, int NT, int QC, int MINW = 1, int ABL = 0>
  static void launch_asm(const AsmParams &p, hipStream_t s)
  {
    hipLaunchKernelGGL((assemble_cells<DIM, P, QSPLIT, NT, QC, MINW, ABL>), dim3(p.cell_count), dim3(NT), 0, s, p);
  }

  // full kernel or its residual-only form (AsmParams::residual_only)
  template <int DIM, int P, int QSPLIT, int NT, int QC>
  static void launch_asm_sel(const AsmParams &p, hipStream_t s)
  {
    if (p.residual_only)
      launch_asm<DIM, P, QSPLIT, NT, QC, 1, 4>(p, s);
    else
      launch_asm<DIM, P, QSPLIT, NT, QC>(p, s);
  }

  int launch_assemble_cells(int dim, int degree, const AsmParams &p, hipStream_t s)
  {
    if (p.cell_count <= 0)
      return 0;
    // <DIM, P, QSPLIT, NT, QC>: NT >= NPC*DIM; QC divides NQ; NTILES*QSPLIT <= NT for a single pass over the tiles
    if (dim == 3 && degree == 2)
      {
        // 105 tiles x 2 lanes, 64 quadrature points in chunks of QC; variants for A/B timing (mi_set_tuning "asm_variant")
        switch ((p.variant == 9 || !p.residual_only) ? p.variant : 0)
          {
            case 0: // sum factorised (default): 4 waves per cell, 41 kB LDS
              if (p.residual_only)
                hipLaunchKernelGGL((assemble_q2sf<true, 0>), dim3(p.cell_count), dim3(64), 0, s, p);
#ifdef MI_EXPERIMENTS
              else if (p.qrec && p.from_records)
                {
                  // round 6: the point pass (gather, gradients, kinematics, records, residual: one wave per cell), then the
                  // tangent from the records
                  hipLaunchKernelGGL((assemble_q2sf<true, 0>), dim3(p.cell_count), dim3(64), 0, s, p);
                  if (p.from_records == 2) // wave 0 alone forms the fields from the records
                    hipLaunchKernelGGL((assemble_q2sf<false, 644>), dim3(p.cell_count), dim3(256), 0, s, p);
                  else                     // every wave a quarter of them
                    hipLaunchKernelGGL((assemble_q2sf<false, 388>), dim3(p.cell_count), dim3(256), 0, s, p);
                }
#endif
              else
                hipLaunchKernelGGL((assemble_q2sf<false, 132>), dim3(p.cell_count), dim3(256), 0, s, p);
              break;
#ifdef MI_EXPERIMENTS // (A/B instantiations: profiles/r05/asm_ab_*.txt, profiles/r06/asm_split_ab_n59.txt)
            case 3: // A/B: the kernel of round 4 (81 fields, block table by wave 0 behind the residual, branching scatter)
              hipLaunchKernelGGL((assemble_q2sf<false, 0>), dim3(p.cell_count), dim3(256), 0, s, p);
              break;
            case 4: // A/B: default + pipelined contraction
              hipLaunchKernelGGL((assemble_q2sf<false, 133>), dim3(p.cell_count), dim3(256), 0, s, p);
              break;
            case 5: // A/B: default + prologue at raised priority
              hipLaunchKernelGGL((assemble_q2sf<false, 134>), dim3(p.cell_count), dim3(256), 0, s, p);
              break;
            case 6: // A/B: 45 fields + block table by wave 3 alone
              hipLaunchKernelGGL((assemble_q2sf<false, 4>), dim3(p.cell_count), dim3(256), 0, s, p);
              break;
            case 7: // A/B: branch-free scatter alone
              hipLaunchKernelGGL((assemble_q2sf<false, 128>), dim3(p.cell_count), dim3(256), 0, s, p);
              break;
            case 8: // A/B: round 4 + pipelined contraction
              hipLaunchKernelGGL((assemble_q2sf<false, 1>), dim3(p.cell_count), dim3(256), 0, s, p);
              break;
#endif
            case 9: // node-pair form (the default until round 2): 16.4 ms per assembly at 5 M DoFs
              launch_asm_sel<3, 2, 2, 256, 8>(p, s);
              break;
            case 1:
              launch_asm<3, 2, 2, 256, 16>(p, s); // 54 kB LDS, 2 workgroups per CU: 23.6 ms per assembly at 5M DoFs
              break;
            case 2:
              launch_asm<3, 2, 2, 256, 4>(p, s); // 27 kB LDS: no faster than QC = 8 (register limited)
              break;
            default:
              launch_asm_sel<3, 2, 2, 256, 8>(p, s); // 36 kB LDS, 3 workgroups per CU: 19.5 ms
          }
      }
    else if (dim == 3 && degree == 1)
      launch_asm_sel<3, 1, 4, 64, 27>(p, s); // 10 tiles x 4
    else if (dim == 3 && degree == 3)
      launch_asm_sel<3, 3, 1, 576, 5>(p, s); // 528 tiles, 125 points in chunks of 5: 61 kB LDS, one pass
    else if (dim == 3 && degree == 4)
      launch_asm_sel<3, 4, 1, 512, 2>(p, s); // 2016 tiles in 4 passes, 216 points in chunks of 2
    else if (dim == 2 && degree == 1)
      launch_asm_sel<2, 1, 4, 64, 9>(p, s); // 3 tiles
    else if (dim == 2 && degree == 2)
      launch_asm_sel<2, 2, 4, 64, 16>(p, s); // 15 tiles x 4
    else if (dim == 2 && degree == 3)
      launch_asm_sel<2, 3, 2, 128, 25>(p, s); // 36 tiles x 2
    else if (dim == 2 && degree == 4)
      launch_asm_sel<2, 4, 2, 192, 18>(p, s); // 91 tiles x 2
    else
      return -1;
    return 0;
  }

  int launch_neumann_faces(int dim, int degree, const AsmParams &p, const int32_t *faces, int face_begin,
                           int face_count, hipStream_t s)
  {
    if (face_count <= 0)
      return 0;
#define MI_NF(D_, P_)                                                                                               \
  if (dim == D_ && degree == P_)                                                                                    \
    {                                                                                                               \
      hipLaunchKernelGGL((neumann_faces<D_, P_>), dim3(face_count), dim3(64), 0, s, p, faces, face_begin);          \
      return 0;                                                                                                     \
    }
    MI_NF(3, 2)
    MI_NF(3, 1)
    MI_NF(3, 3)
    MI_NF(3, 4)
    MI_NF(2, 1)
    MI_NF(2, 2)
    MI_NF(2, 3)
    MI_NF(2, 4)
#undef MI_NF
    return -1;
  }

  void launch_neumann_gather(int dim, const double *slots, const int32_t *node_ids, const int32_t *start, const int32_t *src,
                             const uint8_t *cmask, double *rhs, int nnodes_if, hipStream_t s)
  {
    if (nnodes_if <= 0)
      return;
    const int grid = (nnodes_if * dim + 255) / 256;
    if (dim == 3)
      hipLaunchKernelGGL((neumann_gather<3>), dim3(grid), dim3(256), 0, s, slots, node_ids, start, src, cmask, rhs, nnodes_if);
    else
      hipLaunchKernelGGL((neumann_gather<2>), dim3(grid), dim3(256), 0, s, slots, node_ids, start, src, cmask, rhs, nnodes_if);
  }

  template <int D>
  static void launch_spmv_d(const SpmvParams &p, int grid, hipStream_t s, int variant)
  {
    constexpr int DD = D * D;
    if (variant == 13 || variant == 14) // pure streaming read of the value array, 8- / 16-byte loads
      {
        const int64_t nvals = p.nvalblocks * DD;
        if (variant == 13)
          hipLaunchKernelGGL((stream_read<8>), dim3(grid), dim3(256), 0, s, p.vals, nvals, p.y);
        else
          hipLaunchKernelGGL((stream_read<16>), dim3(grid), dim3(256), 0, s, p.vals, nvals, p.y);
        return;
      }
    hipLaunchKernelGGL((blockrow_spmv_check<D>), dim3(grid), dim3(256), 0, s, p);
  }

  void launch_spmv(int dim, const SpmvParams &p, int grid, hipStream_t s, int variant, int /*maxrow*/)
  {
    if (dim == 3)
      launch_spmv_d<3>(p, grid, s, variant);
    else
      launch_spmv_d<2>(p, grid, s, variant);
  }

  // profiling: events that bracket exactly the NEXT production sliced-ELL launch (kernel start / kernel end from the
  // dispatch itself, as a profiler sees it); plain events recorded around a launch also pick up the tails of its
  // neighbours in the stream
  static thread_local hipEvent_t g_ev_start = nullptr, g_ev_stop = nullptr;
  void set_next_sell_launch_events(hipEvent_t start, hipEvent_t stop)
  {
    g_ev_start = start;
    g_ev_stop  = stop;
  }

  template <int D, bool NTL, bool DOT, bool F32, bool CHEB, bool ICOL>
  static void sell_launch(const SellParams &p, int grid, hipStream_t s)
  {
    const hipEvent_t a = g_ev_start, b = g_ev_stop;
    g_ev_start = g_ev_stop = nullptr;
    if (a && b)
      hipExtLaunchKernelGGL((sell_spmv<D, NTL, DOT, F32, CHEB, ICOL>), dim3(grid), dim3(SELL_WPB * 64), 0, s, a, b, 0, p);
    else
      hipLaunchKernelGGL((sell_spmv<D, NTL, DOT, F32, CHEB, ICOL>), dim3(grid), dim3(SELL_WPB * 64), 0, s, p);
  }
  template <int D, bool DOT, bool F32, bool CHEB, bool ICOL>
  static void sell_launch_split(const SellParams &p, hipStream_t s)
  {
    const hipEvent_t a = g_ev_start, b = g_ev_stop;
    g_ev_start = g_ev_stop = nullptr;
    if (a && b)
      hipExtLaunchKernelGGL((sell_spmv_split<D, DOT, F32, CHEB, ICOL>), dim3(p.nslices), dim3(SELL_SPLIT_W * 64), 0, s, a, b, 0, p);
    else
      hipLaunchKernelGGL((sell_spmv_split<D, DOT, F32, CHEB, ICOL>), dim3(p.nslices), dim3(SELL_SPLIT_W * 64), 0, s, p);
  }
  template <int D, bool NTL, bool ICOL>
  static void sell_dispatch(const SellParams &p, int grid, hipStream_t s)
  {
    const bool dot = p.dotv && p.partials, cheb = (p.cheb_d || p.cheb_b) && !dot, f32 = p.vals32 && !dot;
    if (p.split) // small launch: one workgroup per slice (the caller decides from the slice count, see SellParams::split)
      {
        if (dot)
          sell_launch_split<D, true, false, false, ICOL>(p, s);
        else if (cheb && f32)
          sell_launch_split<D, false, true, true, ICOL>(p, s);
        else if (cheb)
          sell_launch_split<D, false, false, true, ICOL>(p, s);
        else if (f32)
          sell_launch_split<D, false, true, false, ICOL>(p, s);
        else
          sell_launch_split<D, false, false, false, ICOL>(p, s);
        return;
      }
    if (dot)
      sell_launch<D, NTL, true, false, false, ICOL>(p, grid, s);
    else if (cheb && f32)
      sell_launch<D, NTL, false, true, true, ICOL>(p, grid, s);
    else if (cheb)
      sell_launch<D, NTL, false, false, true, ICOL>(p, grid, s);
    else if (f32)
      sell_launch<D, NTL, false, true, false, ICOL>(p, grid, s);
    else
      sell_launch<D, NTL, false, false, false, ICOL>(p, grid, s);
  }
  // unroll (tuning "sell_unroll"): 5 (default) = matrix loads with the non-temporal hint, anything else = plain loads;
  // column indices are generated when SellParams::rowbox is set (lattice meshes), read from memory otherwise
  void launch_sell_spmv(int dim, const SellParams &p, int grid, hipStream_t s, int unroll)
  {
    const bool nt = unroll == 5, icol = p.rowbox != nullptr;
#define MI_SELL_GO(D_)                                                                               \
  do                                                                                                 \
    {                                                                                                \
      if (nt && icol)                                                                                \
        sell_dispatch<D_, true, true>(p, grid, s);                                                   \
      else if (nt)                                                                                   \
        sell_dispatch<D_, true, false>(p, grid, s);                                                  \
      else if (icol)                                                                                 \
        sell_dispatch<D_, false, true>(p, grid, s);                                                  \
      else                                                                                           \
        sell_dispatch<D_, false, false>(p, grid, s);                                                 \
    }                                                                                                \
  while (0)
    if (dim == 3)
      MI_SELL_GO(3);
    else
      MI_SELL_GO(2);
#undef MI_SELL_GO
  }
  void launch_mf_spmv(const MfParams &p, int64_t cell_begin, int32_t cell_count, hipStream_t s, hipEvent_t ev_start,
                      hipEvent_t ev_stop)
  {
    if (cell_count <= 0)
      return;
    static const bool xcd = !(exp_env("MI_MF_XCD") && atoi(exp_env("MI_MF_XCD")) == 0);
    MfParams          q   = p;
    q.count               = cell_count;
    q.xcd_chunk           = xcd ? (cell_count + 7) / 8 : 0;
    const int grid        = xcd ? q.xcd_chunk * 8 : cell_count;
    auto *kern = q.lat.ncol > 0 ?
                   (q.yc ? (q.cellbox ? mf_spmv<true, true, true> : mf_spmv<false, true, true>) :
                           (q.cellbox ? mf_spmv<true, false, true> : mf_spmv<false, false, true>)) :
                   (q.yc ? (q.cellbox ? mf_spmv<true, true, false> : mf_spmv<false, true, false>) :
                           (q.cellbox ? mf_spmv<true, false, false> : mf_spmv<false, false, false>));
#ifdef MI_EXPERIMENTS
    if (q.stamps) // diagnostic: the production shape only (boxes, one launch, lattice ids or not)
      kern = q.lat.ncol > 0 ? mf_spmv<true, true, true, 1> : mf_spmv<true, true, false, 1>;
    static const bool occ5 = exp_env("MI_MF_OCC") && atoi(exp_env("MI_MF_OCC")) == 5; // A/B: five waves per SIMD (spills)
    if (occ5 && q.yc && q.cellbox && q.lat.ncol > 0 && !q.stamps)
      kern = mf_spmv<true, true, true, 0, 5>;
    static const int dbg = exp_env("MI_MF_DBG") ? atoi(exp_env("MI_MF_DBG")) : 0; // timing-only ablations (wrong results)
    if (dbg && q.yc && q.cellbox && q.lat.ncol > 0)
      kern = dbg == 2 ? mf_spmv<true, true, true, 2> : dbg == 4 ? mf_spmv<true, true, true, 4> : dbg == 8 ? mf_spmv<true, true, true, 8> :
             dbg == 14 ? mf_spmv<true, true, true, 14> : dbg == 6 ? mf_spmv<true, true, true, 6> : kern;
#else
    constexpr bool occ5 = false;
    constexpr int  dbg  = 0;
#endif
    // opt-in: fp32 arithmetic on fp32 records (the production shape; MfParams::qrec32 set by the caller for smoother products only)
    if (q.qrec32 && !occ5 && !dbg && !q.stamps && q.yc && q.cellbox && q.lat.ncol > 0)
      kern = mf_spmv<true, true, true, 0, 4, float>;
    if (ev_start || ev_stop)
      hipExtLaunchKernelGGL(kern, dim3(grid), dim3(64), 0, s, ev_start, ev_stop, 0, q, cell_begin);
    else
      hipLaunchKernelGGL(kern, dim3(grid), dim3(64), 0, s, q, cell_begin);
  }

  void launch_band_extract(int dim, const SellParams &p, const int32_t *bperm, double *band, int hbw, hipStream_t s)
  {
    const int grid = (p.nslices + 3) / 4;
    if (dim == 3)
      hipLaunchKernelGGL((band_extract<3>), dim3(grid), dim3(256), 0, s, p, bperm, band, hbw);
    else
      hipLaunchKernelGGL((band_extract<2>), dim3(grid), dim3(256), 0, s, p, bperm, band, hbw);
  }
  int launch_band_cholesky_solve(int dim, double *band, int n, int hbw, const int32_t *bperm, int nnodes, const double *b,
                                 double *x, double *work, int32_t *flag, bool factor, bool solve, hipStream_t s)
  {
    if (hbw >= BAND_MAXH)
      return -1;
    // factorisations of narrow bands (the reference's 2D geometries) run on the LDS-window kernel; MI_BAND_LDS=0: never (A/B)
    static const bool lds_ok = !(exp_env("MI_BAND_LDS") && atoi(exp_env("MI_BAND_LDS")) == 0);
    const bool far = hbw + BAND_NB > BAND_LDS_W;
    if (factor && lds_ok && hbw + BAND_NB <= BAND_LDS_W + BAND_FAR_MAX)
      {
        // MI_BAND_DBG (diagnostic): phase clocks of thread 0 of the first two factorisations, printed after a
        // synchronisation; the solve itself is the production one (stamps only), the buffer lives for the call
        static const bool   dbg_on = exp_env("MI_BAND_DBG") != nullptr;
        static int          shown  = 0;
        unsigned long long *d_dbg  = nullptr;
        if (dbg_on && shown < 2 && hipMalloc((void **)&d_dbg, 8 * sizeof(unsigned long long)) != hipSuccess)
          d_dbg = nullptr;
        if (d_dbg)
          hipMemsetAsync(d_dbg, 0, 8 * sizeof(unsigned long long), s);
        if (dim == 3 && far)
          hipLaunchKernelGGL((band_cholesky_lds<3, true>), dim3(1), dim3(1024), 0, s, band, n, hbw, bperm, nnodes, b, x, work,
                             flag, int(solve), d_dbg);
        else if (dim == 3)
          hipLaunchKernelGGL((band_cholesky_lds<3>), dim3(1), dim3(1024), 0, s, band, n, hbw, bperm, nnodes, b, x, work, flag,
                             int(solve), d_dbg);
        else if (far)
          hipLaunchKernelGGL((band_cholesky_lds<2, true>), dim3(1), dim3(1024), 0, s, band, n, hbw, bperm, nnodes, b, x, work,
                             flag, int(solve), d_dbg);
        else
          hipLaunchKernelGGL((band_cholesky_lds<2>), dim3(1), dim3(1024), 0, s, band, n, hbw, bperm, nnodes, b, x, work, flag,
                             int(solve), d_dbg);
        if (d_dbg)
          {
            unsigned long long h[8];
            hipStreamSynchronize(s);
            hipMemcpy(h, d_dbg, sizeof(h), hipMemcpyDeviceToHost);
            hipFree(d_dbg);
            ++shown;
            fprintf(stderr, "band_cholesky_lds (n = %d, hbw = %d) clocks of thread 0: first tile column %llu, next block column "
                            "beside the other tiles %llu (waves 0-3 through after %llu); before the loop %llu, backward substitution %llu\n", n, hbw, h[0], h[1], h[2], h[3], h[4]);
          }
        return 0;
      }
    // substitutions alone (the linear model's time steps): x in LDS where the system fits
    if (!factor && solve && lds_ok && n <= BAND_SOLVE_XMAX && hbw <= BAND_SOLVE_MAXH)
      {
        if (dim == 3)
          hipLaunchKernelGGL((band_solve_lds<3>), dim3(1), dim3(1024), 0, s, band, n, hbw, bperm, nnodes, b, x);
        else
          hipLaunchKernelGGL((band_solve_lds<2>), dim3(1), dim3(1024), 0, s, band, n, hbw, bperm, nnodes, b, x);
        return 0;
      }
    if (dim == 3)
      hipLaunchKernelGGL((band_cholesky_solve<3>), dim3(1), dim3(1024), 0, s, band, n, hbw, bperm, nnodes, b, x, work, flag,
                         int(factor), int(solve));
    else
      hipLaunchKernelGGL((band_cholesky_solve<2>), dim3(1), dim3(1024), 0, s, band, n, hbw, bperm, nnodes, b, x, work, flag,
                         int(factor), int(solve));
    return 0;
  }
  int launch_dense_inverse_from_sell(int dim, const SellParams &p, int n, double *out, hipStream_t s)
  {
    if (n > DENSE_BIG_MAX)
      return -1;
    if (n > DENSE_MAX)
      {
        if (dim == 3)
          hipLaunchKernelGGL((dense_inverse_from_sell_big<3>), dim3(1), dim3(1024), 0, s, p, n, out);
        else
          hipLaunchKernelGGL((dense_inverse_from_sell_big<2>), dim3(1), dim3(1024), 0, s, p, n, out);
      }
    else if (dim == 3)
      hipLaunchKernelGGL((dense_inverse_from_sell<3>), dim3(1), dim3(256), 0, s, p, n, out);
    else
      hipLaunchKernelGGL((dense_inverse_from_sell<2>), dim3(1), dim3(256), 0, s, p, n, out);
    return 0;
  }
  void launch_dense_apply(const double *inv, const double *b, double *x, int n, hipStream_t s)
  {
    if (n > DENSE_MAX)
      hipLaunchKernelGGL(dense_apply_big, dim3((n + 15) / 16), dim3(256), 0, s, inv, b, x, n);
    else
      hipLaunchKernelGGL(dense_apply, dim3(1), dim3(128), 0, s, inv, b, x, n);
  }

  void launch_mf_gather_cheb(const MfParams &p, const double *b, const double *dinv, double *d, double *xio, double *yres,
                             double c1, double c2, int64_t node0, int64_t nnodes, hipStream_t s)
  {
    if (nnodes <= 0)
      return;
    hipLaunchKernelGGL(mf_gather_cheb, dim3(int((nnodes + 63) / 64)), dim3(192), 0, s, p, b, dinv, d, xio, yres, c1, c2,
                       node0, nnodes);
  }
  void launch_mf_gather_cheb3(const MfParams &p, const double *b, const double *dinv6, const double *xprev, const double *xcur,
                              double *xnext, double c1, double c2, int64_t node0, int64_t nnodes, hipStream_t s)
  {
    if (nnodes <= 0)
      return;
    hipLaunchKernelGGL(mf_gather_cheb3, dim3(int((nnodes + 63) / 64)), dim3(192), 0, s, p, b, dinv6, xprev, xcur, xnext, c1, c2,
                       node0, nnodes);
  }
  void launch_mf_gather(const MfParams &p, int64_t ndofs, hipStream_t s)
  {
    hipLaunchKernelGGL(mf_gather, dim3(int((ndofs + 255) / 256)), dim3(256), 0, s, p, ndofs);
  }
  void launch_mf_gather_dot(const MfParams &p, int64_t ndofs, const double *dotv, double *partials, int grid, int64_t own0,
                            int64_t own_n, hipStream_t s)
  {
    hipLaunchKernelGGL(mf_gather_dot, dim3(grid), dim3(256), 0, s, p, ndofs, dotv, partials, own0, own_n);
  }
  void launch_point_pass_slots(const AsmParams &p, hipStream_t s)
  {
    if (p.cell_count <= 0)
      return;
    AsmParams q = p;
    q.xcd_chunk = (p.cell_count + 7) / 8;
    hipLaunchKernelGGL((assemble_q2sf<true, 1024>), dim3(q.xcd_chunk * 8), dim3(64), 0, s, q);
  }
  void launch_residual_gather(const double *slots3, const int32_t *slot_base, const int32_t *slot_src, const uint8_t *cmask, double *rhs,
                              int64_t ndofs, hipStream_t s)
  {
    hipLaunchKernelGGL(residual_gather, dim3(int((ndofs + 255) / 256)), dim3(256), 0, s, slots3, slot_base, slot_src, cmask, rhs, ndofs);
  }
  void launch_mf_spmv27(const MfParams &p, int32_t cell_count, hipStream_t s, hipEvent_t ev_start, hipEvent_t ev_stop)
  {
    if (cell_count <= 0)
      return;
    MfParams q  = p;
    q.count     = cell_count;
    q.xcd_chunk = ((cell_count + 1) / 2 + 7) / 8; // PAIRS of cells per XCD
    auto *kern  = q.lat.ncol > 0 ? (q.cellbox ? mf_spmv27<true, true> : mf_spmv27<false, true>) :
                                   (q.cellbox ? mf_spmv27<true, false> : mf_spmv27<false, false>);
    if (ev_start || ev_stop)
      hipExtLaunchKernelGGL(kern, dim3(q.xcd_chunk * 8), dim3(64), 0, s, ev_start, ev_stop, 0, q);
    else
      hipLaunchKernelGGL(kern, dim3(q.xcd_chunk * 8), dim3(64), 0, s, q);
  }
  void launch_mf_records27(const MfParams &p, const double *u, const double *du, double *rec27, int32_t cell_count, hipStream_t s)
  {
    if (cell_count <= 0)
      return;
    MfParams q  = p;
    q.count     = cell_count;
    q.xcd_chunk = ((cell_count + 1) / 2 + 7) / 8;
    auto *kern  = q.lat.ncol > 0 ? (q.cellbox ? mf_records27<true, true> : mf_records27<false, true>) :
                                   (q.cellbox ? mf_records27<true, false> : mf_records27<false, false>);
    hipLaunchKernelGGL(kern, dim3(q.xcd_chunk * 8), dim3(64), 0, s, q, u, du, rec27);
  }
  void launch_mf_diag(const MfParams &p, double *slots6, int32_t cell_count, hipStream_t s)
  {
    if (cell_count <= 0)
      return;
    MfParams q  = p;
    q.count     = cell_count;
    q.xcd_chunk = (cell_count + 7) / 8; // as mf_spmv: an XCD takes a contiguous run of cells
    if (q.cellbox)
      hipLaunchKernelGGL(mf_diag<true>, dim3(q.xcd_chunk * 8), dim3(64), 0, s, q, slots6);
    else
      hipLaunchKernelGGL(mf_diag<false>, dim3(q.xcd_chunk * 8), dim3(64), 0, s, q, slots6);
  }
  void launch_mf_diag_gather(const double *slots6, const int32_t *slot_base, const int32_t *slot_src, const uint8_t *cmask,
                             const int32_t *diagpos, double *blk, double *dinv, double *dinv_blk, double *sym6, int64_t nnodes, hipStream_t s)
  {
    hipLaunchKernelGGL(mf_diag_gather, dim3(int((nnodes + 255) / 256)), dim3(256), 0, s, slots6, slot_base, slot_src, cmask, diagpos, blk,
                       dinv, dinv_blk, sym6, nnodes);
  }

  void launch_ebe_spmv(const EbeParams &p, int64_t cell_begin, int32_t cell_count, hipStream_t s, hipEvent_t ev_start,
                       hipEvent_t ev_stop)
  {
    if (cell_count <= 0)
      return;
    if (ev_start && ev_stop)
      hipExtLaunchKernelGGL(ebe_spmv, dim3(cell_count), dim3(384), 0, s, ev_start, ev_stop, 0, p, cell_begin);
    else
      hipLaunchKernelGGL(ebe_spmv, dim3(cell_count), dim3(384), 0, s, p, cell_begin);
  }
  void launch_vals_to_f32(const double *vals, float *vals32, int64_t n, hipStream_t s)
  {
    hipLaunchKernelGGL(vals_to_f32, dim3(int(std::min<int64_t>((n + 255) / 256, 16384))), dim3(256), 0, s, vals, vals32, n);
  }
  void launch_sell_build_cols(const SellParams &p, const int32_t *rowptr, const int32_t *bsr_col, int32_t *sell_col,
                              hipStream_t s)
  {
    hipLaunchKernelGGL(sell_build_cols, dim3((p.nslices + 3) / 4), dim3(256), 0, s, p, rowptr, bsr_col, sell_col);
  }

  void launch_dot_partials(const double *a, const double *b, int64_t n, double *part, int grid, hipStream_t s)
  {
    hipLaunchKernelGGL(dot_partials, dim3(grid), dim3(256), 0, s, a, b, n, part);
  }
  void launch_cheb_step(double *x, double *d, const double *b, const double *q, const double *dinv, double c1, double c2,
                        int64_t n, hipStream_t s)
  {
    hipLaunchKernelGGL(cheb_step, dim3(int((n + 255) / 256)), dim3(256), 0, s, x, d, b, q, dinv, c1, c2, n);
  }
  void launch_cheb4_start(double *x, double *d, double *r, const double *b, const double *q, const double *dinv,
                          double s0, int64_t n, hipStream_t s)
  {
    hipLaunchKernelGGL(cheb4_start, dim3(int((n + 255) / 256)), dim3(256), 0, s, x, d, r, b, q, dinv, s0, n);
  }
  void launch_cheb4_step(double *x, double *d, double *r, const double *q, const double *dinv, double beta, double ca,
                         double cb, int64_t n, hipStream_t s)
  {
    hipLaunchKernelGGL(cheb4_step, dim3(int((n + 255) / 256)), dim3(256), 0, s, x, d, r, q, dinv, beta, ca, cb, n);
  }
  void launch_extract_dinv_blk(int dim, const double *vals, const int32_t *diagpos, double *dinv, double *sym6, int64_t nnodes,
                               hipStream_t s)
  {
    const int grid = int((nnodes + 255) / 256);
    if (dim == 3)
      hipLaunchKernelGGL((extract_dinv_blk<3>), dim3(grid), dim3(256), 0, s, vals, diagpos, dinv, sym6, nnodes);
    else
      hipLaunchKernelGGL((extract_dinv_blk<2>), dim3(grid), dim3(256), 0, s, vals, diagpos, dinv, (double *)nullptr, nnodes);
  }
  void launch_gather_diag_blocks(int dim, const double *vals, const int32_t *diagpos, double *out, int64_t nnodes, hipStream_t s)
  {
    const int64_t n = nnodes * dim * dim;
    if (dim == 3)
      hipLaunchKernelGGL((gather_diag_blocks<3>), dim3(int((n + 255) / 256)), dim3(256), 0, s, vals, diagpos, out, nnodes);
    else
      hipLaunchKernelGGL((gather_diag_blocks<2>), dim3(int((n + 255) / 256)), dim3(256), 0, s, vals, diagpos, out, nnodes);
  }
  void launch_blk_apply(int dim, double *out, const double *a, const double *dinv, int64_t nnodes, hipStream_t s)
  {
    const int grid = int((nnodes + 255) / 256);
    if (dim == 3)
      hipLaunchKernelGGL((blk_apply<3>), dim3(grid), dim3(256), 0, s, out, a, dinv, nnodes);
    else
      hipLaunchKernelGGL((blk_apply<2>), dim3(grid), dim3(256), 0, s, out, a, dinv, nnodes);
  }
  void launch_cheb_step_blk(int dim, double *x, double *d, const double *b, const double *q, const double *dinv,
                            double c1, double c2, int64_t node0, int64_t nnodes, hipStream_t s)
  {
    const int grid = int((nnodes + 255) / 256);
    if (dim == 3)
      hipLaunchKernelGGL(cheb_step_blk3, dim3(int((nnodes + 63) / 64)), dim3(192), 0, s, x, d, b, q, dinv, c1, c2, node0,
                         nnodes);
    else
      hipLaunchKernelGGL((cheb_step_blk<2>), dim3(grid), dim3(256), 0, s, x, d, b, q, dinv, c1, c2, node0, nnodes);
  }
  void launch_vec_scale_mul(double *dst, const double *a, const double *b, double s, int64_t n, hipStream_t st)
  {
    hipLaunchKernelGGL(vec_scale_mul, dim3(int((n + 255) / 256)), dim3(256), 0, st, dst, a, b, s, n);
  }
  void launch_vec_residual(double *res, const double *b, const double *q, int64_t n, hipStream_t s)
  {
    hipLaunchKernelGGL(vec_residual, dim3(int((n + 255) / 256)), dim3(256), 0, s, res, b, q, n);
  }
  void launch_vec_lincomb2(double *x, double a, const double *h1, double b, const double *h2, int64_t n, hipStream_t s)
  {
    hipLaunchKernelGGL(vec_lincomb2, dim3(int((n + 255) / 256)), dim3(256), 0, s, x, a, h1, b, h2, n);
  }
  void launch_scale_start(double *x, double *q, const double *sc, int64_t n, hipStream_t s)
  {
    hipLaunchKernelGGL(scale_start, dim3(int((n + 255) / 256)), dim3(256), 0, s, x, q, sc, n);
  }
  void launch_copy_owned(double *y, const double *x, int64_t n, int64_t own0, int64_t own_n, hipStream_t s)
  {
    hipLaunchKernelGGL(copy_owned, dim3(int((n + 255) / 256)), dim3(256), 0, s, y, x, n, own0, own_n);
  }
  void launch_lattice_interp(int dim, bool add, const LatticeParams &p, double *tgt, const double *src,
                             const uint8_t *cmask_tgt, hipStream_t s)
  {
    const int grid = int((p.n_tgt + 255) / 256);
    if (dim == 3)
      {
        if (add)
          hipLaunchKernelGGL((lattice_interp<3, true>), dim3(grid), dim3(256), 0, s, p, tgt, src, cmask_tgt);
        else
          hipLaunchKernelGGL((lattice_interp<3, false>), dim3(grid), dim3(256), 0, s, p, tgt, src, cmask_tgt);
      }
    else
      {
        if (add)
          hipLaunchKernelGGL((lattice_interp<2, true>), dim3(grid), dim3(256), 0, s, p, tgt, src, cmask_tgt);
        else
          hipLaunchKernelGGL((lattice_interp<2, false>), dim3(grid), dim3(256), 0, s, p, tgt, src, cmask_tgt);
      }
  }
  void launch_lattice_restrict(int dim, const LatticeParams &p, double *coarse, const double *fine,
                               const uint8_t *cmask_coarse, hipStream_t s)
  {
    const int grid = int((p.n_tgt + 255) / 256);
    // lists of at most 3 (nested lattices): 3^D loads per dof, all of them useful -- at every size; of at most 4: 4^D
    // loads of which 27-48 are used, worth it on the latency-bound levels only (loop form: 26 us, this: 42 us at 216 k nodes)
    if (p.rmax >= 1 && p.rmax <= 3)
      {
        const int gu = int((p.n_tgt * dim + 255) / 256);
        if (dim == 3)
          hipLaunchKernelGGL((lattice_restrict_unrolled<3, 3>), dim3(gu), dim3(256), 0, s, p, coarse, fine, cmask_coarse);
        else
          hipLaunchKernelGGL((lattice_restrict_unrolled<2, 3>), dim3(gu), dim3(256), 0, s, p, coarse, fine, cmask_coarse);
      }
    else if (p.rmax == 4 && p.n_tgt <= 100000)
      {
        const int gu = int((p.n_tgt * dim + 255) / 256);
        if (dim == 3)
          hipLaunchKernelGGL((lattice_restrict_unrolled<3, 4>), dim3(gu), dim3(256), 0, s, p, coarse, fine, cmask_coarse);
        else
          hipLaunchKernelGGL((lattice_restrict_unrolled<2, 4>), dim3(gu), dim3(256), 0, s, p, coarse, fine, cmask_coarse);
      }
    else if (dim == 3)
      hipLaunchKernelGGL((lattice_restrict<3>), dim3(grid), dim3(256), 0, s, p, coarse, fine, cmask_coarse);
    else
      hipLaunchKernelGGL((lattice_restrict<2>), dim3(grid), dim3(256), 0, s, p, coarse, fine, cmask_coarse);
  }

  // restriction + first smoother step of the coarse level (3D, lists of at most 4); returns false where that form does not exist
  bool launch_lattice_restrict_first_step(int dim, const LatticeParams &p, double *coarse, const double *fine,
                                          const uint8_t *cmask_coarse, double *x, double *d, const double *dinv_blk, double c2,
                                          int64_t node0, int64_t nnodes, hipStream_t s)
  {
    if (dim != 3 || p.rmax < 1 || p.rmax > 4 || (p.rmax == 4 && p.n_tgt > 100000))
      return false;
    const int g = int((p.n_tgt * 3 + 191) / 192);
    if (p.rmax <= 3)
      hipLaunchKernelGGL((lattice_restrict_first_step3<3>), dim3(g), dim3(192), 0, s, p, coarse, fine, cmask_coarse, x, d, dinv_blk, c2,
                         node0, nnodes);
    else
      hipLaunchKernelGGL((lattice_restrict_first_step3<4>), dim3(g), dim3(192), 0, s, p, coarse, fine, cmask_coarse, x, d, dinv_blk, c2,
                         node0, nnodes);
    return true;
  }

  void launch_cg_update_p(const CgParams &c, int it, int grid, hipStream_t s)
  {
    hipLaunchKernelGGL(cg_update_p, dim3(grid), dim3(256), 0, s, c, it);
  }
  void launch_cg_update_xr(const CgParams &c, int it, int grid, hipStream_t s)
  {
    hipLaunchKernelGGL(cg_update_xr, dim3(grid), dim3(256), 0, s, c, it);
  }
  void launch_cg_update_single(const CgParams &c, int it, int grid, hipStream_t s)
  {
    hipLaunchKernelGGL(cg_update_single, dim3(grid), dim3(256), 0, s, c, it);
  }
  void launch_cg_init_residual(const CgParams &c, const double *b, double *part_bb, int grid, hipStream_t s)
  {
    hipLaunchKernelGGL(cg_init_residual, dim3(grid), dim3(256), 0, s, c, b, part_bb);
  }
  void launch_cg_set_tolerance(const CgParams &c, const double *part_bb, double rel_tol, hipStream_t s)
  {
    hipLaunchKernelGGL(cg_set_tolerance, dim3(1), dim3(256), 0, s, c, part_bb, rel_tol);
  }
  void launch_cg_small(int dim, const SellParams &p, const CgParams &c, const double *b, double rel_tol, int max_it,
                       hipStream_t s)
  {
    if (dim == 3)
      hipLaunchKernelGGL((cg_small<3>), dim3(1), dim3(1024), 0, s, p, c, b, rel_tol, max_it);
    else
      hipLaunchKernelGGL((cg_small<2>), dim3(1), dim3(1024), 0, s, p, c, b, rel_tol, max_it);
  }
  void launch_cg_final_check(const CgParams &c, int it, hipStream_t s)
  {
    hipLaunchKernelGGL(cg_final_check, dim3(1), dim3(256), 0, s, c, it);
  }
  void launch_assemble_linear(int dim, const LinAsmParams &p, hipStream_t s)
  {
    if (p.cell_count <= 0)
      return;
    if (dim == 3)
      hipLaunchKernelGGL(assemble_linear_cells<3>, dim3(p.cell_count), dim3(256), 0, s, p);
    else
      hipLaunchKernelGGL(assemble_linear_cells<2>, dim3(p.cell_count), dim3(256), 0, s, p);
  }
  void launch_extract_dinv(int dim, const double *vals, const int32_t *diagpos, double *dinv, int64_t nnodes,
                           hipStream_t s)
  {
    const int grid = int((nnodes * dim + 255) / 256);
    if (dim == 3)
      hipLaunchKernelGGL((extract_dinv<3>), dim3(grid), dim3(256), 0, s, vals, diagpos, dinv, nnodes);
    else
      hipLaunchKernelGGL((extract_dinv<2>), dim3(grid), dim3(256), 0, s, vals, diagpos, dinv, nnodes);
  }
  void launch_masked_norm(int dim, const double *v, const uint8_t *cmask, int64_t n, double *part, int grid,
                          double *out, hipStream_t s)
  {
    if (dim == 3)
      hipLaunchKernelGGL((masked_norm_partials<3>), dim3(grid), dim3(256), 0, s, v, cmask, n, part);
    else
      hipLaunchKernelGGL((masked_norm_partials<2>), dim3(grid), dim3(256), 0, s, v, cmask, n, part);
    hipLaunchKernelGGL(finish_sum, dim3(1), dim3(256), 0, s, part, grid, out);
  }
  void launch_finish_sum(const double *part, int n, double *out, hipStream_t s)
  {
    hipLaunchKernelGGL(finish_sum, dim3(1), dim3(256), 0, s, part, n, out);
  }
  void launch_reduce_to_totals(const double *pa, int na, double *oa, const double *pb, int nb, double *ob,
                               const int32_t *done, hipStream_t s)
  {
    hipLaunchKernelGGL(reduce_to_totals, dim3(1), dim3(256), 0, s, pa, na, oa, pb, nb, ob, done);
  }
  void launch_team_sum(double *const *bufs, int nranks, int off, int cnt, hipStream_t s)
  {
    hipLaunchKernelGGL(team_sum, dim3(1), dim3(64), 0, s, bufs, nranks, off, cnt);
  }
  void launch_gather_to_slots(int dim, const double *v, const int32_t *nodes, const int32_t *slots, int n, double *out,
                              hipStream_t s)
  {
    const int grid = (n * dim + 255) / 256;
    if (grid == 0)
      return;
    if (dim == 3)
      hipLaunchKernelGGL((gather_to_slots<3>), dim3(grid), dim3(256), 0, s, v, nodes, slots, n, out);
    else
      hipLaunchKernelGGL((gather_to_slots<2>), dim3(grid), dim3(256), 0, s, v, nodes, slots, n, out);
  }
  void launch_zero_constrained(int dim, double *x, const uint8_t *cmask, int64_t n, hipStream_t s)
  {
    const int grid = int((n + 255) / 256);
    if (dim == 3)
      hipLaunchKernelGGL((zero_constrained<3>), dim3(grid), dim3(256), 0, s, x, cmask, n);
    else
      hipLaunchKernelGGL((zero_constrained<2>), dim3(grid), dim3(256), 0, s, x, cmask, n);
  }
  void launch_newmark_acceleration(const NewmarkParams &p, hipStream_t s)
  {
    hipLaunchKernelGGL(newmark_acceleration, dim3(int((p.n + 255) / 256)), dim3(256), 0, s, p);
  }
  void launch_newmark_finish(const NewmarkParams &p, hipStream_t s)
  {
    hipLaunchKernelGGL(newmark_finish, dim3(int((p.n + 255) / 256)), dim3(256), 0, s, p);
  }
  void launch_vec_add(double *y, const double *x, int64_t n, hipStream_t s)
  {
    hipLaunchKernelGGL(vec_add, dim3(int((n + 255) / 256)), dim3(256), 0, s, y, x, n);
  }
  void launch_linear_rhs_prepare(const LinearParams &p, hipStream_t s)
  {
    hipLaunchKernelGGL(linear_rhs_prepare, dim3(int((p.n + 255) / 256)), dim3(256), 0, s, p);
  }
  void launch_linear_rhs_finish(int dim, const LinearParams &p, const double *mv, const double *kw,
                                const uint8_t *cmask, hipStream_t s)
  {
    const int grid = int((p.n + 255) / 256);
    if (dim == 3)
      hipLaunchKernelGGL((linear_rhs_finish<3>), dim3(grid), dim3(256), 0, s, p, mv, kw, cmask);
    else
      hipLaunchKernelGGL((linear_rhs_finish<2>), dim3(grid), dim3(256), 0, s, p, mv, kw, cmask);
  }
  void launch_linear_update_displacement(const LinearParams &p, hipStream_t s)
  {
    hipLaunchKernelGGL(linear_update_displacement, dim3(int((p.n + 255) / 256)), dim3(256), 0, s, p);
  }
  void launch_gather_nodes(int dim, const double *v, const int32_t *nodes, int n, double *out, hipStream_t s)
  {
    const int grid = (n * dim + 255) / 256;
    if (grid == 0)
      return;
    if (dim == 3)
      hipLaunchKernelGGL((gather_nodes<3>), dim3(grid), dim3(256), 0, s, v, nodes, n, out);
    else
      hipLaunchKernelGGL((gather_nodes<2>), dim3(grid), dim3(256), 0, s, v, nodes, n, out);
  }
  void launch_scatter_nodes(int dim, double *v, const int32_t *nodes, int n, const double *in, hipStream_t s)
  {
    const int grid = (n * dim + 255) / 256;
    if (grid == 0)
      return;
    if (dim == 3)
      hipLaunchKernelGGL((scatter_nodes<3>), dim3(grid), dim3(256), 0, s, v, nodes, n, in);
    else
      hipLaunchKernelGGL((scatter_nodes<2>), dim3(grid), dim3(256), 0, s, v, nodes, n, in);
  }
} // namespace mi
