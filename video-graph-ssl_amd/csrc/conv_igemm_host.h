// Host-side problem description shared by the forward / dgrad implicit-GEMM kernels (conv3d.hip: per-tap gathers;
// conv3d_halo.hip: LDS halo tiles): problem classes, kernel parameter block, packed-weight geometry.
#pragma once
#include <algorithm>
#include <vector>
#include "conv_common.h"

namespace gca_conv {

struct IgemmParams {
  int SC, SD, SH, SW;      // gathered (source) tensor: channels + spatial dims
  int DK;                  // destination channels (GEMM M, un-padded)
  int QD, QH, QW;          // iteration sub-grid (GEMM N = NB*QD*QH*QW)
  int DD, DH, DW;          // destination tensor spatial dims
  int dm_d, dm_h, dm_w;    // destination position = q*dm + do
  int do_d, do_h, do_w;
  int m_d, m_h, m_w;       // source position = q*m + o + tap delta
  int o_d, o_h, o_w;
  int ntaps;               // taps of this class (FAST: <= 62)
  int Kpad, Mpad;
  int tilesM, tilesN;
  int tileN_off;           // first column tile of this launch (two-phase launches: tall tiles, then a short-tile tail)
  int splits, kt_per_split;
  int P;                   // stat partials per channel
  int chk;                 // bit0: test D, bit1: test H, bit2: test W
  int accumulate;
  long long Ntot;
  long long src_nstride;   // elements between consecutive images of the gathered tensor
  unsigned src_bytes;      // extent of the gathered tensor for the buffer resource (range check)
  unsigned dst_bytes;      // extent of the destination tensor
  unsigned slab_bytes;     // extent of one fp32 split-K slab [DK][Ntot]
};

// ---- host side: problem classes ---------------------------------------------------------------
struct ClassInfo {
  int na, nb, nc, ntaps;        // tap grid of the class
  int k0[3], ks[3];             // tap = k0 + ks*j per dim
  int dl0[3], dls[3];           // gather delta of tap index j per dim: dl0 + dls*j
  int q[3];                     // iteration sub-grid
  int dm[3], dof[3];            // destination = q*dm + dof
  int m[3], o[3];               // source = q*m + o + delta
  int srcC, M;                  // gathered channels, GEMM M
  long long Kred, Kpad;
  long long table_off;          // int2 units inside the table buffer
  long long pack_off;           // floats inside the packed buffer
  bool vec;
};

inline int cdiv(int a, int b) { return (a + b - 1) / b; }
inline int fast_of(int ntaps) { return ntaps <= 31 ? 1 : (ntaps <= FAST_MAX_TAPS ? 2 : 0); }
inline int pack_rows(int M) { return (int)gca_round_up(M, 32) + 128; }   // every tile height up to 160 stays in bounds

// which: 0 forward, 1 dgrad
// floats reserved for the packed weights of one class: room for either layout (k-major fp32 rows of the gather kernels,
// or [chunk][tap][row][<= 96 B] of the halo kernels), so that class offsets do not depend on the launch configuration
inline long long pack_reserve(const ClassInfo& c) {
  const long long rows = pack_rows(c.M);
  const long long flat = c.Kpad * rows;
  const long long halo = c.srcC >= 8 && c.ntaps <= 64 ? (long long)cdiv(c.srcC, 16) * c.ntaps * rows * 24 : 0;   // halo_geometry()'s limits
  // stem kernels (conv3d_stem.hip): 16 k per pair of (channel, kd, kh) rows, up to 96 bytes (24 floats) per packed row
  const long long stem = c.srcC <= 4 && c.nc <= 8 && c.m[2] == 2 ? (long long)cdiv(c.srcC * c.na * c.nb, 2) * rows * 24 : 0;
  const long long m = halo > stem ? halo : stem;
  return m > flat ? m : flat;
}

inline void build_classes(const gca_conv_geom* g, int which, std::vector<ClassInfo>& out) {
  const int kdim[3] = {g->kd, g->kh, g->kw}, sdim[3] = {g->sd, g->sh, g->sw}, pdim[3] = {g->pd, g->ph, g->pw};
  const int in[3] = {g->D, g->H, g->W}, od[3] = {g->OD, g->OH, g->OW};
  long long toff = 0, poff = 0;
  auto finish = [&](ClassInfo& c) {
    c.ntaps = c.na * c.nb * c.nc;
    c.Kred = (long long)c.srcC * c.ntaps;
    c.Kpad = gca_round_up(c.Kred, BK);
    c.table_off = toff; c.pack_off = poff;
    toff += c.Kpad + 32;                                   // rows + 64-int tap-delta table
    poff += pack_reserve(c);
    // VEC: rows are contiguous runs of the image -> no taps / stride / padding in H and W, float4-aligned planes
    const int SH = which == 0 ? in[1] : od[1], SW = which == 0 ? in[2] : od[2];
    c.vec = c.nb == 1 && c.nc == 1 && c.m[1] == 1 && c.m[2] == 1 && c.o[1] + c.dl0[1] == 0 && c.o[2] + c.dl0[2] == 0 &&
            c.dm[1] == 1 && c.dm[2] == 1 && c.dof[1] == 0 && c.dof[2] == 0 && c.q[1] == SH && c.q[2] == SW &&
            (SH * SW) % 4 == 0 && c.ntaps <= 31 &&
            (which == 1 || g->x_batch_stride % 4 == 0);
    out.push_back(c);
  };
  if (which == 0) {
    ClassInfo c{};
    c.na = g->kd; c.nb = g->kh; c.nc = g->kw;
    for (int d = 0; d < 3; ++d) {
      c.k0[d] = 0; c.ks[d] = 1; c.dl0[d] = 0; c.dls[d] = 1;
      c.q[d] = od[d]; c.dm[d] = 1; c.dof[d] = 0; c.m[d] = sdim[d]; c.o[d] = -pdim[d];
    }
    c.srcC = g->C; c.M = g->K;
    // Temporal convs on short clips: a tap a of dimension d meets real input for some output position only if
    // p - (OD-1) s <= a <= D - 1 + p; the others multiply the zero padding for EVERY output (layer4 of R(2+1)D-18 runs
    // (3,1,1) convs on D = 1: two of three taps).  Such taps are dropped from the class -- no MFMA work, no packed weights.
    if (g->kh == 1 && g->kw == 1 && g->C > 4) {
      const int amin = std::max(0, pdim[0] - (od[0] - 1) * sdim[0]), amax = std::min(kdim[0] - 1, in[0] - 1 + pdim[0]);
      if (amin <= amax && (amin > 0 || amax < kdim[0] - 1)) { c.k0[0] = amin; c.dl0[0] = amin; c.na = amax - amin + 1; }
    }
    finish(c);
    return;
  }
  // dgrad: one class per residue (rho_d, rho_h, rho_w) of the destination position modulo the stride
  for (int rd = 0; rd < sdim[0]; ++rd)
    for (int rh = 0; rh < sdim[1]; ++rh)
      for (int rw = 0; rw < sdim[2]; ++rw) {
        const int rho[3] = {rd, rh, rw};
        ClassInfo c{};
        int cnt[3];
        bool empty = false;
        for (int d = 0; d < 3; ++d) {
          const int k0 = (rho[d] + pdim[d]) % sdim[d];
          cnt[d] = k0 < kdim[d] ? (kdim[d] - 1 - k0) / sdim[d] + 1 : 0;
          c.k0[d] = k0; c.ks[d] = sdim[d];
          c.dl0[d] = (rho[d] + pdim[d] - k0) / sdim[d];    // exact
          c.dls[d] = -1;                                    // next tap of the class is one source step back
          c.q[d] = rho[d] < in[d] ? (in[d] - rho[d] + sdim[d] - 1) / sdim[d] : 0;
          c.dm[d] = sdim[d]; c.dof[d] = rho[d];
          c.m[d] = 1; c.o[d] = 0;
          if (cnt[d] == 0 || c.q[d] == 0) empty = true;
        }
        if (!empty && g->kh == 1 && g->kw == 1 && g->C > 4) {
          // the same for the transposed problem: tap j of the class reads dY position q + dl0 - j, q in [0, c.q); taps that
          // fall outside [0, OD) for every q are dropped
          const int jmin = std::max(0, c.dl0[0] - (od[0] - 1)), jmax = std::min(cnt[0] - 1, c.dl0[0] + c.q[0] - 1);
          if (jmin > jmax) empty = true;
          else if (jmin > 0 || jmax < cnt[0] - 1) { c.k0[0] += sdim[0] * jmin; c.dl0[0] -= jmin; cnt[0] = jmax - jmin + 1; }
        }
        c.na = cnt[0]; c.nb = cnt[1]; c.nc = cnt[2];
        c.srcC = g->K; c.M = g->C;
        if (empty) { c.na = c.nb = c.nc = 0; c.ntaps = 0; c.Kred = c.Kpad = 0; c.table_off = toff; c.pack_off = poff; c.vec = false; out.push_back(c); continue; }
        finish(c);
      }
}

inline void class_params(const gca_conv_geom* g, int which, const ClassInfo& c, IgemmParams& p) {
  if (which == 0) { p.SC = g->C; p.SD = g->D; p.SH = g->H; p.SW = g->W; p.DD = g->OD; p.DH = g->OH; p.DW = g->OW; }
  else { p.SC = g->K; p.SD = g->OD; p.SH = g->OH; p.SW = g->OW; p.DD = g->D; p.DH = g->H; p.DW = g->W; }
  p.DK = c.M;
  p.QD = c.q[0]; p.QH = c.q[1]; p.QW = c.q[2];
  p.dm_d = c.dm[0]; p.dm_h = c.dm[1]; p.dm_w = c.dm[2];
  p.do_d = c.dof[0]; p.do_h = c.dof[1]; p.do_w = c.dof[2];
  p.m_d = c.m[0]; p.m_h = c.m[1]; p.m_w = c.m[2];
  p.o_d = c.o[0]; p.o_h = c.o[1]; p.o_w = c.o[2];
  p.ntaps = c.ntaps;
  p.Kpad = (int)c.Kpad; p.Mpad = (int)gca_round_up(c.M, MPAD);
  p.Ntot = (long long)g->N * c.q[0] * c.q[1] * c.q[2];
  p.src_nstride = which == 0 ? (g->x_batch_stride ? g->x_batch_stride : (long long)g->C * g->D * g->H * g->W)
                             : (long long)g->K * g->OD * g->OH * g->OW;
  {
    const long long es = g->act_f16 ? 2 : 4;                                    // activation element size (fp16 storage: 2)
    const long long span = (long long)g->N * p.src_nstride * es;
    p.src_bytes = span > 0xfffff000LL ? 0xfffff000u : (unsigned)span;
    const long long dspan = (long long)g->N * p.DK * p.DD * p.DH * p.DW * es;    // destination tensor
    p.dst_bytes = dspan > 0xfffff000LL ? 0xfffff000u : (unsigned)dspan;
    const long long sspan = (long long)p.DK * p.Ntot * 4;                        // one split-K slab (always fp32)
    p.slab_bytes = sspan > 0xfffff000LL ? 0xfffff000u : (unsigned)sspan;
  }
  // bounds tests: skip a dimension when every tap of every column stays inside by construction
  const int lim[3] = {p.SD, p.SH, p.SW};
  int chk = 0;
  const int cn[3] = {c.na, c.nb, c.nc};
  for (int d = 0; d < 3; ++d) {
    const int dmin = c.dls[d] > 0 ? c.dl0[d] : c.dl0[d] + c.dls[d] * (cn[d] - 1);
    const int dmax = c.dls[d] > 0 ? c.dl0[d] + c.dls[d] * (cn[d] - 1) : c.dl0[d];
    const int lo = c.o[d] + dmin, hi = (c.q[d] - 1) * c.m[d] + c.o[d] + dmax;
    if (lo < 0 || hi >= lim[d]) chk |= 1 << d;
  }
  p.chk = chk;
}


// Weight re-layout job (gca_conv_pack*).  fmt 0: packed[m][k] fp32, k = (ch, tap) contiguous, zero padded to
// [Mrows][Kpad] (conv3d.hip).  fmt 1: the LDS-halo kernels' operand (conv3d_halo.hip): [step = chunk*ntaps + tap][m < Mrows]
// [part][16 channels of the chunk], bf16 parts PRE-SPLIT (fp32 for arithmetic mode 0), zero padded.
// Class tap grid: tap = ((k0d+sd*a)*KH + (k0h+sh*b))*KW + (k0w+sw*c).  fwd (which 0): value = W[m][ch][tap];
// dgrad (which 1): value = W[ch][m][tap].
struct PackParams {
  int Kred, M, Kpad, Mrows;
  int ntaps, nb, nc;           // class tap grid: ntaps = na*nb*nc
  int k0d, k0h, k0w, sd, sh, sw;
  int KH, KW, T;               // full kernel
  int fmt, SC, nsteps, math;   // fmt 1 (LDS-halo) / 2 (stem): reduction channels, steps, arithmetic mode of the consumer
  long long s_ch, s_m;         // element strides of `ch` and `m` in W
};
// work items of a pack job: fmt 0 one per packed float, fmt 1 one per (step, row, channel pair)
__host__ __device__ inline long long pack_items(const PackParams& p) {
  if (p.fmt == 3) return (long long)p.Mrows * (p.Kpad >> 1);      // pointwise fp16 operand: one item per pair of k
  return p.fmt >= 1 ? (long long)p.nsteps * p.Mrows * 8 : (long long)p.Mrows * p.Kpad;
}

}  // namespace gca_conv
