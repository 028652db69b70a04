// Tap bookkeeping of the F(2x2,2x2) kernels (wino22_conv.hip, wino22_wgrad.hip): a 2-tap unit-stride axis is
// one phase; a stride-2 axis with 4 consecutive taps splits into two source parities of 2 taps each.
#pragma once
#include "common.h"

struct AxisPlan {
  int nph;            // 1 (unit-stride 2-tap axis) or 2 (stride-2 4-tap axis: two source parities)
  int stride;         // source stride
  int par[2];         // parity offset per phase
  int dmin[2];        // patch origin (sub-lattice units)
  int kidx[2][2];     // weight index of Winograd tap a (= sub-lattice offset dmin + a) per phase
};

__host__ __device__ inline int floordiv2(int v) { return (v >= 0) ? (v >> 1) : -((1 - v) >> 1); }

__host__ __device__ inline bool plan_axis(const rehr_axis_taps& t, int s, int b, AxisPlan& ap) {
  if (t.offs != 1 && t.offs != -1) return false;
  if (s == 1 && t.count == 2) {
    ap.nph = 1;
    ap.stride = 1;
    ap.par[0] = 0;
    const int o0 = b + t.off0, o1 = b + t.off0 + t.offs;
    ap.dmin[0] = o0 < o1 ? o0 : o1;
    ap.kidx[0][o0 - ap.dmin[0]] = t.k0;
    ap.kidx[0][o1 - ap.dmin[0]] = t.k0 + t.ks;
    return true;
  }
  if (s == 2 && t.count == 4) {
    ap.nph = 2;
    ap.stride = 2;
    for (int p = 0; p < 2; ++p) {       // taps j = p, p + 2
      const int c = b + t.off0 + t.offs * p;
      const int base = floordiv2(c);
      ap.par[p] = c - 2 * base;
      const int o0 = base, o1 = base + t.offs;   // sub-lattice offsets of the two taps
      ap.dmin[p] = o0 < o1 ? o0 : o1;
      ap.kidx[p][o0 - ap.dmin[p]] = t.k0 + t.ks * p;
      ap.kidx[p][o1 - ap.dmin[p]] = t.k0 + t.ks * (p + 2);
    }
    return true;
  }
  return false;
}

