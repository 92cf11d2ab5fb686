// fe29_inv.h — 1/a in GF(p) by safegcd division steps (modinv30.h) for the 9x29 field: ~10 k instructions
// against ~38 k for the Fermat chain fe29_inv (fe29.h; Invert, internal/field/field_invert.go:11).
// Same value for every input (0 -> 0).  Used where ONE lane inverts for few others: the per-key
// tables (keyed.hip), one inversion per 8 points.
#pragma once
#include "fe29.h"
#include "modinv30.h"

namespace s2k {

__device__ __noinline__ fe29 fe29_inv_gcd(fe29 a) {
  uint32_t w[8], r[8];
  fe29_to_words(w, fe29_normalize(a));
  mi_modinv_words<true>(r, w);
  return fe29_from_words(r);
}

}  // namespace s2k
