// fe26.h — 26-bit limb helpers: the 8x32 <-> 10x26 conversions used by sc26.h (arithmetic modulo
// the group order on 10 x 26-bit limbs) and mad64s, the multiply-add primitive shared with fe29.h.
// (Round 1 ran the whole verification ladder on a 10x26 field defined here; it was replaced by
// the 9x29 field of fe29.h)
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "fe.h"

namespace s2k {

struct fe26 {
  uint32_t n[10];
};

constexpr uint32_t F26_M = 0x3FFFFFFu;

// 8 x 32-bit little-endian words (value < 2^256) -> magnitude-1 limbs
S2K_DEV fe26 fe26_from_words(const uint32_t w[8]) {
  fe26 r;
  r.n[0] = w[0] & F26_M;
  r.n[1] = ((w[0] >> 26) | (w[1] << 6)) & F26_M;
  r.n[2] = ((w[1] >> 20) | (w[2] << 12)) & F26_M;
  r.n[3] = ((w[2] >> 14) | (w[3] << 18)) & F26_M;
  r.n[4] = ((w[3] >> 8) | (w[4] << 24)) & F26_M;
  r.n[5] = (w[4] >> 2) & F26_M;
  r.n[6] = ((w[4] >> 28) | (w[5] << 4)) & F26_M;
  r.n[7] = ((w[5] >> 22) | (w[6] << 10)) & F26_M;
  r.n[8] = ((w[6] >> 16) | (w[7] << 16)) & F26_M;
  r.n[9] = w[7] >> 10;
  return r;
}

// canonical (fully normalised) limbs -> 8 x 32-bit little-endian words
S2K_DEV void fe26_to_words(uint32_t w[8], const fe26& a) {
  w[0] = a.n[0] | (a.n[1] << 26);
  w[1] = (a.n[1] >> 6) | (a.n[2] << 20);
  w[2] = (a.n[2] >> 12) | (a.n[3] << 14);
  w[3] = (a.n[3] >> 18) | (a.n[4] << 8);
  w[4] = (a.n[4] >> 24) | (a.n[5] << 2) | (a.n[6] << 28);
  w[5] = (a.n[6] >> 4) | (a.n[7] << 22);
  w[6] = (a.n[7] >> 10) | (a.n[8] << 16);
  w[7] = (a.n[8] >> 16) | (a.n[9] << 10);
}

// acc += a * k as one v_mad_u64_u32 with a wave-uniform multiplier (SGPR).  Inline asm on
// purpose: the generated products (tools/gen_chain.py) keep every column as one serial chain,
// and hand-placed multiply-adds are not subject to the compiler's 24-bit multiply narrowing
// (pt29.h, fe29_mul_small_norm).  The carry-out goes to VCC and is never read (the callers'
// bounds guarantee no 64-bit overflow), so there is no SGPR hazard to pad.
S2K_DEV void mad64s(uint64_t& acc, uint32_t a, uint32_t k) {
  asm("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(acc) : "v"(a), "s"(k) : "vcc");
}

}  // namespace s2k
