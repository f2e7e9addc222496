// seqsum.h -- the value of a SEQUENTIAL float sum, without the sequential chain.
//
// DenseTracker::denseTrackingCpu accepts an LM step iff `float chi2 - float new_chi2 > 0`, both sums accumulated one sample after the other in ONE float
// (dense_tracking.cpp:229-262, 341-383).  Near convergence the difference is far below the rounding noise of a 19 200-term float sum, so the decision --
// and with it the length of the LM loop and the pose the frame ends at -- is a function of the summation ORDER.  To take the reference's decisions the tracker
// needs the reference's sums bit for bit; a dependent chain of n float adds costs n x 8 cycles on one lane (~65 us at n = 19 200: "trk_seq_chi2").
//
// What follows computes the same bits in parallel.  While the accumulator stays inside one binade [2^e, 2^(e+1)) it is an integer A (24 bits) times
// ulp = 2^(e-23), and adding a term t >= 0 in round-to-nearest-even is the integer update
//     A <- A + q + c,    t / ulp = q + r,    c = [r > 1/2]  or, at r = 1/2 exactly,  c = (A + q) & 1        (the tie goes to the even neighbour)
// so a run of terms is the map  A -> A + D[A & 1]  with two integers D[0], D[1] (the parity of A is all a tie can see), and such maps compose associatively.
// The binade at a given position follows from the EXACT prefix sum P there: the float accumulator is within c * 2^-24 (relative; c = nonzero terms so far) of
// P -- the textbook bound of recursive summation, rigorous.  A segment of terms whose interval [P_start (1 - d), P_end (1 + d)] lies inside one binade is
// "safe": its map at that binade is exact and is found by its own lane; the few segments that may straddle a power of two (and the first ones, where the
// accumulator climbs through many binades) are added the slow way, in order, by the walker.  Every assumption is re-checked where it is used (entry binade of a
// run, no carry out of the binade); a violated check makes the caller fall back to the plain sequential sum, so the result is the sequential sum or nothing.
//
// This header is arithmetic only (host + device); the workgroup orchestration is in dense.hip, a host mirror of it in tests/cpp/seqsum_host.cpp.
#pragma once
#include <stdint.h>
#include <string.h>

#if defined(__HIPCC__)
#define SVS_SEQ_HD __host__ __device__ __forceinline__
#else
#define SVS_SEQ_HD static inline
#endif

// the map of a run of terms on the 24-bit mantissa: entry parity 0 -> + d0, entry parity 1 -> + d0 + dd
struct SvsSeqMap { int32_t d0, dd; };

SVS_SEQ_HD uint32_t svs_seq_bits(float f) { uint32_t u; memcpy(&u, &f, 4); return u; }
SVS_SEQ_HD float svs_seq_float(uint32_t u) { float f; memcpy(&f, &u, 4); return f; }

constexpr int32_t SVS_SEQ_CARRY = 1 << 24;      // a mantissa increment that certainly leaves the binade (saturation value of the maps)

// m <- m followed by the term with bits `tb` (a non-negative float), accumulator in the binade with BIASED exponent `eb`.
// Returns true if the term was a tie (rare; the caller may want to know for statistics only)
SVS_SEQ_HD bool svs_seq_add_term(SvsSeqMap &m, uint32_t tb, int eb) {
  if (tb == 0) return false;
  int et = (int)(tb >> 23);
  uint32_t mt = tb & 0x7fffffu;
  if (et == 0) et = 1; else mt |= 0x800000u;      // subnormal terms: exponent of the smallest normal, no hidden bit
  const int shift = eb - et;                       // t / ulp = mt / 2^shift
  if (shift > 24) return false;                    // t < ulp / 2 (mt < 2^24): the accumulator does not move, and no tie is possible
  if (shift < 0) { m.d0 = SVS_SEQ_CARRY; m.dd = 0; return false; }      // t >= 2^(e+1): certainly a carry (never inside a safe segment)
  const uint32_t q = mt >> shift;
  const uint32_t rem = mt & ((1u << shift) - 1u);
  const uint32_t half = (1u << shift) >> 1;        // shift == 0: rem == 0, half == 0 -> exact, no rounding
  const bool tie = shift > 0 && rem == half;
  if (!tie) {
    const int32_t inc = (int32_t)q + (rem > half ? 1 : 0);
    m.d0 = m.d0 + inc < SVS_SEQ_CARRY ? m.d0 + inc : SVS_SEQ_CARRY;
    return false;
  }
  // tie: the parity of A + q decides.  Entry parity 0: A = (even) + d0, entry parity 1: A = (odd) + d0 + dd
  const int32_t c0 = (m.d0 + (int32_t)q) & 1, c1 = (1 + m.d0 + m.dd + (int32_t)q) & 1;
  const int32_t n0 = m.d0 + (int32_t)q + c0;
  m.d0 = n0 < SVS_SEQ_CARRY ? n0 : SVS_SEQ_CARRY;
  m.dd += c1 - c0;
  return true;
}
// The same update with the float unit doing the rounding: for base = 2^e (mantissa bits zero, even) and 0 <= t < 2^e the sum s = base + t stays in the binade, so
// bits(s) - bits(base) IS q + c with the tie going to the even neighbour of an EVEN accumulator; d = s - base and r = t - d are exact (Sterbenz; r is a multiple
// of t's own ulp no larger than t), and |r| = ulp / 2 exactly marks a tie, with the sign of r telling which way the float unit went: r > 0 it rounded down (q even),
// r < 0 up (q odd).  An accumulator of the other parity goes the other way: + sign(r).  Five vector instructions per term off the tie path.
// Only valid where svs_seq_add_term's carry case cannot occur (t < 2^e: every term of a safe segment); eb >= 27.
SVS_SEQ_HD bool svs_seq_add_term_fast(SvsSeqMap &m, float t, int eb) {
  const float base = svs_seq_float((uint32_t)eb << 23), half_ulp = svs_seq_float((uint32_t)(eb - 24) << 23);
  const float s_ = base + t, d = s_ - base, r = t - d;
  const int32_t inc = (int32_t)(svs_seq_bits(s_) - ((uint32_t)eb << 23));
  const bool tie = (r < 0 ? -r : r) == half_ulp;
  if (!tie) { m.d0 += inc; return false; }
  const int32_t sg = r > 0 ? 1 : -1;
  const int32_t par0 = m.d0 & 1, par1 = (1 + m.d0 + m.dd) & 1;
  const int32_t n0 = m.d0 + inc + (par0 ? sg : 0), n1 = m.d0 + m.dd + inc + (par1 ? sg : 0);
  m.d0 = n0; m.dd = n1 - n0;
  return true;
}
// f then g
SVS_SEQ_HD SvsSeqMap svs_seq_compose(const SvsSeqMap &f, const SvsSeqMap &g) {
  const int32_t f0 = f.d0, f1 = f.d0 + f.dd;
  const int32_t g0 = g.d0, g1 = g.d0 + g.dd;
  const int32_t r0 = f0 + ((f0 & 1) ? g1 : g0);
  const int32_t r1 = f1 + (((1 + f1) & 1) ? g1 : g0);
  SvsSeqMap r;
  r.d0 = r0 < SVS_SEQ_CARRY ? r0 : SVS_SEQ_CARRY;
  r.dd = r1 - r0;
  if (r.dd > 1 || r.dd < -1) r.dd = 0;            // only after saturation (the value is then rejected by the carry check anyway)
  return r;
}
// Is the accumulator certainly inside ONE binade while it runs over a segment with exact prefix sums P_s (before) .. P_e (after) and c_s .. c_e nonzero terms
// before / after?  *eb = biased exponent of that binade.  Bound: |acc - P| <= ((1 + u)^(c - 1) - 1) P < 1.0001 c u P for c u < 0.02, u = 2^-24
// (Higham, Accuracy and Stability of Numerical Algorithms, section 4.2); 1e-12 covers the f64 arithmetic of the prefix sums themselves.
SVS_SEQ_HD bool svs_seq_safe(double P_s, double P_e, int c_s, int c_e, int *eb) {
  if (!(c_s >= 1) || !(P_s > 0.0) || c_e > 300000) return false;
  const double u = 5.9604644775390625e-08;
  const double lo = P_s * (1.0 - (1.0001 * c_s * u + 1e-12)), hi = P_e * (1.0 + (1.0001 * c_e * u + 1e-12));
  uint64_t a, b;
  memcpy(&a, &lo, 8); memcpy(&b, &hi, 8);
  const int ea = (int)((a >> 52) & 0x7ff) - 1023, ebb = (int)((b >> 52) & 0x7ff) - 1023;
  if (ea != ebb || ea < -100 || ea > 100) return false;
  *eb = ea + 127;
  return true;
}
// the walker's step over a safe run: acc -> acc after the run, or false if an assumption does not hold (wrong binade at entry, carry out of the binade)
SVS_SEQ_HD bool svs_seq_apply(float *acc, const SvsSeqMap &m, int eb) {
  const uint32_t bits = svs_seq_bits(*acc);
  if ((int)(bits >> 23) != eb) return false;
  const uint32_t A = (bits & 0x7fffffu) | 0x800000u;
  const int64_t A2 = (int64_t)A + ((A & 1u) ? (int64_t)m.d0 + m.dd : (int64_t)m.d0);
  if (A2 >= (int64_t)SVS_SEQ_CARRY || m.d0 >= SVS_SEQ_CARRY) return false;
  *acc = svs_seq_float(((uint32_t)eb << 23) | ((uint32_t)A2 & 0x7fffffu));
  return true;
}
