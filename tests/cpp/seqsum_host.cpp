// Host mirror of the workgroup orchestration around scavislam_amd/csrc/seqsum.h (dense.hip: exact_seq_sum_f32): NT "lanes" run as loops.
// Built as a shared library by tests/test_seqsum_cpu.py and driven from NumPy:  the emulated sum must equal the plain sequential float sum bit for bit.
#include <stdint.h>
#include <stdlib.h>
#include <vector>

#include "../../scavislam_amd/csrc/seqsum.h"

extern "C" {
// the sum the reference forms: one float accumulator, terms in order (dense_tracking.cpp:229-262)
float svs_host_seq_sum_plain(const float *t, int n) {
  volatile float acc = 0.f;
  for (int i = 0; i < n; ++i) acc = acc + t[i];
  return acc;
}
// stats[0] = terms added the slow way by the walker, stats[1] = segments that were not safe, stats[2] = 1 if a check failed (result from the plain sum),
// stats[3] = ties seen, stats[4] = 1 if svs_seq_add_term_fast disagreed with svs_seq_add_term on a safe segment
float svs_host_seq_sum_emulated(const float *t, int n, int NT, int *stats) {
  const int S = (n + NT - 1) / NT > 0 ? (n + NT - 1) / NT : 1;
  std::vector<double> psum(NT, 0.0), P_s(NT), P_e(NT);
  std::vector<int> cnt(NT, 0), c_s(NT), c_e(NT), eb(NT, 0);
  std::vector<char> safe(NT, 0);
  std::vector<SvsSeqMap> map(NT), pref(NT);
  int ties = 0, mismatch = 0;
  for (int k = 0; k < NT; ++k)
    for (int j = k * S; j < n && j < (k + 1) * S; ++j) { psum[k] += (double)t[j]; cnt[k] += t[j] != 0.f; }
  double P = 0; int c = 0;
  for (int k = 0; k < NT; ++k) { P_s[k] = P; c_s[k] = c; P += psum[k]; c += cnt[k]; P_e[k] = P; c_e[k] = c; }
  for (int k = 0; k < NT; ++k) {
    const bool empty = k * S >= n;
    int e = 0;
    safe[k] = empty ? 1 : svs_seq_safe(P_s[k], P_e[k], c_s[k], c_e[k], &e);
    eb[k] = empty ? -1 : e;                      // an empty segment is the identity in whatever binade
    map[k].d0 = 0; map[k].dd = 0;
    // (two neighbouring safe segments share their binade -- P_s[k + 1] = P_e[k] -- but nothing below relies on the proof: a change of binade ends the run)
    if (safe[k] && !empty && k > 0 && safe[k - 1] && eb[k - 1] != e) safe[k] = 0;
    if (safe[k] && !empty) {
      SvsSeqMap fast{0, 0};      // the float-unit form of the same update (what the kernel runs) must give the same map
      for (int j = k * S; j < n && j < (k + 1) * S; ++j) { ties += svs_seq_add_term(map[k], svs_seq_bits(t[j]), e); svs_seq_add_term_fast(fast, t[j], e); }
      if (map[k].d0 < SVS_SEQ_CARRY && (fast.d0 != map[k].d0 || fast.dd != map[k].dd)) mismatch = 1;
    }
  }
  // segmented inclusive scan (reset behind every unsafe segment), Hillis-Steele as the wave does it
  std::vector<char> flag(NT);
  for (int k = 0; k < NT; ++k) { pref[k] = map[k]; flag[k] = !safe[k] || (k % 64) == 0; }
  for (int d = 1; d < 64; d <<= 1) {
    std::vector<SvsSeqMap> np(pref); std::vector<char> nf(flag);
    for (int k = 0; k < NT; ++k) {
      if ((k % 64) < d) continue;
      if (!flag[k]) { np[k] = svs_seq_compose(pref[k - d], pref[k]); nf[k] = flag[k - d]; }
    }
    pref.swap(np); flag.swap(nf);
  }
  // the walker
  float acc = 0.f;
  int slow = 0, unsafe = 0;
  bool ok = true;
  const int nseg = (n + S - 1) / S;
  for (int chunk = 0; chunk * 64 < nseg && ok; ++chunk) {
    const int limit = nseg - chunk * 64 < 64 ? nseg - chunk * 64 : 64;
    int pos = 0;
    while (pos < limit && ok) {
      int nxt = pos;
      while (nxt < limit && safe[chunk * 64 + nxt]) ++nxt;
      if (nxt > pos) {
        const int last = chunk * 64 + nxt - 1;
        ok = svs_seq_apply(&acc, pref[last], eb[last]);
      }
      if (ok && nxt < limit) {
        const int seg = chunk * 64 + nxt;
        ++unsafe;
        for (int j = seg * S; j < n && j < (seg + 1) * S; ++j) { volatile float a = acc + t[j]; acc = a; ++slow; }
      }
      pos = nxt + 1;
    }
  }
  if (stats) { stats[0] = slow; stats[1] = unsafe; stats[2] = !ok; stats[3] = ties; stats[4] = mismatch; }
  return ok ? acc : svs_host_seq_sum_plain(t, n);
}
}
