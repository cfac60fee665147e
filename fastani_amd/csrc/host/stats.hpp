// stats.hpp — host-side scalar statistics of the mapping stage, evaluated once and uploaded as look-up tables.
//
// Restates skch::Stat (src/map/include/map_stats.hpp): j2md :44-54, md2j :62-66, md_lower_bound :79-111 (GSL branch),
// estimateMinimumHits :120-131, estimateMinimumHitsRelaxed :142-167, estimate_pvalue :179-213,
// recommendedWindowSize :226-256, and the identity expressions of skch::Map::doL2Mapping (computeMap.hpp:375-384).
// Float/double types follow the reference expression by expression so that the thresholds and the reported
// identities are bit-identical.  gsl_cdf_binomial_Q (GSL, not vendored by the reference) is replaced by an exact
// summation of the upper tail; it only feeds comparisons (map_stats.hpp:98, :245).
#pragma once
#include <cmath>
#include <cstdint>
#include <cstring>
#include <vector>

namespace ani {
namespace stat {

// P(X > k), X ~ Binomial(n, p)
inline double binomial_upper_tail(unsigned k, double p, unsigned n)
{
  if (!(p >= 0.0 && p <= 1.0)) return NAN;
  if (k >= n || p == 0.0) return 0.0;
  if (p == 1.0) return 1.0;
  const long double lp = std::log((long double)p), lq = log1pl(-(long double)p);
  const long double odds = (long double)p / (1.0L - (long double)p);
  unsigned i = k + 1;
  long double term = expl(lgammal(n + 1.0L) - lgammal(i + 1.0L) - lgammal((n - i) + 1.0L) + i * lp + (n - i) * lq);
  long double acc = 0.0L;
  for (;;) {
    acc += term;
    if (i == n) break;
    term *= odds * (long double)(n - i) / (long double)(i + 1);
    ++i;
    if (term < acc * 1e-30L && (long double)i > (long double)n * (long double)p) break;
  }
  return (double)acc;
}

inline float j2md(float j, int k)
{
  if (j == 0) return 1.0f;
  if (j == 1) return 0.0f;
  float mash_dist = (float)((-1.0 / k) * std::log(2.0 * j / (1 + j)));
  return mash_dist;
}

inline float md2j(float d, int k)
{
  float jaccard = (float)(1.0 / (2.0 * std::exp((double)(k * d)) - 1.0));
  return jaccard;
}

inline float md_lower_bound(float d, int s, int k, float ci)
{
  const float q2 = (float)((1.0 - ci) / 2);
  const float jd = md2j(d, k);
  int x = (int)std::ceil(s * jd);
  if (x < 1) x = 1;
  while (x <= s) {
    const double tail = binomial_upper_tail((unsigned)(x - 1), jd, (unsigned)s);
    if (tail < q2) { x--; break; }
    x++;
  }
  const float jaccard = (float)x / s;
  return j2md(jaccard, k);
}

inline int estimate_minimum_hits(int s, int k, float perc_identity)
{
  const float mash_dist = (float)(1.0 - perc_identity / 100.0);
  const float jaccard = md2j(mash_dist, k);
  return (int)std::ceil(1.0 * s * jaccard);
}

inline int estimate_minimum_hits_relaxed(int s, int k, float perc_identity)
{
  const int first = estimate_minimum_hits(s, k, perc_identity);
  int relaxed = first;
  for (int i = first; i >= 0; i--) {
    const float jaccard = (float)(1.0 * i / s);
    const float d = j2md(jaccard, k);
    const float d_lower = md_lower_bound(d, s, k, 0.9f);
    const float id_upper = (float)(100.0 * (1.0 - d_lower));
    if (id_upper >= perc_identity) relaxed = i; else break;
  }
  return relaxed;
}

inline double estimate_pvalue(int s, int k, int alphabetSize, float identity, int lengthQuery, uint64_t lengthReference)
{
  const double kmerSpace = std::pow((double)alphabetSize, (double)k);
  const double pX = 1. / (1. + kmerSpace / lengthQuery), pY = pX;
  const double r = pX * pY / (pX + pY - pX * pY);
  const int x = estimate_minimum_hits_relaxed(s, k, identity);
  const double tail = (x == 0) ? 1.0 : binomial_upper_tail((unsigned)(x - 1), r, (unsigned)s);
  return lengthReference * tail;
}

inline int recommended_window_size(double pValue_cutoff, int k, int alphabetSize, float identity, int lengthQuery,
                                   uint64_t lengthReference)
{
  int optimal = -1;
  const int head[3] = {1, 2, 5};
  for (int t = 0; t < 3 && optimal < 0; t++)
    if (estimate_pvalue(head[t], k, alphabetSize, identity, lengthQuery, lengthReference) <= pValue_cutoff) optimal = head[t];
  for (int e = 10; e < lengthQuery && optimal < 0; e += 10)
    if (estimate_pvalue(e, k, alphabetSize, identity, lengthQuery, lengthReference) <= pValue_cutoff) optimal = e;
  if (optimal < 0) return lengthQuery;   // the reference reads an unset variable here; not reachable with CLI-legal values
  int w = (int)(2.0 * lengthQuery / optimal);
  if (w < 1) w = 1;
  return w < lengthQuery ? w : lengthQuery;
}

// computeMap.hpp:375-381
inline void identity(int shared, int s, int k, float *nucIdentity, float *nucIdentityUpperBound)
{
  const float mash_dist = j2md((float)(1.0 * shared / s), k);
  *nucIdentity = 100 * (1 - mash_dist);
  if (nucIdentityUpperBound) {
    const float lb = md_lower_bound(mash_dist, s, k, 0.9f);
    *nucIdentityUpperBound = 100 * (1 - lb);
  }
}

// Tables indexed by sketch size s (1..maxS): minimumHits(s) (computeMap.hpp:301), the smallest shared count whose
// upper identity bound passes the cut-off (:384), and the flattened identity table idLUT[lut_off(s) + shared].
struct Luts {
  int k = 0; float identityCutoff = 0; int maxS = 0;
  std::vector<int32_t> minHits;      // [maxS+1]
  std::vector<int32_t> minShared;    // [maxS+1]
  std::vector<uint32_t> idBits;      // [lut_off(maxS+1)]

  static uint32_t off(int s) { return (uint32_t)(((int64_t)(s - 1) * (s + 2)) / 2); }

  void extend(int k_, float cutoff, int newMaxS)
  {
    if (k_ != k || cutoff != identityCutoff) { k = k_; identityCutoff = cutoff; maxS = 0; minHits.assign(1, 0); minShared.assign(1, 0); idBits.clear(); }
    if (newMaxS <= maxS) return;
    minHits.resize(newMaxS + 1); minShared.resize(newMaxS + 1); idBits.resize(off(newMaxS + 1));
    for (int s = maxS + 1; s <= newMaxS; s++) {
      minHits[s] = estimate_minimum_hits_relaxed(s, k, cutoff);
      int ms = s + 1;                               // "never passes"
      for (int x = 0; x <= s; x++) {
        float id, ub;
        identity(x, s, k, &id, nullptr);
        uint32_t bits; std::memcpy(&bits, &id, 4);
        idBits[off(s) + x] = bits;
        if (ms > s) {                               // bound is monotone in x: stop evaluating it after the first pass
          identity(x, s, k, &id, &ub);
          if (ub >= cutoff) ms = x;
        }
      }
      minShared[s] = ms;
    }
    maxS = newMaxS;
  }
};

}  // namespace stat
}  // namespace ani
