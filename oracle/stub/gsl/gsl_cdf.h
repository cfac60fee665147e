/* TEST INFRASTRUCTURE ONLY (oracle build).
 *
 * Minimal stand-in for <gsl/gsl_cdf.h> so that the untouched reference sources
 * under /root/reference compile in an image without GSL/Boost.  The reference
 * only uses gsl_cdf_binomial_Q (map_stats.hpp:96, :206), and only for threshold
 * comparisons (cdf < q2, pVal <= cutoff), never for a reported number.
 *
 * Q(k; p, n) = P(X > k), X ~ Binomial(n, p), evaluated by direct summation of
 * the probability mass function in extended precision, starting at the largest
 * term side (i = k+1) and walking up with the ratio recurrence.
 * Cross-checked against scipy.stats.binom.sf in tests/test_stats_lut.py.
 */
#ifndef ORACLE_STUB_GSL_CDF_H
#define ORACLE_STUB_GSL_CDF_H
#include <math.h>

static inline double gsl_cdf_binomial_Q(unsigned int k, double p, unsigned int n)
{
  if (!(p >= 0.0 && p <= 1.0)) return NAN;
  if (k >= n) return 0.0;
  if (p == 0.0) return 0.0;
  if (p == 1.0) return 1.0;
  const long double lp = logl((long double)p);
  const long double lq = log1pl(-(long double)p);
  const long double odds = (long double)p / (1.0L - (long double)p);
  long double acc = 0.0L;
  /* pmf(i) for i = k+1 computed in log space, then recurrence upward */
  unsigned int i = k + 1;
  long double term = expl(lgammal((long double)n + 1.0L) - lgammal((long double)i + 1.0L)
                          - lgammal((long double)(n - i) + 1.0L)
                          + (long double)i * lp + (long double)(n - i) * lq);
  for (;;) {
    acc += term;
    if (i == n) break;
    term *= odds * (long double)(n - i) / (long double)(i + 1);
    ++i;
    if (term < acc * 1e-30L && (long double)i > (long double)n * (long double)p) break;
  }
  return (double)acc;
}
#endif
