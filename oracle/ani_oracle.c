/* TEST INFRASTRUCTURE ONLY — CPU restatement ("oracle") of the FastANI hot path in plain C.
 * See ani_oracle.h for the rules about who may call this and for the parity status (PINNED).
 *
 * Each function cites the reference file:line (relative to /root/reference/src) it restates.
 * Nothing here is copied from the reference: the algorithms are re-expressed over flat arrays.
 */
#include "ani_oracle.h"
#include <stdlib.h>
#include <string.h>
#include <math.h>

/* ------------------------------------------------------------------------------------------
 * Hash: MurmurHash3_x64_128(kmer, k, seed=42), low 32 bits of h1.
 * common/murmur3.h:226-303 (body :245-254, tail :265-285, finalisation :290-298, fmix64 :57-66);
 * called through map/include/commonFunc.hpp:71-81 with seed :32.
 * ------------------------------------------------------------------------------------------ */
static inline uint64_t rotl64(uint64_t x, int r) { return (x << r) | (x >> (64 - r)); }
static inline uint64_t fmix64(uint64_t z)
{
  z ^= z >> 33; z *= 0xff51afd7ed558ccdULL;
  z ^= z >> 33; z *= 0xc4ceb9fe1a85ec53ULL;
  z ^= z >> 33; return z;
}
static inline uint64_t load_le(const uint8_t *p, int nbytes)
{
  uint64_t v = 0;
  for (int i = 0; i < nbytes; i++) v |= (uint64_t)p[i] << (8 * i);
  return v;
}

uint32_t orc_hash_kmer(const uint8_t *kmer, int k)
{
  const uint64_t c1 = 0x87c37b91114253d5ULL, c2 = 0x4cf5ad432745937fULL;
  uint64_t h1 = 42, h2 = 42;
  int nblocks = k / 16;
  for (int b = 0; b < nblocks; b++) {
    uint64_t k1 = load_le(kmer + 16 * b, 8), k2 = load_le(kmer + 16 * b + 8, 8);
    k1 *= c1; k1 = rotl64(k1, 31); k1 *= c2; h1 ^= k1;
    h1 = rotl64(h1, 27); h1 += h2; h1 = h1 * 5 + 0x52dce729;
    k2 *= c2; k2 = rotl64(k2, 33); k2 *= c1; h2 ^= k2;
    h2 = rotl64(h2, 31); h2 += h1; h2 = h2 * 5 + 0x38495ab5;
  }
  const uint8_t *tail = kmer + 16 * nblocks;
  int rem = k & 15;
  if (rem > 8) {
    uint64_t k2 = load_le(tail + 8, rem - 8);
    k2 *= c2; k2 = rotl64(k2, 33); k2 *= c1; h2 ^= k2;
  }
  if (rem > 0) {
    uint64_t k1 = load_le(tail, rem > 8 ? 8 : rem);
    k1 *= c1; k1 = rotl64(k1, 31); k1 *= c2; h1 ^= k1;
  }
  h1 ^= (uint64_t)k; h2 ^= (uint64_t)k;
  h1 += h2; h2 += h1;
  h1 = fmix64(h1); h2 = fmix64(h2);
  h1 += h2;
  return (uint32_t)h1;
}

/* commonFunc.hpp:56-66 */
void orc_upper(uint8_t *seq, size_t len)
{
  for (size_t i = 0; i < len; i++) if (seq[i] > 96 && seq[i] < 123) seq[i] -= 32;
}

/* commonFunc.hpp:37-54 — complement of one byte (only A,C,G,T change) */
static inline uint8_t comp(uint8_t b)
{
  switch (b) { case 'A': return 'T'; case 'C': return 'G'; case 'G': return 'C'; case 'T': return 'A'; default: return b; }
}

/* ------------------------------------------------------------------------------------------
 * Winnowing: commonFunc.hpp:91-167.  Monotone queue over canonical hashes; positions whose forward
 * and reverse hashes are equal are skipped entirely (:131); ties keep the newer k-mer (:142 pops >=);
 * an entry is emitted when the front of the queue is an element that has not been emitted yet (:152-161).
 * ------------------------------------------------------------------------------------------ */
size_t orc_winnow(const uint8_t *seq, int32_t len, int k, int w, int32_t seqId, orc_minimizer_t *out)
{
  /* callers guard len >= w && len >= k (winSketch.hpp:153, computeMap.hpp:138); for shorter input the
   * loop below emits nothing anyway (no k-mer, or no complete window) */
  if (len < k) return 0;
  typedef struct { uint32_t hash; int32_t pos; int32_t emitted_wpos; } qent;
  qent *q = (qent *)malloc(sizeof(qent) * (size_t)(w + 2));
  int head = 0, cnt = 0, capq = w + 2;
  size_t n = 0;
  uint8_t rc[16 * 4];
  int have_last = 0; uint32_t last_hash = 0; int32_t last_wpos = 0;
  for (int32_t i = 0; i + k <= len; i++) {
    int32_t cur_win = i - w + 1;
    uint32_t hf = orc_hash_kmer(seq + i, k);
    for (int j = 0; j < k; j++) rc[j] = comp(seq[i + k - 1 - j]);
    uint32_t hb = orc_hash_kmer(rc, k);
    if (hf == hb) continue;
    uint32_t cur = hf < hb ? hf : hb;
    while (cnt > 0 && q[head].pos <= i - w) { head = (head + 1) % capq; cnt--; }
    while (cnt > 0 && q[(head + cnt - 1) % capq].hash >= cur) cnt--;
    qent e = { cur, i, -1 };
    q[(head + cnt) % capq] = e; cnt++;
    if (cur_win >= 0) {
      qent *f = &q[head];
      /* minimizerIndex.back() != Q.front().first  compares (hash, seqId, wpos) */
      int same = have_last && f->hash == last_hash && f->emitted_wpos == last_wpos;
      if (!same) {
        f->emitted_wpos = cur_win;
        out[n].hash = f->hash; out[n].seqId = seqId; out[n].wpos = cur_win; n++;
        have_last = 1; last_hash = f->hash; last_wpos = cur_win;
      }
    }
  }
  free(q);
  return n;
}

/* ------------------------------------------------------------------------------------------
 * Statistics: map_stats.hpp.  The binomial tail replaces gsl_cdf_binomial_Q (third-party GSL,
 * not vendored; call sites map_stats.hpp:96 and :206) by direct summation.
 * ------------------------------------------------------------------------------------------ */
double orc_binomial_Q(unsigned k, double p, unsigned n)
{
  if (!(p >= 0.0 && p <= 1.0)) return NAN;
  if (k >= n) return 0.0;
  if (p == 0.0) return 0.0;
  if (p == 1.0) return 1.0;
  long double lp = logl((long double)p), lq = log1pl(-(long double)p);
  long double odds = (long double)p / (1.0L - (long double)p);
  unsigned i = k + 1;
  long double term = expl(lgammal(n + 1.0L) - lgammal(i + 1.0L) - lgammal((n - i) + 1.0L) + i * lp + (n - i) * lq);
  long double acc = 0;
  for (;;) {
    acc += term;
    if (i == n) break;
    term *= odds * (long double)(n - i) / (long double)(i + 1);
    ++i;
    if (term < acc * 1e-30L && (long double)i > (long double)n * (long double)p) break;
  }
  return (double)acc;
}

/* map_stats.hpp:44-54 */
float orc_j2md(float j, int k)
{
  if (j == 0) return 1.0f;
  if (j == 1) return 0.0f;
  float mash_dist = (float)((-1.0 / k) * log(2.0 * j / (1 + j)));
  return mash_dist;
}
/* map_stats.hpp:62-66 */
float orc_md2j(float d, int k)
{
  float jaccard = (float)(1.0 / (2.0 * exp(k * d) - 1.0));
  return jaccard;
}
/* map_stats.hpp:79-111 (GSL branch) */
float orc_md_lower_bound(float d, int s, int k, float ci)
{
  float q2 = (float)((1.0 - ci) / 2);
  int x = (int)ceil(s * orc_md2j(d, k)); if (x < 1) x = 1;
  while (x <= s) {
    double cdf_complement = orc_binomial_Q((unsigned)(x - 1), orc_md2j(d, k), (unsigned)s);
    if (cdf_complement < q2) { x--; break; }
    x++;
  }
  float jaccard = (float)x / s;
  return orc_j2md(jaccard, k);
}
/* map_stats.hpp:120-131 */
int orc_min_hits(int s, int k, float perc_identity)
{
  float mash_dist = (float)(1.0 - perc_identity / 100.0);
  float jaccard = orc_md2j(mash_dist, k);
  return (int)ceil(1.0 * s * jaccard);
}
/* map_stats.hpp:142-167 */
int orc_min_hits_relaxed(int s, int k, float perc_identity)
{
  int first = orc_min_hits(s, k, perc_identity);
  int relaxed = first;
  for (int i = first; i >= 0; i--) {
    float jaccard = (float)(1.0 * i / s);
    float d = orc_j2md(jaccard, k);
    float d_lower = orc_md_lower_bound(d, s, k, 0.9f);
    float id_upper = (float)(100.0 * (1.0 - d_lower));
    if (id_upper >= perc_identity) relaxed = i; else break;
  }
  return relaxed;
}
/* map_stats.hpp:179-213 */
static double estimate_pvalue(int s, int k, int alphabet, float identity, int lengthQuery, uint64_t lengthReference)
{
  double kmerSpace = pow(alphabet, k);
  double pX, pY; pX = pY = 1. / (1. + kmerSpace / lengthQuery);
  double r = pX * pY / (pX + pY - pX * pY);
  int x = orc_min_hits_relaxed(s, k, identity);
  double cdf_complement = (x == 0) ? 1.0 : orc_binomial_Q((unsigned)(x - 1), r, (unsigned)s);
  return lengthReference * cdf_complement;
}
/* map_stats.hpp:226-256 */
int orc_recommended_window(double pvalue, int k, int alphabet, float identity, int fragLen, uint64_t refSize)
{
  int optimal = 0, found = 0;
  int firsts[3] = {1, 2, 5};
  for (int t = 0; t < 3 && !found; t++)
    if (estimate_pvalue(firsts[t], k, alphabet, identity, fragLen, refSize) <= pvalue) { optimal = firsts[t]; found = 1; }
  for (int e = 10; e < fragLen && !found; e += 10)
    if (estimate_pvalue(e, k, alphabet, identity, fragLen, refSize) <= pvalue) { optimal = e; found = 1; }
  if (!found) return fragLen; /* the reference reads an uninitialised value here; unreachable for sane inputs */
  int w = (int)(2.0 * fragLen / optimal);
  if (w < 1) w = 1;
  return w < fragLen ? w : fragLen;
}
/* computeMap.hpp:375-381 */
void orc_identity(int shared, int s, int k, float *nucIdentity, float *upperBound)
{
  float mash_dist = orc_j2md((float)(1.0 * shared / s), k);
  float lb = orc_md_lower_bound(mash_dist, s, k, 0.9f);
  *nucIdentity = 100 * (1 - mash_dist);
  *upperBound = 100 * (1 - lb);
}

/* Pure functions of (s, shared, k): memoised so that the oracle is not dominated by the binomial tail.
 * (single-threaded test code; one cache per k, dropped when k changes) */
static int g_cache_k = -1, g_cache_cap = 0;
static int *g_minhits = NULL;           /* [s] or -1 */
static float **g_ident = NULL;          /* [s] -> 2*(s+1) floats or NULL */
static void cache_reset(int k, int need)
{
  if (k != g_cache_k) {
    for (int i = 0; i < g_cache_cap; i++) free(g_ident[i]);
    free(g_ident); free(g_minhits); g_ident = NULL; g_minhits = NULL; g_cache_cap = 0; g_cache_k = k;
  }
  if (need >= g_cache_cap) {
    int ncap = need + 64;
    g_minhits = (int *)realloc(g_minhits, sizeof(int) * (size_t)ncap);
    g_ident = (float **)realloc(g_ident, sizeof(float *) * (size_t)ncap);
    for (int i = g_cache_cap; i < ncap; i++) { g_minhits[i] = -1; g_ident[i] = NULL; }
    g_cache_cap = ncap;
  }
}
static int cached_min_hits_relaxed(int s, int k, float identity)
{
  if (identity != 80.0f) return orc_min_hits_relaxed(s, k, identity);
  cache_reset(k, s);
  if (g_minhits[s] < 0) g_minhits[s] = orc_min_hits_relaxed(s, k, identity);
  return g_minhits[s];
}
static void cached_identity(int shared, int s, int k, float *nucIdentity, float *upperBound)
{
  cache_reset(k, s);
  if (!g_ident[s]) {
    g_ident[s] = (float *)malloc(sizeof(float) * 2 * (size_t)(s + 1));
    for (int x = 0; x <= s; x++) g_ident[s][2 * x] = -1.0f;
  }
  float *e = &g_ident[s][2 * shared];
  if (e[0] < 0.0f) orc_identity(shared, s, k, &e[0], &e[1]);
  *nucIdentity = e[0]; *upperBound = e[1];
}

/* ------------------------------------------------------------------------------------------
 * Reference sketch: winSketch.hpp:124-193.
 * ------------------------------------------------------------------------------------------ */
typedef struct { uint32_t hash; uint32_t idx; } hidx_t;
struct orc_sketch {
  int k, w;
  orc_minimizer_t *mi; size_t n, cap;          /* minimizerIndex, position order (:93) */
  int32_t *contigLen; size_t nContigs, capContigs;  /* metadata[].len (:66) */
  int32_t *seqsByFile; size_t nFiles, capFiles;     /* sequencesByFileInfo (:75) */
  hidx_t *byHash; size_t nUnique;              /* ≙ minimizerPosLookupIndex (:84): sorted (hash, position rank) */
};

orc_sketch *orc_sketch_new(int k, int w)
{
  orc_sketch *sk = (orc_sketch *)calloc(1, sizeof *sk);
  sk->k = k; sk->w = w;
  return sk;
}
void orc_sketch_add_contig(orc_sketch *sk, const uint8_t *seq, int32_t len)
{
  if (sk->nContigs == sk->capContigs) { sk->capContigs = sk->capContigs ? 2 * sk->capContigs : 64; sk->contigLen = (int32_t *)realloc(sk->contigLen, 4 * sk->capContigs); }
  int32_t seqId = (int32_t)sk->nContigs;
  sk->contigLen[sk->nContigs++] = len;                 /* :150 metadata for every contig */
  if (len < sk->w || len < sk->k) return;              /* :153 too short: keeps its seqId, no minimizers */
  if (sk->n + (size_t)len > sk->cap) { sk->cap = (sk->n + (size_t)len) * 2; sk->mi = (orc_minimizer_t *)realloc(sk->mi, sizeof(orc_minimizer_t) * sk->cap); }
  sk->n += orc_winnow(seq, len, sk->k, sk->w, seqId, sk->mi + sk->n);
}
void orc_sketch_end_genome(orc_sketch *sk)
{
  if (sk->nFiles == sk->capFiles) { sk->capFiles = sk->capFiles ? 2 * sk->capFiles : 64; sk->seqsByFile = (int32_t *)realloc(sk->seqsByFile, 4 * sk->capFiles); }
  sk->seqsByFile[sk->nFiles++] = (int32_t)sk->nContigs;   /* :167 */
}
static int cmp_hidx(const void *a, const void *b)
{
  const hidx_t *x = (const hidx_t *)a, *y = (const hidx_t *)b;
  if (x->hash != y->hash) return x->hash < y->hash ? -1 : 1;
  return x->idx < y->idx ? -1 : (x->idx > y->idx);
}
void orc_sketch_finish(orc_sketch *sk)
{
  free(sk->byHash);
  sk->byHash = (hidx_t *)malloc(sizeof(hidx_t) * (sk->n ? sk->n : 1));
  for (size_t i = 0; i < sk->n; i++) { sk->byHash[i].hash = sk->mi[i].hash; sk->byHash[i].idx = (uint32_t)i; }
  qsort(sk->byHash, sk->n, sizeof(hidx_t), cmp_hidx);   /* value lists stay in (seqId,wpos) order (:186-190) */
  size_t u = 0;
  for (size_t i = 0; i < sk->n; i++) if (i == 0 || sk->byHash[i].hash != sk->byHash[i - 1].hash) u++;
  sk->nUnique = u;
}
size_t orc_sketch_size(const orc_sketch *sk) { return sk->n; }
const orc_minimizer_t *orc_sketch_data(const orc_sketch *sk) { return sk->mi; }
size_t orc_sketch_unique(const orc_sketch *sk) { return sk->nUnique; }
void orc_sketch_free(orc_sketch *sk)
{
  if (!sk) return;
  free(sk->mi); free(sk->contigLen); free(sk->seqsByFile); free(sk->byHash); free(sk);
}

/* winSketch.hpp:259-270 — first position-ordered entry with (seqId,wpos) >= (seq,pos) */
static size_t search_index(const orc_sketch *sk, int32_t seq, int32_t pos)
{
  size_t lo = 0, hi = sk->n;
  while (lo < hi) {
    size_t mid = lo + (hi - lo) / 2;
    const orc_minimizer_t *m = &sk->mi[mid];
    int less = (m->seqId < seq) || (m->seqId == seq && m->wpos < pos);
    if (less) lo = mid + 1; else hi = mid;
  }
  return lo;
}

/* ------------------------------------------------------------------------------------------
 * Fragment sketch: computeMap.hpp:260-274 (addMinimizers with seqId 0, sort by hash, unique by hash).
 * ------------------------------------------------------------------------------------------ */
static int cmp_u32(const void *a, const void *b) { uint32_t x = *(const uint32_t *)a, y = *(const uint32_t *)b; return x < y ? -1 : (x > y); }
int orc_fragment_sketch(const uint8_t *frag, int32_t len, int k, int w, uint32_t *out)
{
  orc_minimizer_t *tmp = (orc_minimizer_t *)malloc(sizeof(orc_minimizer_t) * (size_t)(len > 0 ? len : 1));
  size_t n = orc_winnow(frag, len, k, w, 0, tmp);
  for (size_t i = 0; i < n; i++) out[i] = tmp[i].hash;
  free(tmp);
  qsort(out, n, 4, cmp_u32);
  size_t s = 0;
  for (size_t i = 0; i < n; i++) if (i == 0 || out[i] != out[i - 1]) out[s++] = out[i];
  return (int)s;
}

/* ------------------------------------------------------------------------------------------
 * Sliding MinHash set: slidingMap.hpp:112-284, kept literally as an ordered container with a pivot.
 * The container is a sorted array; indices play the role of std::map iterators, so an insertion or
 * erasure in front of the pivot shifts the pivot index to keep it on the same element.
 * ------------------------------------------------------------------------------------------ */
#define NAPOS 2147483647
typedef struct { uint32_t hash; int32_t wposQ, wposR; } sm_ent;
typedef struct { sm_ent *e; int n, cap; int pivot; int shared; } slidemap;

static int sm_find(const slidemap *m, uint32_t h, int *found)
{
  int lo = 0, hi = m->n;
  while (lo < hi) { int mid = (lo + hi) / 2; if (m->e[mid].hash < h) lo = mid + 1; else hi = mid; }
  *found = (lo < m->n && m->e[lo].hash == h);
  return lo;
}
static int sm_is_shared(const slidemap *m, int i) { return m->e[i].wposQ != NAPOS && m->e[i].wposR != NAPOS; }

/* insert_ref :137-161 + updateCountersAfterInsert :231-254 */
static void sm_insert_ref(slidemap *m, uint32_t hash, int32_t wpos)
{
  int found; int at = sm_find(m, hash, &found);
  int status; /* 1 UNIQ, 2 CPLD, 3 REV */
  if (!found) {
    if (m->n == m->cap) { m->cap *= 2; m->e = (sm_ent *)realloc(m->e, sizeof(sm_ent) * (size_t)m->cap); }
    memmove(m->e + at + 1, m->e + at, sizeof(sm_ent) * (size_t)(m->n - at));
    m->e[at].hash = hash; m->e[at].wposQ = NAPOS; m->e[at].wposR = wpos; m->n++;
    if (at <= m->pivot) m->pivot++;           /* iterator keeps pointing at the same element */
    status = 1;
  } else {
    status = (m->e[at].wposR == NAPOS) ? 2 : 3;
    m->e[at].wposR = wpos;
  }
  if (hash <= m->e[m->pivot].hash) {
    if (status == 2) m->shared += 1;
    else if (status == 1) { if (sm_is_shared(m, m->pivot)) m->shared -= 1; m->pivot -= 1; }
  }
}
/* delete_ref :167-211 + updateCountersAfterDelete :261-284 */
static void sm_delete_ref(slidemap *m, uint32_t hash, int32_t wpos)
{
  int found; int at = sm_find(m, hash, &found);
  int status; /* 1 DEL, 2 UPD, 3 NOOP */
  int pivotDeleteCase = 0;
  if (found && m->e[at].wposR == wpos) {
    if (m->e[at].wposQ == NAPOS) {
      if (at == m->pivot) {
        m->pivot++;
        if (sm_is_shared(m, m->pivot)) m->shared += 1;
        pivotDeleteCase = 1;
      }
      memmove(m->e + at, m->e + at + 1, sizeof(sm_ent) * (size_t)(m->n - at - 1));
      m->n--;
      if (at < m->pivot) m->pivot--;          /* same element, shifted left */
      status = 1;
    } else { m->e[at].wposR = NAPOS; status = 2; }
  } else status = 3;
  if (!pivotDeleteCase && hash <= m->e[m->pivot].hash) {
    if (status == 2) m->shared -= 1;
    else if (status == 1) { m->pivot += 1; if (sm_is_shared(m, m->pivot)) m->shared += 1; }
  }
}

/* ------------------------------------------------------------------------------------------
 * One query contig: computeMap.hpp:132-190, doL1Mapping :252-304, computeL1CandidateRegions :313-354,
 * doL2Mapping :363-410, computeL2MappedRegions :418-497, MIIteratorL2::next (MIIteratorL2.hpp:74-96),
 * reportL2Mappings with reportAll=true :504-545.
 * ------------------------------------------------------------------------------------------ */
typedef struct { int32_t seqId, wpos; } hit_t;
static int cmp_hit(const void *a, const void *b)
{
  const hit_t *x = (const hit_t *)a, *y = (const hit_t *)b;
  if (x->seqId != y->seqId) return x->seqId < y->seqId ? -1 : 1;
  return x->wpos < y->wpos ? -1 : (x->wpos > y->wpos);
}
typedef struct { int32_t seqId, start, end; } cand_t;

static void push_mapping(orc_mapping_t **out, size_t *n, size_t *cap, const orc_mapping_t *m)
{
  if (*n == *cap) { *cap = *cap ? *cap * 2 : 1024; *out = (orc_mapping_t *)realloc(*out, sizeof(orc_mapping_t) * *cap); }
  (*out)[(*n)++] = *m;
}

int orc_map_contig(const orc_sketch *sk, const uint8_t *seq, int32_t len, int L, float identity,
                   int32_t fragBase, orc_mapping_t **out, size_t *n, size_t *cap)
{
  const int k = sk->k, w = sk->w;
  if (len < w || len < k || len < L) return 0;                         /* :138 */
  int fragmentCount = len / L;                                         /* :152 */
  uint32_t *qh = (uint32_t *)malloc(4 * (size_t)L);
  hit_t *hits = NULL; size_t hcap = 0;
  cand_t *cands = NULL; size_t ccap = 0;
  slidemap sm; sm.cap = 4 * L + 16; sm.e = (sm_ent *)malloc(sizeof(sm_ent) * (size_t)sm.cap);

  for (int f = 0; f < fragmentCount; f++) {
    const uint8_t *frag = seq + (size_t)f * L;
    int s = orc_fragment_sketch(frag, L, k, w, qh);                    /* :260-274 */
    if (s == 0) continue;                                              /* :278 */
    /* :283-299 seed hits */
    size_t H = 0;
    for (int i = 0; i < s; i++) {
      size_t lo = 0, hi = sk->n;
      while (lo < hi) { size_t mid = lo + (hi - lo) / 2; if (sk->byHash[mid].hash < qh[i]) lo = mid + 1; else hi = mid; }
      for (size_t j = lo; j < sk->n && sk->byHash[j].hash == qh[i]; j++) {
        if (H == hcap) { hcap = hcap ? 2 * hcap : 4096; hits = (hit_t *)realloc(hits, sizeof(hit_t) * hcap); }
        const orc_minimizer_t *m = &sk->mi[sk->byHash[j].idx];
        hits[H].seqId = m->seqId; hits[H].wpos = m->wpos; H++;
      }
    }
    int minimumHits = cached_min_hits_relaxed(s, k, identity);           /* :301 */
    if (minimumHits < 1) minimumHits = 1;                              /* :316 */
    qsort(hits, H, sizeof(hit_t), cmp_hit);                            /* :320 */
    size_t C = 0;
    for (size_t a = 0; a + (size_t)minimumHits <= H; a++) {            /* :322-351 */
      size_t b = a + (size_t)minimumHits - 1;
      if (hits[b].seqId == hits[a].seqId && hits[b].wpos - hits[a].wpos < L) {
        cand_t c; c.seqId = hits[a].seqId;
        c.start = hits[b].wpos - L + 1; if (c.start < 0) c.start = 0;
        c.end = hits[a].wpos;
        if (C > 0 && cands[C - 1].seqId == c.seqId && cands[C - 1].end >= c.start) {
          if (c.end > cands[C - 1].end) cands[C - 1].end = c.end;
        } else {
          if (C == ccap) { ccap = ccap ? 2 * ccap : 256; cands = (cand_t *)realloc(cands, sizeof(cand_t) * ccap); }
          cands[C++] = c;
        }
      }
    }
    /* :363-410 */
    for (size_t ci = 0; ci < C; ci++) {
      const cand_t *c = &cands[ci];
      /* computeL2MappedRegions :418-497 */
      size_t beg = search_index(sk, c->seqId, c->start);
      int32_t cmw = L - (w - 1) - (k - 1);
      size_t end = search_index(sk, c->seqId, sk->mi[beg].wpos + cmw);
      size_t last = search_index(sk, c->seqId, c->end + L);
      /* SlideMapper::init :112-129 */
      sm.n = s;
      for (int i = 0; i < s; i++) { sm.e[i].hash = qh[i]; sm.e[i].wposQ = 0; sm.e[i].wposR = NAPOS; }
      sm.pivot = s - 1; sm.shared = 0;
      for (size_t j = beg; j < end; j++) sm_insert_ref(&sm, sk->mi[j].hash, sk->mi[j].wpos);   /* :448 */
      size_t prev_beg = beg, prev_end = end;
      int32_t sw_pos = sk->mi[beg].wpos;
      int best = 0; int32_t beginOptimalPos = 0, lastOptimalPos = 0;
      while (end < last) {                                             /* :455 */
        if (prev_beg != beg) sm_delete_ref(&sm, sk->mi[prev_beg].hash, sk->mi[prev_beg].wpos);
        if (prev_end != end) sm_insert_ref(&sm, sk->mi[prev_end].hash, sk->mi[prev_end].wpos);
        if (sm.shared > best) { best = sm.shared; beginOptimalPos = lastOptimalPos = sk->mi[beg].wpos; }
        else if (sm.shared == best) lastOptimalPos = sk->mi[beg].wpos;
        prev_beg = beg; prev_end = end;
        /* MIIteratorL2::next */
        int32_t beginPos = sw_pos, lastPos = sw_pos + cmw - 1;
        int32_t d1 = sk->mi[beg + 1].wpos - beginPos, d2 = sk->mi[end].wpos - lastPos;
        int32_t adv = d1 < d2 ? d1 : d2;
        sw_pos += adv;
        if (adv == d1) beg++;
        if (adv == d2) end++;
      }
      int32_t meanOptimalPos = (beginOptimalPos + lastOptimalPos) / 2; /* :496 */
      float nucIdentity, upper;
      cached_identity(best, s, k, &nucIdentity, &upper);                 /* :375-381 */
      if (upper >= identity) {                                          /* :384 */
        orc_mapping_t m;
        m.queryLen = L; m.refStartPos = meanOptimalPos; m.refEndPos = meanOptimalPos + L - 1;
        m.queryStartPos = 0; m.queryEndPos = L - 1;
        m.refSeqId = c->seqId; m.querySeqId = fragBase + f;
        m.nucIdentity = nucIdentity; m.nucIdentityUpperBound = upper;
        m.sketchSize = s; m.conservedSketches = best;
        push_mapping(out, n, cap, &m);
      }
    }
  }
  free(qh); free(hits); free(cands); free(sm.e);
  return fragmentCount;
}

/* ------------------------------------------------------------------------------------------
 * Reducer: computeCoreIdentity.hpp:166-298; comparators cgid_types.hpp:31-53.
 * ------------------------------------------------------------------------------------------ */
typedef struct { int32_t refSequenceId, genomeId, querySeqId, refStartPos, queryStartPos, mapRefPosBin; float nucIdentity; } cgimap_t;

static int cmp_query_bucket(const void *a, const void *b)
{
  const cgimap_t *x = (const cgimap_t *)a, *y = (const cgimap_t *)b;
  if (x->genomeId != y->genomeId) return x->genomeId < y->genomeId ? -1 : 1;
  if (x->querySeqId != y->querySeqId) return x->querySeqId < y->querySeqId ? -1 : 1;
  if (x->nucIdentity != y->nucIdentity) return x->nucIdentity < y->nucIdentity ? -1 : 1;
  if (x->refSequenceId != y->refSequenceId) return x->refSequenceId < y->refSequenceId ? -1 : 1;
  if (x->refStartPos != y->refStartPos) return x->refStartPos < y->refStartPos ? -1 : 1;
  return 0;
}
static int cmp_refbin_bucket(const void *a, const void *b)
{
  const cgimap_t *x = (const cgimap_t *)a, *y = (const cgimap_t *)b;
  if (x->refSequenceId != y->refSequenceId) return x->refSequenceId < y->refSequenceId ? -1 : 1;
  if (x->mapRefPosBin != y->mapRefPosBin) return x->mapRefPosBin < y->mapRefPosBin ? -1 : 1;
  if (x->nucIdentity != y->nucIdentity) return x->nucIdentity < y->nucIdentity ? -1 : 1;
  return 0;
}

size_t orc_compute_cgi(const orc_sketch *sk, const orc_mapping_t *maps, size_t n, int L,
                       int32_t totalQueryFragments, int32_t queryGenomeId, orc_cgi_t *out, size_t cap)
{
  if (n == 0) return 0;
  cgimap_t *v = (cgimap_t *)malloc(sizeof(cgimap_t) * n);
  for (size_t i = 0; i < n; i++) {
    v[i].refSequenceId = maps[i].refSeqId; v[i].querySeqId = maps[i].querySeqId;
    v[i].refStartPos = maps[i].refStartPos; v[i].queryStartPos = maps[i].queryStartPos;
    v[i].mapRefPosBin = maps[i].refStartPos / (L - 20);              /* :194 */
    v[i].nucIdentity = maps[i].nucIdentity;
    /* reviseRefIdToGenomeId :31-42 — upper_bound over sequencesByFileInfo */
    size_t lo = 0, hi = sk->nFiles;
    while (lo < hi) { size_t mid = (lo + hi) / 2; if (sk->seqsByFile[mid] <= v[i].refSequenceId) lo = mid + 1; else hi = mid; }
    v[i].genomeId = (int32_t)lo;
  }
  qsort(v, n, sizeof(cgimap_t), cmp_query_bucket);                     /* :214 */
  size_t n1 = 0;                                                       /* :216-231 keep last of each (genome, fragment) */
  for (size_t i = 0; i < n; i++) {
    if (n1 > 0 && v[i].genomeId == v[n1 - 1].genomeId && v[i].querySeqId == v[n1 - 1].querySeqId) v[n1 - 1] = v[i];
    else v[n1++] = v[i];
  }
  qsort(v, n1, sizeof(cgimap_t), cmp_refbin_bucket);                   /* :237 */
  size_t n2 = 0;                                                       /* :239-254 keep last of each (contig, bin) */
  for (size_t i = 0; i < n1; i++) {
    if (n2 > 0 && v[i].refSequenceId == v[n2 - 1].refSequenceId && v[i].mapRefPosBin == v[n2 - 1].mapRefPosBin) v[n2 - 1] = v[i];
    else v[n2++] = v[i];
  }
  size_t m = 0;
  for (size_t i = 0; i < n2;) {                                        /* :267-297 */
    size_t j = i; float sum = 0.0f;
    while (j < n2 && v[j].genomeId == v[i].genomeId) { sum += v[j].nucIdentity; j++; }
    if (m < cap) {
      out[m].qryGenomeId = queryGenomeId; out[m].refGenomeId = v[i].genomeId;
      out[m].countSeq = (int32_t)(j - i); out[m].totalQueryFragments = totalQueryFragments;
      out[m].identity = sum / out[m].countSeq;
    }
    m++; i = j;
  }
  free(v);
  return m;
}

/* ------------------------------------------------------------------------------------------
 * Synthetic genomes: this repo's deterministic, counter-based generator (DESIGN.md §Synthetic data;
 * cluster structure after SURVEY.md §8d).  The HIP generator (kernels/synth.hpp) computes the same
 * function; tests compare them byte for byte.
 * ------------------------------------------------------------------------------------------ */
static inline uint64_t sm64_fin(uint64_t z)
{
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL;
  return z ^ (z >> 31);
}
static inline uint64_t sm64_stream(uint64_t key, uint64_t p) { return sm64_fin(key + (p + 1) * 0x9E3779B97F4A7C15ULL); }
static const int32_t k_rate_permille[20] = {0, 5, 10, 20, 30, 40, 50, 60, 70, 80, 90, 100, 110, 120, 130, 140, 150, 170, 200, 250};

void orc_synth_genome_v(uint64_t seed, uint64_t variant, int32_t genomeId, int32_t len, uint8_t *outAscii) { orc_synth_genome_c(seed, variant, 20, genomeId, len, outAscii); }

/* clusters of clusterSize genomes: members beyond the 20th cycle through the 19 non-zero divergence rates (CPU twin of
 * fastani_amd/csrc/kernels/synth.hpp: synth_rate_index) */
void orc_synth_genome_c(uint64_t seed, uint64_t variant, int32_t clusterSize, int32_t genomeId, int32_t len, uint8_t *outAscii)
{
  static const char base[4] = {'A', 'C', 'G', 'T'};
  uint64_t root = sm64_fin(seed);
  uint64_t c = (uint64_t)(genomeId / clusterSize), m = (uint64_t)(genomeId % clusterSize);
  if (m >= 20) m = 1 + (m - 20) % 19;
  uint64_t keyAnc = sm64_fin(root + 2 * c);
  uint64_t keyMut = sm64_fin(root + 2 * (uint64_t)genomeId + 1) ^ sm64_fin(variant);
  uint32_t thr = (uint32_t)(((uint64_t)k_rate_permille[m] * 16777216ULL + 500ULL) / 1000ULL);
  for (int32_t p = 0; p < len; p++) {
    uint32_t b = (uint32_t)(sm64_stream(keyAnc, (uint64_t)p) >> 62);
    uint64_t u = sm64_stream(keyMut, (uint64_t)p);
    if ((uint32_t)(u >> 40) < thr) {
      uint32_t sub = 1 + (uint32_t)((((u >> 8) & 0xFFFF) * 3) >> 16);
      b = (b + sub) & 3;
    }
    outAscii[p] = (uint8_t)base[b];
  }
}

void orc_synth_genome(uint64_t seed, int32_t genomeId, int32_t len, uint8_t *outAscii) { orc_synth_genome_v(seed, 0, genomeId, len, outAscii); }
