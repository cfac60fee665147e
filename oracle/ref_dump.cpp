/* TEST INFRASTRUCTURE ONLY — never linked into or called by the product.
 *
 * Header-level harness around the UNTOUCHED reference sources that live under
 * /root/reference/src (compiled where they lie; nothing is copied).  It drives the
 * reference's own Sketch / Map / computeCGI objects exactly the way
 * core_genome_identity() does (src/cgi/core_genome_identity.cpp:46-121, single
 * split) and dumps the intermediate results in binary so that the C restatement
 * (oracle/ani_oracle.c) and the HIP path can be compared field by field.
 *
 * Usage: ref_dump <fastANI args: -q/-r/--ql/--rl [-k] [--fragLen]> -o <prefix>
 * Writes:
 *   <prefix>.params      text: kmerSize windowSize fragLen nRef nQry
 *   <prefix>.minimizers  int32[n][3]  (hash, seqId, wpos)  reference minimizerIndex, position order
 *   <prefix>.seqinfo     int32: nContigs, then len per contig; nFiles, then sequencesByFileInfo
 *   <prefix>.q<i>.frags  int32: nFragments, then for each fragment: s, s sorted unique hashes
 *   <prefix>.q<i>.maps   int32/float[n][11]  skch::MappingResult records in callback order
 *   <prefix>.cgi         (int32 ref, int32 qry, int32 count, int32 total, float identity)[m]
 */
#include <chrono>
#include <omp.h>
#include <iostream>
#include <ctime>
#include <functional>
#include <cstdio>
#include <cstring>

#include "map/include/map_parameters.hpp"
#include "map/include/base_types.hpp"
#include "map/include/parseCmdArgs.hpp"
#include "map/include/winSketch.hpp"
#include "map/include/computeMap.hpp"
#include "map/include/commonFunc.hpp"
#include "cgi/include/computeCoreIdentity.hpp"

static void wr(FILE *f, const void *p, size_t n) { if (n && fwrite(p, 1, n, f) != n) { perror("fwrite"); exit(2); } }

/* --stats <kmerSize> <maxS> <prefix>: dump the reference's own scalar statistics
 * (map_stats.hpp:44-167, computeMap.hpp:375-384) as LUTs:
 *   <prefix>.stats  int32 maxS; then for s=1..maxS: int32 minimumHits(s);
 *                   then for s=1..maxS, shared=0..s: float nucIdentity, float nucIdentityUpperBound
 */
static int dump_stats(int k, int maxS, const char *prefix)
{
  FILE *f = fopen((std::string(prefix) + ".stats").c_str(), "wb");
  int32_t m = maxS; wr(f, &m, 4);
  for (int s = 1; s <= maxS; s++) { int32_t v = skch::Stat::estimateMinimumHitsRelaxed(s, k, 80); wr(f, &v, 4); }
  for (int s = 1; s <= maxS; s++)
    for (int x = 0; x <= s; x++) {
      float mash_dist = skch::Stat::j2md(1.0 * x / s, k);
      float lb = skch::Stat::md_lower_bound(mash_dist, s, k, 0.9);
      float id = 100 * (1 - mash_dist);
      float ub = 100 * (1 - lb);
      wr(f, &id, 4); wr(f, &ub, 4);
    }
  fclose(f);
  return 0;
}

/* --window <k> <fragLen>: print recommendedWindowSize with the CLI's fixed arguments
 * (parseCmdArgs.hpp:225-228) */
static int print_window(int k, int L)
{
  printf("%d\n", skch::Stat::recommendedWindowSize(1e-03, k, 4, 80, L, 5000000));
  return 0;
}

int main(int argc, char **argv)
{
  using namespace std::placeholders;
  if (argc >= 5 && !strcmp(argv[1], "--stats")) return dump_stats(atoi(argv[2]), atoi(argv[3]), argv[4]);
  if (argc >= 4 && !strcmp(argv[1], "--window")) return print_window(atoi(argv[2]), atoi(argv[3]));
  skch::Parameters P;
  skch::parseandSave(argc, argv, P);
  std::string prefix = P.outFileName;
  P.outFileName = "/dev/null";

  {
    FILE *f = fopen((prefix + ".params").c_str(), "w");
    fprintf(f, "%d %d %d %zu %zu\n", P.kmerSize, P.windowSize, P.minReadLength, P.refSequences.size(), P.querySequences.size());
    fclose(f);
  }

  skch::Sketch sk(P);

  {
    FILE *f = fopen((prefix + ".minimizers").c_str(), "wb");
    for (auto it = sk.searchIndex(0, 0); it != sk.getMinimizerIndexEnd(); ++it) {
      int32_t rec[3] = {(int32_t)it->hash, it->seqId, it->wpos};
      wr(f, rec, sizeof rec);
    }
    fclose(f);
    f = fopen((prefix + ".seqinfo").c_str(), "wb");
    int32_t n = (int32_t)sk.metadata.size();
    wr(f, &n, 4);
    for (auto &c : sk.metadata) { int32_t l = c.len; wr(f, &l, 4); }
    n = (int32_t)sk.sequencesByFileInfo.size();
    wr(f, &n, 4);
    for (auto v : sk.sequencesByFileInfo) { int32_t x = v; wr(f, &x, 4); }
    fclose(f);
  }

  std::vector<cgi::CGI_Results> finalResults;
  std::string fileName = "/dev/null";

  for (uint64_t q = 0; q < P.querySequences.size(); q++) {
    /* fragment sketches, produced with the reference's own addMinimizers + sort + unique
     * exactly as doL1Mapping does (computeMap.hpp:260-274) */
    {
      FILE *f = fopen((prefix + ".q" + std::to_string(q) + ".frags").c_str(), "wb");
      std::vector<int32_t> out; out.push_back(0);
      int32_t nfr = 0;
      gzFile fp = gzopen(P.querySequences[q].c_str(), "r");
      kseq_t *seq = kseq_init(fp);
      int len;
      while ((len = kseq_read(seq)) >= 0) {
        if (len < P.windowSize || len < P.kmerSize || len < P.minReadLength) continue;
        int fc = len / P.minReadLength;
        for (int i = 0; i < fc; i++) {
          auto copy = *seq;
          copy.seq.s = seq->seq.s + (size_t)i * P.minReadLength;
          copy.seq.l = P.minReadLength;
          std::vector<skch::MinimizerInfo> mt;
          skch::CommonFunc::addMinimizers(mt, &copy, P.kmerSize, P.windowSize, P.alphabetSize);
          std::sort(mt.begin(), mt.end(), skch::MinimizerInfo::lessByHash);
          auto ue = std::unique(mt.begin(), mt.end(), skch::MinimizerInfo::equalityByHash);
          int32_t s = (int32_t)std::distance(mt.begin(), ue);
          out.push_back(s);
          for (auto it = mt.begin(); it != ue; ++it) out.push_back((int32_t)it->hash);
          nfr++;
        }
      }
      kseq_destroy(seq); gzclose(fp);
      out[0] = nfr;
      wr(f, out.data(), out.size() * 4);
      fclose(f);
    }

    skch::MappingResultsVector_t mapResults;
    uint64_t totalQueryFragments = 0;
    auto fn = std::bind(skch::Map::insertL2ResultsToVec, std::ref(mapResults), _1);
    skch::Map mapper(P, sk, totalQueryFragments, (int)q, fn);
    {
      FILE *f = fopen((prefix + ".q" + std::to_string(q) + ".maps").c_str(), "wb");
      static_assert(sizeof(skch::MappingResult) == 44, "MappingResult layout");
      wr(f, mapResults.data(), mapResults.size() * sizeof(skch::MappingResult));
      fclose(f);
    }
    cgi::computeCGI(P, mapResults, mapper, sk, totalQueryFragments, q, fileName, finalResults);
  }
  {
    FILE *f = fopen((prefix + ".cgi").c_str(), "wb");
    static_assert(sizeof(cgi::CGI_Results) == 20, "CGI_Results layout");
    wr(f, finalResults.data(), finalResults.size() * sizeof(cgi::CGI_Results));
    fclose(f);
  }
  return 0;
}
