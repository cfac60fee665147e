// fastANI — drop-in command line over the MI355X-native engine (C-ABI in include/ani_abi.h).
//
// Mirrors the reference driver core_genome_identity() (src/cgi/core_genome_identity.cpp:27-167): same flags and
// defaults (src/map/include/parseCmdArgs.hpp:114-234), same output files and formats (cgi::outputCGI
// src/cgi/include/computeCoreIdentity.hpp:307-344, cgi::outputPhylip :353-448, cgi::outputVisualizationFile :103-153,
// cgi::computeGenomeLengths :48-92), same error messages and exit codes.  Sketch / Map / computeCGI run on the GPU through
// the C-ABI; this file is host-side text I/O only.  `-t` is accepted for compatibility: results do not depend on it
// (tests/fastani_tests.cpp:199-255), except that with `-s` the reference checks each of its T reference splits separately
// (core_genome_identity.cpp:76-79), which is reproduced by sketching the same round-robin splits.
#include <zlib.h>

#include <algorithm>
#include <atomic>
#include <thread>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <iostream>
#include <sstream>
#include <string>
#include <tuple>
#include <unordered_map>
#include <vector>

#include "ani_abi.h"

namespace {

struct Options {
  int kmerSize = 16, fragLen = 3000, threads = 1;
  float minFraction = 0.2f, maxRatioDiff = 100.0f;     // parseCmdArgs.hpp:121,:128 (the help text says 10.0; the code sets 100.0)
  bool visualize = false, matrix = false, sanityCheck = false;
  std::vector<std::string> refs, queries;
  std::string out;
};

[[noreturn]] void usage(const char *argv0, int code)
{
  std::cout <<
    "-----------------\n"
    "fastANI is a fast alignment-free implementation for computing whole-genome Average Nucleotide Identity (ANI) between genomes\n"
    "-----------------\n"
    "Example usage:\n$ fastANI -q genome1.fa -r genome2.fa -o output.txt\n$ fastANI -q genome1.fa --rl genome_list.txt -o output.txt\n\n"
    "SYNOPSIS\n"
    "     " << argv0 << " [-h] [-r <value>] [--rl <value>] [-q <value>] [--ql <value>] [-k <value>] [-t <value>]\n"
    "             [--fragLen <value>] [--minFraction <value>] [--maxRatioDiff <value>] [--visualize] [--matrix]\n"
    "             [-o <value>] [-s] [-v]\n\n"
    "OPTIONS\n"
    "     -h, --help  print this help page\n"
    "     -r, --ref <value>  reference genome (fasta/fastq)[.gz]\n"
    "     --rl, --refList <value>  a file containing list of reference genome files, one genome per line\n"
    "     -q, --query <value>  query genome (fasta/fastq)[.gz]\n"
    "     --ql, --queryList <value>  a file containing list of query genome files, one genome per line\n"
    "     -k, --kmer <value>  kmer size <= 16 [default : 16]\n"
    "     -t, --threads <value>  thread count for parallel execution [default : 1]\n"
    "     --fragLen <value>  fragment length [default : 3,000]\n"
    "     --minFraction <value>  minimum fraction of genome that must be shared for trusting ANI. [default : 0.2]\n"
    "     --maxRatioDiff <value>  maximum difference between (Total Ref. Length/Total Occ. Hashes) and (Total Ref. Length/Total No. Hashes). [default : 10.0]\n"
    "     --visualize  output mappings for visualization [disabled by default]\n"
    "     --matrix    also output ANI values as lower triangular matrix (.matrix) [disabled by default]\n"
    "     -o, --output <value>  output file name\n"
    "     -s, --sanityCheck  run sanity check\n"
    "     -v, --version  show version\n" << std::endl;
  exit(code);
}

// parseCmdArgs.hpp:30-52
void parseFileList(const std::string &fileToRead, std::vector<std::string> &out)
{
  std::ifstream in(fileToRead);
  if (in.fail()) { std::cerr << "ERROR, skch::parseFileList, Could not open " << fileToRead << "\n"; exit(1); }
  std::string line;
  while (std::getline(in, line)) {
    size_t b = 0, e = line.size();
    while (b < e && std::isspace((unsigned char)line[b])) b++;
    while (e > b && std::isspace((unsigned char)line[e - 1])) e--;
    if (e > b) out.push_back(line.substr(b, e - b));
  }
}

std::string listStr(const std::vector<std::string> &v)      // prettyprint.hpp style: [a, b]
{
  std::string s = "[";
  for (size_t i = 0; i < v.size(); i++) { if (i) s += ", "; s += v[i]; }
  return s + "]";
}

Options parse(int argc, char **argv)
{
  Options o;
  std::string refName, refList, qryName, qryList;
  bool help = false, version = false;
  auto need = [&](int &i) -> const char * { if (i + 1 >= argc) usage(argv[0], 1); return argv[++i]; };
  for (int i = 1; i < argc; i++) {
    const std::string a = argv[i];
    if (a == "-h" || a == "--help") help = true;
    else if (a == "-r" || a == "--ref") refName = need(i);
    else if (a == "--rl" || a == "--refList") refList = need(i);
    else if (a == "-q" || a == "--query") qryName = need(i);
    else if (a == "--ql" || a == "--queryList") qryList = need(i);
    else if (a == "-k" || a == "--kmer") o.kmerSize = atoi(need(i));
    else if (a == "-t" || a == "--threads") o.threads = atoi(need(i));
    else if (a == "--fragLen") o.fragLen = atoi(need(i));
    else if (a == "--minFraction") o.minFraction = (float)atof(need(i));
    else if (a == "--maxRatioDiff") o.maxRatioDiff = (float)atof(need(i));
    else if (a == "--visualize") o.visualize = true;
    else if (a == "--matrix") o.matrix = true;
    else if (a == "-o" || a == "--output") o.out = need(i);
    else if (a == "-s" || a == "--sanityCheck") o.sanityCheck = true;
    else if (a == "-v" || a == "--version") version = true;
    else usage(argv[0], 1);
  }
  if (help) usage(argv[0], 0);
  if (version) { std::cerr << "version 1.33\n\n"; exit(0); }                         // parseCmdArgs.hpp:194-198
  if (refName.empty() && refList.empty()) { std::cerr << "Provide reference file (s)\n"; exit(1); }
  if (qryName.empty() && qryList.empty()) { std::cerr << "Provide query file (s)\n"; exit(1); }
  if (!refName.empty()) o.refs.push_back(refName); else parseFileList(refList, o.refs);
  if (!qryName.empty()) o.queries.push_back(qryName); else parseFileList(qryList, o.queries);
  if (o.threads < 1) o.threads = 1;
  return o;
}

// ---------------------------------------------------------------------------------------------------------------
// FASTA/FASTQ(.gz) records with the semantics of the vendored kseq.h (src/common/kseq.h:176-214): the name ends at the first
// white space, sequence lines are appended whole (minus the line terminator) until a line starts with '>', '@' or '+'.
// ---------------------------------------------------------------------------------------------------------------
struct GzReader {
  gzFile fp; unsigned char buf[1 << 16]; int n = 0, p = 0; bool eof = false;
  explicit GzReader(const std::string &path) { fp = gzopen(path.c_str(), "r"); }
  ~GzReader() { if (fp) gzclose(fp); }
  int getc()
  {
    if (p >= n) {
      if (eof || !fp) return -1;
      n = gzread(fp, buf, sizeof buf); p = 0;
      if (n <= 0) { eof = true; return -1; }
    }
    return buf[p++];
  }
};

struct Record { std::string name; std::string seq; };

struct SeqReader {
  GzReader gz; int last = 0;
  explicit SeqReader(const std::string &path) : gz(path) {}
  // returns false at end of file / malformed FASTQ (kseq_read < 0)
  bool next(Record &r)
  {
    int c;
    if (last == 0) {
      while ((c = gz.getc()) >= 0 && c != '>' && c != '@') {}
      if (c < 0) return false;
      last = c;
    }
    r.name.clear(); r.seq.clear();
    while ((c = gz.getc()) >= 0 && !std::isspace(c)) r.name.push_back((char)c);
    if (c < 0 && r.name.empty()) return false;
    if (c >= 0 && c != '\n') while ((c = gz.getc()) >= 0 && c != '\n') {}
    while ((c = gz.getc()) >= 0 && c != '>' && c != '+' && c != '@') {
      if (c == '\n') continue;
      r.seq.push_back((char)c);
      while ((c = gz.getc()) >= 0 && c != '\n') r.seq.push_back((char)c);
      if (!r.seq.empty() && r.seq.back() == '\r') r.seq.pop_back();
    }
    if (c == '>' || c == '@') last = c;
    if (c != '+') { if (c < 0) last = 0; return true; }
    while ((c = gz.getc()) >= 0 && c != '\n') {}
    if (c < 0) return false;
    size_t q = 0;
    while (q < r.seq.size()) {
      size_t line = 0;
      while ((c = gz.getc()) >= 0 && c != '\n') line++;
      q += line;
      if (c < 0) break;
    }
    last = 0;
    return q == r.seq.size();
  }
};

struct Genome { std::vector<std::string> names; std::vector<int32_t> lens; };

struct FileData { std::vector<uint8_t> data; Genome g; };

FileData readFile(const std::string &path)
{
  FileData fd;
  SeqReader rd(path);
  Record r;
  while (rd.next(r)) {
    if (r.seq.size() >= 0x7fffffffull) { std::cerr << "ERROR, contig of " << r.seq.size() << " bases in " << path << " exceeds the 2^31 limit of offset_t" << std::endl; exit(1); }
    fd.data.insert(fd.data.end(), r.seq.begin(), r.seq.end());
    fd.g.names.push_back(r.name); fd.g.lens.push_back((int32_t)r.seq.size());
  }
  return fd;
}

// gz + FASTA parsing is the wall-clock floor of a run once the kernels are fast: files are parsed on `threads` host threads
// (this is what -t buys here; the reference uses it to split the references over OpenMP threads)
std::vector<FileData> readFiles(const std::vector<std::string> &paths, int threads)
{
  std::vector<FileData> out(paths.size());
  std::atomic<size_t> next{0};
  const int nt = std::max(1, std::min<int>(threads, (int)paths.size()));
  auto work = [&]() { for (size_t i; (i = next.fetch_add(1)) < paths.size();) out[i] = readFile(paths[i]); };
  if (nt == 1) work();
  else { std::vector<std::thread> th; for (int t = 0; t < nt; t++) th.emplace_back(work); for (auto &x : th) x.join(); }
  return out;
}

struct HostBatch {
  std::vector<uint8_t> data; std::vector<int64_t> off; std::vector<int32_t> len, gcs{0};
  std::vector<Genome> meta;
  void add(FileData &&fd)
  {
    int64_t o = (int64_t)data.size();
    for (int32_t l : fd.g.lens) { off.push_back(o); len.push_back(l); o += l; }
    data.insert(data.end(), fd.data.begin(), fd.data.end());
    gcs.push_back((int32_t)len.size());
    meta.push_back(std::move(fd.g));
  }
  void add(const std::string &path) { add(readFile(path)); }
  ani_seq_batch_t batch() const
  {
    static const uint8_t dummy = 0;
    ani_seq_batch_t b;
    b.layout = ANI_SEQ_HOST_ASCII; b.nGenomes = (int32_t)meta.size(); b.nContigs = (int32_t)len.size();
    b.genomeContigStart = gcs.data(); b.contigOffset = off.data(); b.contigLen = len.data();
    b.data = data.empty() ? &dummy : data.data();
    return b;
  }
};

void die(const char *what) { std::cerr << "ERROR, " << what << ": " << ani_last_error() << std::endl; exit(1); }

struct VisRow { std::string q, r; float id; int64_t qs, qe, rs, re; };

}  // namespace

int main(int argc, char **argv)
{
  Options o = parse(argc, argv);
  ani_params_t ap;
  if (ani_params_default(&ap, o.kmerSize, o.fragLen)) die("parameters");
  // parseCmdArgs.hpp:96-107
  std::cerr << ">>>>>>>>>>>>>>>>>>" << std::endl;
  std::cerr << "Reference = " << listStr(o.refs) << std::endl;
  std::cerr << "Query = " << listStr(o.queries) << std::endl;
  std::cerr << "Kmer size = " << ap.kmerSize << std::endl;
  std::cerr << "Fragment length = " << ap.fragLen << std::endl;
  std::cerr << "Threads = " << o.threads << std::endl;
  std::cerr << "ANI output file = " << o.out << std::endl;
  std::cerr << "Sanity Check  = " << o.sanityCheck << std::endl;
  std::cerr << ">>>>>>>>>>>>>>>>>>" << std::endl;
  // validateInputFiles, parseCmdArgs.hpp:59-88
  if (o.queries.empty() || o.refs.empty()) { std::cerr << "ERROR, skch::validateInputFiles, Count of query and ref genomes should be non-zero" << std::endl; exit(1); }
  for (auto &e : o.queries) { std::ifstream in(e); if (in.fail()) { std::cerr << "ERROR, skch::validateInputFiles, Could not open " << e << std::endl; exit(1); } }
  for (auto &e : o.refs) { std::ifstream in(e); if (in.fail()) { std::cerr << "ERROR, skch::validateInputFiles, Could not open " << e << std::endl; exit(1); } }

  ani_ctx *ctx = nullptr;
  if (ani_init(0, &ctx)) die("ani_init");
  std::cerr << "INFO [thread 0], skch::Sketch::build, window size for minimizer sampling  = " << ap.windowSize << std::endl;

  // ---- queries are read once (the reference re-reads them in every thread) ----
  HostBatch Q;
  { auto files = readFiles(o.queries, o.threads); for (auto &fd : files) Q.add(std::move(fd)); }
  ani_seq_batch_t qb = Q.batch();

  // ---- reference splits: one sketch, unless -s asks for the reference's per-split sanity check ----
  const int nSplits = o.sanityCheck ? o.threads : 1;
  std::vector<ani_cgi_t> finalResults;
  std::vector<VisRow> vis;
  std::vector<int> failedSplits; std::vector<float> failedRatio;
  for (int sp = 0; sp < nSplits; sp++) {
    std::vector<int> refIdx;
    for (int j = 0; j < (int)o.refs.size(); j++) if (nSplits == 1 || j % nSplits == sp) refIdx.push_back(j);   // computeCoreIdentity.hpp:467-472
    if (refIdx.empty()) continue;                                       // more threads than references: an empty split maps nothing
    HostBatch R;
    { std::vector<std::string> paths; for (int j : refIdx) paths.push_back(o.refs[j]);
      auto files = readFiles(paths, o.threads); for (auto &fd : files) R.add(std::move(fd)); }
    ani_seq_batch_t rb = R.batch();
    ani_sketch *sk = nullptr;
    if (ani_sketch_build(ctx, &ap, &rb, &sk)) die("ani_sketch_build");
    uint64_t occ = 0, uniq = 0, tot = 0;
    ani_sketch_stats(sk, &occ, &uniq, &tot, nullptr, nullptr);
    if (sp == 0) {
      std::cerr << "INFO [thread 0], skch::Sketch::build, minimizers picked from reference = " << occ << std::endl;
      std::cerr << "INFO [thread 0], skch::Sketch::index, unique minimizers = " << uniq << std::endl;
    }
    bool ok = true;
    if (o.sanityCheck) {                                                // winSketch.hpp:298-318
      const float hashRatio = float(tot) / float(occ), uniqHashRatio = float(tot) / float(uniq);
      const float diff = std::abs(hashRatio - uniqHashRatio);
      if (diff > o.maxRatioDiff) { ok = false; failedSplits.push_back(sp); failedRatio.push_back(diff); }
    }
    if (ok) {
      if (!o.visualize) {
        ani_cgi_t *rows = nullptr; size_t m = 0;
        if (ani_map_cgi_batch(ctx, sk, &qb, 0, &rows, &m)) die("ani_map_cgi_batch");
        for (size_t i = 0; i < m; i++) { ani_cgi_t e = rows[i]; e.refGenomeId = refIdx[e.refGenomeId]; finalResults.push_back(e); }
        ani_free(rows);
      } else {
        // per query genome: mappings back to the host for the .visual rows (computeCoreIdentity.hpp:186-261)
        std::vector<int64_t> refOff(R.len.size(), 0);
        for (size_t i = 1; i < R.len.size(); i++) refOff[i] = refOff[i - 1] + R.len[i - 1];
        std::vector<int32_t> contigGenome(R.len.size());
        for (size_t g = 0; g + 1 < R.gcs.size(); g++) for (int32_t c = R.gcs[g]; c < R.gcs[g + 1]; c++) contigGenome[c] = (int32_t)g;
        for (size_t qi = 0; qi < Q.meta.size(); qi++) {
          HostBatch one; one.add(o.queries[qi]);          // --visualize is a one-to-one mode: re-reading the query is cheap
          ani_seq_batch_t ob = one.batch();
          ani_mapping_t *maps = nullptr; size_t n = 0; uint64_t totalFr = 0;
          if (ani_map_query(ctx, sk, &ob, &maps, &n, &totalFr)) die("ani_map_query");
          ani_cgi_t *rows = nullptr; size_t m = 0;
          if (ani_compute_cgi(ctx, sk, maps, n, totalFr, (int32_t)qi, &rows, &m)) die("ani_compute_cgi");
          for (size_t i = 0; i < m; i++) { ani_cgi_t e = rows[i]; e.refGenomeId = refIdx[e.refGenomeId]; finalResults.push_back(e); }
          ani_free(rows);
          // fragment offsets inside the query genome (computeMap.hpp:160-167, computeCoreIdentity.hpp:117-124)
          std::vector<int64_t> qOff;
          { int64_t run = 0;
            for (int32_t c = Q.gcs[qi]; c < Q.gcs[qi + 1]; c++) {
              const int32_t len = Q.len[c];
              if (len < ap.windowSize || len < ap.kmerSize || len < ap.fragLen) { qOff.push_back(run); run += len; continue; }
              const int fc = len / ap.fragLen;
              for (int i = 0; i < fc; i++) { qOff.push_back(run); run += (i != fc - 1) ? ap.fragLen : ap.fragLen + len % ap.fragLen; }
            }
          }
          // 1-way then 2-way selection, as computeCGI does, to know WHICH mappings are reported
          struct M { int32_t refSeq, genome, qSeq, refStart, bin; float id; };
          std::vector<M> v; v.reserve(n);
          for (size_t i = 0; i < n; i++) v.push_back(M{maps[i].refSeqId, contigGenome[maps[i].refSeqId], maps[i].querySeqId, maps[i].refStartPos,
                                                       maps[i].refStartPos / (ap.fragLen - 20), maps[i].nucIdentity});
          ani_free(maps);
          std::sort(v.begin(), v.end(), [](const M &x, const M &y) {
            return std::tie(x.genome, x.qSeq, x.id, x.refSeq, x.refStart) < std::tie(y.genome, y.qSeq, y.id, y.refSeq, y.refStart); });
          std::vector<M> one_way;
          for (auto &e : v) { if (!one_way.empty() && one_way.back().genome == e.genome && one_way.back().qSeq == e.qSeq) one_way.back() = e; else one_way.push_back(e); }
          std::stable_sort(one_way.begin(), one_way.end(), [](const M &x, const M &y) { return std::tie(x.refSeq, x.bin, x.id) < std::tie(y.refSeq, y.bin, y.id); });
          std::vector<M> two_way;
          for (auto &e : one_way) { if (!two_way.empty() && two_way.back().refSeq == e.refSeq && two_way.back().bin == e.bin) two_way.back() = e; else two_way.push_back(e); }
          for (auto &e : two_way)
            vis.push_back(VisRow{o.queries[qi], o.refs[refIdx[e.genome]], e.id, 0 + qOff[e.qSeq], 0 + ap.fragLen - 1 + qOff[e.qSeq],
                                 e.refStart + refOff[e.refSeq], e.refStart + ap.fragLen - 1 + refOff[e.refSeq]});
        }
      }
    }
    ani_sketch_destroy(sk);
  }
  std::cerr << "INFO, skch::main, parallel_for execution finished" << std::endl;
  for (size_t i = 0; i < failedSplits.size(); i++)
    std::cerr << "ERROR :: SPLIT " << failedSplits[i] << "'s ratio difference " << failedRatio[i] << " exceeds maximum thresholds." << std::endl;

  // ---- genome lengths (computeCoreIdentity.hpp:48-92): query files first, then unseen reference files ----
  std::unordered_map<std::string, uint64_t> genomeLengths;
  auto lengthOf = [&](const Genome &g) {
    uint64_t s = 0;
    for (int32_t l : g.lens) if (l >= ap.fragLen) s += (uint64_t)(l / ap.fragLen) * (uint64_t)ap.fragLen;
    return s;
  };
  for (size_t i = 0; i < o.queries.size(); i++) genomeLengths[o.queries[i]] = lengthOf(Q.meta[i]);
  for (auto &f : o.refs)
    if (!genomeLengths.count(f)) { HostBatch one; one.add(f); genomeLengths[f] = lengthOf(one.meta[0]); }

  // ---- outputCGI (computeCoreIdentity.hpp:307-344): query ascending, identity descending ----
  std::stable_sort(finalResults.begin(), finalResults.end(), [](const ani_cgi_t &x, const ani_cgi_t &y) {
    if (x.qryGenomeId != y.qryGenomeId) return x.qryGenomeId < y.qryGenomeId;
    return x.identity > y.identity; });
  auto trusted = [&](const ani_cgi_t &e) {
    const uint64_t minLen = std::min(genomeLengths[o.queries[e.qryGenomeId]], genomeLengths[o.refs[e.refGenomeId]]);
    const uint64_t shared = (uint64_t)e.countSeq * (uint64_t)ap.fragLen;
    return shared >= minLen * o.minFraction;                            // uint64 * float, as in :328
  };
  {
    std::ofstream out(o.out);
    for (auto &e : finalResults)
      if (trusted(e))
        out << o.queries[e.qryGenomeId] << "\t" << o.refs[e.refGenomeId] << "\t" << e.identity << "\t" << e.countSeq << "\t" << e.totalQueryFragments << "\n";
  }
  // ---- outputPhylip (computeCoreIdentity.hpp:353-448) ----
  if (o.matrix) {
    std::unordered_map<std::string, int> g2i; std::vector<std::string> names;
    for (auto &e : o.queries) if (!g2i.count(e)) { g2i[e] = (int)names.size(); names.push_back(e); }
    for (auto &e : o.refs) if (!g2i.count(e)) { g2i[e] = (int)names.size(); names.push_back(e); }
    const int n = (int)names.size();
    std::vector<std::vector<float>> mat(n, std::vector<float>(n, 0.0f));
    for (auto &e : finalResults)
      if (trusted(e)) {
        int a = g2i[o.queries[e.qryGenomeId]], b = g2i[o.refs[e.refGenomeId]];
        if (a == b) continue;
        if (a < b) std::swap(a, b);
        mat[a][b] = mat[a][b] > 0 ? (mat[a][b] + e.identity) / 2 : e.identity;
      }
    std::ofstream out(o.out + ".matrix");
    out << n << "\n";
    for (int i = 0; i < n; i++) {
      out << names[i];
      for (int j = 0; j < i; j++) out << "\t" << (mat[i][j] > 0.0 ? std::to_string(mat[i][j]) : std::string("NA"));
      out << "\n";
    }
  }
  if (o.visualize) {
    std::ofstream out(o.out + ".visual");
    for (auto &r : vis)
      out << r.q << "\t" << r.r << "\t" << r.id << "\tNA\tNA\tNA\t" << r.qs << "\t" << r.qe << "\t" << r.rs << "\t" << r.re << "\tNA\tNA\n";
  }
  ani_shutdown(ctx);
  return 0;
}
