/* TEST INFRASTRUCTURE ONLY: k_index_window_links (fastani_amd/csrc/kernels/index.hpp) on the CPU stand-in for the HIP runtime
 * (tests/emu) against the definition of the window links, entry by entry:
 *   B[e] = e - (first x in [max(cLo, e - cmw1), e] with wpos[x] > wpos[e] - cmw1)
 *   A[j] = (first x in [j + 1, min(j + 2 + cmw1, cHi)) with wpos[x] >= wpos[j + 1] + cmw1, else that bound) - j     (0 for a contig's last entry)
 *   more = wpos[j + A[j]] == wpos[j + 1] + cmw1 (inside the contig)
 * on random contigs: single entries, dense runs (every position), sparse stretches (gaps beyond cmw1), sizes around the
 * workgroup's 1024 entries and its 768-entry halo.  usage: window_links_check <seed> <rounds>; prints "ok <entries>" or the first difference. */
#include <hip/hip_runtime.h>
#include <cstdio>
#include <random>
#include <vector>
#include "kernels/common.hpp"
#include "kernels/index.hpp"

int main(int argc, char **argv)
{
  const unsigned seed = argc > 1 ? (unsigned)atoi(argv[1]) : 1u;
  const int rounds = argc > 2 ? atoi(argv[2]) : 6;
  std::mt19937 rng(seed);
  size_t checked = 0;
  for (int round = 0; round < rounds; round++) {
    const int32_t cmw1 = std::vector<int32_t>{3, 17, 120, 977, 2900}[rng() % 5];
    std::vector<int32_t> seq, wpos, first;
    const int nContigs = 1 + (int)(rng() % 40);
    for (int c = 0; c < nContigs; c++) {
      first.push_back((int32_t)seq.size());
      const int kind = (int)(rng() % 6);
      int len = kind == 0 ? (int)(rng() % 4) : kind == 1 ? 1000 + (int)(rng() % 60) : kind == 2 ? 1700 + (int)(rng() % 200) : (int)(rng() % 3000);
      int32_t w = (int32_t)(rng() % 50);
      for (int i = 0; i < len; i++) {
        seq.push_back(c); wpos.push_back(w);
        const unsigned r = rng() % 100;
        const int32_t typical = std::max(1, cmw1 / 12);
        w += r < 20 ? 1 : r < 90 ? 1 + (int32_t)(rng() % (2 * typical)) : r < 97 ? cmw1 - 1 + (int32_t)(rng() % 3) : cmw1 + 1 + (int32_t)(rng() % (4 * cmw1));
      }
    }
    first.push_back((int32_t)seq.size());
    const uint32_t n = (uint32_t)seq.size();
    if (!n) continue;
    const int32_t expect = (int32_t)(rng() % 3 == 0 ? rng() % 200 : 12);
    std::vector<uint32_t> got(n + 4, 0xdeadbeefu);
    // 16-byte aligned and deliberately misaligned views: the kernel's vector path and its scalar path
    for (int shift = 0; shift < 2; shift++) {
      std::vector<int32_t> seqBuf(n + 8), wposBuf(n + 8); std::vector<uint32_t> outBuf(n + 8, 0xdeadbeefu);
      int32_t *mSeq = (int32_t *)(((uintptr_t)seqBuf.data() + 15) & ~(uintptr_t)15) + shift;
      int32_t *mWpos = wposBuf.data();
      uint32_t *mWin = (uint32_t *)(((uintptr_t)outBuf.data() + 15) & ~(uintptr_t)15) + shift;
      std::copy(seq.begin(), seq.end(), mSeq); std::copy(wpos.begin(), wpos.end(), mWpos);
      hipLaunchKernelGGL(ani::k_index_window_links, dim3((n + ani::kWinBlock - 1) / ani::kWinBlock), dim3(256), 0, (hipStream_t) nullptr, (const int32_t *)mSeq, (const int32_t *)mWpos,
                         (const int32_t *)first.data(), n, cmw1, expect, mWin);
      for (uint32_t j = 0; j < n; j++) {
        const int32_t c = seq[j], cLo = first[c], cHi = first[c + 1];
        int32_t x = std::max<int32_t>(cLo, (int32_t)j - cmw1);
        while (x < (int32_t)j && wpos[x] <= wpos[j] - cmw1) x++;
        const uint32_t b = (uint32_t)((int32_t)j - x);
        uint32_t a = 0, more = 0;
        if ((int32_t)j + 1 < cHi) {
          const int32_t tgt = wpos[j + 1] + cmw1, hi = std::min<int32_t>((int32_t)j + 2 + cmw1, cHi);
          int32_t y = (int32_t)j + 1;
          while (y < hi && wpos[y] < tgt) y++;
          a = (uint32_t)(y - (int32_t)j);
          more = (y < cHi && wpos[y] == tgt) ? ani::kWinMoreBit : 0u;
        }
        const uint32_t want = std::min(a, ani::kWinMask) << ani::kWinShiftA | std::min(b, ani::kWinMask) | more;
        if (mWin[j] != want) {
          printf("round %d shift %d cmw1 %d expect %d entry %u of %u (contig %d [%d, %d)): got %08x want %08x\n", round, shift, cmw1, expect, j, n, c, cLo, cHi, mWin[j], want);
          return 1;
        }
      }
      if (mWin[n] != 0xdeadbeefu) { printf("round %d: wrote past the end\n", round); return 1; }
      checked += n;
    }
  }
  printf("ok %zu\n", checked);
  return 0;
}
