// engine_ingest.hip — sequence batches on their way to the device: validation, table-free SWAR 2-bit
// packing on host threads into page-locked staging, upload; a batch can stay on the device (ani_batch_upload).
#include "host/engine.hpp"

namespace anih {

int check_batch(const ani_seq_batch_t *b)
{
  if (!b || b->nGenomes < 0 || b->nContigs < 0 || (b->nContigs && (!b->genomeContigStart || !b->contigLen)))
    return fail(ANI_ERR_ARG, "invalid sequence batch");
  if (b->layout == ANI_SEQ_DEVICE_BATCH) return b->data ? ANI_OK : fail(ANI_ERR_ARG, "sequence batch without its device batch handle");
  if (b->nContigs && !b->data) return fail(ANI_ERR_ARG, "sequence batch without data");
  if (b->layout != ANI_SEQ_HOST_ASCII && b->layout != ANI_SEQ_DEVICE_PACKED2 && b->layout != ANI_SEQ_HOST_ASCII_PTRS && b->layout != ANI_SEQ_HOST_MIXED_PTRS)
    return fail(ANI_ERR_ARG, "unknown sequence layout %d", b->layout);
  if (b->layout != ANI_SEQ_HOST_ASCII_PTRS && b->nContigs && !b->contigOffset) return fail(ANI_ERR_ARG, "sequence batch without contig offsets");
  if (b->layout == ANI_SEQ_HOST_MIXED_PTRS)
    for (int32_t c = 0; c < b->nContigs; c++) if (b->contigOffset[c] != 0 && b->contigOffset[c] != 1) return fail(ANI_ERR_ARG,
        "contig %d: unknown kind %lld", c,
        (long long)b->contigOffset[c]);
  for (int32_t c = 0; c < b->nContigs; c++) if (b->contigLen[c] < 0) return fail(ANI_ERR_LIMIT, "contig %d has a negative length (>= 2^31 bases?)", c);
  return ANI_OK;
}

// 8 ASCII bases -> 16 bits of 2-bit codes (A0 C1 G2 T3, either case) + "all eight are A/C/G/T", without a table: bits 1..2 of
// the byte give A0 C1 G3 T2, x ^ (x >> 1) swaps the last two; the byte is then rebuilt from its code and compared.
inline uint32_t pack8(uint64_t x, bool *pure)
{
  const uint64_t k01 = 0x0101010101010101ull;
  uint64_t c = (x >> 1) & (3 * k01);
  c ^= (c >> 1) & k01;
  const uint64_t lo = c & k01, hi = (c >> 1) & k01, both = lo & hi;
  // code -> upper-case letter: 'A' + {0, 2, 6, 19}
  const uint64_t rec = 0x41 * k01 + (lo << 1) + (hi << 1) + (hi << 2) + both + (both << 1) + (both << 3);
  *pure = ((x & ~(0x20 * k01)) == rec);
  c = (c | (c >> 6)) & 0x000F000F000F000Full;
  c = (c | (c >> 12)) & 0x000000FF000000FFull;
  c = (c | (c >> 24)) & 0xFFFFull;
  return (uint32_t)c;
}

// bases [lo, hi) of a contig (lo a multiple of 16) -> words dst[lo / 16 ..]; `whole`: hi is the contig's end (two zero words follow).
// Returns false if a byte other than A C G T a c g t occurred (the words are then meaningless).
inline bool pack_range(const uint8_t *sp, int32_t lo, int32_t hi, uint32_t *dst, bool whole)
{
  bool allPure = true;
  int32_t x = lo;
  for (; x + 16 <= hi; x += 16) {
    uint64_t a, bb; memcpy(&a, sp + x, 8); memcpy(&bb, sp + x + 8, 8);
    bool p1, p2;
    const uint32_t wd = pack8(a, &p1) | (pack8(bb, &p2) << 16);
    allPure &= p1 & p2;
    dst[x >> 4] = wd;
  }
  if (x < hi) {                                       // last, partial word of the contig
    uint8_t tail[16]; memset(tail, 'A', 16); memcpy(tail, sp + x, (size_t)(hi - x));
    uint64_t a, bb; memcpy(&a, tail, 8); memcpy(&bb, tail + 8, 8);
    bool p1, p2;
    uint32_t wd = pack8(a, &p1) | (pack8(bb, &p2) << 16);
    allPure &= p1 & p2;
    const int nb = hi - x;
    if (nb < 16) wd &= (1u << (2 * nb)) - 1u;
    dst[x >> 4] = wd;
  }
  if (whole) { const size_t e = ((size_t)hi + 15) / 16; dst[e] = 0; dst[e + 1] = 0; }
  return allPure;
}

int upload_batch(ani_ctx *ctx, const ani_seq_batch_t *b, int32_t g0, int32_t g1, DeviceBatch *out, ani_dev_batch *keep)
{
  if (b->layout == ANI_SEQ_DEVICE_BATCH) {       // a view of genomes [g0, g1) of an uploaded batch
    const ani_dev_batch *src = (const ani_dev_batch *)b->data;
    if (src->device != ctx->device) return fail(ANI_ERR_ARG, "device batch lives on device %d, context on %d", src->device, ctx->device);
    if (g1 > src->db.nGenomes) return fail(ANI_ERR_ARG, "genome range beyond the device batch");
    const DeviceBatch &sb = src->db;
    const int32_t c0 = sb.genomeContigStart[g0], c1 = sb.genomeContigStart[g1];
    out->nGenomes = g1 - g0; out->nContigs = c1 - c0;
    out->genomeContigStart.resize(out->nGenomes + 1);
    for (int32_t g = g0; g <= g1; g++) out->genomeContigStart[g - g0] = sb.genomeContigStart[g] - c0;
    out->contigLen.assign(sb.contigLen.begin() + c0, sb.contigLen.begin() + c1);
    out->totalBases = 0;
    for (int32_t c = 0; c < out->nContigs; c++) out->totalBases += (uint64_t)out->contigLen[c];
    out->dPacked = sb.dPacked; out->dAscii = sb.dAscii;
    out->dContigOff = sb.dContigOff + c0; out->dContigLen = sb.dContigLen + c0; out->dContigMode = sb.dContigMode + c0;
    return ANI_OK;
  }
  const int32_t c0 = b->genomeContigStart[g0], c1 = b->genomeContigStart[g1];
  out->nGenomes = g1 - g0; out->nContigs = c1 - c0;
  out->genomeContigStart.resize(out->nGenomes + 1);
  for (int32_t g = g0; g <= g1; g++) out->genomeContigStart[g - g0] = b->genomeContigStart[g] - c0;
  out->contigLen.assign(b->contigLen + c0, b->contigLen + c1);
  out->contigOff.resize(out->nContigs); out->contigPacked.resize(out->nContigs);
  out->totalBases = 0;
  for (int32_t c = 0; c < out->nContigs; c++) out->totalBases += (uint64_t)out->contigLen[c];
  const size_t nc = (size_t)out->nContigs;
  void *dPacked = nullptr, *dAscii = nullptr;

  if (b->layout == ANI_SEQ_DEVICE_PACKED2) {
    for (int32_t c = 0; c < out->nContigs; c++) { out->contigOff[c] = b->contigOffset[c0 + c]; out->contigPacked[c] = 1; }
    dPacked = const_cast<void *>(b->data);
  } else {
    // Ingest: classify every contig (pure ACGT -> 2 bits per base, anything else -> raw bytes) and pack, on host threads, straight
    // into page-locked staging (contigs are split into segments of 4 Mbases so that one long chromosome still spreads over the
    // threads).  Segments are packed optimistically; a contig with any other byte is copied raw in a second, rare, pass.
    const uint8_t *flat = b->layout == ANI_SEQ_HOST_ASCII ? (const uint8_t *)b->data : nullptr;
    const uint8_t *const *ptrs = b->layout != ANI_SEQ_HOST_ASCII ? (const uint8_t *const *)b->data : nullptr;
    // ANI_SEQ_HOST_MIXED_PTRS: contigs the reader packed already (ani_pack_acgt) are copied, the others go through the same loop as ever
    const bool mixed = b->layout == ANI_SEQ_HOST_MIXED_PTRS;
    auto prepacked = [&](int32_t c) -> bool { return mixed && b->contigOffset[c0 + c] == 1; };
    auto contig_ptr = [&](int32_t c) -> const uint8_t * { return flat ? flat + b->contigOffset[c0 + c] : ptrs[c0 + c]; };
    const int32_t nC = out->nContigs;
    struct Seg { int32_t c; int32_t lo, hi; };
    std::vector<Seg> segs;
    const int32_t kSeg = 1 << 22;                       // multiple of 16: segments pack whole words
    size_t nWordsAll = 0;
    std::vector<size_t> wordOff((size_t)nC + 1, 0);
    for (int32_t c = 0; c < nC; c++) {
      wordOff[c] = nWordsAll; nWordsAll += ((size_t)out->contigLen[c] + 15) / 16 + 2;                          // +2 words of slack for the 3-word fetch
      for (int32_t lo = 0; lo < out->contigLen[c]; lo += kSeg) segs.push_back(Seg{c, lo, std::min(out->contigLen[c], lo + kSeg)});
    }
    wordOff[nC] = nWordsAll;
    uint32_t *hPacked = nullptr;
    TRY(pinned_buffer(ctx, 3, nWordsAll * 4 + 64, (void **)&hPacked));
    std::vector<uint8_t> segImpure(segs.size(), 0);
    parallel_for(segs.size(), out->totalBases, [&](size_t i) {
      const Seg &sg = segs[i];
      uint32_t *dst = hPacked + wordOff[sg.c];
      const bool whole = sg.hi == out->contigLen[sg.c];
      if (prepacked(sg.c)) {                            // words [lo / 16, ceil(hi / 16)) of the reader's copy (+ the two closing words)
        const uint32_t *src = (const uint32_t *)(const void *)contig_ptr(sg.c);
        const size_t w0 = (size_t)sg.lo >> 4, w1 = ((size_t)sg.hi + 15) / 16 + (whole ? 2 : 0);
        memcpy(dst + w0, src + w0, (w1 - w0) * 4);
        return;
      }
      segImpure[i] = !pack_range(contig_ptr(sg.c), sg.lo, sg.hi, dst, whole);
    });
    for (int32_t c = 0; c < nC; c++) if (out->contigLen[c] == 0) { hPacked[wordOff[c]] = 0; hPacked[wordOff[c] + 1] = 0; }
    std::vector<uint8_t> impure(nC, 0);
    for (size_t i = 0; i < segs.size(); i++) impure[segs[i].c] |= segImpure[i];
    size_t nBytes = 0;
    for (int32_t c = 0; c < nC; c++) {
      out->contigPacked[c] = !impure[c];
      if (!impure[c]) out->contigOff[c] = (int64_t)wordOff[c];
      else { out->contigOff[c] = (int64_t)nBytes; nBytes += ((size_t)out->contigLen[c] + 3) & ~(size_t)3; }
    }
    uint8_t *hAscii = nullptr;
    if (nBytes) {
      TRY(pinned_buffer(ctx, 4, nBytes + 64, (void **)&hAscii));
      for (int32_t c = 0; c < nC; c++) if (impure[c]) memcpy(hAscii + out->contigOff[c], contig_ptr(c), (size_t)out->contigLen[c]);
    }
    if (keep) {
      HIP_TRY(pool_malloc(&keep->bufs[0], nWordsAll * 4 + 64)); HIP_TRY(pool_malloc(&keep->bufs[1], nBytes + 64));
      dPacked = keep->bufs[0]; dAscii = keep->bufs[1];
    } else {
      TRY(ctx->seqPacked.ensure(nWordsAll * 4 + 64)); TRY(ctx->seqAscii.ensure(nBytes + 64));
      dPacked = ctx->seqPacked.p; dAscii = ctx->seqAscii.p;
    }
    if (nWordsAll) HIP_TRY(hipMemcpyAsync(dPacked, hPacked, nWordsAll * 4, hipMemcpyHostToDevice, ctx->stream));
    if (nBytes) HIP_TRY(hipMemcpyAsync(dAscii, hAscii, nBytes, hipMemcpyHostToDevice, ctx->stream));
    HIP_TRY(hipStreamSynchronize(ctx->stream));    // the staging buffers are reused by the next upload
  }
  out->dPacked = (const uint32_t *)dPacked; out->dAscii = (const uint8_t *)dAscii;
  void *dOff = nullptr, *dLen = nullptr, *dMode = nullptr;
  if (keep) {
    HIP_TRY(pool_malloc(&keep->bufs[2], nc * 8 + 8)); HIP_TRY(pool_malloc(&keep->bufs[3], nc * 4 + 4)); HIP_TRY(pool_malloc(&keep->bufs[4], nc + 4));
    dOff = keep->bufs[2]; dLen = keep->bufs[3]; dMode = keep->bufs[4];
  } else {
    TRY(ctx->contigOff.ensure(nc * 8 + 8)); TRY(ctx->contigLen.ensure(nc * 4 + 4)); TRY(ctx->contigMode.ensure(nc + 4));
    dOff = ctx->contigOff.p; dLen = ctx->contigLen.p; dMode = ctx->contigMode.p;
  }
  if (nc) {
    HIP_TRY(hipMemcpyAsync(dOff, out->contigOff.data(), nc * 8, hipMemcpyHostToDevice, ctx->stream));
    HIP_TRY(hipMemcpyAsync(dLen, out->contigLen.data(), nc * 4, hipMemcpyHostToDevice, ctx->stream));
    HIP_TRY(hipMemcpyAsync(dMode, out->contigPacked.data(), nc, hipMemcpyHostToDevice, ctx->stream));
    HIP_TRY(hipStreamSynchronize(ctx->stream));
  }
  out->dContigOff = (const int64_t *)dOff; out->dContigLen = (const int32_t *)dLen; out->dContigMode = (const uint8_t *)dMode;
  return ANI_OK;
}

}  // namespace anih

extern "C" {

int ani_pack_acgt(const uint8_t *seq, int32_t len, uint32_t *out)
{
  if (len < 0 || !out || (len && !seq)) return 0;
  if (len == 0) { out[0] = 0; out[1] = 0; return 1; }
  return pack_range(seq, 0, len, out, true) ? 1 : 0;
}

int ani_batch_upload(ani_ctx *ctx, const ani_seq_batch_t *genomes, ani_dev_batch **out)
{
  if (!ctx || !out) return fail(ANI_ERR_ARG, "null argument");
  TRY(check_batch(genomes));
  if (genomes->layout == ANI_SEQ_DEVICE_BATCH || genomes->layout == ANI_SEQ_DEVICE_PACKED2) return fail(ANI_ERR_ARG, "ani_batch_upload takes host sequences");
  HIP_TRY(hipSetDevice(ctx->device));
  ani_dev_batch *b = new ani_dev_batch();
  b->ctx = ctx; b->device = ctx->device;
  const int rc = upload_batch(ctx, genomes, 0, genomes->nGenomes, &b->db, b);
  if (rc != ANI_OK) { ani_batch_free(b); return rc; }
  *out = b;
  return ANI_OK;
}

void ani_batch_free(ani_dev_batch *b)
{
  if (!b) return;
  (void)hipSetDevice(b->device);
  for (void *q : b->bufs) if (q) pool_free(q);
  delete b;
}


}  // extern "C"
