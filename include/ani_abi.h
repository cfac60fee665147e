/* ani_abi.h — C-ABI of the MI355X-native ANI engine (libfastani_amd.so).
 *
 * FastANI has no FFI/plugin interface; its seams are three C++ constructors/functions that
 * core_genome_identity() calls (all paths relative to /root/reference):
 *
 *   skch::Sketch::Sketch(const Parameters&)                         src/map/include/winSketch.hpp:109
 *   skch::Map::Map(param, sketch, totalQueryFragments&, queryno, callback)   src/map/include/computeMap.hpp:93
 *   cgi::computeCGI(param, mapResults, mapper, sketch, totalQueryFragments, queryFileNo, fileName, out)
 *                                                                   src/cgi/include/computeCoreIdentity.hpp:166
 *
 * Each entry point below replaces one of those seams (cited per function).  Plain pointers and sizes only;
 * no C++ or torch types cross this boundary.  The library is implemented with hand-written HIP kernels for
 * gfx950; there is no CPU fallback behind this ABI — every call fails with ANI_ERR_DEVICE if no GPU is usable.
 *
 * Conventions
 *   - every function returns 0 (ANI_OK) or a negative ani_status; ani_last_error() gives the message.
 *   - the caller owns all inputs; host outputs returned through `T **out` are allocated by the library and
 *     released with ani_free(); device outputs are released with ani_device_free().
 *   - one ani_ctx per device and per host thread (mirror of "one Sketch per OpenMP thread",
 *     src/cgi/core_genome_identity.cpp:55-65); a context is not thread-safe.
 *   - POD records are bit-identical to the reference structs (little-endian, 32-bit fields).
 */
#ifndef ANI_ABI_H
#define ANI_ABI_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef enum {
  ANI_OK = 0,
  ANI_ERR_ARG = -1,       /* invalid argument */
  ANI_ERR_DEVICE = -2,    /* no usable GPU / HIP runtime error */
  ANI_ERR_NOMEM = -3,     /* host or device allocation failed */
  ANI_ERR_LIMIT = -4,     /* input exceeds a documented limit (e.g. contig >= 2^31 bases, base_types.hpp:15) */
  ANI_ERR_INTERNAL = -5
} ani_status;

/* skch::MinimizerInfo — src/map/include/base_types.hpp:22-53 (12 bytes) */
typedef struct { uint32_t hash; int32_t seqId; int32_t wpos; } ani_minimizer_t;

/* skch::MappingResult — src/map/include/base_types.hpp:89-102 (44 bytes) */
typedef struct {
  int32_t queryLen, refStartPos, refEndPos, queryStartPos, queryEndPos, refSeqId, querySeqId;
  float nucIdentity, nucIdentityUpperBound;
  int32_t sketchSize, conservedSketches;
} ani_mapping_t;

/* cgi::CGI_Results — src/cgi/include/cgid_types.hpp:68-80 (20 bytes) */
typedef struct { int32_t refGenomeId, qryGenomeId, countSeq, totalQueryFragments; float identity; } ani_cgi_t;

/* The subset of skch::Parameters (src/map/include/map_parameters.hpp:22-41) that the hot path reads.
 * ani_params_default() fills the reference's defaults (src/map/include/parseCmdArgs.hpp:118-130) and
 * derives windowSize the way the CLI does (parseCmdArgs.hpp:225-228 -> Stat::recommendedWindowSize,
 * src/map/include/map_stats.hpp:226-256). */
typedef struct {
  int32_t kmerSize;            /* 1..16 */
  int32_t windowSize;          /* derived; 24 for k=16, fragLen=3000 */
  int32_t fragLen;             /* Parameters::minReadLength */
  float percentageIdentity;    /* 80 */
} ani_params_t;

/* A batch of genomes handed to the library.  Genome g owns contigs [genomeContigStart[g], genomeContigStart[g+1]);
 * contig c has contigLen[c] bases.
 *   layout ANI_SEQ_HOST_ASCII   : `data` is host memory, raw sequence bytes as read from FASTA (newlines removed);
 *                                 contig c starts at byte contigOffset[c].  The library upper-cases a-z
 *                                 (src/map/include/commonFunc.hpp:56-66) and leaves every other byte as is.
 *   layout ANI_SEQ_DEVICE_PACKED2: `data` is DEVICE memory, 2-bit codes A=0 C=1 G=2 T=3, 16 bases per little-endian
 *                                 uint32 (base j of a word in bits 2j..2j+1); contig c starts at WORD contigOffset[c]
 *                                 (contigs are word-aligned).  Only valid for pure-ACGT data.
 *   layout ANI_SEQ_HOST_ASCII_PTRS: as HOST_ASCII, but `data` is an array of nContigs host pointers (const uint8_t *const *), one
 *                                 per contig; contigOffset is ignored.  Lets a reader hand over per-file buffers without
 *                                 concatenating them.
 *   layout ANI_SEQ_DEVICE_BATCH   : `data` is an ani_dev_batch* returned by ani_batch_upload (genomes already packed and
 *                                 resident on the device); the table fields must describe that batch (nGenomes, nContigs,
 *                                 genomeContigStart, contigLen), contigOffset is ignored.
 *   layout ANI_SEQ_HOST_MIXED_PTRS: `data` is an array of nContigs host pointers; contigOffset[c] says what pointer c is:
 *                                 0 = raw sequence bytes (as HOST_ASCII_PTRS), 1 = 2-bit codes made by ani_pack_acgt from a
 *                                 pure-ACGT contig ((len + 15) / 16 + 2 words).  A reader packs every contig it can on its own
 *                                 thread, while the bytes are in its cache, and the upload only copies (round 6: the packing
 *                                 of a slice by the upload thread's pool was the longest stage of the command line's ingest).
 */
typedef enum { ANI_SEQ_HOST_ASCII = 0, ANI_SEQ_DEVICE_PACKED2 = 1, ANI_SEQ_HOST_ASCII_PTRS = 2, ANI_SEQ_DEVICE_BATCH = 3, ANI_SEQ_HOST_MIXED_PTRS = 4 } ani_seq_layout;
typedef struct {
  int32_t layout;
  int32_t nGenomes;
  int32_t nContigs;
  const int32_t *genomeContigStart;   /* [nGenomes+1] host */
  const int64_t *contigOffset;        /* [nContigs]   host; bytes (ASCII) or words (PACKED2) */
  const int32_t *contigLen;           /* [nContigs]   host; bases */
  const void *data;
} ani_seq_batch_t;

typedef struct ani_ctx ani_ctx;
typedef struct ani_sketch ani_sketch;
typedef struct ani_dev_batch ani_dev_batch;
typedef struct ani_fragset ani_fragset;

/* Run-time counters for the measurement contract: the algorithmic-byte figure of a run is
 * computed from these, never estimated. */
typedef struct {
  uint64_t refBases, refMinimizers, refUniqueHashes;
  uint64_t queryGenomes, queryFragments, queryBases, querySketchHashes;   /* Σ s over fragments */
  uint64_t seedHits;          /* H_total */
  uint64_t l1Candidates;
  uint64_t l2WindowEntries;   /* Σ m_c over candidates */
  uint64_t l2Steps;           /* Σ super-window placements evaluated */
  uint64_t l2QueryHashes;     /* Σ s over candidates (each candidate reads its fragment sketch once) */
  uint64_t l2WindowEntriesB, l2QueryHashesB;   /* the share of the two sums above handled by the class-B simulation launches */
  uint64_t l2Launches;        /* launches of the dominant L2 kernel (ani::k_l2_sim) */
  uint64_t l2FastCandidates;  /* candidates finished by the LDS fast path */
  uint64_t l2SlowCandidates;  /* candidates routed to the general kernel (ani::k_l2) */
  uint64_t l2SlowLimit, l2SlowDup, l2SlowOverflow;   /* ... by reason: size limits / same-hash neighbour or wide gap / counter overflow */
  uint64_t mappings;
  uint64_t cgiRows;
  uint64_t indexChunks;       /* index chunks built (one per reference set unless it passes ANI_MAX_INDEX_MINIMIZERS) */
  uint64_t l1Probes;          /* sketch hashes looked up in an index: querySketchHashes x index chunks probed */
  uint64_t l2ChunkHalvings;   /* L2 chunks redone at half size because their code stream passed the 32-bit offset limit */
  uint64_t indexChunkBuilds;  /* index chunks built, rebuilds of a streamed reference set included */
  uint64_t l1BigFragments;    /* query fragments (per index chunk) whose seed hits exceeded every LDS class: batched global-memory L1 path */
  uint64_t l1MidFragments;    /* query fragments (per index chunk) with 2048 < seed hits <= 4096: LDS class M (ani::k_l1<2048, 4096>) */
  uint64_t l1TinyFragments;   /* query fragments (per index chunk) with 1..64 seed hits: one wave each (ani::k_l1_tiny) */
  double msSketch, msIndex, msFragSketch, msL1, msL2, msReduce;   /* HIP-event time per stage, accumulated */
  double msL2Kernel;          /* HIP-event time of the class-A ani::k_l2_sim launches alone (on the launch stream) */
  double msL2Ranges, msL2Codes, msL2Slow;   /* ani::k_l2_ranges, ani::k_l2_codes, ani::k_l2 */
  double msL2SimB;            /* class-B simulation launches (ani::k_l2_sim<L2Geom<319>> + its list compaction) */
  double msL1Probe, msL1Main; /* ani::k_l1_probe; ani::k_l1<0, 2048> (the small-class gather + filter + sort + candidate kernel) */
  double msL1Big;             /* the batched global-memory L1 path (gather + device sort + candidates) */
  double msL1Tiny;            /* ani::k_l1_tiny */
} ani_counters_t;

/* ---- life cycle ---- */
int ani_init(int device, ani_ctx **out);
void ani_shutdown(ani_ctx *ctx);
const char *ani_last_error(void);
void ani_free(void *hostPtr);
void ani_device_free(ani_ctx *ctx, void *devPtr);
/* copy between device buffers of this context (used by the host side to stage records into communication buffers) */
int ani_device_copy(ani_ctx *ctx, void *dst, const void *src, size_t bytes);
/* device memory of this context (released with ani_device_free) and a copy between two contexts' devices — the host side of a
 * multi-GPU run stages minimizer records with these (peer-to-peer over xGMI when the devices can access each other) */
int ani_device_alloc(ani_ctx *ctx, size_t bytes, void **out);
/* free / total memory of the context's device in bytes (the command line sizes the blocks of a reference sketch file with it;
 * no counterpart in the reference, which splits its database by hand: scripts/splitDatabase.sh) */
int ani_device_memory(ani_ctx *ctx, size_t *freeBytes, size_t *totalBytes);
/* A hint, never needed for correctness: take from the driver NOW, on the calling thread, the device memory that sketching and
 * indexing a reference set of about `nMinimizers` minimizers will ask for (~65 bytes per minimizer: 1 GiB segments for the slices'
 * records and fragment sets first, then one segment for the index build), and leave it free in the allocator.  Fresh device memory
 * costs 20 - 40 us per MB on some hosts — 0.6 s of a cold 1000-genome run sat in the index build for that reason; the command line
 * calls this on a side thread while its readers parse the first files (estimate: input bytes x 2 / (w + 1); ANI_CLI_PREWARM=0
 * switches it off).  An estimate that is off only wastes the memory until the allocator is trimmed (ani_shutdown).  No counterpart in
 * the reference. */
int ani_pool_prewarm_index(ani_ctx *ctx, uint64_t nMinimizers);
/* The device allocator of the context's device: out[0] bytes held in segments, [1] of them free, [2] handed out, [3] segments,
 * [4] hipMalloc calls so far, [5] bytes they took, [6] microseconds they took, [7] the largest free extent of the segments that serve requests >= 32 MiB.  (The end-to-end line of
 * bench.py and the command line's trace print [5] / [6]: what fresh device memory cost on the box.)  No counterpart in the reference. */
int ani_pool_stats(ani_ctx *ctx, uint64_t out[8]);
int ani_device_copy_peer(ani_ctx *dstCtx, void *dst, ani_ctx *srcCtx, const void *src, size_t bytes);
/* Ingest: classify + 2-bit pack host sequences on host threads into page-locked staging, copy them to the
 * device and keep them there.  The handle can be passed to every entry point that takes a sequence batch (layout
 * ANI_SEQ_DEVICE_BATCH), any number of times: an all-vs-all run sketches and maps the same upload. */
int ani_batch_upload(ani_ctx *ctx, const ani_seq_batch_t *genomes, ani_dev_batch **out);
/* Host-side helper of the ingest path (no device, no context): packs `len` sequence bytes to 2 bits per base (A0 C1 G2 T3, either
 * case: commonFunc.hpp:56-66 folds a-z) into out[(len + 15) / 16 + 2] if every byte is one of A C G T a c g t and returns 1;
 * returns 0 (out undefined) if any other byte occurs — such a contig stays raw bytes (the reference hashes raw upper-cased ASCII,
 * commonFunc.hpp:71-81).  For ANI_SEQ_HOST_MIXED_PTRS. */
int ani_pack_acgt(const uint8_t *seq, int32_t len, uint32_t *out);
void ani_batch_free(ani_dev_batch *b);
int ani_get_counters(ani_ctx *ctx, ani_counters_t *out);
int ani_reset_counters(ani_ctx *ctx);

/* ---- host-side scalars: skch::Stat (src/map/include/map_stats.hpp) ---- */
int ani_params_default(ani_params_t *p, int kmerSize, int fragLen);          /* parseCmdArgs.hpp:118-130,:225-228 */
int ani_recommended_window(int kmerSize, int fragLen);                       /* map_stats.hpp:226-256 */
int ani_min_hits_relaxed(int sketchSize, int kmerSize, float identity);      /* map_stats.hpp:142-167 */
int ani_identity(int shared, int sketchSize, int kmerSize, float *nucIdentity, float *upperBound); /* computeMap.hpp:375-381 */

/* ---- reference sketch: replaces skch::Sketch::Sketch (winSketch.hpp:109-115: build :124, index :181) ---- */
int ani_sketch_build(ani_ctx *ctx, const ani_params_t *p, const ani_seq_batch_t *refs, ani_sketch **out);
void ani_sketch_destroy(ani_sketch *sk);
/* position-ordered minimizerIndex (winSketch.hpp:93) copied to the host — for parity tests and staging */
int ani_sketch_export(const ani_sketch *sk, ani_minimizer_t **out, size_t *n);
/* the numbers Sketch::sanityCheck needs (winSketch.hpp:298-318): Σ occurrences, #unique hashes, Σ contig length */
int ani_sketch_stats(const ani_sketch *sk, uint64_t *nMinimizers, uint64_t *nUnique, uint64_t *totalLength,
                     int32_t *nContigs, int32_t *nGenomes);

/* A reference set whose minimizers do not fit one 32-bit index (> ANI_MAX_INDEX_MINIMIZERS, default 1.7e9 ~ 4000 bacterial
 * genomes) is held as several index chunks cut at genome borders; every entry point below works on the whole set (global
 * seqIds / genome ids), results are identical to a single index.  This is the device-side form of the
 * reference's own database split (src/cgi/include/computeCoreIdentity.hpp:457-487, scripts/splitDatabase.sh:12-26).
 * Returns the number of chunks and (optionally, up to `cap`) the first genome id of each. */
int ani_sketch_chunks(const ani_sketch *sk, int32_t *nChunks, int32_t *firstGenome, int32_t cap);

/* Reference sets larger than the device memory (BASELINE configs[4]; the reference's own answer is the database split of
 * computeCoreIdentity.hpp:457-487 / scripts/splitDatabase.sh with every query mapped against every split): when the index
 * (~36 bytes per minimizer) does not fit beside a working-set reserve — or when ANI_MAX_RESIDENT_CHUNKS says so — the sketch keeps
 * the 12-byte minimizer records and at most `maxResident` chunks' index arrays; the mapping entry points then walk the set chunk by
 * chunk (build, map every query sub-batch, drop).  Results are identical.  Reports the mode. */
int ani_sketch_residency(const ani_sketch *sk, int32_t *streaming, int32_t *maxResident, int32_t *residentNow);
/* A sketch that holds a shard or a block of a larger reference set: `base` = global id of its first genome, added to refGenomeId in the
 * CGI rows of ani_map_cgi_batch / ani_map_cgi_fragset(s) (mapping records and ani_compute_cgi keep sketch-local ids).  The reference does this
 * per thread after the fact (cgi::correctRefGenomeIds, computeCoreIdentity.hpp:480-487, for its split of :457-474). */
int ani_sketch_set_ref_id_base(ani_sketch *sk, int32_t base);

/* Multi-GPU staging: rank r sketches its share of the reference genomes into device-resident
 * 12-byte records with GLOBAL seqIds (seqIdBase = contigs before this shard), the caller all-gathers the
 * records over RCCL (torch.distributed), and every rank builds the full index from the gathered records. */
int ani_sketch_records(ani_ctx *ctx, const ani_params_t *p, const ani_seq_batch_t *refs, int32_t seqIdBase,
                       void **devRecords, size_t *n);
int ani_sketch_from_records(ani_ctx *ctx, const ani_params_t *p, const void *devRecords, size_t n,
                            const int32_t *contigLen, int32_t nContigs,
                            const int32_t *genomeContigStart, int32_t nGenomes, ani_sketch **out);

/* The same from several record buffers (what an all-gather with one slot per rank delivers): part i holds the n[i] records of
 * genomes [partGenomeStart[i], partGenomeStart[i+1]), position order, global seqIds.  No concatenation copy is made. */
int ani_sketch_from_record_parts(ani_ctx *ctx, const ani_params_t *p, int32_t nParts, const void *const *devRecords,
                                 const uint64_t *n, const int32_t *partGenomeStart,
                                 const int32_t *contigLen, int32_t nContigs,
                                 const int32_t *genomeContigStart, int32_t nGenomes, ani_sketch **out);

/* The same, but the library TAKES the record buffers OVER (they must be device memory of this context that came from
 * ani_sketch_records / ani_sketch_records_self / ani_device_alloc; the caller must not free or touch them afterwards, whatever the
 * return code).  A set that is streamed (see ani_sketch_residency) keeps them as its records — no copy, which a set near the
 * device's capacity has no room for —, a resident set releases each buffer as soon as the index chunks that need it are built.
 * (The reference's counterpart of a set this large is the database split of computeCoreIdentity.hpp:457-487.) */
int ani_sketch_adopt_record_parts(ani_ctx *ctx, const ani_params_t *p, int32_t nParts, void *const *devRecords,
                                  const uint64_t *n, const int32_t *partGenomeStart,
                                  const int32_t *contigLen, int32_t nContigs,
                                  const int32_t *genomeContigStart, int32_t nGenomes, ani_sketch **out);

/* ---- persistent sketch file: the position-ordered minimizer records + contig / genome tables + genome names,
 * sections 4096-byte aligned (mmap-able); the reference has nothing like it (every run and every thread re-sketches).  Saving
 * costs one pass over the index; loading is an mmap, one host-to-device copy of the records and the device-side index build.
 * ani_sketch_load takes a genome range [g0, g1) (g1 < 0: all), so every rank of a multi-GPU job can read its own share. */
int ani_sketch_save(const ani_sketch *sk, const char *path, const char *const *genomeNames /* [nGenomes] or NULL */);
int ani_sketch_load(ani_ctx *ctx, const char *path, int32_t g0, int32_t g1, ani_sketch **out);
/* One file from several sketches, added one after the other (their genomes are appended): how a reference set whose minimizer
 * records exceed the device memory is written block by block.  ani_sketch_save = open + add + close.  A failed or empty writer
 * removes its file at close. */
typedef struct ani_sketch_writer ani_sketch_writer;
int ani_sketch_writer_open(const char *path, ani_sketch_writer **out);
int ani_sketch_writer_add(ani_sketch_writer *w, const ani_sketch *sk, const char *const *genomeNames /* [sk's genomes] or NULL */);
int ani_sketch_writer_close(ani_sketch_writer *w);
/* gives the writer up: the partial file is removed, the handle released (also the way out after a failed add: a close after a failed
 * add fails and removes the file as well) */
void ani_sketch_writer_abort(ani_sketch_writer *w);
int ani_sketch_file_info(const char *path, ani_params_t *p, int32_t *nContigs, int32_t *nGenomes, uint64_t *nMinimizers);
const char *ani_sketch_genome_name(const ani_sketch *sk, int32_t genome);
int ani_sketch_tables(const ani_sketch *sk, const int32_t **contigLen, const int32_t **genomeContigStart);

/* ---- mapping: replaces skch::Map::Map + callback (computeMap.hpp:93-102, mapQuery :112) for ONE query genome.
 * Mappings are returned in the reference's callback order (fragment, then candidate).  *totalQueryFragments is
 * set (not accumulated). */
int ani_map_query(ani_ctx *ctx, const ani_sketch *sk, const ani_seq_batch_t *query,
                  ani_mapping_t **out, size_t *n, uint64_t *totalQueryFragments);
/* fragment sketches of one query genome (computeMap.hpp:260-274): offsets[nFragments+1], concatenated sorted
 * unique hashes — for parity tests */
int ani_query_sketch(ani_ctx *ctx, const ani_params_t *p, const ani_seq_batch_t *query,
                     uint32_t **hashes, uint64_t **offsets, size_t *nFragments);

/* ---- kept fragment sketches.  Map::Map sketches every 3-kb fragment of a query genome (computeMap.hpp:252-274) each time the
 * genome is mapped, and the reference driver re-does that per reference split (core_genome_identity.cpp:81-106).  Here the
 * fragment sketches of a batch of query genomes can be built once and kept on the device:
 *   ani_fragset_build        from query genomes;
 *   ani_sketch_records_self  all-vs-all: the genomes are references AND queries — one pass over their k-mer hashes yields both
 *                            the reference minimizer records (as ani_sketch_records) and the fragment sketches;
 *   ani_map_cgi_fragset      = ani_map_cgi_batch for the genomes of a kept set (any sketch with the same parameters).
 * qryGenomeId = firstQueryId + index of the genome in the set. */
int ani_fragset_build(ani_ctx *ctx, const ani_params_t *p, const ani_seq_batch_t *queries, ani_fragset **out);
int ani_sketch_records_self(ani_ctx *ctx, const ani_params_t *p, const ani_seq_batch_t *genomes, int32_t seqIdBase,
                            void **devRecords, size_t *n, ani_fragset **frags);
int ani_map_cgi_fragset(ani_ctx *ctx, const ani_sketch *sk, const ani_fragset *frags, int32_t firstQueryId,
                        ani_cgi_t **out, size_t *m);
void ani_fragset_free(ani_fragset *frags);
/* several kept sets in one call (set i's genomes get the query ids firstQueryIds[i] + 0, 1, ...): a streamed reference set builds
 * each of its index chunks once per call, so hand over everything that is to be mapped.  Rows: set by set, (query, reference). */
int ani_map_cgi_fragsets(ani_ctx *ctx, const ani_sketch *sk, int32_t nSets, const ani_fragset *const *frags, const int32_t *firstQueryIds,
                         ani_cgi_t **out, size_t *m);
int ani_fragset_info(const ani_fragset *frags, int32_t *nGenomes, int64_t *nFragments, uint64_t *nHashes);
/* A kept set as ONE device buffer (header + tables + hash pool; plain bytes for RCCL send/recv or all-gather) and back: the
 * multi-GPU path shards the REFERENCES, every GPU indexes its shard only, and the query fragment sketches — a third of the size
 * of the minimizer records — travel between the GPUs.  ani_fragset_unpack returns a view: the arrays stay
 * in devBuf, which must outlive the set. */
int ani_fragset_pack_bytes(const ani_fragset *frags, size_t *bytes);
int ani_fragset_pack(ani_ctx *ctx, const ani_fragset *frags, void *devBuf, size_t cap, size_t *bytes);
int ani_fragset_unpack(ani_ctx *ctx, const void *devBuf, size_t bytes, ani_fragset **out);
/* ONE set over several packed sets that lie in one device buffer at a fixed pitch (slot i at devBuf + i * slotBytes) — what an
 * all-gather of the ranks' packed sets delivers.  slotQueryBase[i] = query id of slot i's first genome (ascending over the slots),
 * or < 0 to leave the slot out (the rank's own); rows come back with qryGenomeId = firstQueryId + that id.  The hash pools stay in
 * devBuf (which must outlive the set and span < 2^32 hashes); only the per-fragment tables are copied.  Mapping the merged set is one
 * pass of the kernels instead of one per set — the reference's loop has this shape too: per reference split, ALL queries
 * (core_genome_identity.cpp:55-106). */
int ani_fragset_unpack_merged(ani_ctx *ctx, const void *devBuf, size_t slotBytes, int32_t nSlots, const int32_t *slotQueryBase,
                              ani_fragset **out);

/* ---- reducer: replaces cgi::computeCGI (computeCoreIdentity.hpp:166-298) for one query genome ---- */
int ani_compute_cgi(ani_ctx *ctx, const ani_sketch *sk, const ani_mapping_t *mappings, size_t n,
                    uint64_t totalQueryFragments, int32_t queryFileNo, ani_cgi_t **out, size_t *m);

/* ---- fused many-to-many path: the query loop of core_genome_identity.cpp:81-106 for a whole batch of query
 * genomes, device-resident end to end (Map + computeCGI per query; no mapping records leave the GPU).
 * Row order: query id ascending, then refGenomeId ascending.  qryGenomeId = firstQueryId + index in batch. */
int ani_map_cgi_batch(ani_ctx *ctx, const ani_sketch *sk, const ani_seq_batch_t *queries, int32_t firstQueryId,
                      ani_cgi_t **out, size_t *m);

/* ---- synthetic genomes (benchmark input generator; DESIGN.md §Synthetic data) ----
 * Writes nGenomes genomes of genomeLen bases, 2-bit packed, genome i at word offset i*ceil(genomeLen/16) of devOut
 * (device memory, caller-allocated).  `variant` re-draws the substitutions with the cluster ancestors kept (0 = base set). */
int ani_synth_packed(ani_ctx *ctx, uint64_t seed, uint64_t variant, int32_t firstGenomeId, int32_t nGenomes, int32_t genomeLen, void *devOut);
/* the same population with clusters of `clusterSize` related genomes instead of 20 (a species-dense database: hundreds of strains of
 * one species; members beyond the 20th cycle through the non-zero divergence rates) */
int ani_synth_packed_clusters(ani_ctx *ctx, uint64_t seed, uint64_t variant, int32_t firstGenomeId, int32_t nGenomes, int32_t genomeLen,
                              int32_t clusterSize, void *devOut);

#ifdef __cplusplus
}
#endif
#endif
