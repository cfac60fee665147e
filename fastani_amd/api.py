"""Host-side mirror of the reference seams over the C-ABI (include/ani_abi.h), via ctypes.

    Sketch(engine, params, genomes)        ≙ skch::Sketch::Sketch          (src/map/include/winSketch.hpp:109)
    Sketch.map_query(genome)               ≙ skch::Map::Map + callback     (src/map/include/computeMap.hpp:93)
    Sketch.compute_cgi(maps, total, qid)   ≙ cgi::computeCGI               (src/cgi/include/computeCoreIdentity.hpp:166)
    Sketch.map_cgi_batch(genomes, first)   ≙ the query loop of src/cgi/core_genome_identity.cpp:81-106

Everything here is plumbing: numpy arrays in, numpy record arrays out.  All compute happens in
libfastani_amd.so (hand-written HIP kernels, gfx950); there is no Python or CPU fallback.
"""
import ctypes as C
import weakref

import numpy as np

MINIMIZER_DT = np.dtype([("hash", "<u4"), ("seqId", "<i4"), ("wpos", "<i4")])
MAPPING_DT = np.dtype([("queryLen", "<i4"), ("refStartPos", "<i4"), ("refEndPos", "<i4"), ("queryStartPos", "<i4"),
                       ("queryEndPos", "<i4"), ("refSeqId", "<i4"), ("querySeqId", "<i4"), ("nucIdentity", "<f4"),
                       ("nucIdentityUpperBound", "<f4"), ("sketchSize", "<i4"), ("conservedSketches", "<i4")])
CGI_DT = np.dtype([("refGenomeId", "<i4"), ("qryGenomeId", "<i4"), ("countSeq", "<i4"),
                   ("totalQueryFragments", "<i4"), ("identity", "<f4")])

ANI_SEQ_HOST_ASCII = 0
ANI_SEQ_DEVICE_PACKED2 = 1
ANI_SEQ_HOST_ASCII_PTRS = 2
ANI_SEQ_DEVICE_BATCH = 3
ANI_SEQ_HOST_MIXED_PTRS = 4


class AniError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__("ani error %d: %s" % (code, msg))
        self.code = code


class Params(C.Structure):
    _fields_ = [("kmerSize", C.c_int32), ("windowSize", C.c_int32), ("fragLen", C.c_int32),
                ("percentageIdentity", C.c_float)]


class SeqBatch(C.Structure):
    _fields_ = [("layout", C.c_int32), ("nGenomes", C.c_int32), ("nContigs", C.c_int32),
                ("genomeContigStart", C.c_void_p), ("contigOffset", C.c_void_p), ("contigLen", C.c_void_p),
                ("data", C.c_void_p)]


class Counters(C.Structure):
    _fields_ = [(n, C.c_uint64) for n in ("refBases", "refMinimizers", "refUniqueHashes", "queryGenomes", "queryFragments",
                                          "queryBases", "querySketchHashes", "seedHits", "l1Candidates", "l2WindowEntries",
                                          "l2Steps", "l2QueryHashes", "l2WindowEntriesB", "l2QueryHashesB", "l2Launches", "l2FastCandidates", "l2SlowCandidates", "l2SlowLimit", "l2SlowDup", "l2SlowOverflow", "mappings", "cgiRows", "indexChunks", "l1Probes", "l2ChunkHalvings", "indexChunkBuilds", "l1BigFragments", "l1MidFragments", "l1TinyFragments")] + \
               [(n, C.c_double) for n in ("msSketch", "msIndex", "msFragSketch", "msL1", "msL2", "msReduce", "msL2Kernel", "msL2Ranges", "msL2Codes", "msL2Slow", "msL2SimB", "msL1Probe", "msL1Main", "msL1Big", "msL1Tiny")]

    def as_dict(self):
        return {n: getattr(self, n) for n, _ in self._fields_}


def _bind(lib):
    vp = C.c_void_p
    sig = {
        "ani_init": (C.c_int, [C.c_int, C.POINTER(vp)]),
        "ani_shutdown": (None, [vp]),
        "ani_last_error": (C.c_char_p, []),
        "ani_free": (None, [vp]),
        "ani_device_free": (None, [vp, vp]),
        "ani_device_copy": (C.c_int, [vp, vp, vp, C.c_size_t]),
        "ani_device_alloc": (C.c_int, [vp, C.c_size_t, C.POINTER(vp)]),
        "ani_device_copy_peer": (C.c_int, [vp, vp, vp, vp, C.c_size_t]),
        "ani_pack_acgt": (C.c_int, [vp, C.c_int32, vp]),
        "ani_batch_upload": (C.c_int, [vp, C.POINTER(SeqBatch), C.POINTER(vp)]),
        "ani_batch_free": (None, [vp]),
        "ani_get_counters": (C.c_int, [vp, C.POINTER(Counters)]),
        "ani_reset_counters": (C.c_int, [vp]),
        "ani_params_default": (C.c_int, [C.POINTER(Params), C.c_int, C.c_int]),
        "ani_recommended_window": (C.c_int, [C.c_int, C.c_int]),
        "ani_min_hits_relaxed": (C.c_int, [C.c_int, C.c_int, C.c_float]),
        "ani_identity": (C.c_int, [C.c_int, C.c_int, C.c_int, C.POINTER(C.c_float), C.POINTER(C.c_float)]),
        "ani_sketch_build": (C.c_int, [vp, C.POINTER(Params), C.POINTER(SeqBatch), C.POINTER(vp)]),
        "ani_sketch_destroy": (None, [vp]),
        "ani_sketch_export": (C.c_int, [vp, C.POINTER(vp), C.POINTER(C.c_size_t)]),
        "ani_sketch_stats": (C.c_int, [vp, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64), C.POINTER(C.c_uint64),
                                       C.POINTER(C.c_int32), C.POINTER(C.c_int32)]),
        "ani_sketch_chunks": (C.c_int, [vp, C.POINTER(C.c_int32), vp, C.c_int32]),
        "ani_sketch_records": (C.c_int, [vp, C.POINTER(Params), C.POINTER(SeqBatch), C.c_int32, C.POINTER(vp), C.POINTER(C.c_size_t)]),
        "ani_sketch_from_records": (C.c_int, [vp, C.POINTER(Params), vp, C.c_size_t, vp, C.c_int32, vp, C.c_int32, C.POINTER(vp)]),
        "ani_sketch_from_record_parts": (C.c_int, [vp, C.POINTER(Params), C.c_int32, vp, vp, vp, vp, C.c_int32, vp, C.c_int32, C.POINTER(vp)]),
        "ani_sketch_adopt_record_parts": (C.c_int, [vp, C.POINTER(Params), C.c_int32, vp, vp, vp, vp, C.c_int32, vp, C.c_int32, C.POINTER(vp)]),
        "ani_map_query": (C.c_int, [vp, vp, C.POINTER(SeqBatch), C.POINTER(vp), C.POINTER(C.c_size_t), C.POINTER(C.c_uint64)]),
        "ani_query_sketch": (C.c_int, [vp, C.POINTER(Params), C.POINTER(SeqBatch), C.POINTER(vp), C.POINTER(vp), C.POINTER(C.c_size_t)]),
        "ani_compute_cgi": (C.c_int, [vp, vp, vp, C.c_size_t, C.c_uint64, C.c_int32, C.POINTER(vp), C.POINTER(C.c_size_t)]),
        "ani_sketch_save": (C.c_int, [vp, C.c_char_p, vp]),
        "ani_sketch_load": (C.c_int, [vp, C.c_char_p, C.c_int32, C.c_int32, C.POINTER(vp)]),
        "ani_sketch_writer_open": (C.c_int, [C.c_char_p, C.POINTER(vp)]),
        "ani_sketch_writer_add": (C.c_int, [vp, vp, vp]),
        "ani_sketch_writer_close": (C.c_int, [vp]),
        "ani_sketch_writer_abort": (None, [vp]),
        "ani_device_memory": (C.c_int, [vp, C.POINTER(C.c_size_t), C.POINTER(C.c_size_t)]),
        "ani_pool_prewarm_index": (C.c_int, [vp, C.c_uint64]),
        "ani_pool_stats": (C.c_int, [vp, C.POINTER(C.c_uint64)]),
        "ani_sketch_file_info": (C.c_int, [C.c_char_p, C.POINTER(Params), C.POINTER(C.c_int32), C.POINTER(C.c_int32), C.POINTER(C.c_uint64)]),
        "ani_sketch_genome_name": (C.c_char_p, [vp, C.c_int32]),
        "ani_sketch_tables": (C.c_int, [vp, C.POINTER(vp), C.POINTER(vp)]),
        "ani_fragset_build": (C.c_int, [vp, C.POINTER(Params), C.POINTER(SeqBatch), C.POINTER(vp)]),
        "ani_sketch_records_self": (C.c_int, [vp, C.POINTER(Params), C.POINTER(SeqBatch), C.c_int32, C.POINTER(vp), C.POINTER(C.c_size_t), C.POINTER(vp)]),
        "ani_map_cgi_fragset": (C.c_int, [vp, vp, vp, C.c_int32, C.POINTER(vp), C.POINTER(C.c_size_t)]),
        "ani_fragset_free": (None, [vp]),
        "ani_map_cgi_fragsets": (C.c_int, [vp, vp, C.c_int32, vp, vp, C.POINTER(vp), C.POINTER(C.c_size_t)]),
        "ani_fragset_info": (C.c_int, [vp, C.POINTER(C.c_int32), C.POINTER(C.c_int64), C.POINTER(C.c_uint64)]),
        "ani_fragset_pack_bytes": (C.c_int, [vp, C.POINTER(C.c_size_t)]),
        "ani_fragset_pack": (C.c_int, [vp, vp, vp, C.c_size_t, C.POINTER(C.c_size_t)]),
        "ani_fragset_unpack": (C.c_int, [vp, vp, C.c_size_t, C.POINTER(vp)]),
        "ani_fragset_unpack_merged": (C.c_int, [vp, vp, C.c_size_t, C.c_int32, vp, C.POINTER(vp)]),
        "ani_sketch_residency": (C.c_int, [vp, C.POINTER(C.c_int32), C.POINTER(C.c_int32), C.POINTER(C.c_int32)]),
        "ani_sketch_set_ref_id_base": (C.c_int, [vp, C.c_int32]),
        "ani_map_cgi_batch": (C.c_int, [vp, vp, C.POINTER(SeqBatch), C.c_int32, C.POINTER(vp), C.POINTER(C.c_size_t)]),
        "ani_synth_packed": (C.c_int, [vp, C.c_uint64, C.c_uint64, C.c_int32, C.c_int32, C.c_int32, vp]),
        "ani_synth_packed_clusters": (C.c_int, [vp, C.c_uint64, C.c_uint64, C.c_int32, C.c_int32, C.c_int32, C.c_int32, vp]),
    }
    for name, (res, args) in sig.items():
        fn = getattr(lib, name)
        fn.restype = res
        fn.argtypes = args
    return sorted(sig)


ABI_SYMBOLS = None


class HostGenomes:
    """A batch of genomes in host memory: list of genomes, each a list of contigs (bytes / str / uint8 arrays).
    Bytes are passed as read from FASTA; the library upper-cases them (commonFunc.hpp:56-66)."""

    def __init__(self, genomes):
        self.contig_start = [0]
        lens, offs, parts = [], [], []
        off = 0
        for contigs in genomes:
            for c in contigs:
                if isinstance(c, str):
                    c = c.encode()
                a = np.frombuffer(bytes(c), dtype=np.uint8) if isinstance(c, (bytes, bytearray)) else np.ascontiguousarray(c, dtype=np.uint8)
                parts.append(a)
                lens.append(len(a))
                offs.append(off)
                off += len(a)
            self.contig_start.append(len(lens))
        self.data = np.concatenate(parts) if parts else np.zeros(1, dtype=np.uint8)
        if len(self.data) == 0:
            self.data = np.zeros(1, dtype=np.uint8)
        self.gcs = np.asarray(self.contig_start, dtype=np.int32)
        self.off = np.asarray(offs if offs else [0], dtype=np.int64)
        self.len = np.asarray(lens if lens else [0], dtype=np.int32)
        self.n_genomes = len(genomes)
        self.n_contigs = len(lens)

    def batch(self):
        b = SeqBatch()
        b.layout = ANI_SEQ_HOST_ASCII
        b.nGenomes = self.n_genomes
        b.nContigs = self.n_contigs
        b.genomeContigStart = self.gcs.ctypes.data
        b.contigOffset = self.off.ctypes.data
        b.contigLen = self.len.ctypes.data
        b.data = self.data.ctypes.data
        return b


class DeviceGenomes:
    """Single-contig genomes of equal length, 2-bit packed, already resident in device memory at `dev_ptr`
    (genome i at word offset i*ceil(len/16)) — the synthetic benchmark input."""

    def __init__(self, dev_ptr, n_genomes, genome_len, first=0, count=None):
        count = n_genomes - first if count is None else count
        words = (genome_len + 15) // 16
        self.gcs = np.arange(count + 1, dtype=np.int32)
        self.off = (np.arange(first, first + count, dtype=np.int64) * words)
        self.len = np.full(count, genome_len, dtype=np.int32)
        self.dev_ptr = dev_ptr
        self.n_genomes = count
        self.n_contigs = count

    def batch(self):
        b = SeqBatch()
        b.layout = ANI_SEQ_DEVICE_PACKED2
        b.nGenomes = self.n_genomes
        b.nContigs = self.n_contigs
        b.genomeContigStart = self.gcs.ctypes.data
        b.contigOffset = self.off.ctypes.data
        b.contigLen = self.len.ctypes.data
        b.data = self.dev_ptr
        return b


class UploadedGenomes:
    """Host genomes packed and copied to the device once (ani_batch_upload); usable as references and as queries."""

    def __init__(self, engine, genomes, ptrs=False, mixed=False):
        self.e = engine
        self.host = genomes if isinstance(genomes, HostGenomes) else HostGenomes(genomes)
        b = self.host.batch()
        if ptrs or mixed:                          # per-contig pointer layout (ANI_SEQ_HOST_ASCII_PTRS)
            self._ptrs = (self.host.data.ctypes.data + self.host.off[:max(self.host.n_contigs, 1)]).astype(np.uint64)
            b.layout = ANI_SEQ_HOST_ASCII_PTRS
            b.data = self._ptrs.ctypes.data
            b.contigOffset = None
        if mixed:                                  # ANI_SEQ_HOST_MIXED_PTRS: every contig that ani_pack_acgt accepts goes over as 2-bit codes
            nc = self.host.n_contigs
            self._kind = np.zeros(max(nc, 1), dtype=np.int64)
            self._packed = []
            for c in range(nc):
                ln = int(self.host.len[c])
                out = np.zeros((ln + 15) // 16 + 2, dtype=np.uint32)
                if engine.lib.ani_pack_acgt(C.c_void_p(int(self._ptrs[c])), ln, out.ctypes.data_as(C.c_void_p)) == 1:
                    self._kind[c] = 1
                    self._packed.append(out)
                    self._ptrs[c] = out.ctypes.data
            b.layout = ANI_SEQ_HOST_MIXED_PTRS
            b.contigOffset = self._kind.ctypes.data
            self.packed_contigs = int(self._kind[:nc].sum())
        h = C.c_void_p()
        engine._chk(engine.lib.ani_batch_upload(engine.h, C.byref(b), C.byref(h)))
        self.h = h
        self.n_genomes = self.host.n_genomes
        self.n_contigs = self.host.n_contigs

    def batch(self):
        b = self.host.batch()
        b.layout = ANI_SEQ_DEVICE_BATCH
        b.data = self.h
        return b

    def close(self):
        if getattr(self, "h", None):
            self.e.lib.ani_batch_free(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def _as_batch(g):
    if isinstance(g, (HostGenomes, DeviceGenomes, UploadedGenomes)):
        return g
    return HostGenomes(g)


class _OwnedBuffer:
    """a buffer the library allocated (ani_free releases it), exposed through the array interface"""

    def __init__(self, lib, address, n, dt):
        self._lib, self._address = lib, address
        self.__array_interface__ = dict(shape=(n,), typestr=dt.str, descr=dt.descr, data=(address, False), version=3)

    def __del__(self):
        try:
            self._lib.ani_free(C.c_void_p(self._address))
        except Exception:                       # interpreter shutdown: the process frees it
            pass


class Engine:
    """One ani_ctx: one device, one host thread."""

    def __init__(self, lib, device=0):
        global ABI_SYMBOLS
        self.lib = lib
        ABI_SYMBOLS = _bind(lib)
        h = C.c_void_p()
        self._chk(lib.ani_init(device, C.byref(h)))
        self.h = h
        self._sketches = weakref.WeakSet()      # live sketches are destroyed before the context (see close)

    def _chk(self, rc):
        if rc != 0:
            raise AniError(rc, (self.lib.ani_last_error() or b"").decode())

    def close(self):
        if getattr(self, "h", None):
            for sk in list(self._sketches):
                sk.close()
            self.lib.ani_shutdown(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def params(self, k=16, frag_len=3000):
        p = Params()
        self._chk(self.lib.ani_params_default(C.byref(p), k, frag_len))
        return p

    def counters(self):
        c = Counters()
        self._chk(self.lib.ani_get_counters(self.h, C.byref(c)))
        return c.as_dict()

    def reset_counters(self):
        self._chk(self.lib.ani_reset_counters(self.h))

    def _take(self, ptr, n, dt):
        """the library's malloc'ed result buffer as a numpy array WITHOUT a copy: the array (and every view of it) keeps an owner
        object alive that hands the buffer to ani_free when the last of them is gone.  (A copy of the 785 k result rows of the
        1000 x 1000 job into a fresh array was 2 - 3 ms of page faults and memmove per step.)"""
        if n == 0 or not ptr.value:
            self.lib.ani_free(ptr)
            return np.zeros(0, dtype=dt)
        return np.asarray(_OwnedBuffer(self.lib, ptr.value, n, np.dtype(dt)))

    def query_sketch(self, params, genomes):
        g = _as_batch(genomes)
        b = g.batch()
        hp, op, n = C.c_void_p(), C.c_void_p(), C.c_size_t()
        self._chk(self.lib.ani_query_sketch(self.h, C.byref(params), C.byref(b), C.byref(hp), C.byref(op), C.byref(n)))
        offs = self._take(op, n.value + 1, np.dtype("<u8"))
        hashes = self._take(hp, int(offs[-1]), np.dtype("<u4"))
        return [hashes[int(offs[i]):int(offs[i + 1])] for i in range(n.value)]

    def synth_packed(self, seed, first_genome, n_genomes, genome_len, dev_ptr, variant=0, cluster_size=20):
        self._chk(self.lib.ani_synth_packed_clusters(self.h, seed, variant, first_genome, n_genomes, genome_len, cluster_size, dev_ptr))

    def sketch_records(self, params, genomes, seq_id_base):
        """-> (device pointer to 12-byte records, count); free with device_free."""
        g = _as_batch(genomes)
        b = g.batch()
        p, n = C.c_void_p(), C.c_size_t()
        self._chk(self.lib.ani_sketch_records(self.h, C.byref(params), C.byref(b), seq_id_base, C.byref(p), C.byref(n)))
        return p.value, n.value

    def sketch_records_self(self, params, genomes, seq_id_base):
        """all-vs-all: -> (device pointer to 12-byte records, count, FragmentSet of the same genomes)"""
        g = _as_batch(genomes)
        b = g.batch()
        p, n, f = C.c_void_p(), C.c_size_t(), C.c_void_p()
        self._chk(self.lib.ani_sketch_records_self(self.h, C.byref(params), C.byref(b), seq_id_base, C.byref(p), C.byref(n), C.byref(f)))
        return p.value, n.value, FragmentSet(self, f)

    def fragment_set(self, params, genomes):
        g = _as_batch(genomes)
        b = g.batch()
        f = C.c_void_p()
        self._chk(self.lib.ani_fragset_build(self.h, C.byref(params), C.byref(b), C.byref(f)))
        return FragmentSet(self, f)

    def device_alloc(self, nbytes):
        p = C.c_void_p()
        self._chk(self.lib.ani_device_alloc(self.h, int(nbytes), C.byref(p)))
        return p.value

    def device_free(self, ptr):
        self.lib.ani_device_free(self.h, ptr)

    def pool_prewarm_index(self, n_minimizers):
        """a hint (ani_abi.h): the device memory sketching and indexing about n_minimizers minimizers will take, reserved now"""
        self._chk(self.lib.ani_pool_prewarm_index(self.h, int(n_minimizers)))

    def pool_stats(self):
        """the device allocator (ani_pool_stats): segment / free / live bytes, segments, hipMalloc calls / bytes / microseconds so far"""
        out = (C.c_uint64 * 8)()
        self._chk(self.lib.ani_pool_stats(self.h, out))
        keys = ("segment_bytes", "free_bytes", "live_bytes", "segments", "hipmalloc_calls", "hipmalloc_bytes", "hipmalloc_us", "largest_free_extent")
        return dict(zip(keys, [int(x) for x in out]))

    def device_copy(self, dst, src, nbytes):
        self._chk(self.lib.ani_device_copy(self.h, dst, src, nbytes))


class FragmentSet:
    """Fragment sketches of a batch of query genomes kept on the device (ani_fragset)."""

    def __init__(self, engine, handle, keepalive=None):
        self.e = engine
        self.h = handle
        self._keepalive = keepalive          # the buffer a view (unpack) points into

    def info(self):
        g, f, h = C.c_int32(), C.c_int64(), C.c_uint64()
        self.e._chk(self.e.lib.ani_fragset_info(self.h, C.byref(g), C.byref(f), C.byref(h)))
        return dict(genomes=g.value, fragments=f.value, hashes=h.value)

    def packed_bytes(self):
        n = C.c_size_t()
        self.e._chk(self.e.lib.ani_fragset_pack_bytes(self.h, C.byref(n)))
        return n.value

    def pack_into(self, dev_ptr, cap):
        """the set as one device buffer (header + tables + hash pool) at dev_ptr; returns the bytes used"""
        n = C.c_size_t()
        self.e._chk(self.e.lib.ani_fragset_pack(self.e.h, self.h, dev_ptr, cap, C.byref(n)))
        return n.value

    @staticmethod
    def unpack(engine, dev_ptr, nbytes, keepalive=None):
        """view of a packed set at dev_ptr (the arrays stay in that buffer: pass its owner as keepalive)"""
        h = C.c_void_p()
        engine._chk(engine.lib.ani_fragset_unpack(engine.h, dev_ptr, nbytes, C.byref(h)))
        return FragmentSet(engine, h, keepalive)

    @staticmethod
    def unpack_merged(engine, dev_ptr, slot_bytes, slot_query_base, keepalive=None):
        """ONE set over the packed sets at dev_ptr + i * slot_bytes (an all-gather's output); slot_query_base[i] = query id of slot i's
        first genome, or -1 to leave the slot out.  Map it with first_query_id = 0."""
        q = np.ascontiguousarray(slot_query_base, dtype=np.int32)
        h = C.c_void_p()
        engine._chk(engine.lib.ani_fragset_unpack_merged(engine.h, dev_ptr, slot_bytes, len(q), q.ctypes.data, C.byref(h)))
        return FragmentSet(engine, h, keepalive)

    def close(self):
        if getattr(self, "h", None):
            self.e.lib.ani_fragset_free(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class SketchWriter:
    """one sketch file from several sketches added one after the other (ani_sketch_writer_*): a reference set whose records exceed the
    device memory is written block by block"""
    def __init__(self, engine, path):
        self.e = engine
        h = C.c_void_p()
        engine._chk(engine.lib.ani_sketch_writer_open(str(path).encode(), C.byref(h)))
        self.h = h

    def add(self, sketch, names=None):
        arr = None
        if names is not None:
            arr = (C.c_char_p * len(names))(*[n.encode() if isinstance(n, str) else n for n in names])
        self.e._chk(self.e.lib.ani_sketch_writer_add(self.h, sketch.h, arr))

    def close(self):
        if self.h:
            h, self.h = self.h, None
            self.e._chk(self.e.lib.ani_sketch_writer_close(h))

    def abort(self):
        """give the file up (it is removed)"""
        if self.h:
            h, self.h = self.h, None
            self.e.lib.ani_sketch_writer_abort(h)

    def __enter__(self):
        return self

    def __exit__(self, exc_type, exc, tb):
        if exc_type is None:
            self.close()
        else:
            self.abort()
        return False

    def __del__(self):
        try:
            self.abort()             # a writer dropped without close(): no half-written file stays behind
        except Exception:
            pass


class Sketch:
    def __init__(self, engine, params, genomes=None, records=None, record_parts=None, file=None, genome_range=(0, -1), adopt=False):
        """Either `genomes` (≙ Sketch::Sketch over the reference files), or for the multi-GPU staging path
        records=(dev_ptr, n, contig_len[int32], genome_contig_start[int32]) or
        record_parts=(dev_ptrs, counts, part_genome_start[nParts+1], contig_len, genome_contig_start); adopt=True hands the
        record buffers of record_parts over to the library (ani_sketch_adopt_record_parts: no copy for a streamed set)."""
        self.e = engine
        self.params = params
        h = C.c_void_p()
        if file is not None:
            engine._chk(engine.lib.ani_sketch_load(engine.h, str(file).encode(), genome_range[0], genome_range[1], C.byref(h)))
        elif record_parts is not None:
            ptrs, counts, pgs, clen, gcs = record_parts
            ptrs = np.ascontiguousarray(ptrs, dtype=np.uint64)
            counts = np.ascontiguousarray(counts, dtype=np.uint64)
            pgs = np.ascontiguousarray(pgs, dtype=np.int32)
            clen = np.ascontiguousarray(clen, dtype=np.int32)
            gcs = np.ascontiguousarray(gcs, dtype=np.int32)
            fn = engine.lib.ani_sketch_adopt_record_parts if adopt else engine.lib.ani_sketch_from_record_parts
            engine._chk(fn(engine.h, C.byref(params), len(ptrs), ptrs.ctypes.data, counts.ctypes.data,
                        pgs.ctypes.data, clen.ctypes.data, len(clen), gcs.ctypes.data, len(gcs) - 1, C.byref(h)))
        elif records is not None:
            ptr, n, clen, gcs = records
            clen = np.ascontiguousarray(clen, dtype=np.int32)
            gcs = np.ascontiguousarray(gcs, dtype=np.int32)
            engine._chk(engine.lib.ani_sketch_from_records(engine.h, C.byref(params), ptr, n, clen.ctypes.data, len(clen),
                                                           gcs.ctypes.data, len(gcs) - 1, C.byref(h)))
        else:
            g = _as_batch(genomes)
            b = g.batch()
            engine._chk(engine.lib.ani_sketch_build(engine.h, C.byref(params), C.byref(b), C.byref(h)))
        self.h = h
        engine._sketches.add(self)

    def close(self):
        if getattr(self, "h", None):
            self.e.lib.ani_sketch_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def minimizers(self):
        p, n = C.c_void_p(), C.c_size_t()
        self.e._chk(self.e.lib.ani_sketch_export(self.h, C.byref(p), C.byref(n)))
        return self.e._take(p, n.value, MINIMIZER_DT)

    def save(self, path, names=None):
        arr = None
        if names is not None:
            arr = (C.c_char_p * len(names))(*[n.encode() if isinstance(n, str) else n for n in names])
        self.e._chk(self.e.lib.ani_sketch_save(self.h, str(path).encode(), arr))

    def genome_names(self):
        return [self.e.lib.ani_sketch_genome_name(self.h, g).decode() for g in range(self.stats()["genomes"])]

    def chunks(self):
        """first genome id of every index chunk of the reference set"""
        n = C.c_int32()
        self.e._chk(self.e.lib.ani_sketch_chunks(self.h, C.byref(n), None, 0))
        first = np.zeros(max(n.value, 1), dtype=np.int32)
        self.e._chk(self.e.lib.ani_sketch_chunks(self.h, C.byref(n), first.ctypes.data, n.value))
        return [int(x) for x in first[:n.value]]

    def stats(self):
        a, b, c = C.c_uint64(), C.c_uint64(), C.c_uint64()
        d, f = C.c_int32(), C.c_int32()
        self.e._chk(self.e.lib.ani_sketch_stats(self.h, C.byref(a), C.byref(b), C.byref(c), C.byref(d), C.byref(f)))
        return dict(minimizers=a.value, unique=b.value, totalLength=c.value, contigs=d.value, genomes=f.value)

    def map_query(self, genome):
        """genome = list of contigs of ONE query genome -> (mappings, totalQueryFragments)"""
        g = _as_batch([genome]) if not isinstance(genome, (HostGenomes, DeviceGenomes, UploadedGenomes)) else genome
        b = g.batch()
        p, n, tot = C.c_void_p(), C.c_size_t(), C.c_uint64()
        self.e._chk(self.e.lib.ani_map_query(self.e.h, self.h, C.byref(b), C.byref(p), C.byref(n), C.byref(tot)))
        return self.e._take(p, n.value, MAPPING_DT), tot.value

    def compute_cgi(self, mappings, total_fragments, query_id):
        m = np.ascontiguousarray(mappings, dtype=MAPPING_DT)
        p, n = C.c_void_p(), C.c_size_t()
        self.e._chk(self.e.lib.ani_compute_cgi(self.e.h, self.h, m.ctypes.data if len(m) else None, len(m), total_fragments,
                                               query_id, C.byref(p), C.byref(n)))
        return self.e._take(p, n.value, CGI_DT)

    def map_cgi_fragset(self, fragset, first_query_id=0):
        p, n = C.c_void_p(), C.c_size_t()
        self.e._chk(self.e.lib.ani_map_cgi_fragset(self.e.h, self.h, fragset.h, first_query_id, C.byref(p), C.byref(n)))
        return self.e._take(p, n.value, CGI_DT)

    def map_cgi_fragsets(self, fragsets, first_query_ids):
        """several kept sets in one call (a streamed reference set builds each index chunk once per call)"""
        hs = (C.c_void_p * len(fragsets))(*[f.h for f in fragsets])
        ids = np.ascontiguousarray(first_query_ids, dtype=np.int32)
        p, n = C.c_void_p(), C.c_size_t()
        self.e._chk(self.e.lib.ani_map_cgi_fragsets(self.e.h, self.h, len(fragsets), hs, ids.ctypes.data, C.byref(p), C.byref(n)))
        return self.e._take(p, n.value, CGI_DT)

    def set_ref_id_base(self, base):
        """this sketch is a shard / block of a larger set: CGI rows of the batch entry points report refGenomeId + base"""
        self.e._chk(self.e.lib.ani_sketch_set_ref_id_base(self.h, int(base)))

    def residency(self):
        a, b, c = C.c_int32(), C.c_int32(), C.c_int32()
        self.e._chk(self.e.lib.ani_sketch_residency(self.h, C.byref(a), C.byref(b), C.byref(c)))
        return dict(streaming=bool(a.value), max_resident=b.value, resident_now=c.value)

    def map_cgi_batch(self, genomes, first_query_id=0):
        g = _as_batch(genomes)
        b = g.batch()
        p, n = C.c_void_p(), C.c_size_t()
        self.e._chk(self.e.lib.ani_map_cgi_batch(self.e.h, self.h, C.byref(b), first_query_id, C.byref(p), C.byref(n)))
        return self.e._take(p, n.value, CGI_DT)
