"""ctypes binding of the CPU oracle (oracle/libani_oracle.so) — TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg import this module.
"""
import ctypes as C
import os
import subprocess
import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_DIR = os.path.join(ROOT, "oracle")
LIB_PATH = os.path.join(ORACLE_DIR, "libani_oracle.so")
REF_DUMP = os.path.join(ORACLE_DIR, "_ref", "ref_dump")
REF_BIN = os.path.join(ORACLE_DIR, "_ref", "fastANI_ref")

MINIMIZER_DT = np.dtype([("hash", "<u4"), ("seqId", "<i4"), ("wpos", "<i4")])
MAPPING_DT = np.dtype([("queryLen", "<i4"), ("refStartPos", "<i4"), ("refEndPos", "<i4"), ("queryStartPos", "<i4"),
                       ("queryEndPos", "<i4"), ("refSeqId", "<i4"), ("querySeqId", "<i4"), ("nucIdentity", "<f4"),
                       ("nucIdentityUpperBound", "<f4"), ("sketchSize", "<i4"), ("conservedSketches", "<i4")])
CGI_DT = np.dtype([("refGenomeId", "<i4"), ("qryGenomeId", "<i4"), ("countSeq", "<i4"),
                   ("totalQueryFragments", "<i4"), ("identity", "<f4")])
assert MINIMIZER_DT.itemsize == 12 and MAPPING_DT.itemsize == 44 and CGI_DT.itemsize == 20

_lib = None


def build():
    subprocess.check_call(["make", "-s", "-C", ORACLE_DIR], stdout=subprocess.DEVNULL)


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            build()
        L = C.CDLL(LIB_PATH)
        u8p = C.POINTER(C.c_uint8)
        L.orc_hash_kmer.restype = C.c_uint32
        L.orc_hash_kmer.argtypes = [C.c_void_p, C.c_int]
        L.orc_winnow.restype = C.c_size_t
        L.orc_winnow.argtypes = [C.c_void_p, C.c_int32, C.c_int, C.c_int, C.c_int32, C.c_void_p]
        L.orc_upper.argtypes = [C.c_void_p, C.c_size_t]
        L.orc_binomial_Q.restype = C.c_double
        L.orc_binomial_Q.argtypes = [C.c_uint, C.c_double, C.c_uint]
        L.orc_min_hits_relaxed.restype = C.c_int
        L.orc_min_hits_relaxed.argtypes = [C.c_int, C.c_int, C.c_float]
        L.orc_recommended_window.restype = C.c_int
        L.orc_recommended_window.argtypes = [C.c_double, C.c_int, C.c_int, C.c_float, C.c_int, C.c_uint64]
        L.orc_identity.argtypes = [C.c_int, C.c_int, C.c_int, C.POINTER(C.c_float), C.POINTER(C.c_float)]
        L.orc_sketch_new.restype = C.c_void_p
        L.orc_sketch_new.argtypes = [C.c_int, C.c_int]
        L.orc_sketch_add_contig.argtypes = [C.c_void_p, C.c_void_p, C.c_int32]
        L.orc_sketch_end_genome.argtypes = [C.c_void_p]
        L.orc_sketch_finish.argtypes = [C.c_void_p]
        L.orc_sketch_size.restype = C.c_size_t
        L.orc_sketch_size.argtypes = [C.c_void_p]
        L.orc_sketch_data.restype = C.c_void_p
        L.orc_sketch_data.argtypes = [C.c_void_p]
        L.orc_sketch_unique.restype = C.c_size_t
        L.orc_sketch_unique.argtypes = [C.c_void_p]
        L.orc_sketch_free.argtypes = [C.c_void_p]
        L.orc_fragment_sketch.restype = C.c_int
        L.orc_fragment_sketch.argtypes = [C.c_void_p, C.c_int32, C.c_int, C.c_int, C.c_void_p]
        L.orc_map_contig.restype = C.c_int
        L.orc_map_contig.argtypes = [C.c_void_p, C.c_void_p, C.c_int32, C.c_int, C.c_float, C.c_int32,
                                     C.POINTER(C.c_void_p), C.POINTER(C.c_size_t), C.POINTER(C.c_size_t)]
        L.orc_compute_cgi.restype = C.c_size_t
        L.orc_compute_cgi.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int, C.c_int32, C.c_int32, C.c_void_p, C.c_size_t]
        L.orc_synth_genome.argtypes = [C.c_uint64, C.c_int32, C.c_int32, C.c_void_p]
        L.orc_synth_genome_v.argtypes = [C.c_uint64, C.c_uint64, C.c_int32, C.c_int32, C.c_void_p]
        L.orc_synth_genome_c.argtypes = [C.c_uint64, C.c_uint64, C.c_int32, C.c_int32, C.c_int32, C.c_void_p]
        _lib = L
    return _lib


def _buf(a):
    return a.ctypes.data_as(C.c_void_p)


def as_bytes(seq):
    """str/bytes/uint8 array -> contiguous uint8 array (copy)."""
    if isinstance(seq, str):
        seq = seq.encode()
    if isinstance(seq, (bytes, bytearray)):
        return np.frombuffer(bytes(seq), dtype=np.uint8).copy()
    return np.ascontiguousarray(seq, dtype=np.uint8).copy()


def hash_kmer(kmer):
    b = as_bytes(kmer)
    return int(lib().orc_hash_kmer(_buf(b), len(b)))


def upper(seq):
    b = as_bytes(seq)
    lib().orc_upper(_buf(b), len(b))
    return b


def winnow(seq, k, w, seq_id=0):
    b = upper(seq)
    out = np.zeros(max(len(b), 1), dtype=MINIMIZER_DT)
    n = lib().orc_winnow(_buf(b), len(b), k, w, seq_id, _buf(out))
    return out[:n].copy()


def fragment_sketch(frag, k, w):
    b = upper(frag)
    out = np.zeros(max(len(b), 1), dtype=np.uint32)
    s = lib().orc_fragment_sketch(_buf(b), len(b), k, w, _buf(out))
    return out[:s].copy()


def recommended_window(k=16, frag_len=3000):
    return int(lib().orc_recommended_window(1e-3, k, 4, 80.0, frag_len, 5000000))


def min_hits_relaxed(s, k=16, identity=80.0):
    return int(lib().orc_min_hits_relaxed(s, k, identity))


def identity(shared, s, k=16):
    a, b = C.c_float(), C.c_float()
    lib().orc_identity(shared, s, k, C.byref(a), C.byref(b))
    return np.float32(a.value), np.float32(b.value)


def synth_genome(seed, genome_id, length, variant=0, cluster_size=20):
    out = np.zeros(length, dtype=np.uint8)
    lib().orc_synth_genome_c(seed, variant, cluster_size, genome_id, length, _buf(out))
    return out


class Sketch:
    """Reference sketch over genomes = lists of contigs (each contig a byte string/array)."""

    def __init__(self, genomes, k=16, w=24):
        self.k, self.w = k, w
        self.h = lib().orc_sketch_new(k, w)
        for contigs in genomes:
            for c in contigs:
                b = upper(c)
                lib().orc_sketch_add_contig(self.h, _buf(b), len(b))
            lib().orc_sketch_end_genome(self.h)
        lib().orc_sketch_finish(self.h)

    def minimizers(self):
        n = lib().orc_sketch_size(self.h)
        p = lib().orc_sketch_data(self.h)
        if n == 0:
            return np.zeros(0, dtype=MINIMIZER_DT)
        raw = C.string_at(p, n * 12)
        return np.frombuffer(raw, dtype=MINIMIZER_DT).copy()

    def unique(self):
        return int(lib().orc_sketch_unique(self.h))

    def map_genome(self, contigs, frag_len=3000, identity=80.0):
        """Returns (mappings, totalQueryFragments) for one query genome."""
        out = C.c_void_p(None)
        n, cap = C.c_size_t(0), C.c_size_t(0)
        total = 0
        for c in contigs:
            b = upper(c)
            total += lib().orc_map_contig(self.h, _buf(b), len(b), frag_len, identity, total,
                                          C.byref(out), C.byref(n), C.byref(cap))
        if n.value:
            raw = C.string_at(out, n.value * 44)
            maps = np.frombuffer(raw, dtype=MAPPING_DT).copy()
        else:
            maps = np.zeros(0, dtype=MAPPING_DT)
        if out.value:
            C.CDLL(None).free(out)
        return maps, total

    def compute_cgi(self, maps, total_fragments, query_id, frag_len=3000):
        cap = 1 << 16
        out = np.zeros(cap, dtype=CGI_DT)
        maps = np.ascontiguousarray(maps)
        m = lib().orc_compute_cgi(self.h, _buf(maps), len(maps), frag_len, total_fragments, query_id, _buf(out), cap)
        return out[:m].copy()

    def __del__(self):
        try:
            lib().orc_sketch_free(self.h)
        except Exception:
            pass


# ---------------------------------------------------------------------------------------------
# reference (oracle/_ref) helpers — only usable where oracle/_ref was built (this container, or
# shipped prebuilt to the GPU box)
# ---------------------------------------------------------------------------------------------
def have_ref():
    return os.path.exists(REF_DUMP) and os.path.exists(REF_BIN)


def write_fasta(path, contigs, names=None, width=80):
    with open(path, "wb") as f:
        for i, c in enumerate(contigs):
            b = as_bytes(c).tobytes()
            name = names[i] if names else "c%d" % i
            f.write(b">" + name.encode() + b"\n")
            for o in range(0, len(b), width):
                f.write(b[o:o + width] + b"\n")


def run_ref_dump(tmpdir, ref_genomes, qry_genomes, k=16, frag_len=3000):
    """genomes = list of contig lists. Returns dict with the reference's own intermediate results."""
    rl, ql = [], []
    for i, g in enumerate(ref_genomes):
        p = os.path.join(tmpdir, "r%d.fa" % i)
        write_fasta(p, g)
        rl.append(p)
    for i, g in enumerate(qry_genomes):
        p = os.path.join(tmpdir, "q%d.fa" % i)
        write_fasta(p, g)
        ql.append(p)
    with open(os.path.join(tmpdir, "rl.txt"), "w") as f:
        f.write("\n".join(rl) + "\n")
    with open(os.path.join(tmpdir, "ql.txt"), "w") as f:
        f.write("\n".join(ql) + "\n")
    prefix = os.path.join(tmpdir, "dump")
    subprocess.check_call([REF_DUMP, "--rl", os.path.join(tmpdir, "rl.txt"), "--ql", os.path.join(tmpdir, "ql.txt"),
                           "-k", str(k), "--fragLen", str(frag_len), "-o", prefix],
                          stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    return load_dump(prefix, len(qry_genomes))


def load_dump(prefix, nq):
    res = {}
    k, w, L, nr, nq2 = [int(x) for x in open(prefix + ".params").read().split()]
    res["k"], res["w"], res["L"] = k, w, L
    res["minimizers"] = np.fromfile(prefix + ".minimizers", dtype=MINIMIZER_DT)
    si = np.fromfile(prefix + ".seqinfo", dtype="<i4")
    nc = si[0]
    res["contigLen"] = si[1:1 + nc]
    nf = si[1 + nc]
    res["seqsByFile"] = si[2 + nc:2 + nc + nf]
    res["frags"], res["maps"] = [], []
    for q in range(nq):
        raw = np.fromfile(prefix + ".q%d.frags" % q, dtype="<i4")
        nfr = raw[0]
        fr, o = [], 1
        for _ in range(nfr):
            s = raw[o]
            fr.append(raw[o + 1:o + 1 + s].view("<u4"))
            o += 1 + s
        res["frags"].append(fr)
        res["maps"].append(np.fromfile(prefix + ".q%d.maps" % q, dtype=MAPPING_DT))
    res["cgi"] = np.fromfile(prefix + ".cgi", dtype=CGI_DT)
    return res
