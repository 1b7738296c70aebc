"""ctypes wrapper around oracle/liboracle.so (the CPU restatement) plus the
reference-compatible FASTA/FASTQ reading and TSV formatting the tests need.

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and
bench.py's cpu_baseline leg.  The product (centrifuge_amd/) never imports it.
"""
import ctypes as C
import os
import subprocess
import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
LIB = os.path.join(HERE, "liboracle.so")
REF_DIR = os.path.join(HERE, "_ref")


def build(ref=True):
    """Compile the C restatement and (if /root/reference exists) oracle/_ref."""
    subprocess.check_call(["make", "-s", "-C", HERE, "port"])
    if ref and os.path.isdir("/root/reference") and not have_ref():
        subprocess.check_call(["make", "-s", "-j8", "-C", HERE, "ref"])


def have_ref():
    return all(os.access(os.path.join(REF_DIR, b), os.X_OK)
               for b in ("centrifuge-class", "centrifuge-build-bin"))


class Row(C.Structure):
    _fields_ = [("tax_id", C.c_uint64), ("unique_id", C.c_uint32), ("score", C.c_uint32),
                ("hit_len", C.c_uint32), ("pad", C.c_uint32)]


class Hit(C.Structure):
    _fields_ = [("top", C.c_uint64), ("bot", C.c_uint64), ("bwoff", C.c_uint32), ("len", C.c_uint32)]


class Params(C.Structure):
    _fields_ = [("khits", C.c_int32), ("min_hitlen", C.c_int32), ("rank_slot", C.c_int32),
                ("tree_traverse", C.c_int32),
                ("host_taxids", C.POINTER(C.c_uint64)), ("n_host", C.c_int32),
                ("exclude_taxids", C.POINTER(C.c_uint64)), ("n_exclude", C.c_int32)]


class OpCounts(C.Structure):
    _fields_ = [(n, C.c_uint64) for n in
                ("n_ftab", "n_pair", "n_pair2", "n_single", "n_walk", "n_rows", "n_ranges", "n_reads")]

    def bytes_per_read(self, sa_bytes, read_len):
        """algorithmic bytes per read."""
        n = max(1, self.n_reads)
        sides = (self.n_pair + self.n_pair2 + self.n_single + self.n_walk) / n
        return (128.0 * sides + 16.0 * self.n_ftab / n + sa_bytes * self.n_rows / n
                + (read_len + 3) // 4 + (read_len + 7) // 8 + 32)


RANK_SLOTS = {"strain": 0, "species": 1, "genus": 2, "family": 3, "order": 4, "class": 5, "phylum": 6}

_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB):
            build(ref=False)
        L = C.CDLL(LIB)
        L.cfo_index_open.restype = C.c_void_p
        L.cfo_index_open.argtypes = [C.c_char_p]
        L.cfo_index_close.argtypes = [C.c_void_p]
        L.cfo_last_error.restype = C.c_char_p
        for f in ("cfo_index_len", "cfo_index_nref"):
            getattr(L, f).restype = C.c_uint64
            getattr(L, f).argtypes = [C.c_void_p]
        for f in ("cfo_index_compressed", "cfo_index_offw"):
            getattr(L, f).restype = C.c_int
            getattr(L, f).argtypes = [C.c_void_p]
        L.cfo_index_uid.restype = C.c_char_p
        L.cfo_index_uid.argtypes = [C.c_void_p, C.c_uint64]
        L.cfo_index_ref_taxid.restype = C.c_uint64
        L.cfo_index_ref_taxid.argtypes = [C.c_void_p, C.c_uint64]
        L.cfo_format_seqid.restype = C.c_char_p
        L.cfo_format_seqid.argtypes = [C.c_void_p, C.c_uint32, C.c_uint64]
        L.cfo_tax_rank.restype = C.c_int
        L.cfo_tax_rank.argtypes = [C.c_void_p, C.c_uint64]
        L.cfo_tax_rank_string.restype = C.c_char_p
        L.cfo_tax_rank_string.argtypes = [C.c_int]
        L.cfo_tax_name.restype = C.c_char_p
        L.cfo_tax_name.argtypes = [C.c_void_p, C.c_uint64]
        L.cfo_tax_size.restype = C.c_uint64
        L.cfo_tax_size.argtypes = [C.c_void_p, C.c_uint64]
        L.cfo_gen_rand_seed.restype = C.c_uint32
        L.cfo_gen_rand_seed.argtypes = [C.c_void_p, C.c_void_p, C.c_uint64, C.c_char_p, C.c_uint64, C.c_uint32]
        L.cfo_mate_passes.restype = C.c_int
        L.cfo_mate_passes.argtypes = [C.c_void_p, C.c_uint64]
        L.cfo_classify.restype = C.c_int
        L.cfo_classify.argtypes = [C.c_void_p, C.POINTER(Params), C.c_void_p, C.c_void_p, C.c_void_p,
                                   C.c_uint64, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
        L.cfo_search.restype = C.c_int
        L.cfo_search.argtypes = [C.c_void_p, C.POINTER(Params), C.c_void_p, C.c_uint64,
                                 C.c_void_p, C.c_void_p, C.c_void_p]
        L.cfo_resolve_row.restype = C.c_uint64
        L.cfo_resolve_row.argtypes = [C.c_void_p, C.c_uint64]
        L.cfo_rank.restype = C.c_uint64
        L.cfo_rank.argtypes = [C.c_void_p, C.c_int, C.c_uint64]
        L.cfo_sort_hits.argtypes = [C.c_void_p, C.c_uint32]
        _lib = L
    return _lib


# ------------------------------------------------------------------ read input
# asc2dnacat > 0 (alphabet.cpp:36-58): ACGTN + IUPAC + '-' are kept; asc2dna
# (alphabet.cpp:298-319) maps ACGT -> 0..3, N -> 4 and everything else kept -> 0.
_KEEP = np.zeros(256, dtype=bool)
_CODE = np.zeros(256, dtype=np.uint8)
for ch in "ABCDGHKMNRSTVWXY":
    _KEEP[ord(ch)] = _KEEP[ord(ch.lower())] = True
_KEEP[ord("-")] = True
for ch, v in zip("ACGTN", range(5)):
    _CODE[ord(ch)] = _CODE[ord(ch.lower())] = v


def encode_seq(raw: bytes) -> np.ndarray:
    a = np.frombuffer(raw, dtype=np.uint8)
    return _CODE[a[_KEEP[a]]]


def read_fasta(path):
    """[(name, codes, qual)] the way FastaPatternSource::read does (pat.cpp:725-850)."""
    out = []
    name, chunks = None, []
    with open(path, "rb") as f:
        for ln in f:
            ln = ln.rstrip(b"\r\n")
            if ln.startswith(b">"):
                if name is not None:
                    out.append((name, b"".join(chunks)))
                name, chunks = ln[1:], []
            elif name is not None and not (ln.startswith(b"#") or ln.startswith(b";")):
                chunks.append(ln)
        if name is not None:
            out.append((name, b"".join(chunks)))
    res = []
    for i, (nm, raw) in enumerate(out):
        codes = encode_seq(raw)
        res.append((nm if nm else str(i).encode(), codes, np.full(len(codes), ord("I"), dtype=np.uint8)))
    return res


def read_fastq(path):
    res = []
    with open(path, "rb") as f:
        while True:
            h = f.readline()
            if not h:
                break
            s = f.readline().rstrip(b"\r\n")
            f.readline()
            q = f.readline().rstrip(b"\r\n")
            codes = encode_seq(s)
            res.append((h.rstrip(b"\r\n")[1:], codes, np.frombuffer(q, dtype=np.uint8)[:len(codes)].copy()))
    return res


def read_id(name: bytes) -> bytes:
    """appendReadID aln_sink.h:2203-2217"""
    if len(name) >= 2 and name[-2:-1] == b"/" and name[-1:] in (b"1", b"2", b"3"):
        name = name[:-2]
    for i, ch in enumerate(name):
        if chr(ch).isspace():
            return name[:i]
    return name


def format_taxid(t: int) -> str:
    """appendTaxID aln_sink.h:2236-2250"""
    lo, hi = t & 0xffffffff, t >> 32
    return str(lo) if hi == 0 else "%d.%d" % (lo, hi)


HEADER = "readID\tseqID\ttaxID\tscore\t2ndBestScore\thitLength\tqueryLength\tnumMatches\n"


class Oracle:
    def __init__(self, basename):
        self.L = lib()
        self.h = self.L.cfo_index_open(basename.encode())
        if not self.h:
            raise RuntimeError("oracle: " + self.L.cfo_last_error().decode())

    def close(self):
        if self.h:
            self.L.cfo_index_close(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def params(self, k=5, min_hitlen=22, rank="strain", traverse=True, host=(), exclude=()):
        p = Params()
        p.khits, p.min_hitlen, p.rank_slot, p.tree_traverse = k, min_hitlen, RANK_SLOTS[rank], int(traverse)
        self._host = (C.c_uint64 * max(1, len(host)))(*host)
        self._excl = (C.c_uint64 * max(1, len(exclude)))(*exclude)
        p.host_taxids, p.n_host = self._host, len(host)
        p.exclude_taxids, p.n_exclude = self._excl, len(exclude)
        return p

    def seed(self, codes, qual, name, seed=0):
        return self.L.cfo_gen_rand_seed(codes.ctypes.data, qual.ctypes.data, len(codes), name, len(name), seed)

    @staticmethod
    def pack(reads):
        """reads: list of code arrays -> (seq u8, off u64)."""
        off = np.zeros(len(reads) + 1, dtype=np.uint64)
        if reads:
            off[1:] = np.cumsum([len(r) for r in reads])
        seq = np.concatenate(reads).astype(np.uint8) if reads and off[-1] else np.zeros(1, dtype=np.uint8)
        return np.ascontiguousarray(seq), off

    def classify(self, seq, off, seeds, n_queries, paired, p, ops=None):
        k = p.khits
        rows = (Row * (n_queries * k))()
        n_rows = np.zeros(n_queries, dtype=np.uint32)
        score2 = np.zeros(n_queries, dtype=np.uint32)
        seeds = np.ascontiguousarray(seeds, dtype=np.uint32)
        rc = self.L.cfo_classify(self.h, C.byref(p), seq.ctypes.data, off.ctypes.data, seeds.ctypes.data,
                                 n_queries, int(paired), C.addressof(rows), n_rows.ctypes.data,
                                 score2.ctypes.data, C.addressof(ops) if ops is not None else None)
        if rc:
            raise RuntimeError("cfo_classify failed")
        ra = np.frombuffer(rows, dtype=np.dtype([("tax_id", "<u8"), ("unique_id", "<u4"), ("score", "<u4"),
                                                 ("hit_len", "<u4"), ("pad", "<u4")])).reshape(n_queries, k)
        return ra, n_rows, score2

    def tsv(self, names, qlens, rows, n_rows, score2):
        """TSV body exactly as AlnSinkSam::appendMate prints it (aln_sink.h:2279-2337)."""
        out = []
        for q in range(len(names)):
            rid = read_id(names[q]).decode("latin1")
            if n_rows[q] == 0:
                out.append("%s\tunclassified\t0\t0\t0\t0\t%d\t1\n" % (rid, qlens[q]))
                continue
            for r in range(int(n_rows[q])):
                row = rows[q, r]
                sid = self.L.cfo_format_seqid(self.h, int(row["unique_id"]), int(row["tax_id"])).decode("latin1")
                out.append("%s\t%s\t%s\t%d\t%d\t%d\t%d\t%d\n" % (
                    rid, sid, format_taxid(int(row["tax_id"])), row["score"], score2[q], row["hit_len"],
                    qlens[q], n_rows[q]))
        return "".join(out)

    def classify_files(self, path1, path2=None, fastq=False, global_seed=0, ops=None, **kw):
        rd = read_fastq if fastq else read_fasta
        m1 = rd(path1)
        m2 = rd(path2) if path2 else None
        p = self.params(**kw)
        reads, seeds, names, qlens = [], [], [], []
        for i in range(len(m1)):
            n, c, q = m1[i]
            reads.append(c)
            seeds.append(self.seed(c, q, n, global_seed))
            names.append(n)
            ql = len(c)
            if m2:
                n2, c2, q2 = m2[i]
                reads.append(c2)
                seeds.append(self.seed(c2, q2, n2, global_seed))
                ql += len(c2)
            qlens.append(ql)
        seq, off = self.pack(reads)
        rows, n_rows, score2 = self.classify(seq, off, np.array(seeds, dtype=np.uint32), len(m1), bool(m2), p, ops)
        return HEADER + self.tsv(names, qlens, rows, n_rows, score2)


# ------------------------------------------------------------- reference runs
def ref_build(outdir, base="idx", threads=8, fa="genomes.fa", conv="conv.tsv", nodes="nodes.dmp",
              names="names.dmp", extra=()):
    cmd = [os.path.join(REF_DIR, "centrifuge-build-bin"), "-p", str(threads), "--conversion-table",
           os.path.join(outdir, conv), "--taxonomy-tree", os.path.join(outdir, nodes), "--name-table",
           os.path.join(outdir, names), *extra, os.path.join(outdir, fa), os.path.join(outdir, base)]
    subprocess.run(cmd, check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    return os.path.join(outdir, base)


def ref_classify(index, out_tsv, report, u=None, m1=None, m2=None, fastq=False, threads=1, extra=()):
    cmd = [os.path.join(REF_DIR, "centrifuge-class"), "-q" if fastq else "-f", "-p", str(threads), "--reorder",
           "-x", index, "-S", out_tsv, "--report-file", report, *extra]
    cmd += ["-U", u] if u else ["-1", m1, "-2", m2]
    subprocess.run(cmd, check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    with open(out_tsv) as f:
        return f.read()
