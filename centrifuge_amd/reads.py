"""Read ingest for the Python plumbing layer (tests, bench): FASTA/FASTQ to the
base codes, names and per-read seeds the C ABI takes.  Behaviour follows the
reference's parsers (pat.cpp:725-850 FASTA, :852+ FASTQ; alphabet.cpp:36-58,
298-319): ACGTN plus IUPAC codes and '-' are kept, everything kept but ACGTN
becomes A; FASTA qualities are 'I'.  The C++ front end has its own ingest
(csrc/cf_reads.cpp); this one exists so tests can drive the C ABI directly.
"""
import numpy as np

from . import capi

_KEEP = np.zeros(256, dtype=bool)
_CODE = np.zeros(256, dtype=np.uint8)
for _ch in "ABCDGHKMNRSTVWXY":
    _KEEP[ord(_ch)] = _KEEP[ord(_ch.lower())] = True
_KEEP[ord("-")] = True
for _ch, _v in zip("ACGTN", range(5)):
    _CODE[ord(_ch)] = _CODE[ord(_ch.lower())] = _v


_ALPHA = np.zeros(256, dtype=bool)
for _c in range(256):
    _ALPHA[_c] = chr(_c).isalpha() and _c < 128


def encode(raw: bytes) -> np.ndarray:
    a = np.frombuffer(raw, dtype=np.uint8)
    return _CODE[a[_KEEP[a]]]


def read_fasta(path):
    recs, name, chunks = [], None, []
    with open(path, "rb") as f:
        for ln in f:
            ln = ln.rstrip(b"\r\n")
            if ln.startswith(b">"):
                if name is not None:
                    recs.append((name, b"".join(chunks)))
                name, chunks = ln[1:], []
            elif name is not None and not (ln.startswith(b"#") or ln.startswith(b";")):
                chunks.append(ln)
    if name is not None and any(chunks):       # a record cut off by the end of the file is not a read (pat.cpp:764-783)
        recs.append((name, b"".join(chunks)))
    return [(n if n else str(i).encode(), encode(s), None) for i, (n, s) in enumerate(recs)]


def read_fastq(path):
    out = []
    with open(path, "rb") as f:
        while True:
            h = f.readline()
            if not h:
                break
            s = f.readline().rstrip(b"\r\n")
            f.readline()
            q = f.readline().rstrip(b"\r\n")
            # FASTQ keeps every letter ('.' counts as N; letters other than ACGTN become A), pat.cpp:905-917
            a = np.frombuffer(s.replace(b".", b"N"), dtype=np.uint8)
            c = _CODE[a[_ALPHA[a]]]
            out.append((h.rstrip(b"\r\n")[1:], c, np.frombuffer(q, dtype=np.uint8)[:len(c)].copy()))
    return out


def read_id(name: bytes) -> bytes:
    """readID column (aln_sink.h:2203-2217): drop /1 /2 /3, cut at whitespace."""
    if len(name) >= 2 and name[-2:-1] == b"/" and name[-1:] in (b"1", b"2", b"3"):
        name = name[:-2]
    for i, ch in enumerate(name):
        if chr(ch).isspace():
            return name[:i]
    return name


def load(files, fastq=False, global_seed=0):
    """-> names, qlens, seq(u8), off(u64), seeds(u32), paired"""
    L = capi.lib()
    rd = read_fastq if fastq else read_fasta
    mates = [rd(f) for f in files]
    paired = len(mates) == 2
    names, qlens, reads, seeds = [], [], [], []
    for i in range(len(mates[0])):
        ql = 0
        for m in mates:
            n, c, q = m[i]
            reads.append(c)
            ql += len(c)
            seeds.append(L.cf_gen_rand_seed(c.ctypes.data if len(c) else None,
                                            q.ctypes.data if q is not None and len(q) else None,
                                            len(c), n, len(n), global_seed))
        names.append(mates[0][i][0])
        qlens.append(ql)
    off = np.zeros(len(reads) + 1, dtype=np.uint64)
    if reads:
        off[1:] = np.cumsum([len(r) for r in reads])
    seq = np.concatenate(reads).astype(np.uint8) if reads and off[-1] else np.zeros(1, dtype=np.uint8)
    return names, qlens, np.ascontiguousarray(seq), off, np.array(seeds, dtype=np.uint32), paired


def format_taxid(t: int) -> str:
    lo, hi = t & 0xffffffff, t >> 32
    return str(lo) if hi == 0 else "%d.%d" % (lo, hi)


HEADER = "readID\tseqID\ttaxID\tscore\t2ndBestScore\thitLength\tqueryLength\tnumMatches\n"


def format_tsv(seqid, names, qlens, rows, n_rows, score2):
    """Default 8-column TSV (centrifuge.cpp:520, aln_sink.h:2279-2337)."""
    out = [HEADER]
    for q in range(len(names)):
        rid = read_id(names[q]).decode("latin1")
        if n_rows[q] == 0:
            out.append("%s\tunclassified\t0\t0\t0\t0\t%d\t1\n" % (rid, qlens[q]))
            continue
        for r in range(int(n_rows[q])):
            row = rows[q, r]
            out.append("%s\t%s\t%s\t%d\t%d\t%d\t%d\t%d\n" % (
                rid, seqid(row["unique_id"], row["tax_id"]), format_taxid(int(row["tax_id"])), row["score"],
                score2[q], row["hit_len"], qlens[q], n_rows[q]))
    return "".join(out)
