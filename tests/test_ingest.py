"""The front end's multi-threaded read ingest (cf_ingest.cpp) without a GPU:
`centrifuge-class --dump-reads` prints name, bases, qualities and seed per read; they must
equal what the Python plumbing reader (validated against the reference through the golden
TSVs) produces — for one and several parser threads, and across chunk boundaries."""
import os
import subprocess
import tempfile

import numpy as np
import pytest

import common
from centrifuge_amd import capi, reads

CLI = os.path.join(common.ROOT, "centrifuge_amd", "bin", "centrifuge-class")


def dump(args, env=None):
    r = subprocess.run([CLI, "--dump-reads"] + args, capture_output=True, env=dict(os.environ, **(env or {})))
    assert r.returncode == 0, r.stderr.decode()
    return r.stdout


def expected(path, fastq, trim5=0, trim3=0, seed=0):
    L = capi.lib()
    out = []
    for i, (name, codes, qual) in enumerate((reads.read_fastq if fastq else reads.read_fasta)(path)):
        codes = codes[trim5:]
        q = qual[trim5:] if qual is not None else None
        if trim3:
            codes = codes[:max(0, len(codes) - trim3)] if len(codes) > trim3 else codes[:0]
            if q is not None:
                q = q[:len(codes)]
        codes = np.ascontiguousarray(codes)
        qs = bytes(q) if q is not None else b"I" * len(codes)
        qa = np.frombuffer(qs, dtype=np.uint8).copy() if q is not None and len(qs) else None
        sd = L.cf_gen_rand_seed(codes.ctypes.data if len(codes) else None, qa.ctypes.data if qa is not None else None,
                                len(codes), name, len(name), seed)
        out.append(name + b"\t" + bytes(np.frombuffer(b"ACGTN", dtype=np.uint8)[codes]) + b"\t" + qs + b"\t" + str(sd).encode() + b"\n")
    return b"".join(out)


@pytest.mark.parametrize("threads", [1, 4])
def test_golden_read_files(threads):
    d, _ = common.golden("synth_small")
    for f, fq in (("reads.fa", False), ("reads250.fa", False), ("reads.fq", True), ("r1.fa", False)):
        p = os.path.join(d, f)
        assert dump(["-q" if fq else "-f", "-p", str(threads), "-U", p]) == expected(p, fq), f
    p = os.path.join(d, "reads.fq")
    assert dump(["-q", "-p", str(threads), "-5", "3", "-3", "7", "--seed", "42", "-U", p]) == expected(p, True, 3, 7, 42)


def test_chunk_boundaries_and_odd_records():
    """> 16 MiB of input: several chunks per file; multi-line FASTA, lower case, IUPAC codes,
    empty sequences, CRLF line ends, a quality line starting with '@'"""
    rng = np.random.default_rng(5)
    with tempfile.TemporaryDirectory() as t:
        fa = os.path.join(t, "big.fa")
        n = 260000
        with open(fa, "wb") as f:
            f.write(b"# a comment before the first record\n\n")
            alpha = np.frombuffer(b"ACGTacgtNRYn-", dtype=np.uint8)
            for i in range(n):
                L = int(rng.integers(0, 140))
                s = bytes(alpha[rng.integers(0, len(alpha) if i % 50 == 0 else 4, size=L)])
                nl = b"\r\n" if i % 7 == 0 else b"\n"
                f.write(b">r%d some description/1" % i + nl)
                for k in range(0, L, 60):
                    f.write(s[k:k + 60] + nl)
        want = expected(fa, False)
        assert want.count(b"\n") == n
        assert dump(["-f", "-p", "1", "-U", fa]) == want
        assert dump(["-f", "-p", "6", "-U", fa]) == want
        # many small blocks, dealt out as file ranges (the parsers read them) and as a stream (the I/O thread does)
        for env in ({"CF_INGEST_BLOCK": "300000"}, {"CF_INGEST_BLOCK": "70001", "CF_INGEST_STREAM": "1"}, {"CF_INGEST_BLOCK": "4096"}):
            assert dump(["-f", "-p", "5", "-U", fa], env) == want, env
        fq = os.path.join(t, "big.fq")
        m = 120000
        with open(fq, "wb") as f:
            for i in range(m):
                L = int(rng.integers(1, 150))
                s = bytes(np.frombuffer(b"ACGTN.", dtype=np.uint8)[rng.integers(0, 6 if i % 40 == 0 else 4, size=L)])
                q = bytearray((rng.integers(33, 74, size=L)).astype(np.uint8).tobytes())
                if i % 9 == 0:
                    q[0] = ord("@")
                f.write(b"@q%d extra\n" % i + s + b"\n+\n" + bytes(q) + b"\n")
        want = expected(fq, True)
        assert dump(["-q", "-p", "1", "-U", fq]) == want
        assert dump(["-q", "-p", "6", "-U", fq]) == want
        for env in ({"CF_INGEST_BLOCK": "300000"}, {"CF_INGEST_BLOCK": "70001", "CF_INGEST_STREAM": "1"}, {"CF_INGEST_BLOCK": "4096"}):
            assert dump(["-q", "-p", "5", "-U", fq], env) == want, env


def test_wide_base_runs_stop_at_every_odd_byte():
    """the parsers take upper-case A/C/G/T 32 bytes at a time (codes and the seed's base term); anything else — lower case,
    N, IUPAC letters, '.', a gap, the line end — ends the run wherever it falls in a group, and runs start at any base number"""
    rng = np.random.default_rng(11)
    odd = b"acgtnNRY-."
    with tempfile.TemporaryDirectory() as t:
        fa, fq = os.path.join(t, "w.fa"), os.path.join(t, "w.fq")
        with open(fa, "wb") as f, open(fq, "wb") as g:
            i = 0
            for L in list(range(0, 140)) + [255, 256, 257, 1000]:
                for pos in {0, 1, 15, 16, 31, 32, 33, 63, 64, 65, L - 1, L // 2, None}:
                    b = bytearray(np.frombuffer(b"ACGT", dtype=np.uint8)[rng.integers(0, 4, size=L)].tobytes())
                    if pos is not None and 0 <= pos < L:
                        b[pos] = odd[i % len(odd)]
                    i += 1
                    width = (61, 32, 1000, 7)[i % 4]                      # FASTA: the sequence on lines of this width
                    f.write(b">w%d_%d\n" % (L, i) + b"".join(bytes(b[k:k + width]) + b"\n" for k in range(0, L, width)))
                    if L:
                        q = rng.integers(33, 74, size=L).astype(np.uint8).tobytes()
                        g.write(b"@w%d_%d\n" % (L, i) + bytes(b).replace(b"-", b"A") + b"\n+\n" + q + b"\n")
        for p_ in ("1", "3"):
            assert dump(["-f", "-p", p_, "-U", fa]) == expected(fa, False)
            assert dump(["-q", "-p", p_, "-U", fq]) == expected(fq, True)


@pytest.mark.parametrize("threads", [1, 4])
@pytest.mark.parametrize("trim", [[], ["-5", "1"]])
def test_fastq_errors_are_reported_on_every_ingest_path(threads, trim):
    """the sequential reader, the chunk parsers' direct-write path and their general (trimming) path
    stop with the reference's messages (pat.cpp:1513-1524, qual.h) instead of emitting reads"""
    cases = [(b"@a\nACGT\n+\nIII\n", b"has more read characters than quality values"),
             (b"@a\nACGT\n+\nIIIIII\n", b"has more quality values than read characters"),
             (b"@a\nACGT\n+\nII I\n", b"space in the quality string"),
             (b"@a\nACGT\n+\nII\x1fI\n", b"but expected 33-based Phred qual"),
             (b"@ok\nAC\n+\nII\nACGT\n+\nIIII\n", b"does not look like a FASTQ file")]
    with tempfile.TemporaryDirectory() as t:
        for i, (text, msg) in enumerate(cases):
            p = os.path.join(t, "e%d.fq" % i)
            open(p, "wb").write(b"@first\nGATTACA\n+\nIIIIIII\n" + text)
            r = subprocess.run([CLI, "--dump-reads", "-q", "-p", str(threads)] + trim + ["-U", p], capture_output=True)
            assert r.returncode != 0 and msg in r.stderr, (i, r.stderr)
        # one quality value too many is tolerated (pat.cpp:1075-1078): the extra one is dropped
        p = os.path.join(t, "ok.fq")
        open(p, "wb").write(b"@a\nACGT\n+\nIIIIJ\n")
        assert dump(["-q", "-p", str(threads), "-U", p]) == expected(p, True)


@pytest.mark.parametrize("threads", [1, 4])
def test_mate_files_are_interleaved_in_bulk_and_one_by_one(threads):
    """-1/-2: pairs are assembled by the bulk interleave (chunks of both files rarely end together)
    or record by record (unnamed reads, the -s/-u window); both give mate 1, mate 2, mate 1, ..."""
    rng = np.random.default_rng(9)
    n = 150000
    with tempfile.TemporaryDirectory() as t:
        f1, f2 = os.path.join(t, "m1.fq"), os.path.join(t, "m2.fq")
        with open(f1, "wb") as a, open(f2, "wb") as b:
            for i in range(n):
                l1, l2 = int(rng.integers(30, 151)), int(rng.integers(30, 251))       # different record sizes: chunk ends drift apart
                s1 = bytes(np.frombuffer(b"ACGTN", dtype=np.uint8)[rng.integers(0, 5, l1)])
                s2 = bytes(np.frombuffer(b"ACGTN", dtype=np.uint8)[rng.integers(0, 5, l2)])
                nm = b"" if i in (70000, 70001) else b"p%d" % i
                a.write(b"@" + nm + b"/1\n" + s1 + b"\n+\n" + bytes(rng.integers(33, 74, l1, dtype=np.uint8)) + b"\n")
                b.write(b"@" + nm + b"/2\n" + s2 + b"\n+\n" + bytes(rng.integers(33, 74, l2, dtype=np.uint8)) + b"\n")
        e1, e2 = expected(f1, True).splitlines(True), expected(f2, True).splitlines(True)
        want = [x for pair in zip(e1, e2) for x in pair]
        got = dump(["-q", "-p", str(threads), "-1", f1, "-2", f2]).splitlines(True)
        assert len(got) == 2 * n
        named = [i for i in range(2 * n) if i // 2 not in (70000, 70001)]
        assert [got[i] for i in named] == [want[i] for i in named]
        # reads without a name are named after their ordinal (pat.cpp:838-842); "/1" alone is a name
        assert got[2 * 70000].startswith(b"/1\t") or got[2 * 70000].split(b"\t")[0] in (b"70000", b"/1")
        w = dump(["-q", "-p", str(threads), "-s", "1000", "-u", "5000", "--batch", "777", "-1", f1, "-2", f2]).splitlines(True)
        assert w == want[2000:12000]


def test_every_input_starts_over():
    """the reference works through its inputs one at a time (centrifuge.cpp:3007-3040): -s/-u and the
    ordinals that name unnamed reads restart with every file / every -c sequence"""
    with tempfile.TemporaryDirectory() as t:
        a, b = os.path.join(t, "a.fa"), os.path.join(t, "b.fa")
        open(a, "wb").write(b">a0\nACGTA\n>a1\nCCGTA\n>\nGGGTA\n")
        open(b, "wb").write(b">b0\nTTGTA\n>\nAAGTA\n>b2\nACCTA\n>b3\nACGGA\n")
        for threads in (1, 3):
            out = dump(["-f", "-p", str(threads), "-U", a + "," + b]).splitlines()
            assert [x.split(b"\t")[0] for x in out] == [b"a0", b"a1", b"2", b"b0", b"1", b"b2", b"b3"]
            out = dump(["-f", "-p", str(threads), "-s", "1", "-u", "2", "-U", a + "," + b]).splitlines()
            assert [x.split(b"\t")[0] for x in out] == [b"a1", b"2", b"1", b"b2"]
        out = dump(["-c", "-U", "ACGTACGT,GGGTTTAA,TT"]).splitlines()
        assert [x.split(b"\t")[:2] for x in out] == [[b"0", b"ACGTACGT"], [b"0", b"GGGTTTAA"], [b"0", b"TT"]]


def _ref_names_and_lengths(args):
    """readID and queryLength columns of the compiled reference run on the CPU (test oracle)"""
    from oracle import oracle as O
    d, _ = common.golden("example")
    with tempfile.TemporaryDirectory() as t:
        r = subprocess.run([os.path.join(O.REF_DIR, "centrifuge-class"), "-x", os.path.join(d, "idx"), "--report-file", os.path.join(t, "r.tsv"),
                            "-S", os.path.join(t, "o.tsv")] + args, capture_output=True, text=True)
        assert r.returncode == 0, r.stderr
        rows = [ln.split("\t") for ln in open(os.path.join(t, "o.tsv")).read().splitlines()[1:]]
    out, i = [], 0
    while i < len(rows):                # one entry per read: a read owns numMatches consecutive rows
        out.append((rows[i][0], int(rows[i][6])))
        i += max(1, int(rows[i][7]))
    return out


def _mine_names_and_lengths(args):
    return [(f[0].decode(), len(f[1])) for f in (ln.split(b"\t") for ln in dump(args).splitlines())]


def test_raw_and_command_line_reads_like_the_reference():
    """-r (RawPatternSource, pat.h:1493-1584): first whitespace-free token of a line, every character counts toward
    -5, only letters are bases; -c (VectorPatternSource, pat.cpp:456-546): `seq[:qual]`, trimmed by characters, every
    remaining character is a base.  Names and lengths against the reference binary where it is built."""
    from oracle import oracle as O
    s = "GATCCTCCCCAGGCCCCTACACCCAATGTGGAACCGGGGTCCCGAATGAAAATGCTGCTGTTCCCTGGAGGTGTTTTCCT"
    with tempfile.TemporaryDirectory() as t:
        raw = os.path.join(t, "raw.txt")
        open(raw, "w").write("%s\n\n%s.%s trailing words\n\r\n%s\tTAB%s\nnnnn%s\n" % (s, s[:50], s[51:71], s[:30], s[:10], s[:40]))
        assert _mine_names_and_lengths(["-r", "-U", raw]) == [("0", 80), ("1", 70), ("2", 30), ("3", 44)]
        assert _mine_names_and_lengths(["-r", "-5", "3", "-U", raw]) == [("0", 77), ("1", 67), ("2", 27), ("3", 41)]
        cmd = "%s,%s:%s,%s.%s,acgtnn%s,%s:ABC" % (s, s[:40], "I" * 40, s[:30], s[30:40], s[:30], s[:25])
        assert _mine_names_and_lengths(["-c", "-U", cmd]) == [("0", 80), ("0", 40), ("0", 41), ("0", 36), ("0", 25)]
        out = dump(["-c", "-U", "ACGT:IJKL,AC.T,ACGTACGT:AB"]).splitlines()
        assert [x.split(b"\t")[1:3] for x in out] == [[b"ACGT", b"IJKL"], [b"ACAT", b"IIII"], [b"ACGTACGT", b"ABIIIIII"]]
        if O.have_ref():
            for extra in ([], ["-5", "3", "-3", "2"], ["-5", "60", "-3", "30"]):
                assert _mine_names_and_lengths(["-r"] + extra + ["-U", raw]) == _ref_names_and_lengths(["-r"] + extra + ["-U", raw])
                assert _mine_names_and_lengths(["-c"] + extra + ["-U", cmd]) == _ref_names_and_lengths(["-c"] + extra + ["-U", cmd])
        notraw = os.path.join(t, "x.fa")
        open(notraw, "w").write(">x\nACGT\n")
        r = subprocess.run([CLI, "--dump-reads", "-r", "-U", notraw], capture_output=True)
        assert r.returncode == 1 and b"does not look like a Raw file" in r.stderr and b"please use -f" in r.stderr


def test_unequal_mate_files_are_an_error_in_both_directions():
    with tempfile.TemporaryDirectory() as t:
        a, b = os.path.join(t, "a.fa"), os.path.join(t, "b.fa")
        open(a, "w").write(">r1/1\nACGTACGTAC\n>r2/1\nACGTACGTAA\n>r3/1\nACGTACGTAG\n")
        open(b, "w").write(">r1/2\nTTGTACGTAC\n>r2/2\nTTGTACGTAA\n")
        for threads in ("1", "3"):
            r = subprocess.run([CLI, "--dump-reads", "-f", "-p", threads, "-1", a, "-2", b], capture_output=True)
            assert r.returncode == 1 and b"fewer reads in file specified with -2 than in file specified with -1" in r.stderr
            r = subprocess.run([CLI, "--dump-reads", "-f", "-p", threads, "-1", b, "-2", a], capture_output=True)
            assert r.returncode == 1 and b"fewer reads in file specified with -1 than in file specified with -2" in r.stderr
            # -u below the shorter file's count never gets to see the surplus; -u equal to it does (as the reference)
            assert len(dump(["-f", "-p", threads, "-u", "1", "-1", b, "-2", a]).splitlines()) == 2
            r = subprocess.run([CLI, "--dump-reads", "-f", "-p", threads, "-u", "2", "-1", b, "-2", a], capture_output=True)
            assert r.returncode == 1 and b"fewer reads in file specified with -1" in r.stderr


def test_trailing_empty_fasta_record_and_unnamed_empty_fastq_record():
    """FastaPatternSource::read bails out when the file ends in (or right after) a name line: that record is not a read
    (pat.cpp:764-783), while an empty record in the middle is one.  A FASTQ record without a base letter leaves the reader
    before the default name is set (pat.cpp:985-993): it keeps its empty name."""
    with tempfile.TemporaryDirectory() as t:
        p = os.path.join(t, "x.fa")
        for tail in (">b\n", ">b", ">b\n\n\n", ">b\r\n"):
            open(p, "w", newline="").write(">a\nACGT\n>mid\n>c\nAC\n" + tail)
            for threads in ("1", "3"):
                assert [x.split(b"\t")[0] for x in dump(["-f", "-p", threads, "-U", p]).splitlines()] == [b"a", b"mid", b"c"]
        q = os.path.join(t, "x.fq")
        open(q, "w").write("@\nACGT\n+\nIIII\n@\n\n+\n\n@named\n\n+\n\n@\nGG\n+\nII\n")
        for threads in ("1", "3"):
            out = dump(["-q", "-p", threads, "-U", q]).split(b"\n")[:-1]
            assert [x.rsplit(b"\t", 3)[0] for x in out] == [b"0", b"", b"named", b"3"]
            out = dump(["-q", "-p", threads, "-5", "1", "-U", q]).split(b"\n")[:-1]        # same rule for an empty sequence line under -5
            assert [x.rsplit(b"\t", 3)[0] for x in out] == [b"0", b"", b"named", b"3"]


def test_compressed_and_standard_input():
    """.gz / .bz2 inputs are decompressed on the fly (the reference's wrapper does that, centrifuge:412-419); '-' is stdin"""
    import shutil
    d, _ = common.golden("synth_small")
    with tempfile.TemporaryDirectory() as t:
        src = os.path.join(t, "reads.fq")
        shutil.copy(os.path.join(d, "reads.fq"), src)
        want = dump(["-q", "-p", "3", "-U", src])
        for tool, ext in (("gzip", ".gz"), ("bzip2", ".bz2")):
            if shutil.which(tool) is None:
                continue
            subprocess.run([tool, "-kf", src], check=True)
            assert dump(["-q", "-p", "3", "-U", src + ext]) == want
        r = subprocess.run([CLI, "--dump-reads", "-q", "-p", "3", "-U", "-"], stdin=open(src, "rb"), capture_output=True)
        assert r.returncode == 0 and r.stdout == want


def _bgzf(data, block=40000):
    """`data` as a BGZF file (bgzip's container): independent deflate blocks with a 'BC' size field + the EOF block"""
    import struct
    import zlib
    out = bytearray()
    chunks = [data[i:i + block] for i in range(0, len(data), block)] + [b""]
    for c in chunks:
        z = zlib.compressobj(6, zlib.DEFLATED, -15)
        comp = z.compress(c) + z.flush()
        bsize = 12 + 6 + len(comp) + 8
        out += b"\x1f\x8b\x08\x04\x00\x00\x00\x00\x00\xff" + struct.pack("<H", 6) + b"BC" + struct.pack("<HH", 2, bsize - 1)
        out += comp + struct.pack("<II", zlib.crc32(c) & 0xffffffff, len(c))
    return bytes(out)


def test_gzip_is_inflated_in_process_and_failures_are_errors():
    """no shell sees the file name (a quote in it is just a character); a missing, truncated or corrupt .gz is
    an error with exit code 1, never a silently shorter read set; concatenated members and BGZF blocks are read through."""
    import gzip
    import shutil
    d, _ = common.golden("synth_small")
    raw = open(os.path.join(d, "reads.fq"), "rb").read()
    with tempfile.TemporaryDirectory() as t:
        plain = os.path.join(t, "reads.fq")
        open(plain, "wb").write(raw)
        want = dump(["-q", "-p", "3", "-U", plain])
        # a quote and a space in the name, with a decoy command that must never run
        evil = os.path.join(t, "it's '; touch PWNED; echo '.fq.gz")
        open(evil, "wb").write(gzip.compress(raw))
        assert dump(["-q", "-p", "3", "-U", evil]) == want
        assert not os.path.exists("PWNED") and not os.path.exists(os.path.join(t, "PWNED"))
        if shutil.which("bzip2"):
            evil2 = os.path.join(t, "b'q.fq")
            open(evil2, "wb").write(raw)
            subprocess.run(["bzip2", "-f", evil2], check=True)
            assert dump(["-q", "-p", "2", "-U", evil2 + ".bz2"]) == want
            r = subprocess.run([CLI, "--dump-reads", "-q", "-U", os.path.join(t, "missing.fq.bz2")], capture_output=True)
            assert r.returncode == 1 and b"Could not open read file" in r.stderr
            open(os.path.join(t, "junk.fq.bz2"), "wb").write(b"this is not bzip2 data")
            r = subprocess.run([CLI, "--dump-reads", "-q", "-U", os.path.join(t, "junk.fq.bz2")], capture_output=True)
            assert r.returncode == 1
        # two members back to back (cat a.gz b.gz) and BGZF
        half = raw.index(b"\n@", len(raw) // 2) + 1
        cat = os.path.join(t, "cat.fq.gz")
        open(cat, "wb").write(gzip.compress(raw[:half]) + gzip.compress(raw[half:]))
        assert dump(["-q", "-p", "3", "-U", cat]) == want
        bg = os.path.join(t, "blocks.fq.gz")
        open(bg, "wb").write(_bgzf(raw))
        for threads in ("1", "4"):
            assert dump(["-q", "-p", threads, "-U", bg]) == want
        # failures
        r = subprocess.run([CLI, "--dump-reads", "-q", "-U", os.path.join(t, "missing.fq.gz")], capture_output=True)
        assert r.returncode == 1 and b"Could not open read file" in r.stderr
        z = gzip.compress(raw)
        open(os.path.join(t, "trunc.fq.gz"), "wb").write(z[:len(z) // 2])
        r = subprocess.run([CLI, "--dump-reads", "-q", "-U", os.path.join(t, "trunc.fq.gz")], capture_output=True)
        assert r.returncode == 1 and b"truncated" in r.stderr
        bad = bytearray(z)
        bad[len(bad) // 2] ^= 0xff
        bad[len(bad) // 2 + 1] ^= 0xff
        open(os.path.join(t, "bad.fq.gz"), "wb").write(bytes(bad))
        r = subprocess.run([CLI, "--dump-reads", "-q", "-U", os.path.join(t, "bad.fq.gz")], capture_output=True)
        assert r.returncode == 1
        bb = bytearray(_bgzf(raw))
        bb[len(bb) // 2] ^= 0x55
        open(os.path.join(t, "badblocks.fq.gz"), "wb").write(bytes(bb))
        r = subprocess.run([CLI, "--dump-reads", "-q", "-p", "3", "-U", os.path.join(t, "badblocks.fq.gz")], capture_output=True)
        assert r.returncode == 1
        open(os.path.join(t, "notgz.fq.gz"), "wb").write(raw)
        r = subprocess.run([CLI, "--dump-reads", "-q", "-U", os.path.join(t, "notgz.fq.gz")], capture_output=True)
        assert r.returncode == 1
        # the sequential formats read through the same source
        rawf = os.path.join(t, "r.txt")
        open(rawf, "w").write("ACGTACGTACGTACGTAAAC\nGGGTTTACACACGGGTTTAA\n")
        open(rawf + ".gz", "wb").write(gzip.compress(open(rawf, "rb").read()))
        assert dump(["-r", "-U", rawf + ".gz"]) == dump(["-r", "-U", rawf])


def test_fastq_sequences_over_several_lines():
    """the reference's FASTQ reader takes every letter up to the '+' line, whatever the line breaks (pat.cpp:932-975); the
    qualities are one line.  A file with wrapped sequences reads exactly like its four-line form — on every ingest path, with
    the blocks cut anywhere — and the compiled reference classifies both alike."""
    from oracle import oracle as O
    rng = np.random.default_rng(21)
    d, _ = common.golden("synth_small")
    recs = reads.read_fastq(os.path.join(d, "reads.fq"))[:3000]
    with tempfile.TemporaryDirectory() as t:
        a, b = os.path.join(t, "four.fq"), os.path.join(t, "wrapped.fq")
        with open(a, "wb") as fa, open(b, "wb") as fb:
            for i, (name, codes, qual) in enumerate(recs):
                s = bytes(np.frombuffer(b"ACGTN", dtype=np.uint8)[codes])
                q = bytearray(bytes(qual))
                if i % 13 == 0 and len(q):
                    q[0] = ord("@")                                  # a quality line that starts like a name line
                if i % 17 == 0 and len(q):
                    q[0] = ord("+")                                  # ... or like the separator
                fa.write(b"@" + name + b"\n" + s + b"\n+\n" + bytes(q) + b"\n")
                w = int(rng.integers(1, 71))
                fb.write(b"@" + name + (b"\r\n" if i % 5 == 0 else b"\n") + b"".join(s[k:k + w] + b"\n" for k in range(0, len(s), w)) +
                         (b"+" + name if i % 3 == 0 else b"+") + b"\n" + bytes(q) + b"\n")
        want = dump(["-q", "-p", "1", "-U", a])
        assert want.count(b"\n") == len(recs)
        for env in ({}, {"CF_INGEST_BLOCK": "4096"}, {"CF_INGEST_BLOCK": "10007", "CF_INGEST_STREAM": "1"}):
            for p_ in ("1", "4"):
                assert dump(["-q", "-p", p_, "-U", b], env) == want, (env, p_)
            assert dump(["-q", "-p", "3", "-5", "2", "-3", "1", "-U", b], env) == dump(["-q", "-p", "1", "-5", "2", "-3", "1", "-U", a])
        if O.have_ref():
            base = os.path.join(d, "idx")
            ta = O.ref_classify(base, os.path.join(t, "a.tsv"), os.path.join(t, "a.rep"), u=a, fastq=True)
            tb = O.ref_classify(base, os.path.join(t, "b.tsv"), os.path.join(t, "b.rep"), u=b, fastq=True)
            assert ta == tb and ta.count("\n") > len(recs) // 2


def test_fastq_sequence_lines_with_other_bytes_still_cut_into_blocks():
    """the chunk cutter took a FASTQ record start only when the sequence lines held letters alone, while the parser
    skips whatever is no letter — a file whose sequence lines carry a trailing blank, a tab or digits never got a cut, the
    cutter re-read ever larger windows and the whole file went to one thread.  Such a file reads like its clean form on every
    ingest path, and small blocks stay small (the run finishes quickly: no window regrowth over the file)."""
    import time
    rng = np.random.default_rng(5)
    d, _ = common.golden("synth_small")
    recs = reads.read_fastq(os.path.join(d, "reads.fq"))
    recs = (recs * (1 + 6000 // len(recs)))[:6000]
    with tempfile.TemporaryDirectory() as t:
        a, b = os.path.join(t, "clean.fq"), os.path.join(t, "noisy.fq")
        with open(a, "wb") as fa, open(b, "wb") as fb:
            for i, (name, codes, qual) in enumerate(recs):
                s = bytes(np.frombuffer(b"ACGTN", dtype=np.uint8)[codes])
                q = bytearray(bytes(qual))
                if i % 11 == 0 and len(q):
                    q[0] = ord("@")
                nm = name + b"_%d" % i
                fa.write(b"@" + nm + b"\n" + s + b"\n+\n" + bytes(q) + b"\n")
                tail = (b" ", b"\t", b" 12", b"0", b"")[int(rng.integers(0, 5))]
                fb.write(b"@" + nm + b"\n" + s + tail + b"\n+\n" + bytes(q) + b"\n")
        want = dump(["-q", "-p", "1", "-U", a])
        assert want.count(b"\n") == len(recs)
        for env in ({"CF_INGEST_BLOCK": "4096"}, {"CF_INGEST_BLOCK": "20011"}, {"CF_INGEST_BLOCK": "10007", "CF_INGEST_STREAM": "1"}):
            t0 = time.time()
            assert dump(["-q", "-p", "4", "-U", b], env) == want, env
            assert time.time() - t0 < 20


def test_fastq_dot_no_calls_and_mismatched_qualities_still_cut_into_blocks():
    """the cut rule counted a sequence line's letters while the parser first turns '.' into the N it keeps, so a file
    whose reads carry '.' no-calls never passed the strict check (quality length == sequence length) and found its cut points
    only through the lenient fallback, four blocks late, window after window; and the range path applied that fallback to FASTA
    as well.  Reads with '.' cut into small blocks like their 'N' form; a file whose quality strings are one too long record after
    record is still cut (the lenient rule), on the range and on the stream path, and reads like the one-thread parse."""
    import time
    rng = np.random.default_rng(15)
    d, _ = common.golden("synth_small")
    recs = reads.read_fastq(os.path.join(d, "reads.fq"))
    recs = (recs * (1 + 6000 // len(recs)))[:6000]
    with tempfile.TemporaryDirectory() as t:
        a, b, c = os.path.join(t, "n.fq"), os.path.join(t, "dot.fq"), os.path.join(t, "short.fq")
        with open(a, "wb") as fa, open(b, "wb") as fb, open(c, "wb") as fc:
            for i, (name, codes, qual) in enumerate(recs):
                codes = codes.copy()
                if len(codes) > 8:
                    codes[rng.integers(0, len(codes), size=2)] = 4              # every read holds no-calls
                s = bytes(np.frombuffer(b"ACGTN", dtype=np.uint8)[codes])
                q = bytes(qual)
                nm = name + b"_%d" % i
                fa.write(b"@" + nm + b"\n" + s + b"\n+\n" + q + b"\n")
                fb.write(b"@" + nm + b"\n" + s.replace(b"N", b".") + b"\n+\n" + q + b"\n")
                fc.write(b"@" + nm + b"\n" + s + b"\n+\n" + q + b"I\n")     # one quality value too many: the parser (like the reference's) lets it pass, the strict cut rule never matches
        want = dump(["-q", "-p", "1", "-U", a])
        assert want.count(b"\n") == len(recs)
        for env in ({"CF_INGEST_BLOCK": "4096"}, {"CF_INGEST_BLOCK": "10007", "CF_INGEST_STREAM": "1"}):
            t0 = time.time()
            assert dump(["-q", "-p", "4", "-U", b], env) == want, env
            assert time.time() - t0 < 20
        want_short = dump(["-q", "-p", "1", "-U", c])
        assert want_short.count(b"\n") == len(recs)
        for env in ({"CF_INGEST_BLOCK": "4096"}, {"CF_INGEST_BLOCK": "4096", "CF_INGEST_STREAM": "1"}):
            assert dump(["-q", "-p", "4", "-U", c], env) == want_short, env


def test_packed_form_made_by_the_parser_threads():
    """single-end chunks reach the GPU thread in the packed form of cf_packed_reads (2-bit words, sparse N mask, lengths, seeds),
    made by the parser thread that parsed the chunk (ReadSoA::pack, AVX2 groups of 32 bases + a scalar tail).  --dump-reads with
    CF_DUMP_FROM_PACKED=1 prints the bases back out of that form: it must read like the byte form, N runs and ragged lengths
    included, however the file is cut into blocks"""
    rng = np.random.default_rng(8)
    d, _ = common.golden("synth_small")
    for f, fq in (("reads.fa", False), ("reads250.fa", False), ("reads.fq", True)):
        p = os.path.join(d, f)
        want = expected(p, fq)
        for env in ({}, {"CF_INGEST_BLOCK": "4096"}, {"CF_INGEST_BLOCK": "50021", "CF_INGEST_STREAM": "1"}):
            got = dump(["-q" if fq else "-f", "-p", "4", "-U", p], dict(env, CF_DUMP_FROM_PACKED="1"))
            assert got == want, (f, env)
    # lengths around the 32-base groups, N at every place of a word, all-N and empty reads
    with tempfile.TemporaryDirectory() as t:
        p = os.path.join(t, "odd.fa")
        with open(p, "w") as fh:
            for i, L in enumerate([0, 1, 2, 31, 32, 33, 63, 64, 65, 95, 96, 97, 100, 128, 129, 250, 257, 300] * 6):
                s = rng.integers(0, 4, L)
                letters = np.frombuffer(b"ACGT", dtype=np.uint8)[s].copy()
                if L and i % 3 == 0:
                    letters[rng.integers(0, L, size=max(1, L // 9))] = ord("N")
                if i % 17 == 5:
                    letters[:] = ord("N")
                fh.write(">r%d\n%s\n" % (i, letters.tobytes().decode()))
        want = expected(p, False)
        for env in ({}, {"CF_INGEST_BLOCK": "4096"}):
            assert dump(["-f", "-p", "3", "-U", p], dict(env, CF_DUMP_FROM_PACKED="1")) == want


def test_packed_form_of_mates_is_interleaved_with_their_bytes():
    """mate files: the batch takes whole pairs from the two streams' chunks, their packed words along with their bytes (every
    read starts on a word: a read is a run of whole words; the sparse N words move with it).  The dump out of the packed form
    equals the dump out of the bytes for every way of cutting the two files into blocks, N-rich mates of odd lengths included"""
    rng = np.random.default_rng(12)
    d, _ = common.golden("synth_small")
    a, b = os.path.join(d, "r1.fa"), os.path.join(d, "r2.fa")
    want = dump(["-f", "-p", "2", "-1", a, "-2", b])
    assert want.count(b"\n") > 100
    for env in ({}, {"CF_INGEST_BLOCK": "4096"}, {"CF_INGEST_BLOCK": "30011", "CF_INGEST_STREAM": "1"}):
        assert dump(["-f", "-p", "3", "-1", a, "-2", b], dict(env, CF_DUMP_FROM_PACKED="1")) == want, env
    r = subprocess.run([CLI, "--dump-reads", "-f", "-p", "2", "-1", a, "-2", b], capture_output=True, env=dict(os.environ, CF_DUMP_FROM_PACKED="1"))
    n_packed, n_all = [int(x) for x in r.stderr.decode().split("packed form:")[1].split("batches")[0].replace("of", " ").split()]
    assert n_packed >= 1 and n_packed >= n_all - 1, r.stderr        # (a batch that ends on a lone pair takes it record by record: bytes)
    with tempfile.TemporaryDirectory() as t:
        p1, p2 = os.path.join(t, "a.fa"), os.path.join(t, "b.fa")
        with open(p1, "w") as f1, open(p2, "w") as f2:
            for i in range(700):
                for fh, L in ((f1, int(rng.integers(1, 200))), (f2, int(rng.integers(1, 300)))):
                    letters = np.frombuffer(b"ACGT", dtype=np.uint8)[rng.integers(0, 4, L)].copy()
                    if i % 4 == 0:
                        letters[rng.integers(0, L, size=max(1, L // 7))] = ord("N")
                    fh.write(">p%d\n%s\n" % (i, letters.tobytes().decode()))
        want = dump(["-f", "-p", "1", "-1", p1, "-2", p2])
        for env in ({}, {"CF_INGEST_BLOCK": "4096"}):
            assert dump(["-f", "-p", "4", "-1", p1, "-2", p2], dict(env, CF_DUMP_FROM_PACKED="1")) == want, env
