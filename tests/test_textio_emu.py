"""The front end's ingest and egress on the device (centrifuge_amd/csrc/cf_textio.hpp) in the CPU harness: the bodies of
cf_batch_upload_text's kernels, of the plan stage's k_text_pack and of cf_batch_wait_text's formatter, one call per thread — in
the one-lane build and as wavefronts of 64 lanes (their cross-lane sums) — against

* the HOST parser (`centrifuge-class --dump-reads`, itself checked against the reference's parsers: tests/test_ingest.py,
  tests/fuzz/fuzz_ingest.py): a block the device takes for plain must give the very reads, names and seeds the host parser
  gives, and everything the host parser treats specially must be flagged as not plain;
* a plain Python statement of the default columns (aln_sink.h:2279-2337, readID aln_sink.h:2203-2217) and of the tally
  (aln_sink.h:142-172) for the formatter."""
import ctypes as C
import os
import subprocess
import tempfile

import numpy as np
import pytest

import common
from emu import emu

CLI = os.path.join(common.ROOT, "centrifuge_amd", "bin", "centrifuge-class")
PAD = 128
FASTA, FASTQ = 0, 1


@pytest.fixture(params=[False, True], ids=["lane1", "wave64"])
def L(request):
    was = emu.use_wave64(request.param)
    lib = emu.lib()
    lib.emu_text_parse.restype = C.c_uint32
    lib.emu_text_parse.argtypes = [C.c_void_p, C.c_uint64, C.c_int, C.c_uint32, C.c_uint64, C.c_uint32] + [C.c_void_p] * 6
    lib.emu_text_pack.argtypes = [C.c_void_p, C.c_uint32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
    lib.emu_text_format.restype = C.c_uint64
    yield lib
    emu.use_wave64(was)


def padded(text):
    a = np.zeros(len(text) + PAD + 64, dtype=np.uint8)
    a[:len(text)] = np.frombuffer(text, dtype=np.uint8)
    return a


def vp(a):
    return a.ctypes.data_as(C.c_void_p)


def parse(L, text, fmt, seed=0, rec_cap=None, pos_cap=None):
    """-> (flags, reads) with reads = list of (readID bytes, bases str, seed) made from the bodies' outputs"""
    buf = padded(text)
    rec_cap = rec_cap if rec_cap is not None else (text.count(b">") if fmt == FASTA else text.count(b"\n") // 4) + 16
    pos_cap = pos_cap if pos_cap is not None else (rec_cap if fmt == FASTA else 4 * rec_cap)
    arr = [np.zeros(rec_cap + 80, dtype=np.uint32) for _ in range(5)]
    status = np.zeros(4, dtype=np.uint64)
    n = L.emu_text_parse(vp(buf), len(text), fmt, seed, pos_cap, rec_cap, *[vp(a) for a in arr], vp(status))
    flags = int(status[3])
    if flags:
        return flags, None
    rlen, seeds, seq_off, id_off, id_len = [a[:n] for a in arr]
    assert int(status[0]) == int(((rlen.astype(np.uint64) + 31) // 32).sum()) and int(status[1]) == int(rlen.sum())
    assert int(status[2]) == (int(rlen.max()) if n else 0)
    n_words = int(status[0])
    bases, nmask = np.zeros(n_words + 1, dtype=np.uint64), np.zeros(n_words + 1, dtype=np.uint32)
    L.emu_text_pack(vp(buf), n, vp(seq_off), vp(rlen), vp(bases), vp(nmask))
    out, w = [], 0
    for r in range(n):
        s = []
        for i in range(int(rlen[r])):
            word, j = w + i // 32, i % 32
            s.append("N" if (int(nmask[word]) >> j) & 1 else "ACGT"[(int(bases[word]) >> (2 * j)) & 3])
        w += (int(rlen[r]) + 31) // 32
        out.append((bytes(text[int(id_off[r]):int(id_off[r]) + int(id_len[r])]), "".join(s), int(seeds[r])))
    return 0, out


def read_id(name):
    """aln_sink.h:2203-2217"""
    if len(name) >= 2 and name[-2:] in (b"/1", b"/2", b"/3"):
        name = name[:-2]
    for i, c in enumerate(name):
        if c in b" \t\n\v\f\r":
            return name[:i]
    return name


def host_parse(text, fmt, seed=0):
    """the host parser through --dump-reads: None when it refuses the input, else (readID, bases, seed) per read"""
    with tempfile.NamedTemporaryFile(suffix=".fa" if fmt == FASTA else ".fq") as f:
        f.write(text)
        f.flush()
        r = subprocess.run([CLI, "--dump-reads", "-f" if fmt == FASTA else "-q", "--seed", str(seed), "-U", f.name], capture_output=True)
    if r.returncode != 0:
        return None
    out = []
    for ln in r.stdout.split(b"\n")[:-1]:
        name, seq, _qual, sd = ln.rsplit(b"\t", 3)
        out.append((read_id(name), seq.decode(), int(sd)))
    return out


def make_records(rng, n, fmt, wrap=0, lower=False, ns=True):
    names, recs = [], []
    for i in range(n):
        ln = int(rng.integers(1, 300))
        alphabet = "ACGTN" if ns and rng.random() < 0.3 else "ACGT"
        seq = "".join(rng.choice(list(alphabet), ln))
        if lower and rng.random() < 0.5:
            seq = seq.lower()
        name = ("r%d" % i).encode()
        style = int(rng.integers(0, 7))
        if style == 1:
            name += b" some comment/here"
        elif style == 2:
            name += b"/1"
        elif style == 3:
            name += b"\tx/2"
        elif style == 4:
            name = b"sample/" + name + b"/3"
        elif style == 5:
            name += bytes([200, 255]) + b"x"             # bytes >= 128: the seed takes them sign-extended
        names.append(name)
        if fmt == FASTA:
            body = seq if not wrap else "\n".join(seq[k:k + wrap] for k in range(0, ln, wrap))
            recs.append(b">" + name + b"\n" + body.encode() + b"\n")
        else:
            qual = bytes(int(q) for q in rng.integers(33, 127, ln))
            recs.append(b"@" + name + b"\n" + seq.encode() + b"\n+" + (name if rng.random() < 0.2 else b"") + b"\n" + qual + b"\n")
    return b"".join(recs)


@pytest.mark.parametrize("fmt", [FASTA, FASTQ], ids=["fasta", "fastq"])
def test_plain_blocks_parse_as_the_host_parser_does(L, fmt):
    rng = np.random.default_rng(11 + fmt)
    for trial, (n, wrap, lower) in enumerate([(1, 0, False), (3, 0, False), (70, 0, True), (200, 60, True), (130, 7, False), (64, 0, False), (65, 1, True)]):
        text = make_records(rng, n, fmt, wrap=wrap if fmt == FASTA else 0, lower=lower)
        for seed in (0, 12345):
            flags, got = parse(L, text, fmt, seed)
            assert flags == 0, (trial, flags)
            want = host_parse(text, fmt, seed)
            assert want is not None and got == want, trial
    # a FASTA block whose last line has no '\n' is still plain (the sequence runs to the end of the block)
    if fmt == FASTA:
        text = b">a\nACGT\n>b x\nGGN"
        assert parse(L, text, fmt) == (0, host_parse(text, fmt))
    # an empty block: no reads, nothing flagged
    assert parse(L, b"", fmt) == (0, [])


@pytest.mark.parametrize("fmt", [FASTA, FASTQ], ids=["fasta", "fastq"])
def test_two_blocks_of_mates_in_one_buffer(L, fmt):
    """cf_batch_upload_text with text2: the blocks lie one behind the other, record r of block m is read 2 r + m of the batch, the
    places the later passes use count from the buffer's start"""
    rng = np.random.default_rng(31 + fmt)
    n = 97
    t1, t2 = make_records(rng, n, fmt, lower=True), make_records(rng, n, fmt, wrap=9 if fmt == FASTA else 0)
    at2 = (len(t1) + 63) // 64 * 64 + PAD
    buf = np.zeros(at2 + len(t2) + PAD + 64, dtype=np.uint8)
    buf[:len(t1)] = np.frombuffer(t1, dtype=np.uint8)
    buf[at2:at2 + len(t2)] = np.frombuffer(t2, dtype=np.uint8)
    arr = [np.zeros(2 * n + 200, dtype=np.uint32) for _ in range(5)]
    status = np.zeros(4, dtype=np.uint64)
    L.emu_text_parse2.restype = C.c_uint32
    L.emu_text_parse2.argtypes = [C.c_void_p, C.c_uint64, C.c_int, C.c_uint32, C.c_uint64, C.c_uint32] + [C.c_void_p] * 6 + [C.c_uint32] * 3
    for m, (t, at) in enumerate(((t1, 0), (t2, at2))):
        got = L.emu_text_parse2(buf.ctypes.data + at, len(t), fmt, 7, 4 * (n + 16), n + 16, *[vp(a) for a in arr], vp(status), at, 2, m)
        assert got == n and not status[3]
    rlen, seeds, seq_off, id_off, id_len = [a[:2 * n] for a in arr]
    n_words = int(((rlen.astype(np.uint64) + 31) // 32).sum())
    assert int(status[0]) == n_words and int(status[1]) == int(rlen.sum())
    bases, nmask = np.zeros(n_words + 1, dtype=np.uint64), np.zeros(n_words + 1, dtype=np.uint32)
    L.emu_text_pack(vp(buf), 2 * n, vp(seq_off), vp(rlen), vp(bases), vp(nmask))
    want = [host_parse(t1, fmt, 7), host_parse(t2, fmt, 7)]
    w = 0
    whole = bytes(buf)
    for r in range(2 * n):
        s = []
        for i in range(int(rlen[r])):
            word, j = w + i // 32, i % 32
            s.append("N" if (int(nmask[word]) >> j) & 1 else "ACGT"[(int(bases[word]) >> (2 * j)) & 3])
        w += (int(rlen[r]) + 31) // 32
        assert (whole[int(id_off[r]):int(id_off[r]) + int(id_len[r])], "".join(s), int(seeds[r])) == want[r % 2][r // 2], r


IRREGULAR_FASTA = [
    (b"\n>a\nACGT\n", 1), (b"#comment\n>a\nACGT\n", 1), (b"ACGT\n>a\nACGT\n", 1),      # does not start with '>'
    (b">a\nACGT\n>b", 2), (b">a>b\nACGT\n", 2),                                           # a name line without its end
    (b">\nACGT\n", 4),                                                                    # no name: the host names it by its ordinal
    (b">a\r\nACGT\r\n", 8), (b">a\nAC\rGT\n", 16),
    (b">a\nACRT\n", 16), (b">a\nAC-T\n", 16), (b">a\nAC.T\n", 16), (b">a\nAC GT\n", 16), (b">a\nAC*\n", 16),
    (b">a\n>b\nACGT\n", 32), (b">a\n\n>b\nACGT\n", 32), (b">a\nACGT\n>b\n", 32),
]
IRREGULAR_FASTQ = [
    (b"\n@a\nACGT\n+\nIIII\n", 1), (b"x\n@a\nACGT\n+\nIIII\n", 1),
    (b"@a\nACGT\n+\nIIII", 512), (b"@a\nACGT\n+\nIIII\n\n", 512), (b"@a\nAC\nGT\n+\nIIII\n", 512),
    (b"@\nACGT\n+\nIIII\n", 4), (b"@a\r\nACGT\r\n+\r\nIIII\r\n", 8),
    (b"@a\nAC.T\n+\nIIII\n", 16), (b"@a\nACRT\n+\nIIII\n", 16), (b"@a\nAC T\n+\nIIII\n", 16),
    (b"@a\n\n+\n\n", 32),
    (b"@a\nACGT\n-\nIIII\n", 64), (b"@a\nACGT\n\nIIII\n", 64),
    (b"@a\nACGT\n+\nIII\n", 128), (b"@a\nACGT\n+\nIIIII\n", 128),
    (b"@a\nACGT\n+\nII I\n", 256), (b"@a\nACGT\n+\nII\tI\n", 256),
    (b"@a\nAC\nGT\n+\nII\nII\n@b\nAC\n", 1 | 64),                                      # wrapped lines: the line count happens to fit
]


@pytest.mark.parametrize("fmt,cases", [(FASTA, IRREGULAR_FASTA), (FASTQ, IRREGULAR_FASTQ)], ids=["fasta", "fastq"])
def test_everything_the_host_parser_treats_specially_is_flagged(L, fmt, cases):
    good = make_records(np.random.default_rng(5), 70, fmt)
    for text, bit in cases:
        for block in (text, good + text):
            if block is not text and (bit & 1 or text[:1] in (b"\n", b"#", b"x", b"A")):
                continue                                      # (what is only wrong at the start of a block)
            flags, _ = parse(L, block, fmt)
            assert flags & bit == bit or (flags and bit == 512), (text, flags, bit)
    # too many records / markers for the arrays: left to the host as well
    text = make_records(np.random.default_rng(6), 50, fmt)
    assert parse(L, text, fmt, rec_cap=49)[0] & 1024
    assert parse(L, text, fmt, rec_cap=50)[0] == 0


@pytest.mark.parametrize("fmt", [FASTA, FASTQ], ids=["fasta", "fastq"])
def test_mutated_blocks_are_either_flagged_or_parsed_as_the_host_parser_does(L, fmt):
    """the safety property of the plain form: whatever a block looks like, if the device takes it, its reads are the host parser's"""
    rng = np.random.default_rng(2024 + fmt)
    junk = [b"\r", b">", b"@", b"+", b"\n", b".", b"-", b"R", b"n", b" ", b"\t", b"/", b"/1", b"\n\n", b"*", b"a", b"\x00", b"\xff"]
    taken = 0
    for trial in range(160):
        text = bytearray(make_records(rng, int(rng.integers(1, 8)), fmt, wrap=int(rng.choice([0, 0, 5])) if fmt == FASTA else 0, lower=True))
        for _ in range(int(rng.integers(1, 4))):
            at = int(rng.integers(0, len(text) + 1))
            kind = int(rng.integers(0, 3))
            if kind == 0:
                text[at:at] = junk[int(rng.integers(0, len(junk)))]
            elif kind == 1 and at < len(text):
                del text[at:at + int(rng.integers(1, 4))]
            elif at < len(text):
                text[at:at + 1] = junk[int(rng.integers(0, len(junk)))]
        text = bytes(text)
        flags, got = parse(L, text, fmt)
        if flags:
            continue
        taken += 1
        want = host_parse(text, fmt)
        assert want is not None and got == want, (trial, text)
    assert taken >= 10                                         # (some mutations leave a plain block: the comparison did run)


# ---------------------------------------------------------------------------------------------- the formatter
def format_reference(names, qlens, rows_of, score2, uid, rank_name, tax_col, leaf):
    out = []
    for q, name in enumerate(names):
        rid = read_id(name)
        rows = rows_of[q]
        if not rows:
            out.append(rid + b"\tunclassified\t0\t0\t%d\t0\t%d\t1\n" % (score2[q], qlens[q]))
        for (u, t, score, hit) in rows:
            sid = uid[u] if leaf[t] and u < len(uid) else rank_name[t]
            out.append(rid + b"\t" + sid + b"\t" + tax_col[t] + b"\t%d\t%d\t%d\t%d\t%d\n" % (score, score2[q], hit, qlens[q], len(rows)))
    return b"".join(out)


@pytest.mark.parametrize("paired", [0, 1])
def test_default_columns_and_the_tally_from_narrow_rows(L, paired):
    rng = np.random.default_rng(77 + paired)
    n_taxa, n_refs, nq = 37, 19, 333
    uid = [("NC_%06d.%d" % (rng.integers(0, 999999), rng.integers(1, 9))).encode() for _ in range(n_refs)]
    rank_name = [rng.choice([b"species", b"genus", b"no rank", b"superkingdom"]) for _ in range(n_taxa)]
    tax_col = [(b"%d" % rng.integers(0, 2 ** 32)) + ((b".%d" % rng.integers(1, 2 ** 32)) if rng.random() < 0.2 else b"") for _ in range(n_taxa)]
    leaf = [int(rng.random() < 0.6) for _ in range(n_taxa)]
    strs, uid_off, rank_off, tax_off = bytearray(), [], [], []
    for table, offs in ((uid, uid_off), (rank_name, rank_off), (tax_col, tax_off)):
        for x in table:
            offs.append(len(strs))
            strs += x
        offs.append(len(strs))
    per = 2 if paired else 1
    text = make_records(rng, nq * per, FASTA)
    flags_, parsed = 0, None
    buf = padded(text)
    rec_cap = nq * per + 16
    arr = [np.zeros(rec_cap + 80, dtype=np.uint32) for _ in range(5)]
    status = np.zeros(4, dtype=np.uint64)
    assert L.emu_text_parse(vp(buf), len(text), FASTA, 0, rec_cap, rec_cap, *[vp(a) for a in arr], vp(status)) == nq * per and not status[3]
    rlen, _seeds, _seq_off, id_off, id_len = arr
    names = [text[int(id_off[r * per]) - 0:].split(b"\n", 1)[0] for r in range(nq)]
    qlens = [int(rlen[q * per]) + (int(rlen[q * per + 1]) if paired else 0) for q in range(nq)]
    rows_of, score2, max_score = [], [], []
    for q in range(nq):
        n = int(rng.choice([0, 1, 1, 1, 2, 3, 5]))
        ms = int(rng.choice([50, 7225, 0xffffffff]))
        rows = [(int(rng.integers(0, n_refs + 3)) if rng.random() < 0.8 else 0xffffffff, int(rng.integers(0, n_taxa)),
                 int(rng.choice([ms if ms != 0xffffffff else 9, 7225, 49, 0, 4000000000])), int(rng.integers(0, 100000))) for _ in range(n)]
        rows_of.append(rows); score2.append(int(rng.choice([0, 49, 12345678]))); max_score.append(ms)
    rows = np.array([r for rs in rows_of for r in rs] + [(0, 0, 0, 0)], dtype=np.uint32).reshape(-1, 4)
    qinfo = np.array([len(rs) | 0x40 for rs in rows_of], dtype=np.uint8)
    s2, ms_a = np.array(score2, dtype=np.uint32), np.array(max_score, dtype=np.uint32)
    want = format_reference(names, qlens, rows_of, score2, uid, rank_name, tax_col, leaf)
    out = np.zeros(len(want) + 64, dtype=np.uint8)
    idx_zero = 3
    single = np.zeros(n_taxa, dtype=np.uint64)
    tuples = np.zeros(6 * nq + 16, dtype=np.uint32)
    tw = C.c_uint32(0)
    sb = np.frombuffer(bytes(strs) + bytes(16), dtype=np.uint8)
    mk = lambda x: np.array(x, dtype=np.uint32)
    uo, ro, to, lf = mk(uid_off), mk(rank_off), mk(tax_off), np.array(leaf, dtype=np.uint8)
    L.emu_text_format.argtypes = None
    got_n = L.emu_text_format(vp(buf), vp(id_off), vp(id_len), vp(rlen), vp(rows), vp(qinfo), vp(s2), vp(ms_a), C.c_uint32(nq), C.c_int(paired),
                              vp(sb), vp(uo), vp(ro), vp(to), vp(lf), C.c_uint32(n_refs), C.c_uint32(n_taxa), C.c_uint32(idx_zero),
                              vp(out), C.c_uint64(len(want)), vp(single), vp(tuples), C.c_uint32(len(tuples)), C.byref(tw))
    assert got_n == len(want)
    assert bytes(out[:got_n]) == want
    # the tally (aln_sink.h:142-172): perfect single assignments per taxon (unclassified under taxon 0), perfect tuples
    want_single = np.zeros(n_taxa, dtype=np.uint64)
    want_tuples = []
    for q in range(nq):
        rs, ms = rows_of[q], max_score[q]
        if not rs:
            want_single[idx_zero] += 1
        elif len(rs) == 1:
            if ms != 0xffffffff and rs[0][2] >= ms:
                want_single[rs[0][1]] += 1
        elif ms != 0xffffffff and all(r[2] >= ms for r in rs):
            want_tuples.append(tuple(r[1] for r in rs))
    assert (single == want_single).all()
    got_tuples, i = [], 0
    while i < tw.value:
        n = int(tuples[i])
        got_tuples.append(tuple(int(x) for x in tuples[i + 1:i + 1 + n]))
        i += 1 + n
    assert sorted(got_tuples) == sorted(want_tuples) and want_tuples
