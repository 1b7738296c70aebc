#!/usr/bin/env python3
"""Ingest rate of the front end without a GPU: `centrifuge-class --ingest-bench` parses the input with the I/O + parser
threads and hands the batches straight back (no classification), printing reads, bases and seconds.
usage: tools/ingest_rate.py [reads (default 32e6)] [threads ...]   -> FASTA and FASTQ, GB/s and reads/s per thread count"""
import os
import re
import subprocess
import sys
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CLI = os.path.join(ROOT, "centrifuge_amd", "bin", "centrifuge-class")


def make(path, n, fastq):
    rng = np.random.default_rng(1)
    step = 2_000_000
    with open(path, "wb") as f:
        for a in range(0, n, step):
            m = min(step, n - a)
            seq = np.frombuffer(b"ACGT", dtype=np.uint8)[rng.integers(0, 4, size=(m, 100), dtype=np.uint8)]
            names = np.frombuffer("".join("%08d" % i for i in range(a, a + m)).encode(), dtype=np.uint8).reshape(m, 8)
            w = 215 if fastq else 112
            out = np.empty((m, w), dtype=np.uint8)
            out[:, 0] = ord("@" if fastq else ">"); out[:, 1] = ord("r"); out[:, 2:10] = names; out[:, 10] = 10
            out[:, 11:111] = seq; out[:, 111] = 10
            if fastq:
                out[:, 112] = ord("+"); out[:, 113] = 10
                out[:, 114:214] = rng.integers(35, 74, size=(m, 100), dtype=np.uint8); out[:, 214] = 10
            out.tofile(f)


def main():
    n = int(float(sys.argv[1])) if len(sys.argv) > 1 else 32_000_000
    threads = [int(x) for x in sys.argv[2:]] or [1, 4, 8, 16]
    with tempfile.TemporaryDirectory(dir=os.environ.get("TMPDIR", "/tmp")) as t:
        for fastq in (False, True):
            p = os.path.join(t, "r.fq" if fastq else "r.fa")
            make(p, n, fastq)
            size = os.path.getsize(p)
            for th in threads:
                best = None
                for _ in range(3):
                    r = subprocess.run([CLI, "--ingest-bench", "-q" if fastq else "-f", "-p", str(th), "-U", p], capture_output=True, text=True)
                    m = re.search(r"in ([0-9.]+) s", r.stderr)
                    assert r.returncode == 0 and m, r.stderr
                    s = float(m.group(1))
                    best = s if best is None else min(best, s)
                print("%s  %.2f GB  -p %2d  %.3f s  %.2f GB/s  %.3g reads/s" % ("FASTQ" if fastq else "FASTA", size / 1e9, th, best, size / best / 1e9, n / best), flush=True)
            os.remove(p)


if __name__ == "__main__":
    main()
