"""Shared helpers of the test-suite."""
import json
import os
import tarfile
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLDEN = os.path.join(ROOT, "tests", "golden")
_cache = {}


def golden(name):
    """Extract tests/golden/<name>.tar.xz once per session -> (dir, cases)."""
    if name not in _cache:
        d = tempfile.mkdtemp(prefix="cf_golden_%s_" % name)
        with tarfile.open(os.path.join(GOLDEN, name + ".tar.xz")) as t:
            t.extractall(d)
        _cache[name] = (d, json.load(open(os.path.join(d, "cases.json"))))
    return _cache[name]


def case_kwargs(args):
    """reference CLI args of a golden case -> (classifier kwargs, fastq)."""
    kw, fastq, i = {}, False, 0
    while i < len(args):
        a = args[i]
        if a == "-f":
            pass
        elif a == "-q":
            fastq = True
        elif a == "-k":
            kw["k"] = int(args[i + 1]); i += 1
        elif a == "--no-traverse":
            kw["traverse"] = False
        elif a == "--classification-rank":
            kw["rank"] = args[i + 1]; i += 1
        elif a == "--host-taxids":
            kw["host"] = [int(x) for x in args[i + 1].split(",")]; i += 1
        elif a == "--exclude-taxids":
            kw["exclude"] = [int(x) for x in args[i + 1].split(",")]; i += 1
        elif a == "--min-hitlen":
            kw["min_hitlen"] = int(args[i + 1]); i += 1
        else:
            raise ValueError(a)
        i += 1
    return kw, fastq


def all_cases():
    out = []
    for arch in ("example", "synth_small"):
        _, cases = golden(arch)
        out += [(arch, c["name"]) for c in cases]
    return out


def first_diff(a, b, n=5):
    la, lb = a.splitlines(), b.splitlines()
    msgs = ["%d vs %d lines" % (len(la), len(lb))]
    for x, y in zip(la, lb):
        if x != y:
            msgs.append("got: %s\nref: %s" % (x, y))
            if len(msgs) > n:
                break
    return "\n".join(msgs)
