#!/usr/bin/env python3
"""Writes tests/golden/synthetic_golden.json: SHA-256 digests of the oracle's result streams on seeded synthetic batches
(the batches are regenerated from the seeds by mapper_amd.synth, so only the digests need to be stored)."""
import hashlib
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np  # noqa: E402
import oracle_lib as o  # noqa: E402
from helpers import se_batch, pe_batch  # noqa: E402
from mapper_amd import synth  # noqa: E402


def digest(s):
    h = hashlib.sha256()
    for a in (s.int_off, s.dbl_off, s.ints, np.asarray(s.dbls).view(np.int64)):
        h.update(np.ascontiguousarray(a).tobytes())
    return h.hexdigest()


def cases():
    ref = synth.synthetic_reference(400_000, seed=0xEC011)
    reads, _, _ = synth.synthetic_single_end(ref, 4000, seed=0x5EED0001)
    m1, m2, _, _, _ = synth.synthetic_paired_end(ref, 1500, seed=0x5EED0002)
    return ref, {"single_end_4000": se_batch(reads), "paired_end_1500": pe_batch(m1, m2)}


def full_cases():
    """BASELINE.json configs[1] and configs[2] at full size, exactly the batches bench.py aligns on rank 0 (--config 1 / --config 2): 1,000,000
    single-end reads and 1,000,000 pairs (--spacing 100 50) of 150 bp against the 5 Mb synthetic reference."""
    ref = synth.synthetic_reference(5_000_000, seed=0xEC011)
    reads = synth.synthetic_single_end(ref, 1_000_000, read_len=150, seed=0x5EED0001)[0]
    m1, m2 = synth.synthetic_paired_end(ref, 1_000_000, read_len=150, seed=0x5EED0002)[:2]
    return ref, {"configs1_single_end_1000000": lambda: se_batch(reads), "configs2_paired_end_1000000": lambda: pe_batch(m1, m2, 100.0, 50.0)}


def stated_size_cases():
    """BASELINE.json configs[2] at the size it states: 10,000,000 pairs 2 x 150 bp (--spacing 100 50) against the 5 Mb reference, in one batch (what
    `bench.py --config 2 --reads 10000000` aligns).  The oracle needs ~40 s for it on a 256-core box, a quarter of an hour on eight cores."""
    ref = synth.synthetic_reference(5_000_000, seed=0xEC011)

    def make():
        m1, m2 = synth.synthetic_paired_end(ref, 10_000_000, read_len=150, seed=0x5EED0002)[:2]
        return pe_batch(m1, m2, 100.0, 50.0)
    return ref, {"configs2_paired_end_10000000": make}


if __name__ == "__main__" and len(sys.argv) > 1 and sys.argv[1] == "stated":  # adds stated_digests to the committed file (nothing else is computed again)
    path = os.path.join(ROOT, "tests", "golden", "synthetic_golden.json")
    out = json.load(open(path))
    out.setdefault("stated_digests", {})
    ref, batches = stated_size_cases()
    R = o.OracleReference([("ecoli_syn", ref)])
    for name, make in batches.items():
        s = R.align(make(), o.make_params(), threads=os.cpu_count())
        out["stated_digests"][name] = {"sha256": digest(s), "num_ints": int(len(s.ints)), "num_dbls": int(len(s.dbls)), "oracle_counters": [int(x) for x in s.counters[:9]]}
        print(name, out["stated_digests"][name], flush=True)
    with open(path, "w") as f:
        json.dump(out, f, indent=1)
elif __name__ == "__main__":
    ref, batches = cases()
    R = o.OracleReference([("ecoli_syn", ref)])
    out = {"reference": "synthetic_reference(400000, seed=0xEC011)", "params": "Mapper.main defaults", "digests": {}}
    for name, b in batches.items():
        s = R.align(b, o.make_params())
        out["digests"][name] = {"sha256": digest(s), "num_ints": int(len(s.ints)), "num_dbls": int(len(s.dbls))}
    # the full-size batches (minutes of oracle time on a few cores): digests of the whole result streams + the oracle's work counters
    out["full_reference"] = "synthetic_reference(5000000, seed=0xEC011)"
    out["full_digests"] = {}
    out["stated_digests"] = json.load(open(os.path.join(ROOT, "tests", "golden", "synthetic_golden.json"))).get("stated_digests", {})  # (made by `make_synthetic_golden.py stated`)
    ref, batches = full_cases()
    R = o.OracleReference([("ecoli_syn", ref)])
    for name, make in batches.items():
        s = R.align(make(), o.make_params(), threads=os.cpu_count())
        out["full_digests"][name] = {"sha256": digest(s), "num_ints": int(len(s.ints)), "num_dbls": int(len(s.dbls)), "oracle_counters": [int(x) for x in s.counters[:9]]}
        print(name, out["full_digests"][name], flush=True)
    with open(os.path.join(ROOT, "tests", "golden", "synthetic_golden.json"), "w") as f:
        json.dump(out, f, indent=1)
    print(out)
