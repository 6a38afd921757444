"""Seed-probe rate on an HBM-resident index against max_per_probe and the batch size (what the outputs cost).  usage: gpu_probe_sweep.py [index Mb]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from mapper_amd import api, synth
mb = int(sys.argv[1]) if len(sys.argv) > 1 else 500
ref = np.concatenate([synth.synthetic_reference(min(50_000_000, mb * 1_000_000 - o), seed=0x9B0 + o // 50_000_000) for o in range(0, mb * 1_000_000, 50_000_000)])
db = api.ReferenceDatabase([("probe_ref", ref)], max_query_length=150)
info = db.info()
rng = np.random.default_rng(12345)
sect, _ = api.measure_random_gather(4 << 30, 1 << 26, 0)
print("gather ceiling %.1f G sectors/s" % (sect / 1e9), flush=True)
for n in (4_000_000, 16_000_000, 64_000_000):
    used = rng.integers(info["min_interesting_size"], 151, size=n, dtype=np.int32)
    keys = rng.integers(-2**31, 2**31 - 1, size=n, dtype=np.int64).astype(np.int32)
    for mpp in (0, 1, 2, 4, 7):
        db.seed_probe(used[:4096], keys[:4096], mpp)
        c, _, ms = db.seed_probe(used, keys, mpp)
        c, _, ms2 = db.seed_probe(used, keys, mpp)
        print("n %d max_per_probe %d: %.3f / %.3f ms -> %.1f G probes/s = %.3f of the ceiling" % (n, mpp, ms, ms2, n / (min(ms, ms2) * 1e-3) / 1e9, n / (min(ms, ms2) * 1e-3) / sect), flush=True)
