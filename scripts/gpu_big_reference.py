"""GRCh38-scale smoke of the paths a 5 Mb reference never takes (scripts/README.md): more than 2^32 encoded positions (64-bit position
arrays on the device), several long contigs, the index hashed on the GPU in one or more groups.  No oracle at this size: the check is
that reads sampled from known places come back to them.

    python scripts/gpu_big_reference.py [contigs=10] [contig_len=230000000] [reads_per_sampled_contig=50000]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from mapper_amd import api, synth

nc = int(sys.argv[1]) if len(sys.argv) > 1 else 10
clen = int(sys.argv[2]) if len(sys.argv) > 2 else 230_000_000
per = int(sys.argv[3]) if len(sys.argv) > 3 else 50_000
t = time.time()
contigs = []
for c in range(nc):
    parts = [synth.synthetic_reference(min(50_000_000, clen - o), seed=0xB16 + 1000 * c + o // 50_000_000) for o in range(0, clen, 50_000_000)]
    contigs.append(("chr%02d" % c, np.concatenate(parts)))
print("reference: %d contigs x %d = %.2f G bases generated in %.1f s" % (nc, clen, nc * clen / 1e9, time.time() - t), flush=True)
t = time.time()
db = api.ReferenceDatabase(contigs, max_query_length=150)
i = db.info()
print("index: %.1f s (hash %.1f s on %s, duplication map %.1f s), position bytes %d, %d positions, %.1f GB" %
      (time.time() - t, i["hash_seconds"], "GPU" if i["built_on_device"] else "host", i["duplication_seconds"], i["position_bytes"], i["num_positions"], i["index_bytes"] / 1e9), flush=True)
sampled = sorted(set([0, nc // 2, nc - 1]))
reads, where = [], []
for c in sampled:
    r, starts, strand = synth.synthetic_single_end(contigs[c][1], per, seed=0x5EED + c)
    reads.append(r)
    where.append(np.stack([np.full(per, c), starts, strand], axis=1))
reads = np.concatenate(reads); where = np.concatenate(where)
nq = len(reads)
mc = np.ones(nq, np.int32)
mo = np.zeros(2 * nq, np.int64); mo[0::2] = np.arange(nq, dtype=np.int64) * 150
ml = np.zeros(2 * nq, np.int32); ml[0::2] = 150
t = time.time()
r = db.align_arrays(mc, mo, ml, np.ascontiguousarray(reads.reshape(-1)), np.zeros(nq), np.ones(nq), api.AlignmentParameters())
dt = time.time() - t
io = r.int_off[:-1]
nal = r.ints[io + 1]
one = nal >= 1
contig = r.ints[io[one] + 4]; rev = r.ints[io[one] + 5]; sa = r.ints[io[one] + 7]; sb = r.ints[io[one] + 8]
ok = (contig == where[one, 0]) & (rev == where[one, 2]) & (np.abs((sb - sa) - where[one, 1]) <= 3)
print("aligned %d reads in %.2f s (kernel %.1f ms): %.2f %% aligned, %.2f %% of those at their origin (contig, strand, offset), reruns %d" %
      (nq, dt, r.kernel_ms, 100.0 * one.mean(), 100.0 * ok.mean(), r.counters[11]), flush=True)
# seed lookups on this index (far larger than every cache): bucket lines against CSR probes, next to the random-sector ceiling
rng = np.random.default_rng(12345)
n = 32_000_000
used = rng.integers(i["min_interesting_size"], i["max_hashed_length"] + 1, size=n, dtype=np.int32)
keys = rng.integers(-2**31, 2**31 - 1, size=n, dtype=np.int64).astype(np.int32)
sectors_per_s, _ = api.measure_random_gather(16 << 30, 1 << 27, 0)
print("random 64 B sector gather ceiling: %.1f G sectors/s = %.0f GB/s" % (sectors_per_s / 1e9, sectors_per_s * 64 / 1e9), flush=True)
for no_lines in ("1", "0"):
    os.environ["XM_PROBE_NO_LINES"] = no_lines
    db.seed_probe(used[:4096], keys[:4096], 0)
    c0, _, ms_hdr = db.seed_probe(used, keys, 0)
    c4, pos, ms_pos = db.seed_probe(used[:n // 2], keys[:n // 2], 7)
    fetched = int(np.minimum(np.maximum(c4, 0), 7).sum())
    pb = i["position_bytes"]
    print("seed probe (%s): header only %.1f G probes/s = %.0f GB/s of 64 B sectors (%.1f %% of the 8 TB/s peak, %.0f %% of the gather ceiling); with positions %.1f G probes/s, "
          "%.2f positions per probe, algorithmic %.0f GB/s (8 B header + %d B per position)" %
          ("CSR: bucketOff then positions" if no_lines == "1" else "bucket lines", n / ms_hdr / 1e6, n / ms_hdr / 1e6 * 64, n / ms_hdr / 1e6 * 64 / 80.0, 100.0 * (n / (ms_hdr * 1e-3)) / sectors_per_s,
           (n // 2) / ms_pos / 1e6, fetched / (n // 2), (8.0 * (n // 2) + pb * fetched) / (ms_pos * 1e-3) / 1e9, pb), flush=True)
    if no_lines == "1":
        ref_counts, ref_pos = c4.copy(), pos.copy()
    else:
        assert np.array_equal(ref_counts, c4) and np.array_equal(ref_pos, pos), "bucket lines and CSR probes disagree"
os.environ["XM_PROBE_NO_LINES"] = "0"
assert i["position_bytes"] == (8 if 2 * nc * clen > 0xFFFFFFFF else 4)
assert one.mean() > 0.99 and ok.mean() > 0.99
print("ok")
