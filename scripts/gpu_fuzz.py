"""GPU differential fuzz: random references (repeats, ambiguity codes), read lengths, single/paired mixes, ambiguity in reads and random alignment
parameters; every batch must equal the oracle bit for bit."""
import sys, os, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import oracle_lib as o
from helpers import sprinkle_ambiguity, ambiguous_reference, heavy_ambiguity, low_complexity_reads, IUPAC_2WAY, IUPAC_3WAY
from mapper_amd import api, synth



def run_shapes(rounds=12, seed=99, max_queries=1500, backend="gpu"):
  """The shapes the first fuzz does not draw: several contigs (reads across their ends included), a length per read inside one batch (36 ... 450, one round in
  three up to 1 600: chains at the long-read scratch scales, whose searches run in the form of xm_wsearch.h), mates of unequal length, pairs and single reads mixed."""
  rng = np.random.default_rng(seed)
  bad = 0
  for it in range(rounds):
      n_contigs = int(rng.integers(1, 6))
      contigs = []
      for c in range(n_contigs):
          n = int(rng.integers(3_000, 150_000))
          ref = ambiguous_reference(n, seed=3000 + 10 * it + c, n_runs=int(rng.integers(0, 6)), n_codes=int(rng.integers(0, 40))) if (it + c) % 3 == 0 else synth.synthetic_reference(n, seed=3000 + 10 * it + c).copy()
          if (it + c) % 4 == 1:
              unit = ref[:int(rng.integers(30, 800))].copy()
              for k in range(int(rng.integers(2, 12))):
                  p0 = int(rng.integers(0, n - len(unit)))
                  ref[p0:p0 + len(unit)] = unit
          contigs.append(("c%d_%d" % (it, c), ref))
      whole = np.concatenate([r for _, r in contigs])  # (templates may span a contig end: such reads must come out as the oracle's)
      long_round = it % 3 == 2
      lo, hi = (300, 1600) if long_round else (36, 450)
      nq = int(rng.integers(100, 400)) if long_round else int(rng.integers(300, max_queries))
      sub, ind = float(rng.choice([0.0, 0.01, 0.03])), float(rng.choice([0.0, 0.1, 0.6]))
      params = dict(MutationPenalty=float(rng.choice([1.0, 0.7])), InsertionStart_Penalty=float(rng.choice([1.5, 3.0])), InsertionExtension_Penalty=float(rng.choice([0.6, 0.25])),
                    DeletionStart_Penalty=float(rng.choice([1.5, 1.0])), DeletionExtension_Penalty=float(rng.choice([0.5, 0.2])), MaxErrorRate=float(rng.choice([0.1, 0.05, 0.15])),
                    UnalignedPenalty=float(rng.choice([0.1, 0.3])), AmbiguityPenalty=float(rng.choice([0.1, 0.2])), Max_PenaltySpan=float(rng.choice([0.5, 0.0, 1.5])),
                    MaxNumMatches=int(rng.choice([2**31 - 1, 1, 3])))
      queries = []
      for q in range(nq):
          L1 = int(rng.integers(lo, hi))
          if not long_round and q % 3 == 0:  # a pair, mates of different lengths
              L2 = int(rng.integers(lo, hi))
              Lm = max(L1, L2)
              if Lm + Lm + 410 >= len(whole): continue
              m1, m2 = synth.synthetic_paired_end(whole, 1, read_len=Lm, sub_rate=sub, indel_prob=ind, seed=9000 + 977 * it + q)[:2]
              queries.append(([m1[0][:L1], m2[0][:L2]], float(rng.choice([100.0, 0.0])), float(rng.choice([50.0, 10.0]))))
          else:
              if L1 + 4 >= len(whole): continue
              r = synth.synthetic_single_end(whole, 1, read_len=L1, sub_rate=sub, indel_prob=ind, seed=9000 + 977 * it + q)[0]
              if q % 11 == 0: r = sprinkle_ambiguity(r, q)
              queries.append(([r[0]], 0.0, 1.0))
      b = o.QueryBatch(queries)
      maxlen = int(max(len(m) for ms, _, _ in queries for m in ms))
      if backend == "sim":
          import hostsim_lib
          db = hostsim_lib.SimReference(contigs, mode="mapper")
          r = db.align(b, o.make_params(params))
          r.kernel_ms = 0.0
      else:
          db = api.ReferenceDatabase(contigs, mode="mapper", max_query_length=maxlen)
          r = db.align_arrays(b.mate_count, b.mate_offset, b.mate_length, b.codes, b.expected_inner, b.deviation, api.AlignmentParameters(**params))
      want = o.OracleReference(contigs, mode="mapper").align(b, o.make_params(params), threads=os.cpu_count())
      same = np.array_equal(want.ints, r.ints) and np.array_equal(want.dbls.view(np.int64), r.dbls.view(np.int64)) and np.array_equal(want.int_off, r.int_off)
      aligned = int(sum(1 for q in range(len(queries)) if r.ints[r.int_off[q] + 1] > 0))
      print("shapes round", it, "contigs", [len(x) for _, x in contigs], "lengths", lo, hi, "queries", len(queries), "sub", sub, "indel", ind, "aligned", aligned, "kernel ms %.1f" % r.kernel_ms,
            "reruns", r.counters[11], "IDENTICAL" if same else "DIFFERENT", flush=True)
      if not same:
          bad += 1
          for q in range(len(queries)):
              a0, a1 = want.int_off[q], want.int_off[q + 1]
              if a1 - a0 != r.int_off[q + 1] - r.int_off[q] or not np.array_equal(want.ints[a0:a1], r.ints[r.int_off[q]:r.int_off[q + 1]]):
                  print("  first differing query", q, "mate lengths", [len(m) for m in queries[q][0]], "params", params); break
      if backend != "sim":
          db.close()
  print("shapes fuzz done, differing batches:", bad)
  return bad


def run_ambiguity(rounds=200, seed=515, max_queries=400, backend="gpu"):
  """The third flavour: what real FASTQ holds and the first two never draw.  Per read an ambiguous fraction from {0, 1 %, 10 %, 50 %, 90 %, 100 %} as N, as two-way or as
  three-way IUPAC codes, N runs at either end, homopolymer / dinucleotide / short-period reads, reads shorter than minInterestingSize; single reads and pairs (one or
  both mates affected) in one batch; references plain, with ambiguity codes of their own, or with repeats."""
  rng = np.random.default_rng(seed)
  bad = 0
  acgt = np.array([1, 2, 4, 8], np.uint8)
  for it in range(rounds):
      n = int(rng.integers(8_000, 120_000))
      ref = ambiguous_reference(n, seed=7000 + it, n_runs=int(rng.integers(0, 12)), n_codes=int(rng.integers(0, 60))) if it % 4 == 0 else synth.synthetic_reference(n, seed=7000 + it).copy()
      if it % 4 == 2:
          unit = ref[:int(rng.integers(20, 600))].copy()
          for k in range(int(rng.integers(2, 10))):
              p0 = int(rng.integers(0, n - len(unit)))
              ref[p0:p0 + len(unit)] = unit
      L = int(rng.choice([36, 75, 100, 150, 151, 250]))
      nq = int(rng.integers(96, max_queries))
      sub, ind = float(rng.choice([0.0, 0.01, 0.03])), float(rng.choice([0.0, 0.2]))
      params = dict(MaxErrorRate=float(rng.choice([0.1, 0.05, 0.2])), AmbiguityPenalty=float(rng.choice([0.1, 0.05, 0.2])), UnalignedPenalty=float(rng.choice([0.1, 0.3])),
                    Max_PenaltySpan=float(rng.choice([0.5, 0.0, 1.5])), MaxNumMatches=int(rng.choice([2**31 - 1, 1, 3])))
      def ambiguate(r):  # one read, its own fraction and kind
          r = r.copy()
          kind = int(rng.integers(0, 6))
          f = float(rng.choice([0.0, 0.01, 0.1, 0.5, 0.9, 1.0]))
          m = rng.random(len(r)) < f
          if kind <= 1: r[m] = 15
          elif kind == 2: r[m] = IUPAC_2WAY[rng.integers(0, len(IUPAC_2WAY), int(m.sum()))]
          elif kind == 3: r[m] = IUPAC_3WAY[rng.integers(0, len(IUPAC_3WAY), int(m.sum()))]
          elif kind == 4: r[:int(rng.integers(0, len(r) + 1))] = 15
          else: r[len(r) - int(rng.integers(0, len(r) + 1)):] = 15
          return r
      reads = synth.synthetic_single_end(ref, nq, read_len=L, sub_rate=sub, indel_prob=ind, seed=8000 + it)[0]
      m1, m2 = synth.synthetic_paired_end(ref, nq // 2, read_len=L, sub_rate=sub, indel_prob=ind, seed=8500 + it)[:2]
      low = low_complexity_reads(24, L, seed=it)
      queries = [([ambiguate(r)], 0.0, 1.0) for r in reads]
      queries += [([ambiguate(m1[i]) if i % 3 != 1 else m1[i], ambiguate(m2[i]) if i % 3 != 0 else m2[i]], 100.0, 50.0) for i in range(len(m1))]
      queries += [([r], 0.0, 1.0) for r in low]
      queries += [([ambiguate(reads[i][:int(rng.integers(1, 24))])], 0.0, 1.0) for i in range(16)]
      order = rng.permutation(len(queries))
      queries = [queries[i] for i in order]
      b = o.QueryBatch(queries)
      contigs = [("a%d" % it, ref)]
      if backend == "sim":
          import hostsim_lib
          db = hostsim_lib.SimReference(contigs, mode="mapper")
          r = db.align(b, o.make_params(params))
          r.kernel_ms = 0.0
      else:
          db = api.ReferenceDatabase(contigs, mode="mapper", max_query_length=L)
          r = db.align_arrays(b.mate_count, b.mate_offset, b.mate_length, b.codes, b.expected_inner, b.deviation, api.AlignmentParameters(**params))
      want = o.OracleReference(contigs, mode="mapper").align(b, o.make_params(params), threads=os.cpu_count())
      same = np.array_equal(want.ints, r.ints) and np.array_equal(want.dbls.view(np.int64), r.dbls.view(np.int64)) and np.array_equal(want.int_off, r.int_off)
      aligned = int(sum(1 for q in range(len(queries)) if r.ints[r.int_off[q] + 1] > 0))
      print("ambiguity round", it, "ref", n, "len", L, "queries", len(queries), "aligned", aligned, "kernel ms %.1f" % r.kernel_ms, "reruns", r.counters[11], "IDENTICAL" if same else "DIFFERENT", flush=True)
      if not same:
          bad += 1
          for q in range(len(queries)):
              a0, a1 = want.int_off[q], want.int_off[q + 1]
              if a1 - a0 != r.int_off[q + 1] - r.int_off[q] or not np.array_equal(want.ints[a0:a1], r.ints[r.int_off[q]:r.int_off[q + 1]]):
                  print("  first differing query", q, "mate lengths", [len(m) for m in queries[q][0]], "params", params); break
      if backend != "sim":
          db.close()
  print("ambiguity fuzz done, differing batches:", bad)
  return bad


def run(rounds=24, seed=2026, max_queries=4000, backend="gpu"):
  """backend "gpu": libxmapper_hip.so; "sim": the kernel sources in the host simulation of the tests (tests/hostsim), for the CPU-only tier."""
  rng = np.random.default_rng(seed)
  bad = 0
  for it in range(rounds):
      n = int(rng.integers(20_000, 400_000))
      ref = ambiguous_reference(n, seed=1000 + it, n_runs=int(rng.integers(0, 30)), n_codes=int(rng.integers(0, 200))) if it % 3 == 0 else synth.synthetic_reference(n, seed=1000 + it).copy()
      if it % 4 == 1:  # tandem repeats
          unit = ref[:int(rng.integers(50, 3000))].copy()
          for k in range(int(rng.integers(3, 30))):
              p0 = int(rng.integers(0, n - len(unit)))
              ref[p0:p0 + len(unit)] = unit
      L = int(rng.choice([36, 75, 100, 150, 151, 250, 301]))
      nq = int(rng.integers(500, max_queries))
      sub, ind = float(rng.choice([0.0, 0.005, 0.02, 0.06])), float(rng.choice([0.0, 0.05, 0.5]))
      params = dict(MutationPenalty=float(rng.choice([1.0, 0.7, 2.0])), InsertionStart_Penalty=float(rng.choice([1.5, 1.0, 3.0])), InsertionExtension_Penalty=float(rng.choice([0.6, 0.25, 1.1])),
                    DeletionStart_Penalty=float(rng.choice([1.5, 1.0, 3.0])), DeletionExtension_Penalty=float(rng.choice([0.5, 0.2, 1.0])), MaxErrorRate=float(rng.choice([0.1, 0.05, 0.2])),
                    UnalignedPenalty=float(rng.choice([0.1, 0.3])), AmbiguityPenalty=float(rng.choice([0.1, 0.05, 0.2])), Max_PenaltySpan=float(rng.choice([0.5, 0.0, 1.5])),
                    MaxNumMatches=int(rng.choice([2**31 - 1, 1, 3])))
      queries = []
      if it % 2 == 0:
          reads = synth.synthetic_single_end(ref, nq, read_len=L, sub_rate=sub, indel_prob=ind, seed=5000 + it)[0]
          if it % 5 == 0: reads = sprinkle_ambiguity(reads, it)
          queries = [([r], 0.0, 1.0) for r in reads]
      else:
          m1, m2 = synth.synthetic_paired_end(ref, nq // 2, read_len=L, sub_rate=sub, indel_prob=ind, seed=5000 + it)[:2]
          if it % 5 == 0: m1 = sprinkle_ambiguity(m1, it)
          e, d = float(rng.choice([100.0, 0.0, 300.0])), float(rng.choice([50.0, 10.0]))
          queries = [([m1[i], m2[i]], e, d) for i in range(len(m1))]
          extra = synth.synthetic_single_end(ref, 200, read_len=L, seed=7000 + it)[0]
          queries += [([r], 0.0, 1.0) for r in extra]
      b = o.QueryBatch(queries)
      mode = "api" if it % 7 == 3 else "mapper"
      t = time.time()
      if backend == "sim":
          import hostsim_lib
          db = hostsim_lib.SimReference([("r%d" % it, ref)], mode=mode)
          r = db.align(b, o.make_params(params))
          r.kernel_ms = 0.0
      else:
          db = api.ReferenceDatabase([("r%d" % it, ref)], mode=mode, max_query_length=L)
          r = db.align_arrays(b.mate_count, b.mate_offset, b.mate_length, b.codes, b.expected_inner, b.deviation, api.AlignmentParameters(**params))
      want = o.OracleReference([("r%d" % it, ref)], mode=mode).align(b, o.make_params(params), threads=os.cpu_count())
      same = np.array_equal(want.ints, r.ints) and np.array_equal(want.dbls.view(np.int64), r.dbls.view(np.int64)) and np.array_equal(want.int_off, r.int_off)
      aligned = int(sum(1 for q in range(len(queries)) if r.ints[r.int_off[q] + 1] > 0))
      print("round", it, "ref", n, "len", L, "queries", len(queries), "sub", sub, "indel", ind, "mode", mode, "aligned", aligned, "kernel ms %.1f" % r.kernel_ms, "reruns", r.counters[11],
            "IDENTICAL" if same else "DIFFERENT", flush=True)
      if not same:
          bad += 1
          for q in range(len(queries)):
              a0, a1 = want.int_off[q], want.int_off[q + 1]
              if a1 - a0 != r.int_off[q + 1] - r.int_off[q] or not np.array_equal(want.ints[a0:a1], r.ints[r.int_off[q]:r.int_off[q + 1]]):
                  print("  first differing query", q, "params", params); break
      if backend != "sim":
          db.close()
  print("fuzz done, differing batches:", bad)
  return bad


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "ambiguity":  # gpu_fuzz.py ambiguity [rounds] [seed] [sim]
        sys.exit(1 if run_ambiguity(int(sys.argv[2]) if len(sys.argv) > 2 else 200, int(sys.argv[3]) if len(sys.argv) > 3 else 515, backend=sys.argv[4] if len(sys.argv) > 4 else "gpu") else 0)
    if len(sys.argv) > 1 and sys.argv[1] == "shapes":  # gpu_fuzz.py shapes [rounds] [seed] [sim]
        sys.exit(1 if run_shapes(int(sys.argv[2]) if len(sys.argv) > 2 else 12, int(sys.argv[3]) if len(sys.argv) > 3 else 99, backend=sys.argv[4] if len(sys.argv) > 4 else "gpu") else 0)
    sys.exit(1 if run(int(sys.argv[1]) if len(sys.argv) > 1 else 24, int(sys.argv[2]) if len(sys.argv) > 2 else 2026) else 0)
