"""CPU tier (no GPU): the product's host-side index builder (through the real C ABI, host_only=1) and the kernel sources
compiled for the host (tests/hostsim, a test harness — the shipped library has no CPU path) against the oracle."""
import os
import re
import numpy as np
import pytest

import oracle_lib as o
import hostsim_lib as hs
from helpers import (KAT, streams_equal, first_difference, se_batch, pe_batch, ragged_se_batch, check_align_case, sprinkle_ambiguity, ambiguous_reference,
                     heavy_ambiguity, low_complexity_reads, bound_problems, filter_counters)
from mapper_amd import api, synth, _capi

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_capi_exports_every_declared_symbol():
    """libxmapper_hip.so loads on a machine without GPU and exports every function include/xmapper_hip.h declares."""
    header = open(os.path.join(ROOT, "include", "xmapper_hip.h")).read()
    declared = sorted(set(re.findall(r"\b(xm_[a-z_]+)\s*\(", header)))
    assert len(declared) >= 12
    lib = _capi.lib()
    for sym in declared:
        assert hasattr(lib, sym), sym
    assert sorted(_capi.EXPORTS) == declared


def test_missing_library_is_an_import_error_not_a_build():
    """mapper_amd never builds the library behind the caller's back (a process under a profiler must not spawn make/hipcc) and has no CPU
    fallback: a missing libxmapper_hip.so is an ImportError that says how to build it."""
    import subprocess, sys
    code = "import mapper_amd._capi as c\ntry:\n    c.lib()\nexcept ImportError as e:\n    print('IMPORT-ERROR', e)\n"
    env = dict(os.environ, XM_LIB_PATH="/nonexistent/libxmapper_hip.so")
    out = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, cwd=os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
    assert "IMPORT-ERROR" in out.stdout and "make -j8 -C mapper_amd/csrc" in out.stdout, out.stdout + out.stderr


def test_build_stamp_is_the_digest_of_the_sources():
    """xm_build_stamp(): the library carries the digest of the sources it was compiled from; smoke() and bench.py refuse a stale one."""
    assert api._capi.check_stamp() == api._capi.source_stamp()


def test_importing_the_package_leaves_the_process_environment_alone():
    """GPU_MAX_HW_QUEUES belongs to the process: the entry points set it (python -m mapper_amd, bench.py; _capi.want_hardware_queues for a program that
    embeds the package), importing mapper_amd and loading the library do not."""
    import subprocess, sys
    code = ("import os, sys; sys.path.insert(0, %r); os.environ.pop('GPU_MAX_HW_QUEUES', None); import mapper_amd; from mapper_amd import api, _capi; _capi.lib(); "
            "assert 'GPU_MAX_HW_QUEUES' not in os.environ; import warnings; "
            # a value set AFTER the library was loaded is never seen by the HIP runtime: the function must not report success (round-5 advice)
            "os.environ['GPU_MAX_HW_QUEUES'] = '8'; "
            "w = warnings.catch_warnings(record=True); rec = w.__enter__(); warnings.simplefilter('always'); ok = _capi.want_hardware_queues(8); w.__exit__(None, None, None); "
            "assert ok is False and len(rec) == 1 and 'when the library was loaded' in str(rec[0].message), (ok, [str(r.message) for r in rec]); print('ok')" % ROOT)
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=600)
    assert out.returncode == 0 and "ok" in out.stdout, out.stdout + out.stderr
    # set before the library is loaded (what the entry points do): success, no warning; a user's smaller value: a warning that says so
    code = ("import os, sys; sys.path.insert(0, %r); os.environ.pop('GPU_MAX_HW_QUEUES', None); import warnings; warnings.simplefilter('error'); from mapper_amd import _capi; "
            "assert _capi.want_hardware_queues(8) and os.environ['GPU_MAX_HW_QUEUES'] == '8'; _capi.lib(); assert _capi.want_hardware_queues(8); "
            "os.environ['GPU_MAX_HW_QUEUES'] = '2'; _capi._lib = None; warnings.simplefilter('always'); "
            "w = warnings.catch_warnings(record=True); rec = w.__enter__(); warnings.simplefilter('always'); ok = _capi.want_hardware_queues(8); w.__exit__(None, None, None); "
            "assert ok is False and 'is set to 2' in str(rec[0].message); print('ok')" % ROOT)
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=600)
    assert out.returncode == 0 and "ok" in out.stdout, out.stdout + out.stderr


def test_library_refuses_to_align_without_gpu_index():
    """The product fails loudly instead of falling back to a CPU path."""
    db = api.ReferenceDatabase([("r", synth.synthetic_reference(5000))], host_only=True)
    with pytest.raises(RuntimeError, match="GPU only|host_only"):
        db.align_batch([api.Query("ACGTACGTACGTACGTACGTACGTACGT")], api.AlignmentParameters())
    db.close()


@pytest.mark.parametrize("n", [1, 2, 5, 33, 150, 400])
def test_read_pyramid_matches_oracle(n):
    rng = np.random.default_rng(n)
    codes = np.array([1, 2, 4, 8], dtype=np.uint8)[rng.integers(0, 4, n)]
    a, b = o.pyramid_dump(codes), hs.pyramid_dump(codes)
    assert a.shape == b.shape and np.array_equal(a, b)


@pytest.mark.parametrize("mode,contigs", [("mapper", [120_000]), ("api", [3000]), ("mapper", [40_000, 25_000, 700])])
def test_index_builder_matches_oracle(mode, contigs):
    """Every PackedMap (capacity, per-bucket counts / overfull marks, packed positions) and the duplication keys of the host
    builder in libxmapper_hip.so equal the oracle's literal restatement of HashBlock_Database / DuplicationDetector."""
    refs = [("c%d" % i, synth.synthetic_reference(n, seed=0xEC011 + i)) for i, n in enumerate(contigs)]
    R = o.OracleReference(refs, mode=mode)
    R.align(["ACGTACGTACGTAGCATCGACTAGCAGCATCGAC"], o.make_params())  # triggers prepare()
    P = api.ReferenceDatabase(refs, mode=mode, host_only=True)
    mn, mx = R.index_info()
    info = P.info()
    assert info["min_interesting_size"] == mn and info["max_hashed_length"] == mx
    for L in range(0, mx + 1):
        ta, tb = R.table(L), P.table(L)
        assert ta["capacity"] == tb["capacity"] and ta["maxCount"] == tb["maxCount"], L
        assert np.array_equal(ta["counts"], tb["counts"]), L
        assert np.array_equal(ta["positions"], tb["positions"]), L
    for c in range(len(refs)):
        assert np.array_equal(R.dup_keys(c), P.dup_keys(c))
    # lazy growth (Readable_HashBlock_Database.java:108-113) gives the same tables as hashing them up front
    R.require_size(mx + 20)
    P.ensure_length(R.index_info()[1])
    for L in range(mx + 1, R.index_info()[1] + 1):
        ta, tb = R.table(L), P.table(L)
        assert ta["capacity"] == tb["capacity"] and np.array_equal(ta["counts"], tb["counts"]) and np.array_equal(ta["positions"], tb["positions"]), L
    P.close()


def _layout(info):  # what describes the index itself (not where or how fast it was built)
    return {k: v for k, v in info.items() if k not in ("built_on_device", "reserved", "bucket_line_bytes", "hash_seconds", "duplication_seconds")}


def _same_index(A, B):
    ia, ib = (_layout(x.info()) for x in (A, B))
    assert ia == ib
    for L in range(0, ia["max_hashed_length"] + 1):
        ta, tb = A.table(L), B.table(L)
        assert ta["capacity"] == tb["capacity"] and ta["maxCount"] == tb["maxCount"], L
        assert np.array_equal(ta["counts"], tb["counts"]) and np.array_equal(ta["positions"], tb["positions"]), L
    for c in range(ia["num_contigs"]):
        assert np.array_equal(A.dup_keys(c), B.dup_keys(c))


def test_index_cache_round_trip(tmp_path):
    """--cache-dir (DirCache.java:19-60, HashBlock_Database.java:106-114): the file written after a build gives back the same tables and
    duplication keys; it is only used for the reference and settings it was built with; a damaged file is rebuilt, not trusted."""
    refs = [("c0", synth.synthetic_reference(30_000, seed=5)), ("c1", ambiguous_reference(6000, seed=6))]
    A = api.ReferenceDatabase(refs, host_only=True, cache_dir=tmp_path)
    assert not A.cache_hit and os.path.exists(A.cache_file) and os.path.exists(os.path.join(os.path.dirname(A.cache_file), "metadata"))
    B = api.ReferenceDatabase(refs, host_only=True, cache_dir=tmp_path)
    assert B.cache_hit and B.cache_file == A.cache_file
    _same_index(A, B)
    # tables hashed on demand after the load equal those of the index that was built
    A.ensure_length(A.info()["max_hashed_length"] + 7)
    C2 = api.ReferenceDatabase(refs, host_only=True, cache_dir=tmp_path, max_query_length=A.info()["max_hashed_length"])
    assert C2.cache_hit
    _same_index(A, C2)
    D = api.ReferenceDatabase.load(A.cache_file, host_only=True)
    _same_index(B, D)
    # other settings or another reference: another file
    E = api.ReferenceDatabase(refs, host_only=True, cache_dir=tmp_path, enable_gapmers=False)
    F = api.ReferenceDatabase([("c0", synth.synthetic_reference(30_000, seed=7))], host_only=True, cache_dir=tmp_path)
    assert not E.cache_hit and not F.cache_hit and len({A.cache_file, E.cache_file, F.cache_file}) == 3
    # a file behind the right name that holds something else is refused by the library (and then replaced)
    import shutil
    shutil.copyfile(F.cache_file, A.cache_file)
    with pytest.raises(RuntimeError, match="another reference"):
        L_ = api._capi.lib()
        ref, keep = api._capi.make_ref(A.contigs)
        h = api.C.c_void_p()
        ob = api._capi.XmBuildOpts(); ob.enable_gapmers = 1; ob.min_interesting_size = -1; ob.dup_window = 1000; ob.dup_min_copies = 2; ob.dup_min_length = ob.dup_max_length = -1; ob.host_only = 1
        if L_.xm_index_load(A.cache_file.encode(), api.C.byref(ref), api.C.byref(ob), api.C.byref(h)):
            raise RuntimeError(L_.xm_last_error().decode())
    G = api.ReferenceDatabase(refs, host_only=True, cache_dir=tmp_path)
    assert not G.cache_hit
    _same_index(B, G)
    # truncated file
    data = open(A.cache_file, "rb").read()
    open(A.cache_file, "wb").write(data[: len(data) // 2])
    with pytest.raises(RuntimeError, match="truncated|corrupt"):
        api.ReferenceDatabase.load(A.cache_file, host_only=True)
    H = api.ReferenceDatabase(refs, host_only=True, cache_dir=tmp_path)
    assert not H.cache_hit
    _same_index(B, H)
    # a complete file whose tables are damaged: every offset and position the kernels would index with is bounds-checked on load
    good = open(H.cache_file, "rb").read()
    for frac in (0.35, 0.6, 0.8, 0.95):
        at = int(len(good) * frac) & ~7
        open(A.cache_file, "wb").write(good[:at] + b"\xff" * 64 + good[at + 64:])
        with pytest.raises(RuntimeError, match="corrupt|another reference"):
            api.ReferenceDatabase.load(A.cache_file, host_only=True)
    open(A.cache_file, "wb").write(good)
    I = api.ReferenceDatabase.load(A.cache_file, host_only=True)
    _same_index(B, I)
    I.close()
    for x in (A, B, C2, D, E, F, G, H):
        x.close()


@pytest.mark.parametrize("mode,n", [("mapper", 60_000), ("api", 4000)])
def test_index_builder_with_ambiguous_reference_matches_oracle(mode, n):
    """Reference with N runs and IUPAC codes: the tables the host builder makes from its multi-block restatement (possibilities of every
    block over an ambiguous base, duplicate suppression of PackedMap.add) equal the oracle's literal HashBlock_Database."""
    refs = [("amb", ambiguous_reference(n, seed=0xA3B))]
    R = o.OracleReference(refs, mode=mode)
    R.align(["ACGTACGTACGTAGCATCGACTAGCAGCATCGAC"], o.make_params())
    P = api.ReferenceDatabase(refs, mode=mode, host_only=True)
    mn, mx = R.index_info()
    info = P.info()
    assert info["min_interesting_size"] == mn and info["max_hashed_length"] == mx
    for L in range(0, mx + 1):
        ta, tb = R.table(L), P.table(L)
        assert ta["capacity"] == tb["capacity"] and ta["maxCount"] == tb["maxCount"], L
        assert np.array_equal(ta["counts"], tb["counts"]), L
        assert np.array_equal(ta["positions"], tb["positions"]), L
    assert np.array_equal(R.dup_keys(0), P.dup_keys(0))
    P.close()


@pytest.mark.parametrize("case", ["scattered", "dense", "n_runs", "edges"])
def test_hybrid_build_composition_equals_whole_contig_multi_builder(case, monkeypatch):
    """References with ambiguity codes on the GPU build (xm_index_device.hip): blocks clear of the ambiguous bases by the plain rule + the multi
    blocks of windows around the ambiguous bases (HostIndex::multiRecordsNearAmbiguity, windows cut into pieces).  The same composition run on
    the host (XM_BUILD_HYBRID_ON_HOST=1) must give the tables and duplication keys of the whole-contig multi builder
    (HashBlock_ParentRow.java:69-191), which the oracle pins."""
    rng = np.random.default_rng(5)
    if case == "scattered":
        contigs = [("a", ambiguous_reference(60_000, seed=21, n_runs=10, n_codes=80)), ("b", ambiguous_reference(9_000, seed=22))]
    elif case == "dense":  # ambiguous bases closer together than a window margin: clusters merge
        r = synth.synthetic_reference(30_000, seed=23).copy()
        r[rng.integers(0, 30_000, size=2500)] = 15
        contigs = [("a", r), ("clean", synth.synthetic_reference(8_000, seed=24))]
    elif case == "n_runs":  # runs longer than a piece of a window (65,536 positions) and runs of a few hundred
        r = synth.synthetic_reference(400_000, seed=25).copy()
        r[50_000:190_000] = 15
        r[250_000:250_700] = 15
        r[300_000:300_003] = 15
        r[320_000:323_000] = 15
        r[323_050] = 15      # a long run with other ambiguous bases in its cluster
        r[319_990] = 5
        r2 = synth.synthetic_reference(30_000, seed=27).copy()
        r2[:9_000] = 15      # a long run at the start of a contig, one at the end
        r2[-5_000:] = 15
        contigs = [("a", r), ("b", r2)]
    else:  # ambiguous bases at both ends of a contig and a contig that is nothing but N
        r = synth.synthetic_reference(20_000, seed=26).copy()
        r[:40] = 15
        r[-25:] = 15
        r[7000] = 5
        contigs = [("a", r), ("n", np.full(300, 15, np.uint8))]
    monkeypatch.setenv("XM_DEVICE_BUILD", "0")
    for mis in (-1, 13):  # minInterestingSize from the reference's size (7 here), and the value a 3 Gb reference gets (HashBlock_Database.java:52)
        monkeypatch.delenv("XM_BUILD_HYBRID_ON_HOST", raising=False)
        monkeypatch.delenv("XM_BUILD_SPLICE_MIN", raising=False)
        A = api.ReferenceDatabase(contigs, host_only=True, max_query_length=150, min_interesting_size=mis)
        monkeypatch.setenv("XM_BUILD_HYBRID_ON_HOST", "1")
        B = api.ReferenceDatabase(contigs, host_only=True, max_query_length=150, min_interesting_size=mis)
        _same_index(A, B)
        # runs of N longer than XM_BUILD_SPLICE_MIN (default 2048; here also 256: the 700-run and the 3 000-run split too) are not held whole (their middle is left out of the window; with minInterestingSize 7 the middle
        # of a run emits records of its own, and the window is taken whole after all): same tables again
        monkeypatch.setenv("XM_BUILD_SPLICE_MIN", "256")
        C2 = api.ReferenceDatabase(contigs, host_only=True, max_query_length=150, min_interesting_size=mis)
        _same_index(A, C2)
        A.close(); B.close(); C2.close()


def test_kernel_logic_with_ambiguous_reference():
    """Reads (plain and with ambiguity codes of their own) against a reference with N runs and IUPAC codes: kernel logic vs oracle."""
    ref = ambiguous_reference(200_000, seed=0xA3C, n_runs=60, n_codes=600)
    reads = synth.synthetic_single_end(ref, 4000, seed=51)[0]
    reads[2000:] = sprinkle_ambiguity(reads[2000:], 6)
    R = o.OracleReference([("amb", ref)])
    S = hs.SimReference([("amb", ref)])
    b = se_batch(reads)
    want = R.align(b, o.make_params(), threads=os.cpu_count())
    got = S.align(b, o.make_params())
    assert streams_equal(got, want), first_difference(got, want, len(reads))


@pytest.mark.parametrize("case", KAT["align_cases"], ids=lambda c: c["name"])
def test_kernel_logic_on_reference_kats(case):
    """The reference's own AlignerWorker_Test cases through the kernel sources (host-simulated): expectations hold and the
    result streams are bit-identical to the oracle's."""
    R = o.OracleReference([("reference-0", case["reference"])], mode="api")
    S = hs.SimReference([("reference-0", case["reference"])], mode="api")
    q = [(case["mates"], case["expectedInner"], case["deviation"])]
    p = o.make_params(case["params"])
    sa, sb = R.align(q, p), S.align(q, p)
    assert streams_equal(sa, sb), first_difference(sa, sb, 1)
    check_align_case(case, api.decode_streams(sb.ints, sb.dbls, sb.int_off, sb.dbl_off, 0), o.encode(case["reference"]))


def test_kernel_logic_single_end_synthetic():
    ref = synth.synthetic_reference(200_000)
    reads, starts, strand = synth.synthetic_single_end(ref, 3000)
    b = se_batch(reads)
    R = o.OracleReference([("ecoli_syn", ref)])
    S = hs.SimReference([("ecoli_syn", ref)])
    p = o.make_params()
    sa, sb = R.align(b, p), S.align(b, p)
    assert streams_equal(sa, sb), first_difference(sa, sb, len(reads))
    # sanity of the workload itself: nearly every synthetic read maps back to where it came from
    ok = 0
    for q in range(len(reads)):
        comps = sa.query(q)
        if comps[0]:
            s0 = comps[0][0]["sequences"][0]
            if abs((s0["blocks"][0][1] - s0["blocks"][0][0]) - int(starts[q])) <= 3 and s0["referenceReversed"] == int(strand[q]):
                ok += 1
    assert ok >= 0.99 * len(reads)


def test_kernel_logic_paired_end_synthetic():
    ref = synth.synthetic_reference(300_000)
    m1, m2, _, _, _ = synth.synthetic_paired_end(ref, 1500)
    b = pe_batch(m1, m2)
    R = o.OracleReference([("ecoli_syn", ref)])
    S = hs.SimReference([("ecoli_syn", ref)])
    p = o.make_params()
    sa, sb = R.align(b, p), S.align(b, p)
    assert streams_equal(sa, sb), first_difference(sa, sb, len(m1))


@pytest.mark.parametrize("sub,indel,most_reruns", [(0.02, 0.002, 1), (0.05, 0.05, 1)], ids=["mild", "as_stated"])
def test_kernel_logic_long_reads_in_the_product_pass_sequence(sub, indel, most_reruns):
    """1 kb pieces of 10 kb reads (BASELINE.json configs[4], cut as the command line cuts them) through the pass sequence the product runs for a batch
    of long reads - light pass at scale 4, gapped chain at 16 with the long-read capacities (applyChainCaps) on temporaries of the product's size -
    against the oracle; and nearly all of them without outgrowing a capacity (a read that does is run again from its start, in a pass that lasts
    as long as its slowest read: with the capacities of round 2 every one of the as-stated reads and 1-2 % of the mild ones did)."""
    from mapper_amd import cli
    ref = synth.synthetic_reference(1_000_000, seed=0xEC011)
    n_reads = 8
    starts = (synth.splitmix64(0x5EED0004, n_reads) % np.uint64(len(ref) - 12_600)).astype(np.int64)
    strand = (synth.splitmix64(0x5EED0004 ^ 0x57A, n_reads) >> np.uint64(63)).astype(np.uint8)
    reads = synth.synthetic_long_reads(ref, starts, 10_000, seed=0x5EED0004, sub_rate=sub, indel_rate=indel, strand=strand)
    b = o.QueryBatch([([r[a_:b_].copy()], 0.0, 1.0) for r in reads for a_, b_ in cli.split_sections(10_000, 1000)])
    R = o.OracleReference([("r", ref)])
    S = hs.SimReference([("r", ref)])
    p = o.make_params()
    with o.observe_bound():  # the oracle also evaluates the bound of the product's rejection filter beside every search (and raises if it is not a bound)
        sa = R.align(b, p, threads=os.cpu_count())
    sb = S.align(b, p)
    assert streams_equal(sa, sb), first_difference(sa, sb, b.nq)
    assert sb.counters[11] <= most_reruns, "reads run again at a larger scale: %d of %d" % (sb.counters[11], b.nq)
    # the rejection filter (xm_bound.h) took and rejected exactly the searches and pieces the oracle's observer does; the PathAligner calls and nodes the product did not
    # make are the ones the reference spent in rejected searches and inside rejected pieces
    ok, what = filter_counters(sb.counters, sb.extra, sa.counters)
    assert sb.extra[3] == 1 and ok, what
    ref = what["reference"]
    assert what["oracle_observer"]["searches_examined"] + ref["calls_in_rejected_pieces"] > 0.5 * ref["path_aligner_calls"]
    if indel == 0.05:  # reads that do not align: most of the search's work is in searches that return null, and the filter proves most of it unnecessary
        assert ref["nodes_in_rejected_searches"] + ref["nodes_in_rejected_pieces"] > 0.75 * ref["nodes"], what
        assert what["oracle_observer"]["pieces_rejected"] > 0.5 * what["oracle_observer"]["pieces_examined"] > 0, what


def test_kernel_logic_rejection_filter_decides_like_the_oracle_observer():
    """The rejection filter in front of PathAligner (xm_bound.h, compiled for the host) on random problems: it takes and rejects exactly the searches the
    oracle's observer of the same bound does - the observer runs the reference's search beside its bound and raises when a search the bound rejects
    returns an alignment, so every problem also checks that the bound IS one (searches that align, fail narrowly, fail by far; windows shorter than the
    query, at the ends of the reference, long windows, wide bands, ambiguity codes, prices off the filter's grid)."""
    seen = {}
    for prm, q, rc, sa, ea, ref, sb, eb, off in bound_problems(0xB07D, 2500):
        p = o.make_params(prm)
        verdict, found, _ = o.kat_bound(p, q, rc, sa, ea, ref, sb, eb, off)
        taken, rejected, cells = hs.test_bound(p, q, rc, sa, ea, ref, sb, eb, off)
        assert (taken, rejected) == (1 if verdict else 0, 1 if verdict == 2 else 0), (prm, rc, sa, ea, sb, eb, off, verdict, found, taken, rejected)
        assert not (rejected and found)
        seen[(verdict, found)] = seen.get((verdict, found), 0) + 1
    assert seen.get((2, 0), 0) > 400 and seen.get((1, 1), 0) > 400 and seen.get((0, 0), 0) + seen.get((0, 1), 0) > 200, seen


def test_kernel_logic_rejection_filter_fuzz_through_the_whole_chain(monkeypatch):
    """scripts/cpu_filter_fuzz.py (profiles/r06/fuzz_filter_sim.log has 200 batches of it): batches of long reads with random lengths, error rates, ambiguity codes and prices
    through the kernel sources with the filter on against the oracle with its observer on - streams bit for bit, the filter's counters equal the observer's batch by batch."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("cpu_filter_fuzz", os.path.join(ROOT, "scripts", "cpu_filter_fuzz.py"))
    fuzz = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(fuzz)
    monkeypatch.setattr("sys.argv", ["cpu_filter_fuzz.py", "10", "61"])
    fuzz.main()


def test_kernel_logic_edge_cases():
    """ragged / tiny / unalignable reads, reads hanging over contig ends, repeats, multi-contig reference, long reads."""
    rng = np.random.default_rng(3)
    c0 = synth.synthetic_reference(60_000, seed=11)
    c1 = synth.synthetic_reference(9_000, seed=12)
    rep = np.tile(synth.synthetic_reference(500, seed=13), 8)   # tandem repeat contig
    contigs = api.sort_reference([("c0", c0), ("c1", c1), ("rep", rep)])
    queries = []
    for L in (1, 2, 7, 12, 20, 33, 75, 151, 400, 1000):
        s = int(rng.integers(0, len(c0) - L))
        queries.append(([c0[s:s + L].copy()], 0.0, 1.0))
    queries.append(([np.concatenate([c1[-40:], np.array([1, 2, 4, 8] * 10, dtype=np.uint8)])], 0.0, 1.0))  # hangs over a contig end
    queries.append(([np.concatenate([np.array([8, 4, 2, 1] * 8, dtype=np.uint8), c1[:60]])], 0.0, 1.0))      # hangs over a contig start
    queries.append(([rep[100:250].copy()], 0.0, 1.0))                                                          # many equally good placements
    queries.append(([np.full(150, 1, dtype=np.uint8)], 0.0, 1.0))                                              # poly-A, matches nowhere
    queries.append(([np.array([1, 2, 4, 8], dtype=np.uint8)[rng.integers(0, 4, 150)]], 0.0, 1.0))              # random, unalignable
    queries.append(([c0[5000:5150].copy(), api.reverse_complement(c0[5300:5450])], 100.0, 50.0))               # proper pair
    queries.append(([c0[5000:5150].copy(), api.reverse_complement(c1[300:450])], 100.0, 50.0))                 # mates on different contigs
    queries.append(([c0[5000:5150].copy(), api.reverse_complement(c0[5100:5250])], 100.0, 50.0))               # overlapping mates
    b = o.QueryBatch(queries)
    R = o.OracleReference(contigs)
    S = hs.SimReference(contigs)
    for params in (o.make_params(), o.make_params(MaxNumMatches=3), o.make_params(Max_PenaltySpan=2.0, MaxErrorRate=0.15)):
        sa, sb = R.align(b, params), S.align(b, params)
        assert streams_equal(sa, sb), first_difference(sa, sb, len(queries))


def test_committed_golden_digests_on_cpu():
    """tests/golden/synthetic_golden.json (digests of the oracle's result streams on seeded batches, made by
    tests/golden/make_synthetic_golden.py): the oracle still produces them, and so does the kernel logic in the host simulation — in the
    product's pass sequence and with every PathAligner search deferred to the search "kernel"."""
    import json
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden"))
    from make_synthetic_golden import cases, digest
    golden = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "synthetic_golden.json")))
    ref, batches = cases()
    R = o.OracleReference([("ecoli_syn", ref)])
    S = hs.SimReference([("ecoli_syn", ref)])
    for name, b in batches.items():
        want = golden["digests"][name]["sha256"]
        assert digest(R.align(b, o.make_params(), threads=os.cpu_count())) == want, "oracle: " + name
        assert digest(S.align(b, o.make_params())) == want, "host simulation: " + name


def test_kernel_logic_reads_with_ambiguous_bases():
    """Read-side MultiHashBlock path (blocks over an ambiguous base are lists of conditional possibilities): the kernel logic against the
    oracle's literal restatement of HashBlock_BaseRow / HashBlock_ParentRow / SequenceCondition, single-end and paired."""
    ref = synth.synthetic_reference(300_000, seed=41)
    reads = sprinkle_ambiguity(synth.synthetic_single_end(ref, 5000, seed=42)[0])
    m1, m2 = synth.synthetic_paired_end(ref, 1000, seed=43)[:2]
    m1, m2 = sprinkle_ambiguity(m1, 4), sprinkle_ambiguity(m2, 5)
    R = o.OracleReference([("r", ref)])
    S = hs.SimReference([("r", ref)])
    for b, n in ((se_batch(reads), 5000), (pe_batch(m1, m2, 100.0, 50.0), 1000)):
        want = R.align(b, o.make_params(), threads=os.cpu_count())
        got = S.align(b, o.make_params())
        assert streams_equal(got, want), first_difference(got, want, n)


@pytest.mark.parametrize("trial", range(6))
def test_read_pyramid_with_any_number_of_ambiguous_bases_matches_oracle(trial):
    """The multi blocks of a read's pyramid, possibility by possibility (block, has-block, condition), against the oracle's HashBlock_ParentRow for reads of
    5-150 bases with 1 % ... 100 % of their positions ambiguous (N or IUPAC codes): every level, not only the ones a walk asks for.  The product's pools follow
    the scratch scale; what a scale cannot hold must be reported (None) and fit a larger one."""
    rng = np.random.default_rng(100 + trial)
    codes = np.array([15, 15, 15, 5, 10, 3, 12, 7, 14, 6, 9, 11, 13], np.uint8)
    for k in range(25):
        L = int(rng.integers(5, 151))
        r = np.array([1, 2, 4, 8], np.uint8)[rng.integers(0, 4, L)]
        m = rng.random(L) < [0.01, 0.1, 0.5, 0.9, 1.0][k % 5]
        r[m] = 15 if k % 2 else codes[rng.integers(0, len(codes), int(m.sum()))]
        want = o.pyramid_dump_multi(r)
        got = None
        for scale in (1, 4, 16, 64, 256):
            got = hs.pyramid_dump_multi(r, scale)
            if got is not None:
                break
        assert got is not None and got.shape == want.shape and np.array_equal(got, want), (trial, k, L)


def test_kernel_logic_reads_of_mostly_ambiguous_bases():
    """Reads the reference takes and earlier rounds of this product refused (over 128 ambiguous bases in a mate failed the batch): N and IUPAC codes at 1 % ... 100 %
    of the positions, N runs at either end up to the whole read, single reads, pairs with one or both mates affected, reads shorter than minInterestingSize,
    homopolymers and short-period reads - the kernel logic in the product's pass sequence against the oracle, bit for bit."""
    ref = synth.synthetic_reference(300_000, seed=41)
    R = o.OracleReference([("r", ref)])
    S = hs.SimReference([("r", ref)])
    p = o.make_params()
    reads = heavy_ambiguity(synth.synthetic_single_end(ref, 480, seed=42)[0])
    m1, m2 = synth.synthetic_paired_end(ref, 320, seed=43)[:2]
    m1h, m2h = heavy_ambiguity(m1, 4), heavy_ambiguity(m2, 5)
    m2one = m2.copy(); m2one[::2] = 15                                      # every other pair: one mate of nothing but N
    short = [r[:int(L)] for r, L in zip(heavy_ambiguity(synth.synthetic_single_end(ref, 160, seed=44)[0], 6), np.tile([1, 2, 3, 5, 8, 11, 13, 20], 20))]
    low = low_complexity_reads(120, 150)
    batches = [("single", se_batch(reads)), ("pairs", pe_batch(m1h, m2h)), ("one all-N mate", pe_batch(m1, m2one)), ("short", ragged_se_batch(short)),
               ("low complexity", se_batch(low)), ("all N", se_batch(np.full((3, 150), 15, np.uint8))),
               ("135 N inside", se_batch(np.concatenate([reads[15:16, :10], np.full((1, 135), 15, np.uint8), reads[15:16, 145:]], axis=1)))]
    for name, b in batches:
        want = R.align(b, p, threads=os.cpu_count())
        got = S.align(b, p)
        assert streams_equal(got, want), name + ": " + str(first_difference(got, want, b.nq))
    all_n = R.align(batches[5][1], p)
    assert all(all_n.ints[all_n.int_off[q] + 1] == 0 for q in range(3))  # (the reference walks such a read and finds nothing)


def test_kernel_logic_on_a_repeat_rich_reference(monkeypatch):
    """synth.repeat_rich_reference (segmental duplications at 90-99.5 % identity, tandem repeats, a 28-mer whose buckets overflow): the branch of the
    path a real genome sends reads into - a read in a duplicated window gets no early accept (Readable_DuplicationDetector.java:28-47 via
    AlignerWorker.java:494-587), its candidates are all enumerated, overfull buckets are skipped (HashBlock_Database.java:569-577).  Index tables and
    duplication keys against the oracle's, then reads and pairs: streams and work counters."""
    st = {}
    ref = synth.repeat_rich_reference(600_000, n_segdups=14, n_tandem=30, n_hot=40, stats=st)
    assert st["fraction_in_repeats"] >= 0.3
    R = o.OracleReference([("rep", ref)])
    S = hs.SimReference([("rep", ref)])
    p = o.make_params()
    b = se_batch(synth.synthetic_single_end(ref, 3000, seed=91)[0])
    want = R.align(b, p, threads=os.cpu_count())
    m1, m2 = synth.synthetic_paired_end(ref, 1000, seed=92)[:2]
    pb = pe_batch(m1, m2)
    wantp = R.align(pb, p, threads=os.cpu_count())
    quick = want.counters[8] / b.nq
    assert 0.2 < quick < 0.85, "quick accepts on the repeat-rich reference: %.3f of the reads (i.i.d. reference: 0.94)" % quick
    assert want.counters[5] > 1.5 * b.nq, "candidates extended per read: %.2f" % (want.counters[5] / b.nq)
    got = S.align(b, p)
    assert streams_equal(got, want), first_difference(got, want, b.nq)
    assert list(want.counters[5:9]) == list(got.counters[4:8])  # candidates extended, PathAligner calls, nodes put, quick accepts
    gotp = S.align(pb, p)
    assert streams_equal(gotp, wantp), first_difference(gotp, wantp, pb.nq)


def test_kernel_logic_random_configurations():
    """The differential fuzz of the GPU tier (scripts/gpu_fuzz.py: random references with repeats and ambiguity codes, read lengths 36-301, single / paired
    mixes, random alignment parameters) through the host simulation: every batch must equal the oracle bit for bit.  Round 3 of this seed is a batch of pairs
    sampled from a reference with ambiguity codes that reach getUnpairedAlignments: the seeding state of both mates, their possibilities and three aligners in
    one region - the case that decides how much room the aligners may take (xm_worker.h, qmaInit)."""
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "scripts"))
    import gpu_fuzz
    assert gpu_fuzz.run(rounds=6, seed=77, max_queries=1500, backend="sim") == 0


def test_kernel_logic_random_shapes():
    """The second flavour of the fuzz (gpu_fuzz.run_shapes): several contigs with reads across their ends, a length per read inside one batch (long reads among
    them: chains at the long-read scales), mates of unequal length, pairs and single reads mixed - through the host simulation, equal to the oracle."""
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "scripts"))
    import gpu_fuzz
    assert gpu_fuzz.run_shapes(rounds=3, seed=99, max_queries=700, backend="sim") == 0


def test_kernel_logic_ambiguity_fuzz():
    """The third flavour of the fuzz (gpu_fuzz.run_ambiguity: per read an ambiguous fraction from {0, 1 %, 10 %, 50 %, 90 %, 100 %} as N / IUPAC codes, N runs at the ends,
    low-complexity and very short reads, single reads and pairs mixed) through the host simulation, equal to the oracle."""
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "scripts"))
    import gpu_fuzz
    assert gpu_fuzz.run_ambiguity(rounds=5, seed=31, max_queries=300, backend="sim") == 0


def test_kernel_logic_reads_no_arena_memory_it_has_not_written():
    """The host simulation built with -DXM_ARENA_POISON (every arena allocation filled with 0xA5 first: hostsim_lib.build, XMSIM_POISON=1) on reads, pairs and long
    reads: equal to the oracle, i.e. no structure in a lane's region or temporaries is read before it is written - on the
    GPU, where a lane's arena holds what the lane's previous read left, such a read would make a result depend on the lane's history (profiles/r04/NOTES.md 14)."""
    import subprocess, sys
    code = r'''
import os, sys
sys.path.insert(0, %r); sys.path.insert(0, os.path.join(%r, "tests"))
import numpy as np
import oracle_lib as o, hostsim_lib
from helpers import se_batch, pe_batch, streams_equal
from mapper_amd import synth
assert hostsim_lib.build().endswith("_poison.so")
ref = synth.synthetic_reference(400_000, seed=0xEC011)
R = o.OracleReference([("e", ref)])
if True:
    S = hostsim_lib.SimReference([("e", ref)])
    b = se_batch(synth.synthetic_single_end(ref, 4000, read_len=150, seed=11, sub_rate=0.02, indel_prob=0.4)[0])
    assert streams_equal(S.align(b, o.make_params()), R.align(b, o.make_params(), threads=8))
    m1, m2 = synth.synthetic_paired_end(ref, 1500, read_len=150, seed=12, sub_rate=0.02, indel_prob=0.3)[:2]
    pb = pe_batch(m1, m2)
    assert streams_equal(S.align(pb, o.make_params()), R.align(pb, o.make_params(), threads=8))
    lb = se_batch(synth.synthetic_single_end(ref, 120, read_len=1000, seed=13, sub_rate=0.02, indel_prob=0.5)[0])
    assert streams_equal(S.align(lb, o.make_params()), R.align(lb, o.make_params(), threads=8))
print("poison ok")
''' % (ROOT, ROOT)
    env = dict(os.environ, XMSIM_POISON="1")
    out = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=1500)
    assert out.returncode == 0 and "poison ok" in out.stdout, out.stdout[-2000:] + out.stderr[-4000:]
