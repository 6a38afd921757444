"""The in-process multi-GPU driver (mapper_amd/multi.py; SURVEY.md section 8e: index replicated, batches k mod N, results in batch order).
CPU tier: the dealing and ordering logic over stand-in replicas; GPU tier: two contexts on one device (xm_index_replicate + two host threads
+ their streams) return exactly the single-GPU streams, through the API and through the command line."""
import io
import os
import threading
import time
import numpy as np
import pytest

from helpers import se_batch, pe_batch, streams_equal
from mapper_amd import api, multi, synth


class FakeReplica:
    """Stands in for ReferenceDatabase.align_stream: 'aligns' a batch by returning (replica id, batch payload) after a delay that depends on
    the replica, so that the replicas finish out of order."""
    def __init__(self, gid, delay):
        self.gid, self.delay, self.seen = gid, delay, []

    def align_stream(self, batches, parameters):
        for b in batches:
            time.sleep(self.delay)
            self.seen.append(b)
            yield (self.gid, b)

    def close(self):
        pass


def fake_db(delays):
    db = multi.MultiGpuDatabase.__new__(multi.MultiGpuDatabase)
    db.devices = list(range(len(delays)))
    db.replicas = [FakeReplica(g, d) for g, d in enumerate(delays)]
    db.contigs = []
    return db


def test_batches_are_dealt_round_robin_and_come_back_in_order():
    db = fake_db([0.02, 0.0, 0.01])
    out = list(db.align_stream(iter(range(20)), None))
    assert [b for _, b in out] == list(range(20))                 # batch order
    assert [g for g, _ in out] == [k % 3 for k in range(20)]      # batch k was aligned by replica k mod N
    assert db.replicas[1].seen == list(range(1, 20, 3))
    assert list(db.align_stream(iter([]), None)) == []


def test_a_failing_gpu_fails_the_stream():
    class Broken(FakeReplica):
        def align_stream(self, batches, parameters):
            for b in batches:
                if b == 4:
                    raise RuntimeError("Failed to align: device lost")
                yield (self.gid, b)
    db = fake_db([0.0, 0.0])
    db.replicas[0] = Broken(0, 0.0)
    got = []
    with pytest.raises(RuntimeError, match="device lost"):
        for r in db.align_stream(iter(range(10)), None):
            got.append(r[1])
    assert got == [0, 1, 2, 3]
    assert threading.active_count() < 20


@pytest.mark.gpu
def test_two_contexts_equal_one_gpu():
    ref = synth.synthetic_reference(400_000, seed=0xEC011)
    reads = synth.synthetic_single_end(ref, 9000, seed=31)[0]
    m1, m2 = synth.synthetic_paired_end(ref, 2000, seed=32)[:2]
    batches = [se_batch(reads[i:i + 1500]) for i in range(0, 9000, 1500)] + [pe_batch(m1[i:i + 500], m2[i:i + 500]) for i in range(0, 2000, 500)]
    arrays = [(b.mate_count, b.mate_offset, b.mate_length, b.codes, b.expected_inner, b.deviation) for b in batches]
    params = api.AlignmentParameters()
    one = api.ReferenceDatabase([("r", ref)], device=0)
    want = [one.align_arrays(*a, params) for a in arrays]
    two = multi.MultiGpuDatabase([("r", ref)], [0, 0])
    assert two.replicas[1].info()["index_bytes"] == two.replicas[0].info()["index_bytes"] and two.replicas[1].info()["num_positions"] == two.replicas[0].info()["num_positions"]
    got = list(two.align_stream(iter(arrays), params))
    assert len(got) == len(want)
    for g, w in zip(got, want):
        assert streams_equal(g, w)
    # the replica is a full index of its own: it outlives the one it was copied from, and grows its tables on demand
    two.replicas[0].close()
    long_reads = synth.synthetic_single_end(ref, 64, read_len=400, seed=33)[0]
    lb = se_batch(long_reads)
    a = two.replicas[1].align_arrays(lb.mate_count, lb.mate_offset, lb.mate_length, lb.codes, lb.expected_inner, lb.deviation, params)
    b = one.align_arrays(lb.mate_count, lb.mate_offset, lb.mate_length, lb.codes, lb.expected_inner, lb.deviation, params)
    assert streams_equal(a, b)
    two.close()
    one.close()


@pytest.mark.gpu
def test_cli_gpus_flag(tmp_path):
    from mapper_amd import cli
    ref = synth.synthetic_reference(120_000, seed=5)
    reads = synth.synthetic_single_end(ref, 600, seed=6)[0]
    with open(tmp_path / "ref.fasta", "w") as f:
        f.write(">chrSyn\n" + api.decode(ref) + "\n")
    with open(tmp_path / "reads.fastq", "w") as f:
        for i, r in enumerate(reads):
            f.write("@r%d\n%s\n+\n%s\n" % (i, api.decode(r), "I" * len(r)))
    outs = []
    for extra in ([], ["--devices", "0,0", "--batch-size", "100"], ["--contexts", "3", "--batch-size", "64"]):
        sam_path = tmp_path / ("out%d.sam" % len(outs))
        buf = io.StringIO()
        assert cli.run(["--reference", str(tmp_path / "ref.fasta"), "--queries", str(tmp_path / "reads.fastq"), "--out-sam", str(sam_path)] + extra, out=buf) == 0
        outs.append((open(sam_path).read(), buf.getvalue()))
    assert outs[0] == outs[1] == outs[2] and outs[0][0].count("\n") > 600


@pytest.mark.gpu
def test_contexts_share_one_index():
    """xm_context_new (SURVEY.md section 8(b) "Threading"; HashBlock_Database.java:129-133, Mapper.java:1026-1040): contexts of one index read the same
    tables - no second copy in host memory or in HBM - align at the same time from their own host threads with the results one context gives alone,
    see the tables any of them grows, and are freed in any order."""
    import psutil
    ref = synth.synthetic_reference(40_000_000, seed=0xC0)
    first = api.ReferenceDatabase([("r", ref)], device=0, max_query_length=150)
    index_bytes = first.info()["index_bytes"]
    assert index_bytes > 300 << 20
    proc = psutil.Process()
    free0, rss0 = api.device_memory(0)[0], proc.memory_info().rss
    ctx = [first] + [first.new_context() for _ in range(3)]
    free1, rss1 = api.device_memory(0)[0], proc.memory_info().rss
    # (three more streams and their events cost the HIP runtime some tens of MB of host memory; a copy of the tables would be 3 x 330 MB)
    assert free0 - free1 < index_bytes // 4, "a context must not copy the tables in HBM"
    assert rss1 - rss0 < index_bytes // 2, "a context must not copy the host tables"
    params = api.AlignmentParameters()
    reads = synth.synthetic_single_end(ref, 40_000, seed=0xC1, indel_prob=0.3)[0]
    parts = [se_batch(reads[i * 10_000:(i + 1) * 10_000]) for i in range(4)]
    arrays = [(b.mate_count, b.mate_offset, b.mate_length, b.codes, b.expected_inner, b.deviation) for b in parts]
    want = [first.align_arrays(*a, params) for a in arrays]
    n, share = api.divide_scratch(ctx, 0)
    assert n == 4 and share >= 8 << 30
    got = [None] * 4
    errors = []

    def work(i):
        try:
            for _ in range(3):
                got[i] = ctx[i].align_arrays(*arrays[i], params)
        except BaseException as e:  # noqa: BLE001
            errors.append(e)
    th = [threading.Thread(target=work, args=(i,)) for i in range(4)]
    [t.start() for t in th]
    [t.join() for t in th]
    assert not errors, errors
    for g, w in zip(got, want):
        assert streams_equal(g, w)
    # one context grows the tables (longer mates) while the others keep aligning: all of them see the grown tables afterwards
    long_reads = se_batch(synth.synthetic_single_end(ref, 300, read_len=400, seed=0xC2)[0])
    la = (long_reads.mate_count, long_reads.mate_offset, long_reads.mate_length, long_reads.codes, long_reads.expected_inner, long_reads.deviation)

    def grow():
        try:
            got[3] = ctx[3].align_arrays(*la, params)
        except BaseException as e:  # noqa: BLE001
            errors.append(e)
    th = [threading.Thread(target=work, args=(i,)) for i in range(3)] + [threading.Thread(target=grow)]
    [t.start() for t in th]
    [t.join() for t in th]
    assert not errors, errors
    for i in range(3):
        assert streams_equal(got[i], want[i])
    assert all(c.info()["max_hashed_length"] >= 400 for c in ctx)
    again = ctx[1].align_arrays(*la, params)
    assert streams_equal(again, got[3])
    # any order of release; the last handle keeps the tables alive
    ctx[0].close(); ctx[2].close()
    last = ctx[1].align_arrays(*arrays[1], params)
    assert streams_equal(last, want[1])
    ctx[1].close(); ctx[3].close()


def test_contexts_of_a_gpu_are_counted_with_their_pile_ups(monkeypatch):
    """api.divide_scratch: what every context allocates beside its scratch (--out-mutations: one pile-up per context, 40-48 bytes per reference base) comes
    out of the free HBM before the contexts are counted - two contexts on a 3.1 Gb reference would ask for 2 x 149 GB of pile-up - and the reserve is a
    share of what is free on a small or busy GPU, not a fixed claim."""
    class Ctx:
        def __init__(self):
            self.scratch = None

        def set_scratch(self, n):
            self.scratch = n
    GiB = 1 << 30
    monkeypatch.setattr(api, "device_memory", lambda d: (250 * GiB, 288 * GiB))
    a, b = Ctx(), Ctx()
    n, share = api.divide_scratch([a, b], 0)
    assert n == 2 and share == (250 - 24) * GiB // 2 and a.scratch == share and b.scratch == share
    pile_up = 48 * 3_088_269_832 + (64 << 20)
    a, b = Ctx(), Ctx()
    n, share = api.divide_scratch([a, b], 0, per_context_extra=pile_up)
    assert n == 1 and b.scratch == 0 and share == min(200 * GiB, 250 * GiB - 24 * GiB - pile_up)
    a, b = Ctx(), Ctx()
    n, share = api.divide_scratch([a, b], 0, per_context_extra=48 * 500_000_000)
    assert n == 2 and share == (250 * GiB - 24 * GiB - 2 * 48 * 500_000_000) // 2
    monkeypatch.setattr(api, "device_memory", lambda d: (10 * GiB, 288 * GiB))
    a, b = Ctx(), Ctx()
    n, share = api.divide_scratch([a, b], 0)
    assert n == 1 and share == 10 * GiB - (10 * GiB) // 4
