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
