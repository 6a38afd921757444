import os, sys, time
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
import numpy as np, ctypes as C
mode = sys.argv[1] if len(sys.argv) > 1 else "2"
os.environ["XMSIM_WAVE"] = mode
import oracle_lib as o, hostsim_lib as hs
from helpers import streams_equal, first_difference, se_batch, pe_batch
from mapper_amd import synth
refLen = int(sys.argv[2]) if len(sys.argv) > 2 else 300000
nq = int(sys.argv[3]) if len(sys.argv) > 3 else 2000
kind = sys.argv[4] if len(sys.argv) > 4 else "se"
ref = synth.synthetic_reference(refLen, seed=0xEC011)
if kind == "se":
    reads, starts, strand = synth.synthetic_single_end(ref, nq, seed=0x5EED0001)
    batch = se_batch(reads)
else:
    m1, m2, _, _, _ = synth.synthetic_paired_end(ref, nq, seed=0x5EED0002)
    batch = pe_batch(m1, m2)
t=time.time(); R = o.OracleReference([("ecoli_syn", ref)], mode="mapper"); want = R.align(batch, o.make_params()); print("oracle %.1fs" % (time.time()-t))
t=time.time(); S = hs.SimReference([("ecoli_syn", ref)], mode="mapper"); got = S.align(batch, o.make_params()); print("sim %.1fs" % (time.time()-t))
print("wave status counts", list(hs.wave_status_counts())); w=(C.c_longlong*64)(); hs.lib().xmsim_wave_why_counts(w); print("why", {i:w[i] for i in range(64) if w[i]})
print("equal:", streams_equal(got, want))
if not streams_equal(got, want): print(first_difference(got, want, nq))
print("counters sim   ", list(got.counters[:11])); print("counters oracle", list(want.counters[:11]) if hasattr(want,'counters') and want.counters is not None else None)
