"""Index build on the GPU vs the host builder: wall time at several reference sizes (scripts/README.md)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mapper_amd import api, synth
sizes = sys.argv[1:] or ["5000000", "50000000"]  # "N" = GPU, host, GPU;  "N:gpu" = GPU only
for spec in sizes:
    n = int(spec.split(":")[0])
    ref = synth.synthetic_reference(n, seed=0xEC011)
    for dev in (("1", "1") if spec.endswith(":gpu") else ("1", "0", "1")):
        os.environ["XM_DEVICE_BUILD"] = dev
        t = time.time()
        db = api.ReferenceDatabase([("r", ref)], max_query_length=150)
        dt = time.time() - t
        i = db.info()
        print("ref %d: %s build %.3f s (hash %.3f s, duplication map %.3f s), %d positions, index %d bytes" %
              (n, "GPU " if i["built_on_device"] else "host", dt, i["hash_seconds"], i["duplication_seconds"], i["num_positions"], i["index_bytes"]), flush=True)
        db.close()
