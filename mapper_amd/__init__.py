"""mapper_amd: MI355X-native drop-in for X-Mapper's per-read seed-and-extend path.

`from mapper_amd import api` loads libxmapper_hip.so; there is no CPU fallback.  `mapper_amd.synth` (synthetic inputs)
and `mapper_amd.sam` (SAM text of the result types) are pure-Python helpers.
"""
import os as _os

# Before anything in the process starts the HIP runtime: contexts beyond two per GPU need hardware queues of their own (mapper_amd/_capi.py has the measurement).
_os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
