"""mapper_amd: MI355X-native drop-in for X-Mapper's per-read seed-and-extend path.

`from mapper_amd import api` loads libxmapper_hip.so; there is no CPU fallback.  `mapper_amd.synth` (synthetic inputs)
and `mapper_amd.sam` (SAM text of the result types) are pure-Python helpers.
"""
