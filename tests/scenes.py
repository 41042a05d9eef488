"""Scenes of the parity tests: the workload definitions live in the package (volume_renderer_amd/workloads.py),
so that bench.py does not import from tests/."""
from volume_renderer_amd.workloads import *  # noqa: F401,F403
from volume_renderer_amd.workloads import SEED, BACKGROUND_SKY  # noqa: F401
