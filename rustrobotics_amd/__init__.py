"""rustrobotics_amd -- MI355X (gfx950) backend for RustRobotics' pose-graph optimization path.

Only what the hot path needs: `csrc/` (HIP kernels + the C ABI of librr_pgo.so)
and `mapping` (the host-side mirror of `robotics::mapping::{PoseGraph, PoseGraphSolver}`);
`sharding` drives ONE graph sharded over ranks (torch.distributed / RCCL plumbing).
"""
from .mapping import PoseGraph, PoseGraphSolver, PoseGraphError, synthetic_grid_arrays  # noqa: F401
