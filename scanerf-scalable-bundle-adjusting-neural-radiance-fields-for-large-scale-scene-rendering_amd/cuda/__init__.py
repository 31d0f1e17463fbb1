"""Package `cuda` of the reference (cuda/__init__.py:2-23): re-exports the hot-path ops of CUDA_EXT."""
from .lib.CUDA_EXT import (adam_step_cuda, adam_step_cuda_fp16, background_sampling_cuda,  # noqa: F401
                           compute_ray_backward, compute_ray_forward, ray_aabb_intersection,
                           ray_aabb_intersection_v2, sample_insideout_block, sample_points_contract,
                           sample_points_grid, voxelize_mesh)
