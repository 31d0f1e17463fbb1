"""MI355X-native implementation of ScaNeRF's per-tile volume-rendering hot path.

Layout (mirrors the reference's operator surface for this path, nothing else):

  cuda/            package `cuda` of the reference  -> lib/CUDA_EXT names   (cuda/binding.cpp:10-54)
  hashgrid/        package `hashgrid`               -> lib/HASHGRID names   (hashgrid/binding.cpp:9-44)
  render.py        fused fast path (one launch per render_batch_rays)
  network.py       ShallowMLP mirror + decoder-blob packing
  consensus.py     ADMM camera consensus as an RCCL all-reduce
  csrc/            HIP kernels + the C ABI (include/scanerf_hip.h) -> lib/libscanerf_hip.so

The directory name is not a Python identifier; import it through the `scanerf_amd`
alias module at the repository root.
"""
from . import _capi  # noqa: F401

__all__ = ["_capi"]
