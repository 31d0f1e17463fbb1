"""Package `hashgrid` of the reference: the HASHGRID op names plus the encoder modules."""
from .lib.HASHGRID import (SKIP_UNSAMPLED, Sampler, accumulate_color, bg_pts_inference, bg_pts_inference_v2,  # noqa: F401
                           embedding_backward_cuda, embedding_bg_backward_cuda, embedding_bg_forward_cuda,
                           embedding_forward_cuda, get_last_block, inverse_z_sampling, prepare_points,
                           process_occupied_grid, pts_inference, pts_inference_tracing, ray_block_intersection, ray_firsthit_block,
                           rendering_cuda, sample_points, sort_by_key, sort_tracing_blocks, tracing_fusable, update_outgoing_bidx,
                           update_outgoing_bidx_v2)
from .PyHashGrid import HashEmbedding, PyHashGrid  # noqa: F401
from .PyHashGridBG import HashEmbeddingBG, PyHashGridBG, level_resolutions  # noqa: F401
from .grid import INFERENCE, TRAIN, HashGrid  # noqa: F401,E402  (the reference defines HashGrid in hashgrid/__init__.py:32)
