"""Package `hashgrid` of the reference: the HASHGRID op names plus the encoder modules."""
from .lib.HASHGRID import (Sampler, embedding_backward_cuda, embedding_bg_backward_cuda,  # noqa: F401
                           embedding_bg_forward_cuda, embedding_forward_cuda, rendering_cuda)
from .PyHashGrid import HashEmbedding, PyHashGrid  # noqa: F401
from .PyHashGridBG import HashEmbeddingBG, PyHashGridBG, level_resolutions  # noqa: F401
