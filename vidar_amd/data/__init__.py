"""Sample assembly between a frame loader and the model (the data format on
the input side of the hot path).  No dataset reader lives here -- records are plain dicts."""
from .assemble import (frame_index_lists, frame_meta_from_info, transform_matrix, union2one,
                       usable_indices)  # noqa: F401
