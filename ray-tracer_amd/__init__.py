"""MI355X-native trace core: host bindings for librt_mi355.so (see include/rt_mi355.h).

Importing the package does not load the shared library; the first use does, and raises if it
has not been built - there is no CPU fallback.
"""
from ._lib import (BOX_DTYPE, HIT_DTYPE, MATERIAL_DTYPE, RT_COMM_ID_BYTES, RT_ERR_COMM, RT_ERR_HIP, RT_ERR_INVALID, RT_ERR_LIMIT, RT_ERR_NO_DEVICE,  # noqa: F401
                   RT_ERR_STATE, RT_GRID_HIERARCHICAL, RT_GRID_TRIANGLES, RT_MISS, RT_NOALLOC_MAX_BOXES, RT_OWNER_SKY, TRIANGLE_DTYPE,
                   CameraUniform, GridMetadata, RayDebugInfo, RtParams, RtShard, RtStats)
from .host import (DEFAULT_FOV, Camera, RayTracer, RtError, TraceCore, default_params, shard_pixels,  # noqa: F401
                   shared_stream, triangles_from_xyz9)
