"""folddisco_amd — MI355X (gfx950) implementation of Folddisco's geometric-hash index-build and
motif-query hot path behind a C ABI (include/fdgpu.h, libfdgpu.so).  See DESIGN.md."""
from .api import (Batch, Context, FdgpuError, FolddiscoIndex, FolddiscoIndexSet, PackedStructures, count_query,  # noqa: F401
                  count_query_batch, count_query_maps, count_query_set, get_geometric_hash_as_u32, idf_of_lengths, length_penalty)
