"""frankensearch_amd — MI355X (gfx950) semantic tier for frankensearch.

The product is the HIP library `libfsgpu.so` (C ABI in include/fsgpu.h); this package is the thin
host-side mirror of the reference's VectorIndex / embedder interfaces used by tests and bench.py.
"""
from . import _lib
from .embed import Model2VecEmbedder, NativeEmbedder
from .errors import (DeviceError, DimensionMismatch, IndexCorrupted, IndexVersionMismatch, InvalidConfig, IoError,
                     ModelLoadFailed, NoDevice, SearchError)
from .index import (ClassifiedHits, NativeShardedIndex, VectorHit, VectorIndex, encode_f32_to_f16, pack_bitmap, widen_f16_to_f32,
                    write_fsvi)

__all__ = ["write_fsvi", "VectorIndex", "NativeShardedIndex", "VectorHit", "ClassifiedHits", "Model2VecEmbedder", "NativeEmbedder", "SearchError", "DimensionMismatch",
           "InvalidConfig", "IndexCorrupted", "IndexVersionMismatch", "IoError", "DeviceError", "NoDevice", "ModelLoadFailed",
           "encode_f32_to_f16", "widen_f16_to_f32", "pack_bitmap", "_lib"]
