"""SearchError variants of the reference (crates/frankensearch-core/src/error.rs:57-176) as exceptions."""
from __future__ import annotations

from . import _lib


class SearchError(Exception):
    code = -1


class DimensionMismatch(SearchError):
    code = _lib.ERR_DIMENSION_MISMATCH


class InvalidConfig(SearchError):
    code = _lib.ERR_INVALID_CONFIG


class IndexCorrupted(SearchError):
    code = _lib.ERR_INDEX_CORRUPTED


class IndexVersionMismatch(SearchError):
    code = _lib.ERR_INDEX_VERSION_MISMATCH


class IoError(SearchError):
    code = _lib.ERR_IO


class DeviceError(SearchError):
    code = _lib.ERR_DEVICE


class NoDevice(SearchError):
    code = _lib.ERR_NO_DEVICE


class NullArgument(SearchError):
    code = _lib.ERR_NULL_ARGUMENT


class EmbeddingFailed(SearchError):
    code = _lib.ERR_EMBEDDING_FAILED


class ModelLoadFailed(SearchError):
    code = _lib.ERR_MODEL_LOAD_FAILED


_BY_CODE = {c.code: c for c in (DimensionMismatch, InvalidConfig, IndexCorrupted, IndexVersionMismatch, IoError,
                                DeviceError, NoDevice, NullArgument, EmbeddingFailed, ModelLoadFailed)}


def error_for(status: int, detail: str) -> SearchError:
    """The exception of a status code with a detail string the caller already holds (a call that failed on another thread)."""
    return _BY_CODE.get(status, SearchError)(detail)


def check(status: int) -> None:
    if status != _lib.OK:
        raise _BY_CODE.get(status, SearchError)(_lib.last_error())
