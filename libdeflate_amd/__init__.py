"""libdeflate_amd -- MI355X-native engine for LibDeflate's compress/decompress hot path.

`LibDeflate` mirrors the reference's Lua method table (same names, argument meaning and
error behaviour) over the C ABI of libdeflate_hip.so; `batch` holds the batched entry
points (host arrays or device-resident torch tensors).
"""
from .LibDeflate import LibDeflate  # noqa: F401
from . import batch  # noqa: F401
