"""ctypes binding of libdeflate_hip.so (include/libdeflate_hip.h).

The HIP library is the product path; there is no CPU fallback.  Importing this module
fails loudly when the shared object has not been built (run `python -c "import
__graft_entry__ as g; g.build()"` or `make -C libdeflate_amd/csrc`).
"""
import ctypes as C
import os

HERE = os.path.dirname(os.path.abspath(__file__))
SO_PATH = os.path.join(HERE, "libdeflate_hip.so")

if not os.path.exists(SO_PATH):
    raise ImportError(
        "libdeflate_hip.so is missing (%s). Build it with `make -C libdeflate_amd/csrc`; "
        "this package has no CPU fallback." % SO_PATH)



def _share_torch_hip_runtime():
    """torch wheels bundle their own libamdhip64 (same SONAME as /opt/rocm's, but torch asks for it by another
    name), so a process that loads libdeflate_hip.so BEFORE torch ends up with two HIP runtimes, and the second
    one finds no device.  Loading torch's copy first (without importing torch) makes both resolve to that one,
    whatever the import order; without torch installed the system runtime is used."""
    try:
        import importlib.util
        spec = importlib.util.find_spec("torch")
        if spec is None or not spec.origin:
            return
        cand = os.path.join(os.path.dirname(spec.origin), "lib", "libamdhip64.so")
        if os.path.exists(cand):
            C.CDLL(cand, mode=C.RTLD_GLOBAL)
    except Exception:
        pass


_share_torch_hip_runtime()
lib = C.CDLL(SO_PATH)

u8p, u32p, u64p, i32p = C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p
sz = C.c_size_t

_SIGS = {
    "ldh_init": (C.c_int, [C.c_int]),
    "ldh_shutdown": (None, []),
    "ldh_device_count": (C.c_int, []),
    "ldh_last_error": (C.c_char_p, []),
    "ldh_stream": (C.c_void_p, []),
    "ldh_sync": (C.c_int, []),
    "ldh_init_devices": (C.c_int, [C.POINTER(C.c_int), C.c_int]),
    "ldh_pool_size": (C.c_int, []),
    "ldh_use_context": (C.c_int, [C.c_int]),
    "ldh_partition_by_bytes": (C.c_int, [u64p, sz, C.c_int, C.c_void_p]),
    "ldh_compress_bound": (sz, [sz]),
    "ldh_host_register": (C.c_int, [C.c_void_p, sz]),
    "ldh_host_unregister": (C.c_int, [C.c_void_p]),
    "ldh_host_alloc": (C.c_void_p, [sz]),
    "ldh_host_free": (None, [C.c_void_p]),
    "ldh_dict_create": (C.c_void_p, [u8p, sz]),
    "ldh_dict_free": (None, [C.c_void_p]),
    "ldh_dict_adler32": (C.c_uint32, [C.c_void_p]),
    "ldh_dict_len": (sz, [C.c_void_p]),
    "ldh_compress": (C.c_int, [u8p, sz, C.c_int, C.c_int, C.c_int, C.c_void_p, u8p, sz,
                               C.POINTER(sz), C.POINTER(C.c_int)]),
    "ldh_decompress": (C.c_int, [u8p, sz, C.c_int, C.c_void_p, u8p, sz, C.POINTER(sz), C.POINTER(sz),
                                 C.POINTER(C.c_int)]),
    "ldh_adler32": (C.c_int, [u8p, sz, C.POINTER(C.c_uint32)]),
    "ldh_codec_create": (C.c_void_p, [u8p, sz, u8p, sz, u8p, sz, C.POINTER(C.c_char_p)]),
    "ldh_codec_free": (None, [C.c_void_p]),
    "ldh_codec_encode_batch": (C.c_int, [C.c_void_p, u8p, u64p, sz, u8p, sz, u32p]),
    "ldh_codec_encode_batch_dev": (C.c_int, [C.c_void_p, u8p, u64p, sz, u8p, sz, u32p]),
    "ldh_codec_decode_batch": (C.c_int, [C.c_void_p, u8p, u64p, sz, u8p, sz, u32p, u8p]),
    "ldh_codec_decode_batch_dev": (C.c_int, [C.c_void_p, u8p, u64p, sz, u8p, sz, u32p, u8p]),
    "ldh_print6_encode": (C.c_int, [u8p, sz, u8p, sz, C.POINTER(sz)]),
    "ldh_print6_decode": (C.c_int, [u8p, sz, u8p, sz, C.POINTER(sz), C.POINTER(C.c_int)]),
    "ldh_compress_batch": (C.c_int, [u8p, u64p, sz, C.c_int, C.c_int, C.c_int, C.c_void_p, u8p, sz, u32p, u8p]),
    "ldh_compress_batch_dev": (C.c_int, [u8p, u64p, sz, sz, C.c_int, C.c_int, C.c_int, C.c_void_p, u8p, sz,
                                         u32p, u8p]),
    "ldh_decompress_batch": (C.c_int, [u8p, u64p, sz, C.c_int, C.c_void_p, u8p, sz, u32p, u32p, i32p]),
    "ldh_decompress_batch_dev": (C.c_int, [u8p, u64p, sz, C.c_int, C.c_void_p, u8p, sz, u32p, u32p, i32p]),
    "ldh_decompress_batch_dev_noalloc": (C.c_int, [u8p, u64p, sz, C.c_int, C.c_void_p, u8p, sz, u32p, u32p, i32p]),
    "ldh_decompress_slots_dev": (C.c_int, [u8p, sz, u32p, sz, C.c_int, C.c_void_p, u8p, sz, u32p, u32p, i32p]),
    "ldh_adler32_batch": (C.c_int, [u8p, u64p, sz, u32p]),
    "ldh_adler32_batch_dev": (C.c_int, [u8p, u64p, sz, u32p]),
    "ldh_print6_encode_batch": (C.c_int, [u8p, u64p, sz, u8p, sz, u32p]),
    "ldh_print6_encode_batch_dev": (C.c_int, [u8p, u64p, sz, u8p, sz, u32p]),
    "ldh_print6_decode_batch": (C.c_int, [u8p, u64p, sz, u8p, sz, u32p, u8p]),
    "ldh_print6_decode_batch_dev": (C.c_int, [u8p, u64p, sz, u8p, sz, u32p, u8p]),
    "ldh_profile_enable": (C.c_int, [C.c_int]),
    "ldh_profile_reset": (C.c_int, []),
    "ldh_profile_count": (C.c_int, []),
    "ldh_profile_get": (C.c_int, [C.c_int, C.c_char_p, sz, C.POINTER(C.c_double), C.POINTER(C.c_uint64)]),
    "ldh_timer_start": (C.c_int, []),
    "ldh_timer_stop": (C.c_int, [C.POINTER(C.c_double)]),
    "ldh_debug_fetch": (C.c_int, [C.c_int, sz, C.c_void_p, sz, C.POINTER(sz)]),
    "ldh_debug_alloc_count": (C.c_uint64, []),
}

EXPORTS = sorted(_SIGS)

for _name, (_res, _args) in _SIGS.items():
    _f = getattr(lib, _name)  # AttributeError here = header/library mismatch
    _f.restype = _res
    _f.argtypes = _args


class LdhError(RuntimeError):
    pass


def check(rc):
    if rc != 0:
        raise LdhError("libdeflate_hip error %d: %s" % (rc, (lib.ldh_last_error() or b"").decode()))


_inited = None


def init(device=0):
    global _inited
    if _inited == device:
        return
    check(lib.ldh_init(device))
    _inited = device


def init_devices(devices=None):
    """A pool of contexts, one per listed device (None: every visible GPU; a device may be listed twice):
    the host-pointer batch calls then split their streams over it by bytes (include/libdeflate_hip.h)."""
    global _inited
    if devices:
        arr = (C.c_int * len(devices))(*devices)
        check(lib.ldh_init_devices(arr, len(devices)))
    else:
        check(lib.ldh_init_devices(None, 0))
    if _inited is None:
        _inited = devices[0] if devices else 0


def partition_by_bytes(in_off, parts):
    """first[0..parts] of ldh_partition_by_bytes for a numpy uint64 offsets array (host arithmetic only)."""
    import numpy as np
    off = np.ascontiguousarray(in_off, dtype=np.uint64)
    first = (sz * (parts + 1))()
    check(lib.ldh_partition_by_bytes(off.ctypes.data, len(off) - 1, parts, first))
    return [int(x) for x in first]
