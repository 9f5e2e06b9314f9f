"""ctypes mirror of include/optable_hip.h and the loader for liboptable_hip.so.

The library is the product: there is no Python or CPU fallback behind it.  `load()` raises
`EngineUnavailable` when the shared object has not been built (run `python -c "import
__graft_entry__ as g; g.build()"` or `make -C optable_amd/csrc`).
"""
import ctypes as C
import os

ABI_VERSION = 13
_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "csrc", "liboptable_hip.so")


class EngineUnavailable(RuntimeError):
    pass


class OtNode(C.Structure):
    _fields_ = [
        ("M", C.c_double * 9), ("origin", C.c_double * 3), ("aabb", C.c_double * 6),
        ("lbox", C.c_double * 6), ("p", C.c_double * 8),
        ("reflectivity", C.c_double), ("transmission", C.c_double),
        ("focal_length", C.c_double), ("roc", C.c_double),
        ("kind", C.c_int32), ("end", C.c_int32), ("flags", C.c_int32), ("shape", C.c_int32),
        ("interaction", C.c_int32), ("mat1", C.c_int32), ("mat2", C.c_int32), ("roc_kind", C.c_int32),
        ("max_interact_count", C.c_int32), ("count_slot", C.c_int32), ("aux", C.c_int32),
        ("leaf_id", C.c_int32),
    ]


class OtMaterial(C.Structure):
    _fields_ = [("n", C.c_double), ("B", C.c_double * 3), ("C", C.c_double * 3),
                ("kind", C.c_int32), ("_pad", C.c_int32)]


class OtSceneDesc(C.Structure):
    _fields_ = [
        ("nodes", C.POINTER(OtNode)), ("n_nodes", C.c_int32),
        ("materials", C.POINTER(OtMaterial)), ("n_materials", C.c_int32),
        ("aux", C.POINTER(C.c_double)), ("n_aux", C.c_int32),
        ("n_count_slots", C.c_int32), ("max_children", C.c_int32), ("unit", C.c_double),
        ("root_grid", C.c_int32), ("_pad", C.c_int32),
    ]


RAY_FIELDS = ("ox", "oy", "oz", "dx", "dy", "dz", "wavelength", "q_re", "q_im", "intensity", "n", "pathlength")
SEG_FIELDS = ("ox", "oy", "oz", "dx", "dy", "dz", "length", "intensity", "q_re", "q_im", "n", "pathlength")


class OtRays(C.Structure):
    _fields_ = [(f, C.c_void_p) for f in RAY_FIELDS] + [("id", C.c_void_p), ("flags", C.c_void_p), ("length", C.c_void_p)]


class OtSegments(C.Structure):
    _fields_ = [(f, C.c_void_p) for f in SEG_FIELDS] + [("ray", C.c_void_p), ("surface", C.c_void_p)]


class OtSegmentBlock(C.Structure):
    """Append layout: one allocation of 14 planes of `capacity` slots (12 reals in SEG_FIELDS order, int32 ray, int32 surface)."""
    _fields_ = [("base", C.c_void_p), ("capacity", C.c_int64)]


class OtMonitor(C.Structure):
    _fields_ = [("M", C.c_double * 9), ("origin", C.c_double * 3),
                ("half_width", C.c_double), ("half_height", C.c_double)]


# flags / enums (== header)
NODE_GROUP, NODE_LEAF = 0, 1
NODE_CHECK_AABB, NODE_GRID, NODE_BOX_TRUSTED = 1, 2, 4
MAT_CONST, MAT_SELLMEIER, MAT_CHEB = 0, 1, 2
RAY_HAS_Q, RAY_DEAD = 1, 2
OPT_NT_STORES, OPT_MIN_WAVES, OPT_BLOCKS_PER_CU, OPT_KERNEL, OPT_LDS_LIMIT_KB, OPT_LIST_CAP, OPT_PAIR_STORES, OPT_MIX_GENERATIONS, OPT_FLAT_QUEUE, OPT_LDS_RECORDS = 1, 2, 3, 4, 5, 6, 7, 8, 9, 10
OPT_APPEND_CHUNK, OPT_INSTANCING, OPT_GEN_REUSE, OPT_BLOCK_POOL, OPT_GEN_DROP_DOOMED, OPT_REFILL, OPT_REFILL_TICKET, OPT_POOL_JITTER, OPT_GEN_ONEPASS, OPT_GEN_AHEAD, OPT_TREES_LDS_ENTRIES, OPT_TREES_REFILL_AT, OPT_TREES_FLAT, OPT_GEN_PARENT_INDEX, OPT_TREES_GLOBAL_IMAGE = 11, 12, 13, 14, 15, 16, 17, 18, 19, 20, 21, 22, 23, 24, 25

# every symbol the header declares, with its ctypes signature
_vp, _i32, _i64 = C.c_void_p, C.c_int32, C.c_int64
_TRACE_ARGS = [_vp, C.POINTER(OtRays), _i64, _i32, C.POINTER(OtSegments), _vp, _vp, _i32]
SYMBOLS = {
    "ot_abi_version": (C.c_int, []),
    "ot_last_error": (C.c_char_p, []),
    "ot_runtime_info": (C.c_int, [C.c_char_p, _i32, C.POINTER(_i32)]),
    "ot_ctx_create": (C.c_int, [C.c_int, _vp, C.POINTER(_vp)]),
    "ot_ctx_destroy": (C.c_int, [_vp]),
    "ot_ctx_synchronize": (C.c_int, [_vp]),
    "ot_ctx_set_stream": (C.c_int, [_vp, _vp]),
    "ot_scene_upload": (C.c_int, [_vp, C.POINTER(OtSceneDesc)]),
    "ot_trace_f64": (C.c_int, _TRACE_ARGS),
    "ot_trace_f32": (C.c_int, _TRACE_ARGS),
    "ot_trace_tiled_f64": (C.c_int, [_vp, C.POINTER(OtRays), _i64, _i32, _vp, _i64, _vp, _vp, _i32]),
    "ot_trace_tiled_f32": (C.c_int, [_vp, C.POINTER(OtRays), _i64, _i32, _vp, _i64, _vp, _vp, _i32]),
    "ot_bench_stream_tiled_f64": (C.c_int, [_vp, C.POINTER(OtRays), _i64, _i32, _vp, _i64, _vp]),
    "ot_bench_stream_tiled_f32": (C.c_int, [_vp, C.POINTER(OtRays), _i64, _i32, _vp, _i64, _vp]),
    "ot_trace_append_f64": (C.c_int, [_vp, C.POINTER(OtRays), _i64, _i32, C.POINTER(OtSegmentBlock), _vp, _vp, _vp, _i32]),
    "ot_trace_append_f32": (C.c_int, [_vp, C.POINTER(OtRays), _i64, _i32, C.POINTER(OtSegmentBlock), _vp, _vp, _vp, _i32]),
    "ot_trace_generation_f64": (C.c_int, [_vp, C.POINTER(OtRays), _vp, _i64, _vp, C.POINTER(OtSegments), _i64,
                                          _vp, C.POINTER(OtRays), _vp, _i64, _vp, _vp, _i32]),
    "ot_trace_generation_f32": (C.c_int, [_vp, C.POINTER(OtRays), _vp, _i64, _vp, C.POINTER(OtSegments), _i64,
                                          _vp, C.POINTER(OtRays), _vp, _i64, _vp, _vp, _i32]),
    "ot_trace_tree_f64": (C.c_int, [_vp, C.POINTER(OtRays), _vp, _i64, _vp, C.POINTER(OtSegments), _i64, _vp, C.POINTER(OtRays), _vp,
                                    C.POINTER(OtRays), _vp, _i64, _vp, _i32, C.c_double, _vp]),
    "ot_trace_tree_f32": (C.c_int, [_vp, C.POINTER(OtRays), _vp, _i64, _vp, C.POINTER(OtSegments), _i64, _vp, C.POINTER(OtRays), _vp,
                                    C.POINTER(OtRays), _vp, _i64, _vp, _i32, C.c_double, _vp]),
    "ot_trace_trees_f64": (C.c_int, [_vp, C.POINTER(OtRays), _i64, _i32, C.POINTER(OtSegments), _vp, _vp, _i32]),
    "ot_trace_trees_f32": (C.c_int, [_vp, C.POINTER(OtRays), _i64, _i32, C.POINTER(OtSegments), _vp, _vp, _i32]),
    "ot_trace_trees_append_f64": (C.c_int, [_vp, C.POINTER(OtRays), _i64, _i32, C.POINTER(OtSegmentBlock), _vp, _vp, _vp, _i32]),
    "ot_trace_trees_append_f32": (C.c_int, [_vp, C.POINTER(OtRays), _i64, _i32, C.POINTER(OtSegmentBlock), _vp, _vp, _vp, _i32]),
    "ot_trace_trees_plan": (C.c_int, [_vp, _i32, _i32, _i64, _vp]),
    "ot_monitor_record_f64": (C.c_int, [_vp, C.POINTER(OtMonitor), C.POINTER(OtSegments), _i64, _vp, _i64, _vp, _vp,
                                        _vp, _vp, _vp, _vp]),
    "ot_timing_enable": (C.c_int, [_vp, C.c_int]),
    "ot_timing_read": (C.c_int, [_vp, C.POINTER(C.c_double), C.POINTER(_i64)]),
    "ot_timing_reset": (C.c_int, [_vp]),
    "ot_set_option": (C.c_int, [_vp, _i32, _i32]),
    "ot_debug_generation_mismatches": (C.c_int, [_vp, C.POINTER(_i64)]),
    "ot_debug_last_launch": (C.c_int, [_vp, C.POINTER(_i32 * 8)]),
    "ot_trace_plan": (C.c_int, [_vp, _i32, _i64, _i32, C.POINTER(_i32 * 8)]),
    "ot_probe_layouts": (C.c_int, [_vp, _i32, C.POINTER(C.c_double), C.POINTER(C.c_double)]),
    "ot_bench_stream_f64": (C.c_int, [_vp, C.POINTER(OtRays), _i64, _i32, C.POINTER(OtSegments), _vp]),
    "ot_bench_stream_f32": (C.c_int, [_vp, C.POINTER(OtRays), _i64, _i32, C.POINTER(OtSegments), _vp]),
}

_lib = None


def load():
    """dlopen the engine and bind every declared symbol; raises if it is missing."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise EngineUnavailable(
            f"{LIB_PATH} not found: the HIP engine is not built and optable_amd has no CPU fallback. "
            "Build it with `python -c \"import __graft_entry__ as g; g.build()\"`.")
    # PyTorch brings a HIP runtime of its own; liboptable_hip.so links the system's.  Whichever initialises the device
    # first decides whether the other can: torch after the library finds "No HIP GPUs" (seen with a test that called the
    # C-ABI before its first tensor), the library after torch works.  So torch goes first — a no-op without a GPU.
    try:
        import torch

        torch.cuda.is_available() and torch.cuda.init()
    except Exception:  # noqa: BLE001 - no torch / no driver: the library's own checks report what is missing
        pass
    try:
        lib = C.CDLL(LIB_PATH)
    except OSError as exc:
        raise EngineUnavailable(f"cannot load {LIB_PATH}: {exc}") from exc
    for name, (restype, argtypes) in SYMBOLS.items():
        fn = getattr(lib, name)  # AttributeError if the header and the library disagree
        fn.restype, fn.argtypes = restype, argtypes
    if lib.ot_abi_version() != ABI_VERSION:
        raise EngineUnavailable(f"ABI mismatch: library {lib.ot_abi_version()} != binding {ABI_VERSION}")
    # one HIP runtime per process: if PyTorch's copy and another one are both mapped, device arrays of one are foreign to the other
    copies = hip_runtimes_in_process()
    if len(copies) > 1:
        raise EngineUnavailable("two HIP runtimes are loaded in this process: " + ", ".join(sorted(copies)) + " — liboptable_hip.so binds to the "
                                "libamdhip64.so.7 that is loaded first; import torch (or dlopen its lib/libamdhip64.so) before anything that "
                                "pulls in the ROCm installation's copy (INTEGRATION.md)")
    _lib = lib
    return lib


def hip_runtimes_in_process():
    """Paths of the libamdhip64 copies mapped into this process (/proc/self/maps)."""
    found = set()
    try:
        with open("/proc/self/maps") as fh:
            for line in fh:
                if "libamdhip64" in line:
                    found.add(os.path.realpath(line.split()[-1]))
    except OSError:
        pass
    return found


def runtime_info():
    """(path of the libamdhip64 the engine is bound to, hipRuntimeGetVersion)."""
    lib = load()
    buf, ver = C.create_string_buffer(1024), _i32()
    check(lib.ot_runtime_info(buf, 1024, C.byref(ver)), lib)
    return buf.value.decode(), int(ver.value)


def check(status, lib=None):
    if status != 0:
        lib = lib or load()
        msg = lib.ot_last_error()
        raise RuntimeError(f"optable_hip error {status}: {msg.decode() if msg else '?'}")
