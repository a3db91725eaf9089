"""ctypes binding of libpdhip.so (include/pdhip.h).  No fallback: if the HIP library is missing or a
call fails, this raises -- the product never computes on the CPU or through the oracle."""
import ctypes as C
import os
import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, 'libpdhip.so')


class PdhipError(RuntimeError):
    pass


_lib = None

vp, i32, i64, f32, f64, sz = C.c_void_p, C.c_int, C.c_int64, C.c_float, C.c_double, C.c_size_t

_SIGS = {
    'pdhip_version': (C.c_int, []),
    'pdhip_lab_build': (C.c_int, []),
    'pdhip_last_error': (C.c_char_p, []),
    'pdhip_project_points': (C.c_int, [vp, i32, vp, i32, vp, i32, i32, f64, vp, vp, vp, vp, vp, vp, vp, vp]),
    'pdhip_project_points_shapes': (C.c_int, [vp, i32, i32, vp, i32, vp, i32, i32, f64, vp, vp, vp, vp, vp, vp, vp, vp]),
    'pdhip_raster_mesh_shapes': (C.c_int, [vp, i32, i32, i32, vp, i32, i32, vp, sz, vp, vp, vp, vp]),
    'pdhip_hidden_point_removal_shapes': (C.c_int, [vp, i32, vp, i32, i32, f64, vp, vp, vp, vp]),
    'pdhip_sparse_views_shapes': (C.c_int, [vp, vp, vp, vp, i32, i32, i32, i32, i32, i32, f64, vp, vp, vp, vp, vp, vp, vp]),
    'pdhip_texel_visibility_shapes': (C.c_int, [vp, i32, i32, vp, vp, i32, vp, vp, f64, vp, i32, f32, vp, vp]),
    'pdhip_nbf_shrink_shapes': (C.c_int, [vp, vp, i32, i32, i32, vp, i32, vp, vp, vp]),
    'pdhip_view_select_blend_shapes': (C.c_int, [vp, i32, i32, vp, vp, vp, i32, vp, i32, vp, vp, vp, f64, vp, vp, i32, vp, i32, vp, i32,
                                                 vp, vp, vp, vp]),
    'pdhip_raster_mesh': (C.c_int, [vp, i32, i32, vp, i32, i32, vp, vp, vp, vp, vp]),
    'pdhip_raster_mesh_ws_bytes': (sz, [i32, i32, i32]),
    'pdhip_raster_mesh_ws': (C.c_int, [vp, i32, i32, vp, i32, i32, vp, sz, vp, vp, vp, vp]),
    'pdhip_debug_set_raster_path': (C.c_int, [i32]),
    'pdhip_raster_barycentrics': (C.c_int, [vp, i32, i32, vp, i32, vp, vp, vp]),
    'pdhip_interpolate': (C.c_int, [vp, i32, vp, vp, vp, C.c_longlong, vp, vp]),
    'pdhip_rescale_vertices': (C.c_int, [vp, i32, i32, vp, vp, vp, f64, vp]),
    'pdhip_optimize_color_ws_bytes': (sz, [i32, i32, i32]),
    'pdhip_optimize_color': (C.c_int, [vp, i32, vp, vp, i32, i32, vp, i32, vp, f64, i32, vp, vp, vp]),
    'pdhip_resize_mask': (C.c_int, [vp, i32, i32, i32, vp, i32, i32, vp]),
    'pdhip_point_visibility': (C.c_int, [i32, vp, vp, vp, i32, i32, f32, vp, vp, vp]),
    'pdhip_hpr_ws_bytes': (sz, [i32, i32]),
    'pdhip_hidden_point_removal': (C.c_int, [vp, i32, vp, i32, f64, vp, vp, vp, vp]),
    'pdhip_hpr_read_counters': (C.c_int, [vp, i32, vp, vp]),
    'pdhip_point_pixels': (C.c_int, [vp, i32, i32, i32, vp, vp]),
    'pdhip_point_visibility_pixels': (C.c_int, [i32, vp, vp, vp, i32, i32, f32, vp, i32, vp, vp]),
    'pdhip_sparse_views_ws_bytes': (sz, [i32, i32, i32]),
    'pdhip_sparse_views': (C.c_int, [vp, vp, vp, vp, i32, i32, i32, i32, i32, f64, vp, vp, vp, vp, vp, vp, vp]),
    'pdhip_nearest_fill_ws_ints': (sz, [i32, i32, i32]),
    'pdhip_nearest_fill': (C.c_int, [vp, vp, i32, i32, i32, i32, i64, i64, i64, vp, i32, i64, vp, vp]),
    'pdhip_linear_fill_ws_bytes': (sz, [i32, i32, i32]),
    'pdhip_linear_fill': (C.c_int, [vp, vp, i32, i32, i32, i32, vp, i32, i64, vp, vp, vp]),
    'pdhip_linear_fill_unresolved': (C.c_int, [vp, i32, i32, i32, vp, vp]),
    'pdhip_debug_set_linear_local': (C.c_int, [i32]),
    'pdhip_debug_set_unproject_generic': (C.c_int, [i32]),
    'pdhip_texel_visibility': (C.c_int, [vp, i32, vp, vp, i32, vp, vp, f64, vp, i32, f32, vp, vp]),
    'pdhip_nbf_shrink': (C.c_int, [vp, vp, i32, i32, vp, i32, vp, vp, vp]),
    'pdhip_pack_bits': (C.c_int, [vp, C.c_longlong, vp, vp]),
    'pdhip_unpack_bits': (C.c_int, [vp, C.c_longlong, vp, vp]),
    'pdhip_nbf_triptych': (C.c_int, [vp, vp, i32, i32, i32, vp, vp, vp]),
    'pdhip_view_select_blend': (C.c_int, [vp, i32, vp, vp, vp, i32, vp, vp, vp, vp, f64, vp, vp, i32, vp, i32, vp, i32,
                                          vp, vp, vp, vp]),
    'pdhip_compact_texels': (C.c_int, [vp, vp, i32, vp, vp, vp, vp, vp, vp, vp]),
    'pdhip_io_ply_count': (C.c_longlong, [C.c_char_p]),
    'pdhip_io_read_ply_xyzrgb': (C.c_int, [C.c_char_p, vp, vp, C.c_longlong]),
    'pdhip_io_write_obj_mtl': (C.c_int, [C.c_char_p, C.c_char_p, C.c_char_p, vp, C.c_longlong, vp, C.c_longlong, vp, vp, C.c_longlong]),
    'pdhip_io_write_png': (C.c_int, [C.c_char_p, vp, i32, i32, i32, i32]),
    'pdhip_chw_f32_to_hwc_u8': (C.c_int, [vp, i32, i32, i32, vp, vp]),
    'pdhip_mark_unpainted_faces': (C.c_int, [vp, vp, i32, i32, vp, vp]),
    'pdhip_vertex_texel_fetch': (C.c_int, [vp, i32, vp, vp, i32, vp, vp, vp, vp]),
    'pdhip_neighbor_diffuse_round': (C.c_int, [vp, vp, i32, vp, i32, vp, vp, vp, vp, vp]),
    'pdhip_scatter_vertex_colors': (C.c_int, [vp, vp, i32, vp, vp, vp, i32, vp]),
}


def register(name, restype, argtypes):
    """Used by the nn binding module to add its entry points to the same table."""
    _SIGS[name] = (restype, argtypes)
    if _lib is not None:
        fn = getattr(_lib, name)
        fn.restype, fn.argtypes = restype, argtypes


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise PdhipError(f"{LIB_PATH} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                             "(make -C pointdreamer_amd/csrc).  There is no CPU fallback.")
        _lib = C.CDLL(LIB_PATH)
        for name, (res, args) in _SIGS.items():
            fn = getattr(_lib, name)          # AttributeError here == ABI mismatch, fail loudly
            fn.restype, fn.argtypes = res, args
        if _lib.pdhip_lab_build() != 0 and not os.environ.get('PDHIP_ALLOW_LAB_BUILD'):
            n, _lib = _lib.pdhip_lab_build(), None
            raise PdhipError(f"{LIB_PATH} was built with a wrong-result PD_LAB_* timing switch in {n} translation unit(s): rebuild it "
                             "(make -C pointdreamer_amd/csrc clean all) or set PDHIP_ALLOW_LAB_BUILD=1 for a timing experiment")
    return _lib


def check(rc, what):
    if rc != 0:
        msg = lib().pdhip_last_error()
        raise PdhipError(f"{what} failed (code {rc}): {msg.decode() if msg else ''}")


# (the host needs ~10 us per kernel launch on this path; torch's python wrappers around "current device" / "current stream" were
# a quarter of it, so the raw bindings are used where this torch build has them)
_raw_device = getattr(torch._C, '_cuda_getDevice', None)
_raw_stream = getattr(torch._C, '_cuda_getCurrentRawStream', None)


def _current_device():
    return _raw_device() if _raw_device is not None else torch.cuda.current_device()


def stream():
    if _raw_stream is not None and _raw_device is not None:
        return C.c_void_p(_raw_stream(_raw_device()))
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def resolve_device(device):
    """torch.device with an explicit index ('cuda' -> the current device)."""
    d = torch.device(device)
    if d.type == 'cuda' and d.index is None:
        d = torch.device('cuda', torch.cuda.current_device())
    return d


def ptr(t, dtype=None, allow_none=False):
    """Device pointer of a contiguous CUDA(HIP) tensor, with dtype check.  The kernels are launched on the CURRENT device's
    current stream (stream()), so a tensor living on another device is refused here instead of faulting in a kernel."""
    if t is None:
        if allow_none:
            return C.c_void_p(0)
        raise PdhipError("null tensor")
    if not t.is_cuda:
        raise PdhipError("pointdreamer_amd needs tensors on the GPU (cuda:N == HIP device); got a CPU tensor. "
                         "There is no CPU path.")
    if t.device.index != _current_device():
        raise PdhipError(f"tensor on {t.device} but the current device is cuda:{torch.cuda.current_device()}: call "
                         "torch.cuda.set_device / use `with torch.cuda.device(...)` around pointdreamer_amd calls")
    if not t.is_contiguous():
        raise PdhipError("tensor must be contiguous")
    if dtype is not None and t.dtype != dtype:
        raise PdhipError(f"expected dtype {dtype}, got {t.dtype}")
    return C.c_void_p(t.data_ptr())


def as_u8(t):
    """torch.bool <-> uint8 views share storage (bool is one byte, values 0/1)."""
    return t.view(torch.uint8) if t.dtype == torch.bool else t


# ---- derived constants of long-lived inputs (a mesh's int32 face table, the stacked camera parameters, the eye positions on the
# device ...): recomputed only when the source object changes.  An entry is valid while the very same tensor object is alive and
# unmodified (weak reference + torch's version counter), so a recycled address can never serve a stale result.
import weakref as _weakref
_MEMO = {}


def memo(src, tag, fn):
    """fn(src) cached per (tensor object, tag).  `src`: a tensor, or a tuple / list of tensors (all must be the same objects)."""
    items = tuple(src) if isinstance(src, (tuple, list)) else (src,)
    key = (tag,) + tuple(id(t) for t in items)
    hit = _MEMO.get(key)
    if hit is not None:
        refs, vers, val = hit
        if all(r() is t for r, t in zip(refs, items)) and vers == tuple(t._version for t in items):
            return val
    val = fn(src)
    _settle(val)
    if len(_MEMO) > 256:
        if torch.cuda.is_available():
            torch.cuda.synchronize()                 # (an evicted value may still be in use on another stream)
        _MEMO.clear()
    _MEMO[key] = (tuple(_weakref.ref(t) for t in items), tuple(t._version for t in items), val)
    return val


def _settle(val):
    """A cached value is handed to whichever HIP stream asks next (pipeline.colorize_meshes_batched runs shapes on streams of their
    own): finish computing it before it enters the cache.  Once per new entry."""
    if torch.is_tensor(val) and val.is_cuda:
        torch.cuda.current_stream(val.device).synchronize()


_CONST = {}


def const_vec(n, value, dev):
    """A read-only [n] float32 vector of one value on `dev` (the kernels' form of the reference's python-scalar crop parameters)."""
    key = (int(n), float(value), str(dev))
    v = _CONST.get(key)
    if v is None:
        v = _CONST[key] = torch.full((int(n),), float(value), device=dev)
        _settle(v)
    return v

