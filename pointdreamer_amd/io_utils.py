"""I/O edges of the texturing path (row O1 + PLY input), formats identical to the reference:
utils/utils_2d.py:351-399 (PNG save/load, uint8 truncation), models/get3d/get3d_utils/utils_3d.py:27-64
(OBJ/MTL), utils/other_utils.py:122-163 (binary little-endian PLY x,y,z f32 + red,green,blue u8).
The per-shape edges -- PLY read, OBJ/MTL write, PNG encode, float -> uint8 conversion -- run in native code
(`csrc/io_native.hip` behind `pdhip_io_*`); the rarely used ones (PNG load, OBJ load, PLY write) are numpy / PIL.
"""
import os
import ctypes as C
import numpy as np
import torch

from . import _lib


def _host_u8(t):
    """uint8 CUDA tensor -> (host numpy view, wait).  With the write queue on (set_async) the copy goes into PINNED memory without
    blocking the host -- `wait()` (called by the writer thread before it encodes) synchronises on an event recorded behind the copy --
    so a directory run never stalls its launch thread on a device-to-host copy (round 5: 49 blocking `.cpu()` calls per shape were
    16 ms of a 34 ms shape).  Synchronous mode: a plain blocking copy, `wait` does nothing."""
    if _pool is None:
        return t.cpu().numpy(), (lambda: None)
    # the producing kernel was enqueued on _lib.stream() = the CURRENT device's current stream; the copy and the event go onto the same
    # stream only if the tensor lives on that device -- anything else would let the writer read a half-converted image
    if t.device.index is not None and t.device.index != torch.cuda.current_device():
        raise _lib.PdhipError(f"io_utils: image on {t.device} while cuda:{torch.cuda.current_device()} is current (set the device first)")
    host = torch.empty(t.shape, dtype=t.dtype, pin_memory=True)
    host.copy_(t, non_blocking=True)
    ev = torch.cuda.Event()
    ev.record(torch.cuda.current_stream(t.device))
    keep = (t, host)                                       # (the device tensor must outlive the copy, the pinned one the encode)

    def wait():
        ev.synchronize()
        return keep
    return host.numpy(), wait


def _u8_hwc(img, channels, want_wait=False):
    """uint8 [H,W,C] host array of a CHW float image in [0,1]: `(img * 255).clip(0, 255).astype(uint8)` (utils_2d.py:351-372).
    GPU tensors are converted on the device so only H*W*C bytes cross PCIe.  want_wait: also return the callable that makes the
    array's contents valid (asynchronous device-to-host copy, `_host_u8`)."""
    wait = lambda: None
    if torch.is_tensor(img) and img.is_cuda:
        L = _lib.lib()
        img = img.float().contiguous()
        Cn, H, W = img.shape
        out = torch.empty((H, W, Cn), dtype=torch.uint8, device=img.device)
        _lib.check(L.pdhip_chw_f32_to_hwc_u8(_lib.ptr(img), Cn, H, W, _lib.ptr(out), _lib.stream()), 'pdhip_chw_f32_to_hwc_u8')
        if want_wait:
            arr, wait = _host_u8(out)
        else:
            arr = out.cpu().numpy()
    else:
        a = img.detach().cpu().numpy() if torch.is_tensor(img) else np.asarray(img)
        arr = np.ascontiguousarray((a.astype(np.float32).transpose(1, 2, 0) * 255.0).clip(0, 255).astype(np.uint8))
    assert arr.shape[2] == channels, f"expected {channels} channels, got {arr.shape[2]}"
    return (arr, wait) if want_wait else arr


# ---- deferred writes.  The encoders are native calls that release the GIL, so a directory run (demo.py over many clouds) can
# overlap the PNG / OBJ encoding of one shape with the GPU work of the next: `set_async(True)` queues the encode + write on a
# small thread pool (the pixel data has already been converted and copied to the host), `flush()` waits and re-raises the first
# error.  Default is synchronous: the file exists when the save function returns, as in the reference.
_pool = None
_pool_workers = 0
_pending = []


def usable_cpus():
    """CPUs this process may actually use: scheduler affinity AND the cgroup CPU quota (the pool's GPU boxes show 256 logical CPUs and
    grant the container cpu.max = 16)."""
    n = os.cpu_count() or 1
    try:
        n = min(n, len(os.sched_getaffinity(0)))
    except (AttributeError, OSError):
        pass
    try:
        q, per = open('/sys/fs/cgroup/cpu.max').read().split()[:2]
        if q != 'max':
            n = max(1, min(n, int(int(q) / int(per) + 0.5)))
    except (OSError, ValueError):
        pass
    try:
        q = int(open('/sys/fs/cgroup/cpu/cpu.cfs_quota_us').read()); per = int(open('/sys/fs/cgroup/cpu/cpu.cfs_period_us').read())
        if q > 0:
            n = max(1, min(n, int(q / per + 0.5)))
    except (OSError, ValueError):
        pass
    return n


def cpus_per_rank():
    """usable_cpus() shared between the ranks of this node (torchrun's LOCAL_WORLD_SIZE; one process per GPU)."""
    return max(1, usable_cpus() // max(1, int(os.environ.get('LOCAL_WORLD_SIZE', 1))))


def set_async(on, workers=8):
    """Turn the background write queue on / off.  A queue that exists with another worker count is drained and rebuilt (ADVICE r5:
    `workers` used to be ignored once a pool existed)."""
    global _pool, _pool_workers
    flush()
    if _pool is not None and (not on or workers != _pool_workers):
        _pool.shutdown(wait=True)
        _pool = None
    if on and _pool is None:
        from concurrent.futures import ThreadPoolExecutor
        _pool = ThreadPoolExecutor(max_workers=workers, thread_name_prefix='pdhip-io')
        _pool_workers = workers


def flush():
    """Wait for every queued write; raises the first failure."""
    global _pending
    todo, _pending = _pending, []
    err = None
    for f in todo:
        try:
            f.result()
        except Exception as e:          # noqa: BLE001 -- re-raised below, after every write has been waited for
            err = err or e
    if err is not None:
        raise err


def _submit(fn):
    if _pool is None:
        fn()
    else:
        _pending.append(_pool.submit(fn))


def _save_png(img, file_name, channels):
    L = _lib.lib()
    arr, wait = _u8_hwc(img, channels, want_wait=True)
    path = os.fspath(file_name).encode()

    def write():
        wait()
        _lib.check(L.pdhip_io_write_png(path, arr.ctypes.data_as(C.c_void_p), arr.shape[0], arr.shape[1], channels, 1),
                   'pdhip_io_write_png')
    _submit(write)


def save_HWC_u8_img(arr, file_name, wait=None):
    """uint8 [H,W,3|4] host array -> PNG (used for images composed on the device in 8-bit).  wait: callable that makes the array's
    contents valid (`_host_u8`), called by the writer before it encodes."""
    L = _lib.lib()
    if wait is None:
        arr = np.ascontiguousarray(arr, np.uint8)
    assert arr.dtype == np.uint8 and arr.flags['C_CONTIGUOUS']
    path = os.fspath(file_name).encode()
    ch = arr.shape[2]

    def write():
        if wait is not None:
            wait()
        _lib.check(L.pdhip_io_write_png(path, arr.ctypes.data_as(C.c_void_p), arr.shape[0], arr.shape[1], ch, 1),
                   'pdhip_io_write_png')
    _submit(write)


def save_CHW_RGB_img(img, file_name):
    _save_png(img, file_name, 3)


def save_CHW_RGBA_img(img, file_name):
    _save_png(img, file_name, 4)


def load_CHW_RGB_img(file_name):
    import PIL.Image
    im = PIL.Image.open(file_name)
    if im.mode != 'RGB':
        im = im.convert('RGB')
    a = torch.from_numpy(np.array(im))[:, :, :3].float() / 255.
    return a.permute(2, 0, 1)


def read_ply_xyzrgb(path):
    """Binary-LE or ASCII PLY with vertex properties x,y,z and red,green,blue -> (xyz float32 [n,3], rgb uint8 [n,3])."""
    L = _lib.lib()
    p = os.fspath(path).encode()
    n = L.pdhip_io_ply_count(p)
    if n < 0:
        raise _lib.PdhipError(L.pdhip_last_error().decode())
    xyz = np.empty((n, 3), np.float32)
    rgb = np.empty((n, 3), np.uint8)
    _lib.check(L.pdhip_io_read_ply_xyzrgb(p, xyz.ctypes.data_as(C.c_void_p), rgb.ctypes.data_as(C.c_void_p), n), 'pdhip_io_read_ply_xyzrgb')
    return xyz, rgb


def save_colored_pc_ply(coords, colors, path):
    """other_utils.py:122-146: colours are floats in [0,1], stored as uint8(c*255)."""
    n = coords.shape[0]
    v = np.empty(n, dtype=[('x', '<f4'), ('y', '<f4'), ('z', '<f4'), ('red', 'u1'), ('green', 'u1'), ('blue', 'u1')])
    v['x'], v['y'], v['z'] = (coords[:, i].astype('f4') for i in range(3))
    v['red'], v['green'], v['blue'] = ((colors[:, i].astype('f4') * 255).astype('u1') for i in range(3))
    hdr = ("ply\nformat binary_little_endian 1.0\nelement vertex %d\nproperty float x\nproperty float y\nproperty float z\n"
           "property uchar red\nproperty uchar green\nproperty uchar blue\nend_header\n" % n)
    with open(path, 'wb') as f:
        f.write(hdr.encode('ascii'))
        f.write(v.tobytes())


def load_obj_mesh(path, with_uv=False):
    """Vertices / triangle faces of an OBJ (what kal.io.obj.import_mesh provides at demo.py:395).  with_uv=True also returns
    the `vt` table and the per-corner texture indices (None, None when the file has no complete `f v/vt` records)."""
    vs, fs, vts, fts = [], [], [], []
    has_uv = True
    with open(path) as f:
        for line in f:
            t = line.split()
            if not t:
                continue
            if t[0] == 'v':
                vs.append([float(x) for x in t[1:4]])
            elif t[0] == 'vt':
                vts.append([float(x) for x in t[1:3]])
            elif t[0] == 'f':
                parts = [x.split('/') for x in t[1:]]
                idx = [int(p[0]) for p in parts]
                tix = [int(p[1]) if len(p) > 1 and p[1] else 0 for p in parts]
                has_uv &= all(tix)
                for k in range(1, len(idx) - 1):
                    fs.append([idx[0] - 1, idx[k] - 1, idx[k + 1] - 1])
                    fts.append([tix[0] - 1, tix[k] - 1, tix[k + 1] - 1])
    v, f = np.array(vs, np.float32), np.array(fs, np.int64)
    if not with_uv:
        return v, f
    if has_uv and vts:
        return v, f, np.array(vts, np.float32), np.array(fts, np.int64)
    return v, f, None, None


def savemeshtes2(pointnp_px3, tcoords_px2, facenp_fx3, facetex_fx3, fname):
    """utils_3d.py:27-64, byte-identical text output (native formatter)."""
    L = _lib.lib()
    fol, na = os.path.split(fname)
    na, _ = os.path.splitext(na)
    pts = np.ascontiguousarray(np.asarray(pointnp_px3)[:, :3], np.float64)
    tcs = np.ascontiguousarray(np.asarray(tcoords_px2)[:, :2], np.float64)
    f1 = np.ascontiguousarray(facenp_fx3, np.int64)
    f2 = np.ascontiguousarray(facetex_fx3, np.int64)
    vp = lambda a: a.ctypes.data_as(C.c_void_p)
    obj, mtl = os.fspath(fname).encode(), os.path.join(fol, 'model_normalized.mtl').encode()

    def write():
        _lib.check(L.pdhip_io_write_obj_mtl(obj, mtl, na.encode(), vp(pts), len(pts), vp(tcs), len(tcs), vp(f1), vp(f2), len(f1)),
                   'pdhip_io_write_obj_mtl')
    _submit(write)
