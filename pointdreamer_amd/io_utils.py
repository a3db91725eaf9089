"""I/O edges of the texturing path (row O1 + PLY input), formats identical to the reference:
utils/utils_2d.py:351-399 (PNG save/load, uint8 truncation), models/get3d/get3d_utils/utils_3d.py:27-64
(OBJ/MTL), utils/other_utils.py:122-163 (binary little-endian PLY x,y,z f32 + red,green,blue u8).
Pure numpy / PIL; no plyfile / trimesh dependency.
"""
import os
import numpy as np
import PIL.Image
import torch


def _to_u8_hwc(img_chw):
    img = np.asarray(img_chw, np.float32).transpose(1, 2, 0) * 255.0      # works on a copy (the reference scales in place)
    return np.ascontiguousarray(img.clip(0, 255).astype(np.uint8))


def save_CHW_RGB_img(img, file_name):
    PIL.Image.fromarray(_to_u8_hwc(img), 'RGB').save(file_name)


def save_CHW_RGBA_img(img, file_name):
    PIL.Image.fromarray(_to_u8_hwc(img), 'RGBA').save(file_name)


def load_CHW_RGB_img(file_name):
    im = PIL.Image.open(file_name)
    if im.mode != 'RGB':
        im = im.convert('RGB')
    a = torch.from_numpy(np.array(im))[:, :, :3].float() / 255.
    return a.permute(2, 0, 1)


def read_ply_xyzrgb(path):
    """Binary-LE or ASCII PLY with vertex properties x,y,z (float) and red,green,blue (uchar)."""
    with open(path, 'rb') as f:
        header = []
        while True:
            line = f.readline()
            if not line:
                raise ValueError("bad PLY header")
            header.append(line.decode('ascii', 'replace').strip())
            if header[-1] == 'end_header':
                break
        fmt = [h for h in header if h.startswith('format')][0].split()[1]
        n = 0
        props = []
        in_vertex = False
        for h in header:
            t = h.split()
            if t[:1] == ['element']:
                in_vertex = t[1] == 'vertex'
                if in_vertex:
                    n = int(t[2])
            elif t[:1] == ['property'] and in_vertex:
                props.append((t[2], t[1]))
        np_t = {'float': '<f4', 'float32': '<f4', 'double': '<f8', 'float64': '<f8', 'uchar': 'u1', 'uint8': 'u1',
                'int': '<i4', 'int32': '<i4', 'uint': '<u4', 'short': '<i2', 'ushort': '<u2', 'char': 'i1'}
        if fmt == 'binary_little_endian':
            dt = np.dtype([(name, np_t[ty]) for name, ty in props])
            data = np.frombuffer(f.read(dt.itemsize * n), dtype=dt, count=n)
        elif fmt == 'ascii':
            rows = np.loadtxt(f, max_rows=n, ndmin=2)
            data = {name: rows[:, i] for i, (name, _) in enumerate(props)}
        else:
            raise ValueError(f"unsupported PLY format {fmt}")
    xyz = np.stack([data['x'], data['y'], data['z']], -1)
    rgb = np.stack([data['red'], data['green'], data['blue']], -1)
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


def load_obj_mesh(path):
    """Vertices / triangle faces of an OBJ (what kal.io.obj.import_mesh provides at demo.py:395)."""
    vs, fs = [], []
    with open(path) as f:
        for line in f:
            t = line.split()
            if not t:
                continue
            if t[0] == 'v':
                vs.append([float(x) for x in t[1:4]])
            elif t[0] == 'f':
                idx = [int(x.split('/')[0]) for x in t[1:]]
                for k in range(1, len(idx) - 1):
                    fs.append([idx[0] - 1, idx[k] - 1, idx[k + 1] - 1])
    return np.array(vs, np.float32), np.array(fs, np.int64)


def savemeshtes2(pointnp_px3, tcoords_px2, facenp_fx3, facetex_fx3, fname):
    """utils_3d.py:27-64, byte-identical text output."""
    fol, na = os.path.split(fname)
    na, _ = os.path.splitext(na)
    with open(os.path.join(fol, 'model_normalized.mtl'), 'w') as fid:
        fid.write('newmtl material_0\nKd 1 1 1\nKa 0 0 0\nKs 0.4 0.4 0.4\nNs 10\nillum 2\nmap_Kd %s.png\n' % na)
    out = ['mtllib %s.mtl\n' % na]
    out += ['v %f %f %f\n' % (p[0], p[1], p[2]) for p in pointnp_px3]
    out += ['vt %f %f\n' % (p[0], p[1]) for p in tcoords_px2]
    out.append('usemtl material_0\n')
    f1 = np.asarray(facenp_fx3) + 1
    f2 = np.asarray(facetex_fx3) + 1
    out += ['f %d/%d %d/%d %d/%d\n' % (a[0], b[0], a[1], b[1], a[2], b[2]) for a, b in zip(f1, f2)]
    with open(fname, 'w') as fid:
        fid.write(''.join(out))
