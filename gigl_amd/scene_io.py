"""Scene and fog asset I/O in the reference's formats.

- OBJ/MTL -> ray-tracing arrays, following the RT half of Scene::loadObjects
  (/root/reference/Source/Common/Scene.cpp:28-47, :102-117, :136-153) on top of TinyOBJ 0.9.6's
  shape splitting, per-shape vertex cache and triangle-fan conversion
  (Source/Include/TinyOBJ/tiny_obj_loader.cc:169-208, :232-255, :495-698).
- `.3dt` volume files: 3 x int32 resolution, then the raw payload
  (Source/Fog/DensityField.cpp:213-246 for u8 density, :329-362 for f32 preintegrated density).
"""
import os

import numpy as np

INV_PI = np.float32(0.318309873)  # Source/Common/Constants.h:9


def _load_mtl(path):
    mats, cur = {}, None
    with open(path, "r") as f:
        for line in f:
            tok = line.split()
            if not tok or tok[0].startswith("#"):
                continue
            if tok[0] == "newmtl":
                cur = {"name": tok[1], "Kd": [0.0, 0.0, 0.0], "Ks": [0.0, 0.0, 0.0], "Ns": 1.0,
                       "Ke": [0.0, 0.0, 0.0]}
                mats[tok[1]] = cur
            elif cur is not None and tok[0] in ("Kd", "Ks", "Ke"):
                # TinyOBJ matches the key case-sensitively; "kE" lines are ignored.
                cur[tok[0]] = [float(x) for x in tok[1:4]]
            elif cur is not None and tok[0] == "Ns":
                cur["Ns"] = float(tok[1])
    return mats


def _fix_index(idx, n):
    # tiny_obj_loader.cc fixIndex: positive -> idx-1, negative -> n+idx
    return idx - 1 if idx > 0 else (n + idx if idx < 0 else 0)


def load_obj(path):
    """Returns (verts f32[n,3], tris u32[m,4] = i0,i1,i2,mat_id, materials f32[k,12], mat_names).

    materials rows follow rt::PhongMaterial (Source/RT/RTBase.h:72-77):
    k_d(3) pad k_s(3) n_s k_e(3) pad.
    """
    base = os.path.dirname(path)
    v = []
    mats = {}
    material = {"name": "", "Kd": [0.0] * 3, "Ks": [0.0] * 3, "Ns": 1.0, "Ke": [0.0] * 3}
    shapes = []
    face_group = []

    def flush():
        if not face_group:
            return
        positions, cache, indices = [], {}, []
        for face in face_group:
            for k in range(2, len(face)):
                for vi in (face[0], face[k - 1], face[k]):
                    if vi not in cache:
                        cache[vi] = len(positions)
                        positions.append(v[vi])
                    indices.append(cache[vi])
        shapes.append({"positions": positions, "indices": indices, "material": dict(material)})

    with open(path, "r") as f:
        for line in f:
            tok = line.split()
            if not tok or tok[0].startswith("#"):
                continue
            if tok[0] == "v":
                v.append(tuple(np.float32(float(x)) for x in tok[1:4]))
            elif tok[0] == "f":
                face = []
                for t in tok[1:]:
                    face.append(_fix_index(int(t.split("/")[0]), len(v)))
                face_group.append(face)
            elif tok[0] == "usemtl":
                material = mats.get(tok[1], {"name": "", "Kd": [0.0] * 3, "Ks": [0.0] * 3, "Ns": 1.0,
                                             "Ke": [0.0] * 3})
            elif tok[0] == "mtllib":
                mats = _load_mtl(os.path.join(base, tok[1]))
            elif tok[0] in ("g", "o"):
                flush()
                face_group = []
    flush()
    if not shapes:
        raise ValueError("Object file loading failed!")  # Scene.cpp:30-33
    # Scene.cpp:35-39: std::sort by material name; libstdc++ uses insertion sort below 16
    # elements, i.e. a stable order.  Python's sort is stable for any size.
    shapes.sort(key=lambda s: s["material"]["name"])
    verts, tris, materials, names = [], [], [], []
    cur_name = None
    for s in shapes:
        m = s["material"]
        if cur_name is None or cur_name != m["name"]:
            cur_name = m["name"]
            names.append(cur_name)
            rho_d = np.array(m["Kd"], np.float32)
            rho_s = np.array(m["Ks"], np.float32)
            ssum = rho_d + rho_s
            if (ssum > np.float32(1.0)).any():           # Scene.cpp:105-109
                rho_d = rho_d / ssum
                rho_s = rho_s / ssum
            n_s = np.float32(m["Ns"])
            k_d = rho_d * INV_PI                         # Scene.cpp:110
            k_s = ((rho_s * (n_s + np.float32(2.0))) * np.float32(0.5)) * INV_PI
            k_e = np.array(m["Ke"], np.float32)
            materials.append(np.concatenate([k_d, [0], k_s, [n_s], k_e, [0]]).astype(np.float32))
        mat_id = len(names) - 1
        start = len(verts)                               # Scene.cpp:136-153
        verts.extend(s["positions"])
        idx = s["indices"]
        for k in range(0, len(idx), 3):
            tris.append((start + idx[k], start + idx[k + 1], start + idx[k + 2], mat_id))
    return (np.array(verts, np.float32).reshape(-1, 3), np.array(tris, np.uint32).reshape(-1, 4),
            np.array(materials, np.float32).reshape(-1, 12), names)


def read_3dt(path, dtype=np.uint8):
    """Reads a `.3dt` volume; returns array [z][y][x] (x fastest)."""
    with open(path, "rb") as f:
        res = np.frombuffer(f.read(12), dtype="<i4")
        data = np.frombuffer(f.read(), dtype=dtype)
    rx, ry, rz = int(res[0]), int(res[1]), int(res[2])
    if data.size != rx * ry * rz:
        raise ValueError("3dt payload size does not match its header")
    return data.reshape(rz, ry, rx).copy()


def write_3dt(path, vol):
    vol = np.ascontiguousarray(vol)
    rz, ry, rx = vol.shape
    with open(path, "wb") as f:
        f.write(np.array([rx, ry, rz], "<i4").tobytes())
        f.write(vol.tobytes())


# Defaults of the reference application (Source/GIGL.cpp:36-41, Source/Common/Constants.h).
CAM_POS = (278.0, 273.0, -800.0)
CAM_DIR = (0.0, 0.0, 1.0)
CAM_UP = (0.0, 1.0, 0.0)
CAM_HFOV = float(np.float32(0.215) * np.float32(3.14159274))
WINDOW_RES = 1024
KD_PARAMS = dict(inters_cost=10, trav_cost=1, max_depth=30, min_num_prims=2, empty_bonus=0.25)
PRIM_PL_POS = (278.2, 600.0, 279.5)
PRIM_PL_INTENS = (85000.0, 75000.0, 75000.0)
ABS_K, SCA_K, MAJ_EXT_K = 1e-5, 1e-2, 0.010001
MAX_FOG_HEIGHT = 548.8
MAX_N_VPLS = 150
