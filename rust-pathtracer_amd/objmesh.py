"""Minimal Wavefront OBJ reader with the semantics the reference gets from tobj
(src/parsing/meshes.rs:17-157: LoadOptions{single_index: true, triangulate: true}):

* one model per `o`/`g` group and per `usemtl` change inside it,
* a unique vertex per distinct (v, vt, vn) index triple, in order of first use,
* polygons fan-triangulated (0,1,2), (0,2,3), ...
"""
import numpy as np


class ObjModel:
    def __init__(self, name, material):
        self.name = name
        self.material = material  # usemtl name or None
        self.positions = []
        self.normals = []
        self.indices = []
        self._remap = {}

    def arrays(self):
        p = np.asarray(self.positions, dtype=np.float32).reshape(-1, 3)
        n = np.asarray(self.normals, dtype=np.float32).reshape(-1, 3) if self.normals else None
        i = np.asarray(self.indices, dtype=np.uint32).reshape(-1, 3)
        return p, n, i


def load_obj(path):
    v, vn = [], []
    models = []
    cur = None
    name, mat = "unnamed_object", None

    def start():
        nonlocal cur
        cur = ObjModel(name, mat)
        models.append(cur)

    for line in open(path):
        t = line.split()
        if not t or t[0].startswith("#"):
            continue
        if t[0] == "v":
            v.append([float(x) for x in t[1:4]])
        elif t[0] == "vn":
            vn.append([float(x) for x in t[1:4]])
        elif t[0] in ("o", "g"):
            name = t[1] if len(t) > 1 else "unnamed_object"
            cur = None
        elif t[0] == "usemtl":
            mat = t[1]
            cur = None
        elif t[0] == "f":
            if cur is None:
                start()
            corner = []
            for c in t[1:]:
                parts = c.split("/")
                vi = int(parts[0]); vi = vi - 1 if vi > 0 else len(v) + vi
                ni = None
                if len(parts) > 2 and parts[2]:
                    ni = int(parts[2]); ni = ni - 1 if ni > 0 else len(vn) + ni
                ti = parts[1] if len(parts) > 1 else ""
                key = (vi, ti, ni)
                idx = cur._remap.get(key)
                if idx is None:
                    idx = len(cur.positions)
                    cur._remap[key] = idx
                    cur.positions.append(v[vi])
                    if ni is not None:
                        cur.normals.append(vn[ni])
                corner.append(idx)
            for k in range(1, len(corner) - 1):
                cur.indices.append([corner[0], corner[k], corner[k + 1]])
    return models
