"""Point-cloud plumbing either side of the NDT hot path (host side, numpy).

* binary/ascii PCD v0.7 reader + binary writer -- the on-disk format of the
  reference's fixtures (ndt_omp/data/*.pcd) and of lidar_subscriber_node
  (lidar_subscriber/src/lidar_subscriber_node.cpp:35-54).
* centroid voxel down-sampling -- what every caller does before NDT
  (ndt_omp/apps/align.cpp:60-69, ndt_omp_mapping_node.cpp:142-148).
* the synthetic workloads of SURVEY.md section 8(d) (sets "U" and "S").
"""
import numpy as np


# --------------------------------------------------------------------------- PCD
def read_pcd(path):
    """Return (N, F) float32 array of the PCD's float fields and the field names."""
    with open(path, "rb") as f:
        header = {}
        while True:
            line = f.readline()
            if not line:
                raise ValueError("PCD: no DATA line")
            s = line.decode("ascii", "replace").strip()
            if not s or s.startswith("#"):
                continue
            k, _, v = s.partition(" ")
            header[k.upper()] = v.split()
            if k.upper() == "DATA":
                break
        fields = header["FIELDS"]
        sizes = [int(x) for x in header["SIZE"]]
        types = header["TYPE"]
        counts = [int(x) for x in header.get("COUNT", ["1"] * len(fields))]
        n = int(header["POINTS"][0]) if "POINTS" in header else int(header["WIDTH"][0]) * int(header["HEIGHT"][0])
        kind = header["DATA"][0].lower()
        np_t = {("F", 4): "<f4", ("F", 8): "<f8", ("U", 1): "u1", ("U", 2): "<u2", ("U", 4): "<u4",
                ("I", 1): "i1", ("I", 2): "<i2", ("I", 4): "<i4"}
        dt = np.dtype([(nm, np_t[(t, s)], (c,)) if c != 1 else (nm, np_t[(t, s)])
                       for nm, t, s, c in zip(fields, types, sizes, counts)])
        if kind == "binary":
            raw = np.frombuffer(f.read(n * dt.itemsize), dtype=dt, count=n)
        elif kind == "ascii":
            raw = np.loadtxt(f, dtype=np.float64, ndmin=2)
            out = raw.astype(np.float32)
            return out, fields
        else:
            raise ValueError("PCD: DATA %s not supported" % kind)
    out = np.stack([raw[nm].astype(np.float32) for nm in fields], axis=1)
    return out, fields


def write_pcd_xyz(path, xyz):
    xyz = np.ascontiguousarray(xyz[:, :3], dtype="<f4")
    n = xyz.shape[0]
    hdr = ("# .PCD v0.7 - Point Cloud Data file format\nVERSION 0.7\nFIELDS x y z\nSIZE 4 4 4\nTYPE F F F\n"
           "COUNT 1 1 1\nWIDTH %d\nHEIGHT 1\nVIEWPOINT 0 0 0 1 0 0 0\nPOINTS %d\nDATA binary\n" % (n, n))
    with open(path, "wb") as f:
        f.write(hdr.encode("ascii"))
        f.write(xyz.tobytes())


# ------------------------------------------------------------- voxel down-sample
def voxel_downsample(xyz, leaf):
    """Centroid per occupied voxel, output ordered by ascending linear voxel index
    (the order pcl::VoxelGrid emits).  Index math in f32 like PCL
    (floor(x * inv_leaf) - min_b); centroid accumulated in f64, rounded to f32."""
    xyz = np.ascontiguousarray(xyz[:, :3], dtype=np.float32)
    inv = np.float32(1.0) / np.float32(leaf)
    mn = xyz.min(axis=0)
    mx = xyz.max(axis=0)
    min_b = np.floor(mn * inv).astype(np.int64)
    max_b = np.floor(mx * inv).astype(np.int64)
    div = max_b - min_b + 1
    ijk = (np.floor(xyz * inv) - min_b.astype(np.float32)).astype(np.int64)
    key = ijk[:, 0] + ijk[:, 1] * div[0] + ijk[:, 2] * div[0] * div[1]
    order = np.argsort(key, kind="stable")
    key_s = key[order]
    uniq, start, cnt = np.unique(key_s, return_index=True, return_counts=True)
    sums = np.add.reduceat(xyz[order].astype(np.float64), start, axis=0)
    return (sums / cnt[:, None]).astype(np.float32)


# ------------------------------------------------------------ rigid transforms
def rot_xyz(rx, ry, rz):
    cx, sx, cy, sy, cz, sz = np.cos(rx), np.sin(rx), np.cos(ry), np.sin(ry), np.cos(rz), np.sin(rz)
    Rx = np.array([[1, 0, 0], [0, cx, -sx], [0, sx, cx]])
    Ry = np.array([[cy, 0, sy], [0, 1, 0], [-sy, 0, cy]])
    Rz = np.array([[cz, -sz, 0], [sz, cz, 0], [0, 0, 1]])
    return Rx @ Ry @ Rz


def make_T(t, rpy):
    T = np.eye(4)
    T[:3, :3] = rot_xyz(*rpy)
    T[:3, 3] = t
    return T


T_GT_DEFAULT = make_T([0.30, -0.20, 0.10], np.deg2rad([0.5, -0.3, 1.0]))


def apply_T(T, xyz):
    return (xyz.astype(np.float64) @ T[:3, :3].T + T[:3, 3]).astype(np.float32)


# ------------------------------------------------------------ synthetic clouds
SEED = 20250614


def target_uniform(m, seed=SEED, half=(50.0, 50.0, 5.0)):
    """Set "U" target: m points uniform in [-50,50]x[-50,50]x[-5,5] m."""
    rng = np.random.default_rng(seed)
    h = np.asarray(half)
    return (rng.random((m, 3)) * 2 * h - h).astype(np.float32)


def target_surfaces(m, seed=SEED, extent=100.0, n_boxes=60):
    """Set "S" target: ground plane (40 %) + axis-aligned box walls (60 %)."""
    rng = np.random.default_rng(seed)
    n_ground = int(m * 0.4)
    half = extent / 2
    ground = np.stack([rng.uniform(-half, half, n_ground), rng.uniform(-half, half, n_ground),
                       rng.normal(0, 0.01, n_ground)], axis=1)
    n_wall = m - n_ground
    cx = rng.uniform(-half * 0.9, half * 0.9, n_boxes)
    cy = rng.uniform(-half * 0.9, half * 0.9, n_boxes)
    wx = rng.uniform(4, 20, n_boxes) * extent / 100.0
    wy = rng.uniform(4, 20, n_boxes) * extent / 100.0
    hz = rng.uniform(3, 10, n_boxes)
    # area-weighted choice of box, then of one of its 4 walls
    area = 2 * (wx + wy) * hz
    box = rng.choice(n_boxes, size=n_wall, p=area / area.sum())
    u = rng.random(n_wall)
    v = rng.random(n_wall)
    per = 2 * (wx[box] + wy[box])
    s = u * per
    x = np.empty(n_wall)
    y = np.empty(n_wall)
    bx, by, bwx, bwy = cx[box], cy[box], wx[box], wy[box]
    m0 = s < bwx
    m1 = (~m0) & (s < bwx + bwy)
    m2 = (~m0) & (~m1) & (s < 2 * bwx + bwy)
    m3 = ~(m0 | m1 | m2)
    x[m0] = bx[m0] - bwx[m0] / 2 + s[m0]
    y[m0] = by[m0] - bwy[m0] / 2
    x[m1] = bx[m1] + bwx[m1] / 2
    y[m1] = by[m1] - bwy[m1] / 2 + (s[m1] - bwx[m1])
    x[m2] = bx[m2] + bwx[m2] / 2 - (s[m2] - bwx[m2] - bwy[m2])
    y[m2] = by[m2] + bwy[m2] / 2
    x[m3] = bx[m3] - bwx[m3] / 2
    y[m3] = by[m3] + bwy[m3] / 2 - (s[m3] - 2 * bwx[m3] - bwy[m3])
    z = v * hz[box]
    walls = np.stack([x, y, z], axis=1) + rng.normal(0, 0.01, (n_wall, 3))
    pts = np.concatenate([ground, walls], axis=0)
    rng.shuffle(pts, axis=0)
    return pts.astype(np.float32)


def source_from_target(target, n, T_gt=T_GT_DEFAULT, seed=SEED + 1, noise=0.02):
    """n target points (no replacement) + N(0, noise) noise, moved by T_gt^-1, so that
    registering source->target recovers T_gt."""
    rng = np.random.default_rng(seed)
    sel = rng.choice(target.shape[0], size=n, replace=False)
    pts = target[sel].astype(np.float64) + np.random.default_rng(seed + 1).normal(0, noise, (n, 3))
    Ti = np.linalg.inv(T_gt)
    return (pts @ Ti[:3, :3].T + Ti[:3, 3]).astype(np.float32)


def random_T(rng, max_t=0.5, max_deg=2.0):
    t = rng.uniform(-max_t, max_t, 3)
    r = np.deg2rad(rng.uniform(-max_deg, max_deg, 3))
    return make_T(t, r)


def mapbuild_scan(target, k, n=100000, max_t=0.5, max_deg=2.0):
    """Scan k of the map-build workload (BASELINE configs[3]): n target points + noise moved by its
    own T_gt,k drawn U(+-0.5 m, +-2 deg).  Seeds depend on k only, so the workload is the same however it is split over
    ranks.  -> (scan, T_gt)"""
    T = random_T(np.random.default_rng(SEED + 100 + k), max_t, max_deg)
    return source_from_target(target, n, T_gt=T, seed=SEED + 1000 + 2 * k), T
