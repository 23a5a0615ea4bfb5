"""Synthetic inputs of the shapes BASELINE.json names.

The reference ships only the Cornell box, so the 1M / 10M-triangle scenes and the incoherent
bounce-ray batches are generated here.  They are INPUTS: the same arrays are handed to the HIP
engine and to the checker, so nothing in this file has a parity obligation of its own.

Scene S(k): each Cornell quad (two consecutive triangles of the loaded scene) is split into
k x k bilinear sub-quads, two triangles each in TinyOBJ fan order (0,1,2),(0,2,3); interior grid
vertices are displaced along the quad normal by (h - 0.5) with h a 32-bit integer hash in [0,1),
so the tessellation is not coplanar.  k=158 -> 998,560 triangles, k=500 -> 10,000,000.

Rays R(n, seed): ray i picks a uniform triangle, a uniform point on it and a cosine-weighted
direction about the triangle normal (spanTangentSpace * genCosRandHemiDir,
/root/reference/Source/RT/RTBase.h:33-50, RTBase.cpp:27-36), built with the origin-lifting
3-argument Ray constructor (RTBase.cpp:117-122).  No sorting: the batch is incoherent.
"""
import numpy as np

F32 = np.float32
SCENE_SEED = 0x5EED0001
RAY_SEED_16M = 0xC0FFEE03
RAY_SEED_128M = 0xC0FFEE04
FLT_MAX = np.finfo(np.float32).max


def _hash32(x):
    """lowbias32 integer hash on uint32 arrays."""
    x = np.asarray(x, dtype=np.uint32).copy()
    x ^= x >> np.uint32(16)
    x *= np.uint32(0x7FEB352D)
    x ^= x >> np.uint32(15)
    x *= np.uint32(0x846CA68B)
    x ^= x >> np.uint32(16)
    return x


def _unit(h):
    return (h >> np.uint32(8)).astype(F32) * F32(1.0 / (1 << 24))


def tessellate(verts, tris, k, amplitude=1.0, seed=SCENE_SEED):
    """S(k) from a quad scene given as consecutive triangle pairs (0,1,2),(0,2,3)."""
    verts = np.asarray(verts, F32)
    tris = np.asarray(tris, np.uint32)
    assert len(tris) % 2 == 0
    n_quads = len(tris) // 2
    out_v, out_t = [], []
    g = np.arange(k + 1, dtype=np.uint32)
    ii, jj = np.meshgrid(g, g, indexing="xy")          # ii varies fastest
    s = (ii.astype(F32) / F32(k))[..., None]
    t = (jj.astype(F32) / F32(k))[..., None]
    interior = ((ii > 0) & (ii < k) & (jj > 0) & (jj < k))[..., None]
    ci, cj = np.meshgrid(np.arange(k, dtype=np.uint32), np.arange(k, dtype=np.uint32), indexing="xy")
    v00 = (cj * (k + 1) + ci).ravel()
    v10, v11, v01 = v00 + 1, v00 + (k + 1) + 1, v00 + (k + 1)
    base = 0
    for q in range(n_quads):
        ta, tb = tris[2 * q], tris[2 * q + 1]
        assert ta[0] == tb[0] and ta[2] == tb[1], "scene is not made of fan-ordered quads"
        c0, c1, c2, c3 = verts[ta[0]], verts[ta[1]], verts[ta[2]], verts[tb[2]]
        one = F32(1.0)
        p = ((one - s) * (one - t)) * c0 + (s * (one - t)) * c1 + (s * t) * c2 + ((one - s) * t) * c3
        nrm = np.cross(c1 - c0, c3 - c0).astype(F32)
        nrm = nrm / F32(np.sqrt((nrm * nrm).sum(dtype=F32)))
        h = _hash32(np.uint32(seed) ^ _hash32(np.uint32((q * 0x9E3779B1) & 0xFFFFFFFF)
                                              + ii * np.uint32(0x85EBCA77) + jj * np.uint32(0xC2B2AE3D)))
        disp = (F32(amplitude) * (_unit(h) - F32(0.5)))[..., None]
        p = np.where(interior, p + disp * nrm, p).astype(F32)
        out_v.append(p.reshape(-1, 3))
        mat = np.full(k * k, ta[3], np.uint32)
        b = np.uint32(base)
        out_t.append(np.stack([np.stack([b + v00, b + v10, b + v11, mat], 1),
                               np.stack([b + v00, b + v11, b + v01, mat], 1)], 1).reshape(-1, 4))
        base += (k + 1) * (k + 1)
    return np.concatenate(out_v).astype(F32), np.concatenate(out_t).astype(np.uint32)


def bounce_rays(verts, tris, n, seed, first=0, chunk=1 << 20, n_threads=1):
    """R(n, seed) rays [first, first+n) as (o_tmin f32[n,4], d_tmax f32[n,4]).  Ray i depends on
    (seed, first + i) only, so chunks are independent: n_threads > 1 generates them on a thread pool
    (NumPy releases the GIL in its array loops) with the same result."""
    verts = np.asarray(verts, F32)
    tris = np.asarray(tris, np.uint32)
    n_tris = len(tris)
    ro = np.empty((n, 4), F32)
    rd = np.empty((n, 4), F32)

    def one(lo):
        m = min(chunk, n - lo)
        idx = (np.arange(lo, lo + m, dtype=np.uint64) + np.uint64(first))
        c = _hash32((idx & np.uint64(0xFFFFFFFF)).astype(np.uint32) ^ np.uint32(seed))
        c ^= _hash32((idx >> np.uint64(32)).astype(np.uint32) + np.uint32(0x632BE5AB))
        h = [_hash32(c + np.uint32(0x9E3779B9 * (j + 1) & 0xFFFFFFFF)) for j in range(5)]
        tri = ((h[0].astype(np.uint64) * np.uint64(n_tris)) >> np.uint64(32)).astype(np.int64)
        b1, b2 = _unit(h[1]), _unit(h[2])
        flip = (b1 + b2) > F32(1.0)
        b1 = np.where(flip, F32(1.0) - b1, b1)
        b2 = np.where(flip, F32(1.0) - b2, b2)
        u1, u2 = _unit(h[3]), _unit(h[4])
        p0 = verts[tris[tri, 0]]
        e0 = verts[tris[tri, 1]] - p0
        e1 = verts[tris[tri, 2]] - p0
        pt = p0 + b1[:, None] * e0 + b2[:, None] * e1
        N = np.cross(e0, e1).astype(F32)
        N = N * (F32(1.0) / np.sqrt((N * N).sum(1, dtype=F32)))[:, None]
        # spanTangentSpace(N)
        c1 = np.cross(N, np.array([0, 0, 1], F32)).astype(F32)
        c2 = np.cross(N, np.array([0, 1, 0], F32)).astype(F32)
        l1 = np.sqrt((c1 * c1).sum(1, dtype=F32))
        l2 = np.sqrt((c2 * c2).sum(1, dtype=F32))
        c3 = np.where((l1 > l2)[:, None], c1, c2)
        tang = -c3
        tang = tang * (F32(1.0) / np.sqrt((tang * tang).sum(1, dtype=F32)))[:, None]
        bitan = np.cross(N, tang).astype(F32)
        # genCosRandHemiDir(u1, u2)
        phi = F32(6.28318548) * u1
        sin_t = np.sqrt(F32(1.0) - u2)
        mx, my, mz = sin_t * np.cos(phi), sin_t * np.sin(phi), np.sqrt(u2)
        d = tang * mx[:, None] + bitan * my[:, None] + N * mz[:, None]
        d = d * (F32(1.0) / np.sqrt((d * d).sum(1, dtype=F32)))[:, None]
        ro[lo:lo + m, :3] = pt + F32(1e-4) * N
        ro[lo:lo + m, 3] = 0.0
        rd[lo:lo + m, :3] = d
        rd[lo:lo + m, 3] = FLT_MAX

    starts = list(range(0, n, chunk))
    if n_threads > 1 and len(starts) > 1:
        from concurrent.futures import ThreadPoolExecutor
        with ThreadPoolExecutor(max_workers=n_threads) as ex:
            list(ex.map(one, starts))
    else:
        for lo in starts:
            one(lo)
    return ro, rd
