"""Procedural triangle meshes for the SDF bake (BASELINE config 1): closed surfaces with a known analytic distance.

Winding: the bake derives the triangle normal as normalize(cross(v0 - v2, v0 - v1)) (reference AssetPipeline/SceneSDF.cpp:273),
the negative of the counter-clockwise normal, so triangles are emitted clockwise seen from outside and that normal points outward.
"""
import numpy as np


def _finish(positions, tris, outward_ref=None):
    positions = np.asarray(positions, np.float32)
    tris = np.asarray(tris, np.uint32).reshape(-1, 3)
    v0, v1, v2 = positions[tris[:, 0]], positions[tris[:, 1]], positions[tris[:, 2]]
    n = np.cross(v0 - v2, v0 - v1)
    centre = (v0 + v1 + v2) / 3.0
    ref = centre - (positions.mean(axis=0) if outward_ref is None else outward_ref(centre))
    flip = (n * ref).sum(axis=1) < 0
    tris[flip] = tris[flip][:, [0, 2, 1]]
    keep = np.linalg.norm(n, axis=1) > 1e-12  # drop the degenerate triangles at the poles
    return positions, tris[keep].reshape(-1)


def uv_sphere(radius=1.0, segments=24, rings=12, centre=(0.0, 0.0, 0.0)):
    pos = []
    for r in range(rings + 1):
        th = np.pi * r / rings
        for s in range(segments):
            ph = 2.0 * np.pi * s / segments
            pos.append((radius * np.sin(th) * np.cos(ph), radius * np.cos(th), radius * np.sin(th) * np.sin(ph)))
    tris = []
    for r in range(rings):
        for s in range(segments):
            a = r * segments + s; b = r * segments + (s + 1) % segments
            c = (r + 1) * segments + s; d = (r + 1) * segments + (s + 1) % segments
            tris += [(a, b, c), (b, d, c)]
    p = np.asarray(pos, np.float32) + np.asarray(centre, np.float32)
    return _finish(p, tris)


def box(half=(1.0, 1.0, 1.0), centre=(0.0, 0.0, 0.0), subdiv=2):
    """Axis-aligned box, each face a subdiv x subdiv grid of quads."""
    half = np.asarray(half, np.float32)
    pos, tris = [], []
    for axis in range(3):
        for sgn in (-1.0, 1.0):
            u, v = (axis + 1) % 3, (axis + 2) % 3
            base = len(pos)
            for i in range(subdiv + 1):
                for j in range(subdiv + 1):
                    p = np.zeros(3, np.float32)
                    p[axis] = sgn * half[axis]
                    p[u] = (2.0 * i / subdiv - 1.0) * half[u]
                    p[v] = (2.0 * j / subdiv - 1.0) * half[v]
                    pos.append(p)
            for i in range(subdiv):
                for j in range(subdiv):
                    a = base + i * (subdiv + 1) + j; b = a + 1; c = a + subdiv + 1; d = c + 1
                    tris += [(a, b, c), (b, d, c)]
    p = np.asarray(pos, np.float32) + np.asarray(centre, np.float32)
    return _finish(p, tris)


def torus(major=1.5, minor=0.5, segments=24, sides=12, centre=(0.0, 0.0, 0.0)):
    pos, tris = [], []
    for i in range(segments):
        a = 2.0 * np.pi * i / segments
        for j in range(sides):
            b = 2.0 * np.pi * j / sides
            r = major + minor * np.cos(b)
            pos.append((r * np.cos(a), minor * np.sin(b), r * np.sin(a)))
    for i in range(segments):
        for j in range(sides):
            a = i * sides + j; b = i * sides + (j + 1) % sides
            c = ((i + 1) % segments) * sides + j; d = ((i + 1) % segments) * sides + (j + 1) % sides
            tris += [(a, b, c), (b, d, c)]
    c3 = np.asarray(centre, np.float32)
    p = np.asarray(pos, np.float32)

    def ring_centre(q):
        q = q.copy()
        xz = q[:, [0, 2]]
        n = np.linalg.norm(xz, axis=1, keepdims=True)
        out = np.zeros_like(q)
        out[:, [0, 2]] = xz / np.maximum(n, 1e-9) * major
        return out
    positions, idx = _finish(p, tris, outward_ref=ring_centre)
    return positions + c3, idx


def bounds(positions):
    p = np.asarray(positions, np.float32)
    return p.min(axis=0), p.max(axis=0)
