"""Polygon predicates and areas for the detection metric (the reference uses shapely, which is not a dependency here).

Intersection area of two simple polygons through signed fan triangulation: 1_A = sum_i s_i 1_{T_i} almost everywhere, so
area(A & B) = sum_ij s_i s_j area(T_i & T_j); triangle/triangle intersections are convex clips (Sutherland-Hodgman)."""
import numpy as np


def _area2(p):
    x, y = p[:, 0], p[:, 1]
    return float(np.dot(x, np.roll(y, -1)) - np.dot(y, np.roll(x, -1)))


def area(poly):
    return abs(_area2(np.asarray(poly, np.float64))) * 0.5


def _clip(subject, a, b):
    """keep the part of `subject` (list of points) on the left of the directed line a->b"""
    out = []
    n = len(subject)
    for i in range(n):
        p, q = subject[i], subject[(i + 1) % n]
        sp = (b[0] - a[0]) * (p[1] - a[1]) - (b[1] - a[1]) * (p[0] - a[0])
        sq = (b[0] - a[0]) * (q[1] - a[1]) - (b[1] - a[1]) * (q[0] - a[0])
        if sp >= 0:
            out.append(p)
        if (sp > 0 and sq < 0) or (sp < 0 and sq > 0):
            t = sp / (sp - sq)
            out.append((p[0] + t * (q[0] - p[0]), p[1] + t * (q[1] - p[1])))
    return out


def _tri_inter_area(t1, t2):
    if _area2(np.asarray(t2)) < 0:
        t2 = t2[::-1]
    poly = list(t1)
    for i in range(3):
        if not poly:
            return 0.0
        poly = _clip(poly, t2[i], t2[(i + 1) % 3])
    return abs(_area2(np.asarray(poly))) * 0.5 if len(poly) >= 3 else 0.0


def _fan(poly):
    p = np.asarray(poly, np.float64)
    tris = []
    for i in range(1, len(p) - 1):
        t = [tuple(p[0]), tuple(p[i]), tuple(p[i + 1])]
        a2 = _area2(np.asarray(t))
        if a2 != 0:
            tris.append((1.0 if a2 > 0 else -1.0, t))
    return tris


def intersection_area(pa, pb):
    ta, tb = _fan(pa), _fan(pb)
    sa = 1.0 if _area2(np.asarray(pa, np.float64)) >= 0 else -1.0
    sb = 1.0 if _area2(np.asarray(pb, np.float64)) >= 0 else -1.0
    s = 0.0
    for s1, t1 in ta:
        for s2, t2 in tb:
            s += (s1 * sa) * (s2 * sb) * _tri_inter_area(t1, t2)
    return max(s, 0.0)


def union_area(pa, pb):
    return area(pa) + area(pb) - intersection_area(pa, pb)


def _seg_intersect(p1, p2, p3, p4):
    def orient(a, b, c):
        v = (b[0] - a[0]) * (c[1] - a[1]) - (b[1] - a[1]) * (c[0] - a[0])
        return (v > 0) - (v < 0)

    def on(a, b, c):
        return min(a[0], b[0]) <= c[0] <= max(a[0], b[0]) and min(a[1], b[1]) <= c[1] <= max(a[1], b[1])
    o1, o2, o3, o4 = orient(p1, p2, p3), orient(p1, p2, p4), orient(p3, p4, p1), orient(p3, p4, p2)
    if o1 != o2 and o3 != o4:
        return True
    return (o1 == 0 and on(p1, p2, p3)) or (o2 == 0 and on(p1, p2, p4)) or (o3 == 0 and on(p3, p4, p1)) or (o4 == 0 and on(p3, p4, p2))


def is_valid_simple(poly):
    """Polygon(points).is_valid and .is_simple of the reference: >= 3 vertices, non-zero area, no two non-adjacent edges
    touch or cross, no repeated vertex."""
    p = [tuple(map(float, q)) for q in np.asarray(poly).reshape(-1, 2)]
    if len(p) >= 2 and p[0] == p[-1]:
        p = p[:-1]
    n = len(p)
    if n < 3 or _area2(np.asarray(p)) == 0 or len(set(p)) != n:
        return False
    for i in range(n):
        for j in range(i + 1, n):
            if j == i + 1 or (i == 0 and j == n - 1):
                continue
            if _seg_intersect(p[i], p[(i + 1) % n], p[j], p[(j + 1) % n]):
                return False
    return True
