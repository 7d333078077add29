"""Katz hidden-point removal as the reference uses it (DepthPrompting.py:273-290:
open3d ``PointCloud.hidden_point_removal(camera, radius)`` for each viewpoint) -- CPU restatement.

TEST INFRASTRUCTURE ONLY (like the rest of oracle/).  open3d is absent and unpinned; its published
algorithm (geometry/PointCloud.cpp, HiddenPointRemoval) is restated: spherical flipping
``p' = v + 2 (radius - |v|) v / |v|`` with ``v = p - camera`` (|v| = 0 -> 1e-4), the origin appended, convex
hull of the flipped set by **qhull** (the library open3d calls; here through scipy.spatial.ConvexHull),
visible = hull vertices other than the appended origin.  Which near-coplanar points qhull keeps as
vertices depends on its options and version, so this is a reference for COUNTS and for the selected
view, not for bit-level parity.
"""
import numpy as np


def hidden_point_removal(points, camera, radius):
    """-> indices of the points visible from `camera` (ascending)."""
    from scipy.spatial import ConvexHull
    v = np.asarray(points, np.float64) - np.asarray(camera, np.float64)
    norm = np.linalg.norm(v, axis=1)
    norm[norm == 0] = 0.0001
    flipped = v + 2.0 * (radius - norm)[:, None] * v / norm[:, None]
    flipped = np.vstack([flipped, np.zeros((1, 3))])
    hull = ConvexHull(flipped)
    vis = hull.vertices[hull.vertices != len(points)]
    return np.sort(vis)


def visible_counts(points, viewpoints, radius):
    """Number of visible points per viewpoint (DepthPrompting.viewpoint_select sums the mask)."""
    return np.array([len(hidden_point_removal(points, c, radius)) for c in viewpoints], np.int64)
