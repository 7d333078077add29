"""The hidden-point removal's alternative paths give the oracle's masks too.

By default almost every point that reaches the wave-per-point kernels is decided by the decision rounds (csrc/hpr.hip,
hpr_decide) and the exact walk -- the definition those rounds prove things about -- runs for one point in a thousand.  The
switches are read once per process, so each variant runs tests/test_gpu_hpr.py (masks against oracle/genpc_oracle_hpr.c and
qhull: random clouds at six radii, real scans, lattices, polygons over 128 / 1024 vertices, edge cases, the 120-configuration
differential fuzz, best-view counts) in a child process of its own.
"""
import os
import subprocess
import sys

import pytest

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))

VARIANTS = {
    "walk_only": {"GENPC_HPR_NOCULL": "128"},                       # no decision rounds: every parked point takes the exact walk
    "rounds_inside_the_wave_kernel": {"GENPC_HPR_DECIDE_KERNEL": "0"},
    "three_home_tiles_and_block_verify": {"GENPC_HPR_HOME_TILES": "3", "GENPC_HPR_NOCULL": "512"},      # round 3's first kernel
    "one_chunk_of_the_home_tile": {"GENPC_HPR_HOME_CHUNKS": "1"},   # the loosest polygons the rounds ever start from
    "one_kernel_form": {"GENPC_HPR_SPLIT": "0", "_deselect": "best_view"},      # (views are only dropped in the two-kernel form)
    "two_kernel_form_everywhere": {"GENPC_HPR_SPLIT": "1"},
    # the polygon store holds 4096 parked points: the others are listed and restart from the box in the wave-per-point pass
    "polygon_store_full": {"GENPC_HPR_SPLIT": "1", "GENPC_HPR_PARK_MB": "1"},
}


@pytest.mark.gpu
@pytest.mark.parametrize("name", sorted(VARIANTS))
def test_hpr_masks_on_every_path(name):
    env = dict(os.environ)
    var = dict(VARIANTS[name])
    skip = var.pop("_deselect", None)
    env.update(var)
    r = subprocess.run([sys.executable, "-m", "pytest", os.path.join(ROOT, "tests", "test_gpu_hpr.py"), "-x", "-q", "-m", "gpu",
                        "-p", "no:cacheprovider"] + (["-k", "not " + skip] if skip else []), cwd=ROOT, env=env, capture_output=True,
                       text=True, timeout=900)
    assert r.returncode == 0, "%s (%s):\n%s\n%s" % (name, VARIANTS[name], r.stdout[-3000:], r.stderr[-1500:])
