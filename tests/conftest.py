import os
import sys

import numpy as np
import pytest

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def pytest_collection_modifyitems(config, items):
    # `-m gpu` tests must FAIL, not skip, when the HIP path is unavailable on a GPU
    # box; on a CPU-only box they are deselected by `-m "not gpu"`.
    pass


@pytest.fixture(scope="session")
def oracle():
    from oracle import oracle as O
    O.build()
    return O


@pytest.fixture(scope="session")
def golden():
    def load(name):
        return np.load(os.path.join(GOLDEN, name), allow_pickle=False)
    return load


def gen_pair(seed, s1, s2, shift=0.5):
    """The survey's input convention: x1 then x2 from one default_rng(seed)."""
    rng = np.random.default_rng(seed)
    a = rng.random(s1, dtype=np.float32) - np.float32(shift)
    b = rng.random(s2, dtype=np.float32) - np.float32(shift)
    return a, b


def write_glb(path, verts, faces, colors=None, node=None, indices_u16=True):
    """Minimal binary glTF writer for the tests (one mesh, one primitive, one node)."""
    import json
    import struct
    v = np.asarray(verts, "<f4")
    idx = np.asarray(faces, "<u2" if indices_u16 else "<u4").reshape(-1)
    blobs = [v.tobytes(), idx.tobytes() + b"\x00" * (-len(idx.tobytes()) % 4)]
    views = [{"buffer": 0, "byteOffset": 0, "byteLength": len(blobs[0])},
             {"buffer": 0, "byteOffset": len(blobs[0]), "byteLength": len(idx.tobytes())}]
    accs = [{"bufferView": 0, "componentType": 5126, "count": len(v), "type": "VEC3"},
            {"bufferView": 1, "componentType": 5123 if indices_u16 else 5125, "count": len(idx), "type": "SCALAR"}]
    attrs = {"POSITION": 0}
    if colors is not None:
        c = np.asarray(colors, "<f4")
        views.append({"buffer": 0, "byteOffset": sum(map(len, blobs)), "byteLength": c.nbytes})
        blobs.append(c.tobytes())
        accs.append({"bufferView": 2, "componentType": 5126, "count": len(c), "type": "VEC3"})
        attrs["COLOR_0"] = 2
    n = dict(node or {})
    n["mesh"] = 0
    g = {"asset": {"version": "2.0"}, "scene": 0, "scenes": [{"nodes": [0]}], "nodes": [n],
         "meshes": [{"primitives": [{"attributes": attrs, "indices": 1}]}], "accessors": accs, "bufferViews": views,
         "buffers": [{"byteLength": sum(map(len, blobs))}]}
    js = json.dumps(g).encode()
    js += b" " * (-len(js) % 4)
    binc = b"".join(blobs)
    total = 12 + 8 + len(js) + 8 + len(binc)
    with open(path, "wb") as f:
        f.write(struct.pack("<4sII", b"glTF", 2, total))
        f.write(struct.pack("<I4s", len(js), b"JSON") + js)
        f.write(struct.pack("<I4s", len(binc), b"BIN\x00") + binc)
