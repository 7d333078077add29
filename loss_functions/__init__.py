"""Top-level alias so that the reference's drivers (`from loss_functions import
chamfer_3DDist, emdModule`, utils/loss_util.py:6) import the gfx950 implementation
unchanged when this repository is first on sys.path.  See INTEGRATION.md."""
from genpc_amd.loss_functions import chamfer_3DDist, emdModule

__all__ = ["chamfer_3DDist", "emdModule"]
