"""Same two names the reference's ``loss_functions`` package exports
(loss_functions/__init__.py:2-3)."""
from .Chamfer3D.dist_chamfer_3D import chamfer_3DDist
from .emd.emd_module import emdModule

__all__ = ["chamfer_3DDist", "emdModule"]
