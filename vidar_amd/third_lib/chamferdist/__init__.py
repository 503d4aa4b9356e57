"""Drop-in for the reference's pip-installed `chamferdist` package
(third_lib/chamfer_dist/chamferdist/chamferdist/__init__.py)."""
from .chamfer import ChamferDistance, knn_points, knn_gather  # noqa: F401
from . import _C  # noqa: F401
