"""Host-side mirror of the reference's `projects/mmdet3d_plugin` registry surface for the hot path
(same registered type names, constructor kwargs and parameter names).  Importing this package
registers every module, like `importlib.import_module('projects.mmdet3d_plugin')` does in the
reference (tools/train.py:113-137)."""
