import sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[1] / "tests"))
from test_ray_ops_gpu import dense_rays  # noqa: F401,E402
