"""Drop-in modules for the reference's third_lib extensions: `dvr`, `dvxlr`, `dvxlr_v2`
(third_lib/dvr/dvr.cpp, third_lib/dvxlr/dvxlr.cpp, third_lib/dvxlr/dvxlr_v2.cpp) and `chamferdist`.
Import them as  `from vidar_amd.third_lib import dvr, dvxlr, dvxlr_v2`."""
from . import dvr, dvxlr, dvxlr_v2  # noqa: F401
