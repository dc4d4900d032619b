"""genome-on-diet_amd: MI355X (gfx950) implementation of Genome-on-Diet's per-read mapping hot path.

The product is the C-ABI shared library ``libgdiet_hip.so`` (include/gdiet_hip.h); this package is the thin
Python host mirror used by the tests and the benchmark.  There is no CPU fallback: if the library is missing or no
gfx950 device is present, construction of :class:`Context` raises.

The directory name carries a hyphen, so import it with ``importlib`` (see ``tests/conftest.py:load_pkg``) --
``__graft_entry__.py`` does exactly that.
"""
from .hip_abi import Context, GdietError, KswScore, library_path, load_library, pack, PRESET_SCORES  # noqa: F401
from .map_api import Mapper, MapOpt, PRESETS as MAP_PRESETS, map_multi, read_ranges_by_cost_c  # noqa: F401,E402
from .fastx import FastxReader  # noqa: F401,E402
from .shard import JobClock, effective_cpus, rank_seed, read_range, read_ranges_by_cost  # noqa: F401,E402
