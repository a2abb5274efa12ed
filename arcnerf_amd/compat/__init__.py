"""Drop-in module shims named like the reference's native dependencies.  Put `arcnerf_amd/compat` on sys.path (or call
arcnerf_amd.compat.install()) and `import _volume_func` / `import tinycudann` resolve to these HIP-backed modules, so
arcnerf/ops/*.py and the tcnn-backed encoders/networks of an unmodified ArcNerf checkout run on MI355X."""
import os
import sys


def install():
    here = os.path.dirname(os.path.abspath(__file__))
    if here not in sys.path:
        sys.path.insert(0, here)
