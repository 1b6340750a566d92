"""rust-pathtracer_amd — MI355X-native spectral path-tracing hot path.

The product is the C-ABI shared library built from ``csrc/`` (hand-written HIP for gfx950,
entry points declared in ``include/pt_api.h``).  This Python package is plumbing around it:
ctypes bindings (``api``), host-side scene assembly (``scene``) and multi-GPU film sharding
(``sharding``).  There is no CPU fallback: ``load()`` raises if the HIP library is missing.

Import with ``importlib.import_module("rust-pathtracer_amd")`` (the name has a hyphen).
"""
import os

from . import api, scene, scene_file, sharding  # noqa: F401

PACKAGE_DIR = os.path.dirname(os.path.abspath(__file__))
REPO_ROOT = os.path.dirname(PACKAGE_DIR)
# (PT_AMD_LIBRARY: another build of the same engine, for A/B measurements of kernel variants — tools/ab_libs.sh)
LIBRARY_PATH = os.environ.get("PT_AMD_LIBRARY") or os.path.join(PACKAGE_DIR, "csrc", "libptamd.so")

_library = None


def load():
    """The HIP engine (libptamd.so).  Raises FileNotFoundError when it has not been built:
    run ``python -c 'import __graft_entry__ as g; g.build()'`` or ``make -C rust-pathtracer_amd/csrc``."""
    global _library
    if _library is None:
        if not os.path.exists(LIBRARY_PATH):
            raise FileNotFoundError(
                "HIP engine not built: %s is missing (make -C rust-pathtracer_amd/csrc). "
                "There is no CPU fallback for the product path." % LIBRARY_PATH)
        _library = api.Library(LIBRARY_PATH, "pt_")
    return _library
