"""Import alias: the package directory carries the repository's full (hyphenated) name, which the ``import`` statement
cannot spell.  ``import dgtd`` returns that package, and ``dgtd.<sub>`` names resolve to the SAME module objects (no second
copy of any submodule is ever created)."""
import importlib
import sys

_REAL = "depth-guided-texture-diffusion-for-image-semantic-segmentation_amd"
_pkg = importlib.import_module(_REAL)
for _name, _mod in list(sys.modules.items()):
    if _name == _REAL or _name.startswith(_REAL + "."):
        sys.modules["dgtd" + _name[len(_REAL):]] = _mod
sys.modules[__name__] = _pkg
