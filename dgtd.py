"""Import alias: the package directory carries the repository's full (hyphenated) name, which the
``import`` statement cannot spell.  ``import dgtd`` returns that package."""
import importlib
import sys

_pkg = importlib.import_module("depth-guided-texture-diffusion-for-image-semantic-segmentation_amd")
sys.modules[__name__] = _pkg
