"""Import shim: the product package lives in ``feature-predictor-for-speech-codec_amd/``
(a directory name Python cannot import directly).  ``import fpcodec_amd`` loads that
directory as the package ``fpcodec_amd``."""
import importlib.util
import os
import sys

_here = os.path.dirname(os.path.abspath(__file__))
_pkg_dir = os.path.join(_here, "feature-predictor-for-speech-codec_amd")
_spec = importlib.util.spec_from_file_location(
    "fpcodec_amd", os.path.join(_pkg_dir, "__init__.py"), submodule_search_locations=[_pkg_dir]
)
_mod = importlib.util.module_from_spec(_spec)
sys.modules["fpcodec_amd"] = _mod
_spec.loader.exec_module(_mod)
