"""`simple_knn._C.distCUDA2` of the reference (submodules/simple-knn/ext.cpp) -> ex4dgs_amd.simple_knn._C.distCUDA2."""
import os
import sys

_ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
if _ROOT not in sys.path:
    sys.path.append(_ROOT)

from ex4dgs_amd.simple_knn._C import distCUDA2  # noqa: E402,F401
