"""Drop-in for the reference's `simple_knn` package (submodules/simple-knn):
`from simple_knn._C import distCUDA2`."""
import os
import sys

_PKG_ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if _PKG_ROOT not in sys.path:
    sys.path.insert(0, _PKG_ROOT)
