"""Alias: `import ray_tracing_amd` -> the package in the directory `ray-tracing_amd/` (whose
hyphen cannot appear in an import statement); `python -m ray_tracing_amd ...` runs its launcher."""
import importlib
import os
import sys

_root = os.path.dirname(os.path.abspath(__file__))
if _root not in sys.path:
    sys.path.insert(0, _root)
_pkg = importlib.import_module("ray-tracing_amd")
if __name__ == "__main__":
    sys.modules["ray_tracing_amd"] = _pkg
    importlib.import_module("ray-tracing_amd.__main__").main()
else:
    sys.modules[__name__] = _pkg
