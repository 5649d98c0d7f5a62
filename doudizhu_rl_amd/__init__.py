"""Importable alias of the package directory `doudizhu-rl_amd/` (a hyphen is not a valid
Python identifier): `import doudizhu_rl_amd` gives the same module object."""
import importlib
import sys

sys.modules[__name__] = importlib.import_module("doudizhu-rl_amd")
