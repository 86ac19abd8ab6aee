"""Makes `kgcn.<name>` the very same module object as `kgcn_amd.<name>` (flags like `enabled_bspmm` are module
globals that callers set from outside: a re-exporting copy would split them)."""
import importlib
import sys


def alias(name):
    mod = importlib.import_module("kgcn_amd." + name)
    sys.modules["kgcn." + name] = mod
    return mod
