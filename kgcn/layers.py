"""`kgcn.layers` -> `kgcn_amd.layers` (see kgcn/__init__.py); importing this name yields that module object itself."""
from ._alias import alias

alias("layers")
