"""`kgcn.bspmm_call` -> `kgcn_amd.bspmm_call` (see kgcn/__init__.py); importing this name yields that module object itself."""
from ._alias import alias

alias("bspmm_call")
