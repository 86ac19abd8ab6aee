"""`kgcn.batched_call` -> `kgcn_amd.batched_call` (see kgcn/__init__.py); importing this name yields that module object itself."""
from ._alias import alias

alias("batched_call")
