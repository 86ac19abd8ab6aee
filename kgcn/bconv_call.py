"""`kgcn.bconv_call` -> `kgcn_amd.bconv_call` (see kgcn/__init__.py); importing this name yields that module object itself."""
from ._alias import alias

alias("bconv_call")
