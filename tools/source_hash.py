"""Hash of kernel sources with comments and white space removed: the staleness key of profiles/traffic_<cfg>.json (bench.py,
tools/profile_summary.py, tools/profile_config_summary.py).  A comment edit does not make measured HBM traffic stale; any code edit does."""
import glob
import hashlib
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "kgcn_amd", "csrc")
_COMMENT = re.compile(r"/\*.*?\*/|//[^\n]*", re.S)


def code_only(text):
    return re.sub(r"\s+", " ", _COMMENT.sub(" ", text)).strip()


def sources_sha256(names=None):
    """names: file names under kgcn_amd/csrc (None: every .hip / .h there)."""
    files = sorted(glob.glob(os.path.join(CSRC, "*.hip")) + glob.glob(os.path.join(CSRC, "*.h"))) if names is None else \
        [os.path.join(CSRC, n) for n in names]
    h = hashlib.sha256()
    for f in files:
        h.update(os.path.basename(f).encode())
        h.update(code_only(open(f, encoding="utf-8", errors="replace").read()).encode())
    return h.hexdigest()


CFG2_FILES = ("fused.hip", "spmm.hip", "dense.hip", "kgcn_common.h")
