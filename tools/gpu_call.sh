#!/bin/bash
REPO=${GRAFT_REPO_ROOT:-/root/repo}; cd /tmp; export TMPDIR=/tmp
rocprofv3 --kernel-trace --memory-copy-trace -d /tmp/mc -o mc -- python $REPO/bench.py --config cfg1 --profile --steps 20 --warmup 3 > /tmp/mc.log 2>&1
python - <<'PY'
import glob, sqlite3, collections
for db in glob.glob('/tmp/mc/**/*.db', recursive=True):
    con = sqlite3.connect(db); cur = con.cursor()
    tabs = [r[0] for r in cur.execute("select name from sqlite_master where type in ('table','view')")]
    print([t for t in tabs if 'cop' in t.lower() or 'mem' in t.lower()][:20])
    for t in tabs:
        if 'memory_cop' in t.lower() and 'rocpd_' not in t:
            cols = [r[1] for r in cur.execute("pragma table_info(%s)" % t)]
            print(t, cols)
            rows = list(cur.execute("select * from %s order by start" % t))
            print(len(rows), "copies; last 12:")
            for r in rows[-12:]: print(r)
    ks = list(cur.execute("select name, start, duration from kernels order by start"))
    print("last 40 kernels:")
    for n, s, d in ks[-40:]: print(s, d, n[:60])
PY
