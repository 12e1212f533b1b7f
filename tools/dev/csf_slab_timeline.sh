# timeline of the kernels of a few steps of the lone-big-slab case (tools/dev/csf_slab_rank.py) -> gpurun_out/csf_slab_timeline.txt
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rm -rf $R/gpurun_out/csf_tl
CSF_TL_ONLY=${1:-mixed} rocprofv3 --kernel-trace -d $R/gpurun_out/csf_tl -o x -- python $R/tools/dev/csf_slab_rank.py > $R/gpurun_out/csf_tl.log 2>&1
db=$(find $R/gpurun_out/csf_tl -name "x_results.db" | head -1)
python - "$db" > $R/gpurun_out/csf_slab_timeline.txt <<'PY'
import sqlite3, sys
c = sqlite3.connect(sys.argv[1])
tabs = [r[0] for r in c.execute("select name from sqlite_master where type in ('table','view')")]
print("# tables:", [t for t in tabs if "kernel" in t.lower()][:12])
v = [t for t in tabs if t == "kernels"] or [t for t in tabs if "kernel" in t.lower()]
cols = [r[1] for r in c.execute("pragma table_info(%s)" % v[0])]
print("# columns of", v[0], cols)
rows = list(c.execute("select name, start, end, %s from %s order by start" % ("stream_id" if "stream_id" in cols else ("queue_id" if "queue_id" in cols else "0"), v[0])))
print("# dispatches:", len(rows))
# the last 40 % of the dispatches: print 60 consecutive ones around the 80 % mark
k0 = int(len(rows) * 0.8)
t0 = rows[k0][1]
prev_end = t0
for name, s, e, q in rows[k0:k0 + 70]:
    print("%9.1f us  +%7.1f gap  %8.1f us  q%-4s %s" % ((s - t0) / 1e3, (s - prev_end) / 1e3, (e - s) / 1e3, q, name[:60].replace("(anonymous namespace)::", "")))
    prev_end = max(prev_end, e)
PY
rm -rf $R/gpurun_out/csf_tl
head -75 $R/gpurun_out/csf_slab_timeline.txt
