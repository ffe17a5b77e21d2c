export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; cd /tmp
for cfg in "25 25" "50 25" "50 50" "100 100"; do set -- $cfg
  for c in FETCH_SIZE WRITE_SIZE; do
    rm -rf /tmp/tp; timeout -s KILL 120 rocprofv3 --pmc $c -d /tmp/tp -- python3 $R/tools/traffic_probe.py 50 262144 $1 $2 > /tmp/tp.log 2>&1 || { tail -5 /tmp/tp.log; exit 1; }
    python3 - $c $1 $2 <<'P'
import glob, sqlite3, sys
db = sqlite3.connect(glob.glob("/tmp/tp/**/*.db", recursive=True)[0])
v = [r[0] for r in db.execute("select value from counters_collection where counter_name = ? and kernel_name like '%admm_kernel%'", (sys.argv[1],))]
print("iters %s check %s %s: %d launches, mean %.1f MB (raw KB counter)" % (sys.argv[2], sys.argv[3], sys.argv[1], len(v), sum(v) / len(v) / 1024))
P
  done
done
