set -e
export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; cd /tmp
rm -rf /tmp/kt; timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kt -- python3 $R/bench.py --workload mpc_6dof_B4096_N20 --steps 200 --pipeline 1 --no-cpu-baseline > /tmp/kt.log 2>&1 || { tail /tmp/kt.log; exit 1; }
f=$(find /tmp/kt -name "*kernel_stats.csv" | head -1); head -4 $f | cut -c1-400
t=$(find /tmp/kt -name "*kernel_trace.csv" | head -1); python3 - $t <<'P'
import csv, sys
rows = [r for r in csv.DictReader(open(sys.argv[1])) if "admm_kernel" in r["Kernel_Name"]]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
d = [int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in rows]
g = [int(rows[i + 1]["Start_Timestamp"]) - int(rows[i]["End_Timestamp"]) for i in range(len(rows) - 1)]
import statistics
print("launches", len(d), "median duration us", statistics.median(d) / 1e3, "median gap us", statistics.median(g) / 1e3)
P
