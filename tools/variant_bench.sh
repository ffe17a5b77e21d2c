#!/bin/bash
# Development aid (GPU box): the bench workloads with alternative builds of the library, one after
# the other; prints workload, library, solves/s, executed fraction.
#   tools/variant_bench.sh lib1.so lib2.so ...     ("default" = the in-tree library)
cd "$(dirname "$0")/.." || exit 1
for wl in mpc_6dof_B262144_N50 mpc_6dof_B4096_N20 mpc_12state_B4096_N20; do
  for lib in default "$@"; do
    if [ "$lib" = default ]; then unset TG_MPC_LIB; else export TG_MPC_LIB="$PWD/$lib"; fi
    out=$(timeout -k 10 120 python bench.py --workload $wl --no-cpu-baseline --no-check-fine 2>/dev/null) || { echo "$wl $lib FAILED"; exit 1; }
    python - "$wl" "$lib" "$out" <<'E'
import json, sys
d = json.loads(sys.argv[3].strip().splitlines()[-1])
print("%-24s %-50s %.4g solves/s  frac %.3f  iters %.2f" % (sys.argv[1], sys.argv[2], d["value"], d["roofline"]["frac"], d["iters_mean"]), flush=True)
E
  done
done
