for cf in 0 5; do for gc in -1 0; do
timeout -k 10 120 python bench.py --no-cpu-baseline --workload mpc_6dof_B32768_N50 --steps 10 --warmup 3 --check-fine $cf --grid-cap $gc > gpurun_out/v_${cf}_${gc}.json 2>gpurun_out/v.err || exit 1
python -c "
import json;d=json.load(open('gpurun_out/v_${cf}_${gc}.json'));print('fine',$cf,'cap',$gc,'ms',d['ms_per_step'],'mean it',d['iters_mean'],'max',d['iters_max'])"
done; done
