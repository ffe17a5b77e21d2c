# step-size schedules of the 12-state stages (tools/rho_policy_probe.py), one log per stage
set -e
GZ="100:1,10,1,1/200:1,100,1,1;100:1,10,1,1/200:1,30,1,1;100:1,5,1,1/200:1,50,1,1;75:1,10,1,1/200:1,100,1,1;100:1,10,1,1/300:1,100,1,1"
GXY="100:1,1,3,1;75:1,1,3,1;100:1,1,5,1;100:1,1,3,1/200:1,1,10,1;100:1,1,3,3;100:1,1,3,0.3;100:1,2,3,1;100:1,0.5,3,1;150:1,1,3,1;100:1,1,3,1/300:1,3,3,1;100:1,1,2,1"
for st in yaw z xy; do
  rho=500; [ $st = yaw ] && rho=200
  G="$GZ"; [ $st = xy ] && G="$GXY"
  timeout -k 10 700 python -u tools/rho_policy_probe.py --model 12state --stage $st --N 20 --B 4096 --seed-off 3 --rho $rho --eps $1 --max-iter $2 --set g --base 1,10,3,3 --grid "$G" > gpurun_out/r3_policy12b_${st}.log 2>&1
done
