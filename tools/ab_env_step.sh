#!/bin/bash
# ab_env_step.sh "ENV=.. ENV=.." "ENV=.." ... : bench.py ms/step per environment setting (A/B of kernel switches) -> gpurun_out/ab_env_step.txt
out=gpurun_out/ab_env_step.txt; : > $out
for cfg in "$@"; do
  r=$(env $cfg python bench.py --no_cpu_baseline --profile_steps 0 --steps 60 --warmup 15 2>/dev/null | python -c "import sys,json; print(json.loads(sys.stdin.readline())['ms_per_step'])")
  echo "$cfg ms_per_step=$r" >> $out
done
cat $out
