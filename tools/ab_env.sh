#!/bin/bash
# A/B of environment knobs on one box, back to back:  tools/ab_env.sh <out-prefix> "<bench args>" "VAR=1 VAR2=x" "..." ...
# (first variant "" = defaults); prints ms_per_step per variant, two rounds each (boxes drift: compare within a round)
pre=$1; shift
bargs=$1; shift
mkdir -p gpurun_out
for round in 1 2; do
  i=0
  for v in "$@"; do
    i=$((i+1))
    out=gpurun_out/${pre}_v${i}_r${round}.json
    env $v python3 bench.py $bargs --no-cpu-baseline --no-extra-configs --no-kernel-timer > $out 2> gpurun_out/${pre}_v${i}_r${round}.err || { echo "variant '$v' failed"; tail -5 gpurun_out/${pre}_v${i}_r${round}.err; }
    python3 - "$out" "$v" "$round" <<'PY'
import json, sys
try:
    j = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print(f"round {sys.argv[3]}  [{sys.argv[2] or 'default':50s}]  {j['ms_per_step']:8.3f} ms  {j['value']:8.2f} img/s", flush=True)
except Exception as e:
    print("no result for", sys.argv[2], e, flush=True)
PY
  done
done
