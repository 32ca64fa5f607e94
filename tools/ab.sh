#!/bin/bash
# A/B of environment knobs on ONE box: tools/ab.sh "VAR=a" "VAR=b" ... ; each setting is run REPS times, interleaved.
# ("-" = no variable).  Prints sprites/s of the default bench leg (300 timed steps behind the standard warm-up).
cd "$(dirname "$0")/.."
REPS=${REPS:-3}
for r in $(seq $REPS); do
  for kv in "$@"; do
    if [ "$kv" = "-" ]; then v=$(python3 bench.py --steps 300 --no-cpu-baseline --hybrid-steps 0 --fp8-steps 0 --config2-steps 0 --highend-steps 0 --prof-steps 0 2>/dev/null)
    else v=$(env $kv python3 bench.py --steps 300 --no-cpu-baseline --hybrid-steps 0 --fp8-steps 0 --config2-steps 0 --highend-steps 0 --prof-steps 0 2>/dev/null); fi
    echo "$kv $(echo "$v" | python3 -c 'import sys,json; print(round(json.loads(sys.stdin.read())["value"]))')"
  done
done
