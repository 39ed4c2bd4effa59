#!/bin/bash
# steps/s of bench.py's headline leg under one option, against the default;  usage: bash tools/opt_probe.sh "name=value" "N B" ...
opt=$1; shift
for cfg in "$@"; do
  set -- $cfg
  a=$(python3 bench.py --landmarks $1 --trajectories $2 --no-cpu-baseline --no-single 2>/dev/null | python3 -c "import json,sys; print(round(json.loads(sys.stdin.read().strip().splitlines()[-1])['value']))")
  b=$(python3 bench.py --landmarks $1 --trajectories $2 --no-cpu-baseline --no-single --option $opt 2>/dev/null | python3 -c "import json,sys; print(round(json.loads(sys.stdin.read().strip().splitlines()[-1])['value']))")
  echo "N=$1 B=$2: default $a steps/s, $opt $b steps/s"
done
