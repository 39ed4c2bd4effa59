# usage: bench_opts.sh "<bench args>" opt1 opt2 ...   (each opt is a comma-separated NAME=VALUE list or "-")
args="$1"; shift
for o in "$@"; do
  extra=""
  if [ "$o" != "-" ]; then for kv in $(echo $o | tr ',' ' '); do extra="$extra --option $kv"; done; fi
  echo "== $args $o"
  python bench.py $args --no-cpu-baseline --no-single $extra | python -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print('steps/s %.0f  flush %.1f us x %.1f steps  frac %.3f mfma %.1f TF' % (d['value'], r['avg_launch_ms']*1e3, r['steps_per_launch'], r['frac'], r['mfma']['achieved']))"
done
