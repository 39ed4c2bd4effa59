run() { python bench.py --no-cpu-baseline --no-single "$@" | python -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print('$*', 'steps/s %.0f  ms/step %.4f  flush %.1f us x %.1f  kernel %s' % (d['value'], d['ms_per_step'], r['avg_launch_ms']*1e3, r['steps_per_launch'], r['kernel']))"; }
for b in 8 10 12 14 16 20 24 28 32; do run --landmarks 2000 --trajectories $b; done
