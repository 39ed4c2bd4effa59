# the chained cadence's panel launch in the latency regime: the triangular solve (k_panels_cad_tf, default) against the replay with
# one wave per workgroup and the DPP broadcast (k_panels_cad<1>: "panel_tform" = 0, "panel_shape" = 2) and the row-split replay
# (k_panels_cad_ks: "panel_tform" = 0).  steps/s of config 3, config 2 and config 5.   bash tools/panel_form_probe.sh
for opts in "" "--option panel_tform=0 --option panel_shape=2" "--option panel_tform=0"; do
  echo "== options: '$opts'"
  python3 bench.py --leg single_trajectory $opts 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read())['single_trajectory']; print('config3', round(d['value']), {k: round(v,1) for k,v in d['cadence_us'].items() if k != 'note'})"
  python3 bench.py --leg config2 $opts 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read())['config2']; print('config2', round(d['value']))"
  python3 bench.py --leg config5 $opts 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read())['config5']; print('config5', {k: round(v['value']) for k,v in d.items() if isinstance(v,dict) and 'value' in v})"
done
