# where the panel launch's time goes (profiles/r06_panel_launch.txt section 4): the headline's kernel trace on diagnostic builds of the
# cadence kernels that leave a part out (timing only, wrong results).  Build them first, here or on the box:
#   for v in "nodd -DCADP_SKIP_DD" "nost -DCADP_SKIP_STORE" "nog -DCADP_SKIP_GATHER" "none -DCADP_SKIP_DD -DCADP_SKIP_STORE -DCADP_SKIP_GATHER"; do
#     set -- $v; t=$1; shift; make -C slam-duckietown_amd/csrc variant_cad TAG=$t EXTRA="$*"; done
export TMPDIR=/tmp
OUT=gpurun_out/pbrk; rm -rf $OUT; mkdir -p $OUT
for v in "" nodd nost nog none; do
  EKFSLAM_HIP_VARIANT=$v rocprofv3 --kernel-trace --output-format csv -d $OUT/v_$v -o run -- python3 bench.py --no-cpu-baseline --no-single --steps 100 > $OUT/v_$v.log 2>&1
  echo "== variant '$v'" >> $OUT/summary.txt
  python3 tools/kernel_times.py $OUT/v_$v | grep "k_panels_cad" >> $OUT/summary.txt
done
cat $OUT/summary.txt
