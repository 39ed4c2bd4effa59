export TMPDIR=/tmp
OUT=gpurun_out/pbrk; rm -rf $OUT; mkdir -p $OUT
for v in "" nodd nost nog none; do
  EKFSLAM_HIP_VARIANT=$v rocprofv3 --kernel-trace --output-format csv -d $OUT/v_$v -o run -- python3 bench.py --no-cpu-baseline --no-single --steps 100 > $OUT/v_$v.log 2>&1
  echo "== variant '$v'" >> $OUT/summary.txt
  python3 tools/kernel_times.py $OUT/v_$v | grep "k_panels_cad" >> $OUT/summary.txt
done
cat $OUT/summary.txt
