# the pass at N=8000 x 1 against the row stride of P (capacity n_max -> ld = ceil(n_max / 64) * 64)
for nm in 16003 16127 16191 16255 16319 16383 16447 16511 16639 16895 17407 18431 20479; do
  python3 tools/flush_time.py --landmarks 8000 --trajectories 1 --steps 40 --nmax $nm 2>/dev/null
  python3 tools/flush_time.py --landmarks 8000 --trajectories 1 --steps 40 --nmax $nm --option pass_kernel=0 2>/dev/null
done
