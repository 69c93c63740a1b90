for t in 0 1; do
  echo "HV_WGRAD_THIN=$t"
  HV_WGRAD_THIN=$t CHECK=1 python tools/bench_wgrad.py 2 40 48 4 16 5 1 2 5 2>&1 | grep "^W "
  HV_WGRAD_THIN=$t CHECK=1 python tools/bench_wgrad.py 3 37 29 3 16 5 1 2 5 2>&1 | grep "^W "
  HV_WGRAD_THIN=$t python tools/bench_wgrad.py 16 256 256 4 16 5 1 2 2>&1 | grep "^W "
done
for w in 256 512 2048 4096; do echo "WGS=$w"; HV_WGRAD_THIN_WGS=$w python tools/bench_wgrad.py 16 256 256 4 16 5 1 2 2>&1 | grep "^W "; done
