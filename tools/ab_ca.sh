cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout -k 10 600 python -m pytest tests/test_generator_gpu.py tests/test_infer_gpu.py -m gpu -x -q 2>&1 | tail -2 || exit 1
for cfg in "0 0" "1 1"; do
  set -- $cfg
  export HV_CA_FUSE_XCD=$1 HV_CA_SOFTMAX_WAVE=$2
  rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/ca_$1$2 -o p -- python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-inference --no-extra --serial > gpurun_out/ca_$1$2.log 2>&1 || exit 1
  echo "== FUSE_XCD=$1 SOFTMAX_WAVE=$2"
  grep -E "ca_fuse|ca_softmax" gpurun_out/ca_$1$2/p_kernel_stats.csv | cut -d, -f1-4 | cut -c1-120
done
unset HV_CA_FUSE_XCD HV_CA_SOFTMAX_WAVE
for v in "0 0" "1 1" "0 0" "1 1"; do set -- $v; HV_CA_FUSE_XCD=$1 HV_CA_SOFTMAX_WAVE=$2 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-inference --no-extra 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('$v', d['ms_per_step'], d['fine_generator_forward']['graph_replay']['ms'])"; done
