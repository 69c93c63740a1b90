# A/B of the attention block's late round-3 switches inside ONE gpurun call: parity tests first, then per-kernel times (rocprofv3 --stats over a short
# serial bench) and step / fine-forward times with the switches off and on.   usage: bash tools/ab_ca.sh   (repo root, GPU box)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout -k 10 600 python -m pytest tests/test_generator_gpu.py tests/test_infer_gpu.py -m gpu -x -q 2>&1 | tail -2 || exit 1
for v in 0 1; do
  export HV_CA_GS_XCD=$v HV_CA_PBWD_VEC=$v
  rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/cb_$v -o p -- python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-inference --no-extra --serial > gpurun_out/cb_$v.log 2>&1 || exit 1
  echo "== GS_XCD=PBWD_VEC=$v"
  python3 - <<PY
import csv
for r in csv.DictReader(open('gpurun_out/cb_$v/p_kernel_stats.csv')):
    if r['Name'].startswith('ca_') or 'ca_' in r['Name'][:40]:
        print('   %-60s %3s %7.1f us' % (r['Name'][:60], r['Calls'], float(r['AverageNs']) / 1e3))
PY
done
unset HV_CA_GS_XCD HV_CA_PBWD_VEC
for v in 0 1 0 1; do HV_CA_GS_XCD=$v HV_CA_PBWD_VEC=$v python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-inference --no-extra 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('$v', d['ms_per_step'], d['fine_generator_forward']['graph_replay']['ms'])"; done
