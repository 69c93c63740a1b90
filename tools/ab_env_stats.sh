# Per-kernel times of a kernel family under a list of environment settings, inside ONE gpurun call (short serial bench under rocprofv3 --stats each).
# usage: bash tools/ab_env_stats.sh <kernel-name regex> "VAR=a" "VAR=b" ...      (repo root, GPU box)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
pat=$1; shift
i=0
for cfg in "$@"; do
  i=$((i+1))
  env $cfg true || exit 1
  ( export $cfg; rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/abenv_$i -o p -- python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-inference --no-extra --serial > gpurun_out/abenv_$i.log 2>&1 ) || exit 1
  echo "== $cfg"
  python3 - "$pat" gpurun_out/abenv_$i/p_kernel_stats.csv <<'PY'
import csv, re, sys
pat, path = sys.argv[1], sys.argv[2]
tot = 0.0
for r in csv.DictReader(open(path)):
    if re.search(pat, r['Name']):
        tot += float(r['TotalDurationNs'])
        print('   %-70s %4s %7.1f us' % (r['Name'][:70], r['Calls'], float(r['AverageNs']) / 1e3))
print('   family total %.3f ms over the run' % (tot / 1e6))
PY
done
