cd $GRAFT_REPO_ROOT
run() { env "$@" python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-inference --no-extra 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print(d['ms_per_step'], d['config']['launch'])"; }
for i in 1 2; do
echo -n "plain batched   "; run HV_BATCH_D=1
echo -n "plain split     "; run HV_BATCH_D=0
echo -n "dp schedule     "; run HV_DDP_FORCE=1
done
