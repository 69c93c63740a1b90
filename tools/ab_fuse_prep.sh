set -e
cd /root/repo
python -m pytest tests/test_generator_gpu.py -q -x -k "score_fusion or attention" 2>&1 | tail -5
for i in 1 2; do
for v in 0 1; do
HV_CA_FUSE_PREP=$v python bench.py --steps 40 --warmup 10 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('FUSE_PREP=$v', d['ms_per_step'])"
done; done
