# interleaved A/B of bench.py: `tools/ab.sh "VAR=1" ""` compares env settings; SDRX_LIB=... compares builds
for i in 1 2 3; do for v in "$@"; do echo -n "[${v:-default}] "; env $v python bench.py --no-cpu --steps 40 ${ABARGS} | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], {k:v['avg_ms'] for k,v in d['kernels'].items()})"; done; done
