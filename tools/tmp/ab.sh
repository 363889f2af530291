run() { python bench.py --no-cpu-baseline --no-extras --steps 10 --warmup 2 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('$1', d['ms_per_step'], d['value'], d['roofline']['frac'])"; }
V=$PWD/bayesian_ensembling_amd/libbayens_epi3.so
run base
BAYENS_LIB=$V run epi3
run base
BAYENS_LIB=$V run epi3
echo base; python tools/gemm_bench.py 2>&1 | tail -13
echo epi3; BAYENS_LIB=$V python tools/gemm_bench.py 2>&1 | tail -13
