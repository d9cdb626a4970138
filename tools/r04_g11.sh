cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r04k
O=gpurun_out/r04k
(timeout -k 10 1000 python -m pytest tests -m gpu -x -q --durations=8 2>&1 | tail -25) > $O/tests.txt 2>&1
tail -16 $O/tests.txt
timeout -k 10 600 python bench.py --steps 20 --warmup 5 > $O/bench_k20.json 2> $O/bench_k20.err; tail -c 3000 $O/bench_k20.json
