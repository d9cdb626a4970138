cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r04l
O=gpurun_out/r04l
(timeout -k 10 1000 python -m pytest tests -m gpu -x -q 2>&1 | tail -8) > $O/tests.txt 2>&1
tail -5 $O/tests.txt
timeout -k 10 900 python bench.py --steps 20 --warmup 5 > $O/bench_driver_config.json 2> $O/bench_driver_config.err; tail -c 1500 $O/bench_driver_config.json; tail -3 $O/bench_driver_config.err
GPUART_BENCH_BACKEND=gloo timeout -k 10 600 python bench.py --gpus 2 --steps 8 --warmup 2 --no-cpu-baseline > $O/bench_gloo2.json 2> $O/bench_gloo2.err; tail -c 1200 $O/bench_gloo2.json; tail -5 $O/bench_gloo2.err
