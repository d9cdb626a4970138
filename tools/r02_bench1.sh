set -e
cd $GRAFT_REPO_ROOT
timeout -k 10 900 python -m pytest tests -m gpu -x -q > gpurun_out/r02_pytest3.log 2>&1 || { tail -40 gpurun_out/r02_pytest3.log; exit 1; }
tail -2 gpurun_out/r02_pytest3.log
timeout -k 10 600 python bench.py > gpurun_out/r02_bench_default.json 2> gpurun_out/r02_bench_default.err || { tail -20 gpurun_out/r02_bench_default.err; exit 1; }
cat gpurun_out/r02_bench_default.json
GPUART_BENCH_BACKEND=gloo timeout -k 10 600 python bench.py --gpus 2 --steps 16 --warmup 4 --verify-gather --no-cpu-baseline > gpurun_out/r02_bench_g2.json 2> gpurun_out/r02_bench_g2.err || { tail -30 gpurun_out/r02_bench_g2.err; exit 1; }
grep verify gpurun_out/r02_bench_g2.err; cat gpurun_out/r02_bench_g2.json | cut -c1-400
