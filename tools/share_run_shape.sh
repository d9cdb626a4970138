# the N = 8 / K = 20 share of the 1080p frame by run shape: one k_run run (default) against launch-pipeline runs of several sizes (profiles/r04/birth_order.txt, 4)
mkdir -p gpurun_out
{
run() { echo "-- $*"; env "$@" TILE_N=8 TILE_K=20 timeout -k 10 120 python3 tools/tile_overhead.py | grep "N=8" || exit 1; }
for rep in 1 2; do
run X=1
run GPUART_HIP_SMALL_KPATHS=0
run GPUART_HIP_SMALL_KPATHS=0 GPUART_HIP_MIN_RUN_KPATHS=1024
run GPUART_HIP_SMALL_KPATHS=0 GPUART_HIP_MIN_RUN_KPATHS=512
run GPUART_HIP_SMALL_KPATHS=0 GPUART_HIP_MIN_RUN_KPATHS=256
run GPUART_HIP_SMALL_KPATHS=0 GPUART_HIP_MIN_RUN_KPATHS=512 GPUART_HIP_MAX_BATCH=4
run GPUART_HIP_SMALL_KPATHS=0 GPUART_HIP_MIN_RUN_KPATHS=512 GPUART_HIP_MAX_BATCH=3
run GPUART_HIP_SMALL_KPATHS=0 GPUART_HIP_MIN_RUN_KPATHS=256 GPUART_HIP_MAX_BATCH=2
done
} > gpurun_out/share_run_shape.txt 2>&1
