# same-box A/B of the library's own birth order (GPUART_HIP_TILE_ORDER=0 / 1): tools/knob_ab.py, best of 3 sequences per process
mkdir -p gpurun_out
{
for w in cfg3 dragon871k cfg2; do echo "== $w, one pass alone"; timeout -k 10 300 python3 tools/knob_ab.py -k 1 -r 5 "WORKLOAD=$w TILE_ORDER=0" "WORKLOAD=$w TILE_ORDER=1" || exit 1; done
for w in cfg3 cfg2; do for k in 2 20; do echo "== $w K=$k (the order is not applied: must be equal)"; timeout -k 10 300 python3 tools/knob_ab.py -k $k -r 3 "WORKLOAD=$w TILE_ORDER=0" "WORKLOAD=$w TILE_ORDER=1" || exit 1; done; done
} > gpurun_out/tile_order_ab2.txt 2>&1
