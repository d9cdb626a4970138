# Same-box A/B of one library variant (gpuart_amd/lib_ab/nopipe; edit the name) against the product build on the other workloads:
#   K = 3 passes per pass and one pass alone, twice each, alternating.   bash tools/ab_workloads.sh   (on the GPU box, from the repo root)
for w in cfg2 cluster tree; do for v in nopipe default nopipe default; do if [ $v = nopipe ]; then export GPUART_LIBDIR=$PWD/gpuart_amd/lib_ab/nopipe; else unset GPUART_LIBDIR; fi; python bench.py --workload $w --steps 3 --warmup 2 --repeats 5 --no-profile --no-cpu-baseline 2>/dev/null | python3 -c "
import json,sys; d=json.loads(sys.stdin.read().strip().split(chr(10))[-1]); print('$w', '$v', 'K=3 per pass', d['ms_per_step'], 'single', d['ms_per_frame_single'])"; done; done
