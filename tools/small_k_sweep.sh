# mode 0 with the k_run threshold off (launch pipeline) and wide open (k_run as one run) for short pass sequences: where is the crossover?
cd $GRAFT_REPO_ROOT
for k in 1 2 3 4 6 8; do
  a=$(GPUART_HIP_SMALL_KPATHS=0 python3 tools/run_passes.py $k 6 | grep ms/pass | awk '{print $3}' | sort -n | head -1)
  b=$(GPUART_HIP_SMALL_KPATHS=65536 python3 tools/run_passes.py $k 6 | grep ms/pass | awk '{print $3}' | sort -n | head -1)
  echo "K=$k  pipeline $a  k_run $b ms/pass"
done
