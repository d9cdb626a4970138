set -e
cd $GRAFT_REPO_ROOT
export GPUART_MODE=3
for w in 8 12 16; do
  echo "== WAVES_PER_CU $w"; GPUART_HIP_WAVES_PER_CU=$w python3 tools/ab.py -k 64 -r 3 default w6r4 w8r4
done
