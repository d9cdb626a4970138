set -e
cd $GRAFT_REPO_ROOT
for w in 8 12; do
  echo "== WAVES_PER_CU $w"; GPUART_HIP_WAVES_PER_CU=$w python3 tools/ab.py -k 64 -r 3 default w7 w7r6 w8r6
done
