# k_run occupancy variants: K = 1, 2, 3 (mode 0), K = 64 as one k_run (mode 5), and rank 0's share at N = 8 with K = 20
set -e
cd $GRAFT_REPO_ROOT
python3 tools/ab.py -k 1 -r 4 "$@"
python3 tools/ab.py -k 3 -r 3 "$@"
GPUART_MODE=5 python3 tools/ab.py -k 64 -r 3 "$@"
for n in "$@"; do
  if [ "$n" = default ]; then d=""; else d=$GRAFT_REPO_ROOT/gpuart_amd/lib_ab/$n; fi
  echo -n "$n  N=8 K=20: "; GPUART_LIBDIR=$d TILE_K=20 TILE_N=8 python3 tools/tile_overhead.py | tail -1 | cut -c1-22
done
