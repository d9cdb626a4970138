# same-box A/B of library variants: one pass observed alone (K=1, k_run), K=2, 3, and K=64 through k_run (mode 5)
set -e
cd $GRAFT_REPO_ROOT
python3 tools/ab.py -k 1 -r 5 "$@"
python3 tools/ab.py -k 2 -r 3 "$@"
python3 tools/ab.py -k 3 -r 3 "$@"
GPUART_MODE=5 python3 tools/ab.py -k 64 -r 3 "$@"
