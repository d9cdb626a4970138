set -e
cd $GRAFT_REPO_ROOT
timeout -k 10 600 python -m pytest tests -m gpu -x -q 2>&1 | tail -3
python3 tools/ab.py -k 64 -r 3 default top1 top15 top127 top255 wg2 wg8
