set -e
cd $GRAFT_REPO_ROOT
python3 tools/ab.py -k 64 -r 3 default ta1 ta2 ta4
