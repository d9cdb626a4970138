set -e
cd $GRAFT_REPO_ROOT
python3 tools/ab.py -k 64 -r 3 wg1top1 wg1top7 wg1top15 wg1top31 wg4top63
