set -e
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r04a
(timeout -k 10 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -15) > gpurun_out/r04a/tests.txt 2>&1 || true
cat gpurun_out/r04a/tests.txt | tail -5
python3 tools/ab.py -k 64 -r 4 base default nshadow > gpurun_out/r04a/ab_k64.txt 2>&1
cat gpurun_out/r04a/ab_k64.txt
python3 tools/ab.py -k 20 -r 4 base default nshadow > gpurun_out/r04a/ab_k20.txt 2>&1
cat gpurun_out/r04a/ab_k20.txt
python3 tools/ab.py -k 1 -r 5 base default nshadow > gpurun_out/r04a/ab_k1.txt 2>&1
cat gpurun_out/r04a/ab_k1.txt
python3 tools/ab.py -k 2 -r 3 base default nshadow > gpurun_out/r04a/ab_k2.txt 2>&1
cat gpurun_out/r04a/ab_k2.txt
