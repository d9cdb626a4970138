cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r04z
python3 tools/ab.py -k 64 -r 3 default slabprobe > gpurun_out/r04z/slabprobe.txt 2>&1; cat gpurun_out/r04z/slabprobe.txt
python3 tools/ab.py -k 1 -r 3 default slabprobe >> gpurun_out/r04z/slabprobe.txt 2>&1; tail -2 gpurun_out/r04z/slabprobe.txt
