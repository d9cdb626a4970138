# Round-2 extras beside tools/profile_round.sh:  bash tools/profile_extras.sh <name>  -> gpurun_out/<name>/
set -e
NAME=${1:-prof}
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/$NAME
mkdir -p $OUT
cd $R
bash tools/small_k_sweep.sh > $OUT/k_run_vs_pipeline_small_k.txt 2>&1
python3 tools/single_pass_time.py > $OUT/single_pass_by_frame_size.txt 2>&1
if [ -d gpuart_amd/lib_ab/tl ]; then GPUART_LIBDIR=$R/gpuart_amd/lib_ab/tl python3 tools/run_timeline.py 1 > $OUT/k_run_timeline.txt 2>&1; fi
bash tools/l1_calibration.sh > $OUT/l1_access_calibration_raw.txt 2>&1
cat $OUT/k_run_vs_pipeline_small_k.txt $OUT/single_pass_by_frame_size.txt
