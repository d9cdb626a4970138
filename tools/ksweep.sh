for k in 4 8 20 64; do SWEEP_K=$k timeout -k 10 300 python tools/sweep.py "BATCH_MPATHS=16" 2>&1 | grep BATCH | sed "s/^/K=$k /"; done
