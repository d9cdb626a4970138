# ms per pass against the length K of a pass sequence (K passes between two observations of the result); extra
# arguments are passed to tools/sweep.py (knob=value lists)
for k in 1 2 4 8 20 64; do SWEEP_K=$k timeout -k 10 300 python tools/sweep.py "$@" 2>&1 | grep "^{" | sed "s/^/K=$k /"; done
