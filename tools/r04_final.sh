cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r04y
O=gpurun_out/r04y
python3 -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.txt 2>&1; tail -1 $O/smoke.txt
GPUART_LIBDIR=$GRAFT_REPO_ROOT/gpuart_amd/lib_ab/tl python3 tools/run_timeline.py 1 > $O/k_run_timeline.txt 2>&1; tail -30 $O/k_run_timeline.txt
(timeout -k 10 500 python3 tests/fuzz_parity.py --wild2 500000 8000 > $O/fuzz_wild2.txt 2>&1; tail -1 $O/fuzz_wild2.txt)
(timeout -k 10 500 python3 tests/fuzz_parity.py --lattice 500000 8000 > $O/fuzz_lattice.txt 2>&1; tail -1 $O/fuzz_lattice.txt)
(timeout -k 10 300 python3 tests/fuzz_parity.py --renderer 500000 500 > $O/fuzz_renderer.txt 2>&1; tail -1 $O/fuzz_renderer.txt)
(timeout -k 10 300 python3 tests/fuzz_parity.py --renderer --lattice 500000 500 > $O/fuzz_renderer_lattice.txt 2>&1; tail -1 $O/fuzz_renderer_lattice.txt)
(timeout -k 10 300 python3 tests/fuzz_parity.py --renderer --wild 500000 300 > $O/fuzz_renderer_wild.txt 2>&1; tail -1 $O/fuzz_renderer_wild.txt)
(timeout -k 10 300 python3 tests/fuzz_parity.py --shares 500000 500 > $O/fuzz_shares.txt 2>&1; tail -1 $O/fuzz_shares.txt)
(timeout -k 10 300 python3 tests/fuzz_parity.py --shares --lattice 500000 300 > $O/fuzz_shares_lattice.txt 2>&1; tail -1 $O/fuzz_shares_lattice.txt)
