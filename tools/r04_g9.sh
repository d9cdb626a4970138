cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r04i
O=gpurun_out/r04i
(timeout -k 10 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -15) > $O/tests.txt 2>&1
tail -3 $O/tests.txt
(timeout -k 10 600 python3 tests/fuzz_parity.py --wild2 100000 5000 > $O/fuzz_wild2.txt 2>&1; tail -3 $O/fuzz_wild2.txt)
(timeout -k 10 600 python3 tests/fuzz_parity.py --wild 100000 7000 > $O/fuzz_wild.txt 2>&1; tail -3 $O/fuzz_wild.txt)
(timeout -k 10 600 python3 tests/fuzz_parity.py 107000 3000 > $O/fuzz_plain.txt 2>&1; tail -1 $O/fuzz_plain.txt)
(timeout -k 10 600 python3 tests/fuzz_parity.py --renderer 200000 600 > $O/fuzz_renderer.txt 2>&1; tail -1 $O/fuzz_renderer.txt)
(timeout -k 10 600 python3 tests/fuzz_parity.py --renderer --lattice 200000 600 > $O/fuzz_renderer_lattice.txt 2>&1; tail -1 $O/fuzz_renderer_lattice.txt)
(timeout -k 10 600 python3 tests/fuzz_parity.py --shares 200000 400 > $O/fuzz_shares.txt 2>&1; tail -1 $O/fuzz_shares.txt)
