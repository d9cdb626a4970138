# CPU-only: the parallel BVH build under ThreadSanitizer and AddressSanitizer + UBSan (300 000 triangles with many tied keys, 8 threads
# against 1 thread; the compiled trees must be equal).   bash tools/sanitize_host_build.sh [triangles]
set -e
ROOT=$(cd "$(dirname "$0")/.." && pwd)
H=$ROOT/gpuart_amd/csrc/host
OUT=${TMPDIR:-/tmp}/gpuart_sanitize
mkdir -p $OUT
for san in thread address,undefined; do
  g++ -O1 -g -std=c++17 -fsanitize=$san -ffp-contract=off -I$H -I$ROOT/include -o $OUT/bvh_${san%%,*} $ROOT/tools/bvh_sanitize.cpp $H/bvh.cpp $H/core.cpp -lpthread
  echo "== -fsanitize=$san"
  $OUT/bvh_${san%%,*} ${1:-300000}
done
