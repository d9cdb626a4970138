# Builds variants of the device library for same-box A/B timing:  bash tools/ab_build.sh name1 "-DFLAG=1" name2 "-DFLAG=0" ...
# (always with the test hooks of include/gpuart_hip_test.h: the variants are for measurement and tests, never shipped)
# -> gpuart_amd/lib_ab/<name>/ (git-ignored; travels to the GPU box); run them with tools/ab.py name1 name2 ...
set -e
ROOT=$(cd "$(dirname "$0")/.." && pwd)
while [ $# -ge 2 ]; do
  name=$1; flags=$2; shift 2
  d=$ROOT/gpuart_amd/lib_ab/$name
  mkdir -p $d
  make -s -C $ROOT/gpuart_amd/csrc LIBDIR=$d BINDIR=$d EXTRA_HIPFLAGS="-DGPUART_HIP_TEST_HOOKS $flags" $d/libgpuart_hip.so $d/libgpuart.so
  echo "built $name ($flags)"
done
