# csrc/hip/bounded.h (bounded calls on a helper thread, error log, phase watchdog) under ThreadSanitizer and AddressSanitizer + UBSan (CPU only).
set -e
ROOT=$(cd "$(dirname "$0")/.." && pwd)
for san in thread address,undefined; do
  g++ -std=c++17 -O1 -g -fsanitize=$san -fno-omit-frame-pointer -I$ROOT/gpuart_amd/csrc/hip -o /tmp/bounded_san $ROOT/tools/bounded_sanitize.cpp -lpthread
  echo "== -fsanitize=$san"; /tmp/bounded_san
done
