#!/bin/sh
# CPU-side sanitizer runs of the host driver (linked against the oracle backend): AddressSanitizer + UBSan, then
# ThreadSanitizer with 8 host threads. Usage: tools/sanitize_host.sh <profiles.txt> [popdel call options...]
# (GPU sanitizers are not available on the pool; the HIP library is covered by the parity tests instead.)
set -e
ROOT=$(cd "$(dirname "$0")/.." && pwd)
H=$ROOT/popdel_amd/csrc/host
OUT=${TMPDIR:-/tmp}/popdel_sanitize
mkdir -p "$OUT"
SRC="$ROOT/oracle/oracle_call_main.cpp $ROOT/oracle/genotype_oracle.cpp $H/histogram.cpp $H/profile_io.cpp $H/chrom_profile.cpp $H/call_params.cpp $H/merge_calls.cpp $H/vcf_writer.cpp $H/call_driver.cpp"
g++ -std=c++17 -O1 -g -fsanitize=address,undefined -fno-omit-frame-pointer -I"$ROOT/include" -I"$ROOT/oracle" -I"$H" -o "$OUT/call_asan" $SRC -lz -lpthread
g++ -std=c++17 -O1 -g -fsanitize=thread -fno-omit-frame-pointer -I"$ROOT/include" -I"$ROOT/oracle" -I"$H" -o "$OUT/call_tsan" $SRC -lz -lpthread
POPDEL_THREADS=4 "$OUT/call_asan" "$@" -o "$OUT/asan.vcf" > /dev/null
POPDEL_THREADS=8 "$OUT/call_tsan" "$@" -o "$OUT/tsan.vcf" > /dev/null
echo "sanitizer runs finished without findings"
