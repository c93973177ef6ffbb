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
# the planned-segment path: the closed-form walk behind pd_walk_submit (the oracle offers it with POPDEL_ORACLE_WALK=1), i.e. the
# hand-over thread reading the slot tables in place beside the next round's decode, and the same with jobs that carry copies
check() {
    grep -v '^##fileDate' "$OUT/$1.vcf" > "$OUT/$1.body"
    grep -v '^##fileDate' "$OUT/asan.vcf" | cmp -s - "$OUT/$1.body" || { echo "$1.vcf differs from the host-walk output" >&2; exit 1; }
}
POPDEL_ORACLE_WALK=1 POPDEL_THREADS=4 "$OUT/call_asan" "$@" -o "$OUT/asan_walk.vcf" > /dev/null
POPDEL_ORACLE_WALK=1 POPDEL_THREADS=8 "$OUT/call_tsan" "$@" -o "$OUT/tsan_walk.vcf" > /dev/null
POPDEL_ORACLE_WALK=1 POPDEL_COPY_HAND_OVER=1 POPDEL_THREADS=4 "$OUT/call_asan" "$@" -o "$OUT/asan_walk_copy.vcf" > /dev/null
POPDEL_ORACLE_WALK=1 POPDEL_COPY_HAND_OVER=1 POPDEL_THREADS=8 "$OUT/call_tsan" "$@" -o "$OUT/tsan_walk_copy.vcf" > /dev/null
for f in asan_walk tsan_walk asan_walk_copy tsan_walk_copy; do check $f; done
echo "sanitizer runs finished without findings"
