#!/bin/bash
# Host code (config/GTF builder, reads decoder, table finalisation) under AddressSanitizer + UBSan, or ThreadSanitizer, through the
# CPU tests.  Sanitizers cover host code only: GPU ASan is not available on this pool.
#   bash tools/sanitize_host.sh [address|thread]
set -e
cd "$(dirname "$0")/.."
KIND=${1:-address}
OUT=$(mktemp -d /tmp/mmsan.XXXXXX)
cat > "$OUT/stub.cpp" <<'EOS'
#include "mmannot_amd.h"
// the device half of the library is not part of this build
extern "C" int mm_index_create(mm_index **, const mm_elements *, int32_t, size_t, const int32_t *, const int32_t *, const int32_t *, const uint8_t *, const uint8_t *, int, int) { return MM_ERR_HIP; }
extern "C" const char *mm_last_error(void) { return "device code not built"; }
extern "C" int mm_get_stats(mm_ctx *, uint64_t *) { return MM_ERR_HIP; }
extern "C" int mm_get_rows(mm_ctx *, uint64_t *, double *, size_t, size_t *) { return MM_ERR_HIP; }
extern "C" int mm_set_row_sums(mm_ctx *, const uint64_t *, const double *, size_t) { return MM_ERR_HIP; }
extern "C" int mm_ratio_resolve_n(mm_ctx *, size_t) { return MM_ERR_HIP; }
extern "C" int mm_ratio_handoff(mm_ctx *, mm_ctx *, uint32_t, uint32_t) { return MM_ERR_HIP; }
extern "C" int mm_ratio_replay(mm_ctx *, size_t, size_t, uint32_t, uint32_t) { return MM_ERR_HIP; }
extern "C" int mm_ratio_release(mm_ctx *) { return MM_ERR_HIP; }
extern "C" int mm_sync(mm_ctx *) { return MM_ERR_HIP; }
extern "C" size_t mm_ratio_pending(const mm_ctx *) { return 0; }
extern "C" uint32_t mm_row_chunk(uint64_t mask, uint32_t n) { return n <= 1 ? 0u : (uint32_t) ((mask * 0x9E3779B97F4A7C15ull) >> 40) % n; }
EOS
if [ "$KIND" = thread ]; then SAN="-fsanitize=thread"; RT=$(g++ -print-file-name=libtsan.so); TESTS="tests/test_decoder.py tests/test_partition.py";
else SAN="-fsanitize=address,undefined"; RT=$(g++ -print-file-name=libasan.so); TESTS="tests/test_decoder.py tests/test_builder.py tests/test_kat.py tests/test_synth_e2e.py tests/test_addnh.py tests/test_partition.py"; fi
g++ -std=c++17 -g -O1 $SAN -fno-omit-frame-pointer -shared -fPIC -I include -o "$OUT/libmm_host_san.so" \
    mmannot_amd/csrc/mm_annot.cpp mmannot_amd/csrc/mm_reads.cpp mmannot_amd/csrc/mm_host.cpp mmannot_amd/csrc/mm_multi.cpp "$OUT/stub.cpp" -lz -lpthread
LD_PRELOAD=$RT ASAN_OPTIONS=detect_leaks=0 TSAN_OPTIONS="report_signal_unsafe=0" MMANNOT_AMD_LIB="$OUT/libmm_host_san.so" \
    python -m pytest $TESTS -x -q -m "not gpu"
rm -rf "$OUT"
