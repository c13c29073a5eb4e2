#!/usr/bin/env bash
# TEST INFRASTRUCTURE ONLY.  Builds the *reference itself* (mzytnicki/mmannot v1.1)
# from the sources where they lie under $REF (default /root/reference) into
# oracle/_ref/.  Nothing from the reference is copied into the repository: the
# instrumented variants are produced by piping `sed` straight into `g++ -x c++ -`,
# so no patched source file ever exists on disk.  oracle/_ref/ is git-ignored.
#
# Recipe == the reference's own Makefile:8
#   g++ mmannot.cpp -DMMSTANDALONE -Wpedantic -std=c++11 -pthread -lz -O3 -o mmannot
#
# Variants
#   mmannot               verbatim
#   mmannot_patched       + one-line strand fix: setFlags(unsigned int f){flags=f;...}
#   mmannot_dump          verbatim + interval dump (after std::sort, mmannot.cpp:1267-1275)
#                                  + per-hit dump   (after scan,      mmannot.cpp:1776)
#   mmannot_patched_dump  both
# The dumps go to the file named by $MM_DUMP (never stderr: cerr carries the
# comma_numpunct locale, mmannot.cpp:2092).
set -euo pipefail
REF="${REF:-/root/reference}"
HERE="$(cd "$(dirname "${BASH_SOURCE[0]}")" && pwd)"
OUT="$HERE/_ref"
SRC="$REF/mmannot.cpp"
if [ ! -f "$SRC" ]; then
  echo "build_ref.sh: $SRC not present (GPU box?) - keeping prebuilt oracle/_ref as is" >&2
  exit 0
fi
mkdir -p "$OUT"
if [ -x "$OUT/mmannot_patched_dump" ] && [ -x "$OUT/addNH" ] && [ "$OUT/mmannot_patched_dump" -nt "$SRC" ] && [ "$OUT/mmannot_patched_dump" -nt "${BASH_SOURCE[0]}" ]; then
  echo "build_ref.sh: oracle/_ref is up to date" >&2
  exit 0
fi
FLAGS="-DMMSTANDALONE -Wpedantic -std=c++11 -pthread -O3"

# anchors must match exactly once, otherwise the reference changed under us
need_one() { # pattern
  local n; n=$(grep -c -- "$1" "$SRC" || true)
  if [ "$n" != "1" ]; then echo "build_ref.sh: anchor '$1' matched $n times" >&2; exit 1; fi
}
A_FLAGS='void setFlags (unsigned int) {'
A_IV='//cerr << "\\t" << intervals\[i\] << endl;'
A_HIT='intervalList.scan(read, regions, intervals, position);'
need_one "$A_FLAGS"; need_one "$A_IV"; need_one "$A_HIT"

SED_PATCH='s|void setFlags (unsigned int) {|void setFlags (unsigned int f) { flags = f;|'
# interval dump: idx chrId chrName start end type strand(1=F,2=R) id
SED_IV='s|//cerr << "\\t" << intervals\[i\] << endl;|{ static FILE *mmd = getenv("MM_DUMP")? fopen(getenv("MM_DUMP"), "w"): NULL; if (mmd) { fprintf(mmd, "IV\\t%u\\t%u\\t%s\\t%lu\\t%lu\\t%zu\\t%d\\t%s\\n", i, intervals[i].getChromosomeId(), chromosomes[intervals[i].getChromosomeId()].c_str(), intervals[i].getStart(), intervals[i].getEnd(), intervals[i].getType(), (int) intervals[i].getStrand(), intervals[i].getId().c_str()); fflush(mmd); } }|'
# hit dump: name chr start end strand nHits regions(comma list or .)
SED_HIT='s|intervalList.scan(read, regions, intervals, position);|intervalList.scan(read, regions, intervals, position); { static FILE *mmh = getenv("MM_DUMP")? fopen(getenv("MM_DUMP"), "a"): NULL; if (mmh) { fprintf(mmh, "HIT\\t%s\\t%s\\t%lu\\t%lu\\t%d\\t%u\\t", read.getName().c_str(), read.getChromosome().c_str(), read.getStart(), read.getEnd(), (int) read.getStrand(), read.getNHits()); if (regions.empty()) fputc(0x2e, mmh); for (size_t mmi = 0; mmi < regions.size(); ++mmi) fprintf(mmh, mmi? ",%zu": "%zu", regions[mmi]); fputc(10, mmh); } }|'

g++ "$SRC" $FLAGS -lz -o "$OUT/mmannot"
sed -e "$SED_PATCH" "$SRC" | g++ -x c++ - $FLAGS -lz -o "$OUT/mmannot_patched"
sed -e "$SED_IV" -e "$SED_HIT" "$SRC" | g++ -x c++ - $FLAGS -lz -o "$OUT/mmannot_dump"
sed -e "$SED_PATCH" -e "$SED_IV" -e "$SED_HIT" "$SRC" | g++ -x c++ - $FLAGS -lz -o "$OUT/mmannot_patched_dump"
if [ -f "$REF/addNH.cpp" ]; then g++ "$REF/addNH.cpp" $FLAGS -lz -o "$OUT/addNH"; fi
echo "build_ref.sh: built $(ls "$OUT" | tr '\n' ' ')" >&2
