#!/bin/sh
# Build oracle/_ref/libref_faiss.so: the vendored, patched faiss KMeans path of the reference
# (lib/faiss, faiss 1.10.0 + patolette's patch), compiled with g++ from its own sources where
# they lie -- only the translation units listed in faiss_sources.txt, AVX2-flavour flags of
# lib/faiss/faiss/CMakeLists.txt:253-258,359,379-384.  BLAS/LAPACK entry points are renamed at
# compile time (-Dsgemm_=scipy_sgemm_ ...) to the OpenBLAS that ships in this image (scipy's
# wheel): a real library, not a stand-in.  Usage: build_ref_faiss.sh <reference root> <openblas .so>
set -e
REF="$1"; OB="$2"
HERE="$(cd "$(dirname "$0")" && pwd)"
F="$REF/lib/faiss"
[ -f "$F/faiss/Clustering.cpp" ] || { echo "no reference tree at $REF"; exit 1; }
[ -f "$OB" ] || { echo "no OpenBLAS found"; exit 1; }
TMP="${TMPDIR:-/tmp}/patolette_ref_faiss_obj"
mkdir -p "$TMP" "$HERE/_ref"
REN=""
for s in sgemm_ dgemm_ ssyrk_ sgesvd_ sgeqrf_ dsyev_ dgesvd_ sorgqr_ sgetrf_ sgetri_ dgetrf_ dgetri_ sgelsd_; do
  REN="$REN -D$s=scipy_$s"
done
FLAGS="-std=c++17 -O3 -DNDEBUG -fPIC -mavx2 -mfma -mf16c -mpopcnt -fopenmp -DFINTEGER=int $REN -I $F"
while read -r s; do
  [ -n "$s" ] || continue
  o="$TMP/$(echo "$s" | tr '/' '_' | sed 's/\.cpp$/.o/')"
  echo "g++ $FLAGS -c $F/$s -o $o"
done < "$HERE/faiss_sources.txt" | xargs -P "$(nproc)" -I{} sh -c "{}"
g++ -shared -fopenmp -o "$HERE/_ref/libref_faiss.so" "$TMP"/*.o "$OB" -Wl,-rpath,"$(dirname "$OB")" -Wl,--no-undefined
rm -rf "$TMP"
echo "built $HERE/_ref/libref_faiss.so"
