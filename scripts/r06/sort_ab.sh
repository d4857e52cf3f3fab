#!/bin/bash
# Collect-all searches on the library's own radix sort, then on a variant library that still calls rocPRIM (rounds 1-5): built by
# hand — `git show 6e4621d:frankensearch_amd/csrc/sort_general.hip` over the file, build, copy to libfsgpu_variant_rocprim.so, restore.
L=frankensearch_amd/libfsgpu.so
V=frankensearch_amd/libfsgpu_variant_rocprim.so
echo "own radix sort (csrc/sort_general.hip)"; python scripts/r06/sort_ab.py 2>/dev/null
[ -f $V ] || { echo "(no rocPRIM variant library here)"; exit 0; }
cp $L /tmp/libfsgpu_default.so
cp $V $L
echo "rocPRIM radix_sort_keys_desc (rounds 1-5)"; python scripts/r06/sort_ab.py 2>/dev/null
cp /tmp/libfsgpu_default.so $L
