#!/bin/bash
# how often a repetition of an int8-filtered batch differs: determinism.py over many seeds, int8 filter only
n=0; for s in $(seq ${1:-100} ${2:-139}); do python scripts/r03/determinism.py $s 300 i8 2>&1 | grep -E "^case .* (rep|waves)|differing" | grep -v ": 0 differing" ; done; echo "seeds ${1:-100}..${2:-139} done"
