#!/bin/bash
for t in h i j; do echo "== $t"; timeout 300 ./tools/r4/rotgemm_$t 2>&1 | tail -4; done
