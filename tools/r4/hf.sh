#!/bin/bash
for t in 0 4 8 16 32; do echo "== RVT_COPY_THREADS=$t"; RVT_COPY_THREADS=$t ./tools/host_feed_bench --modes bed,int8 2>&1 | cut -c1-170; done
