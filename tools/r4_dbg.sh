#!/bin/bash
# phase stamps of the tile kernels at several database sizes / tile sizes
for S in ${SIZES:-67750 542000}; do for T in ${TILES:-13 14}; do
  echo "== S=$S tile 2^$T"
  FDGPU_QT_TILE=$T FDGPU_QT_DBG=1 python tools/profile_query_host.py --structures $S --reps 2 --chunk ${CHUNK:-32} --no-profile 2>&1 | grep "^\[qt\]\|full batched" | tail -4
done; done
