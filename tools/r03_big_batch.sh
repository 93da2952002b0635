#!/bin/bash
# tile-kernel time per M strings against batch size and segment length (serial path)
for n in 1000000 2000000 3000000 5000000 10000000; do
  echo "n=$n default"; timeout -k 10 120 python3 tools/quick_bench.py $n 0 20 2>&1 | grep pipeline
done
for sg in 768 512 384 256 192 140; do
  echo "n=10000000 seg=$sg"; LATOK_AB_SEG_TILES=$sg timeout -k 10 120 python3 tools/quick_bench.py 10000000 0 20 2>&1 | grep pipeline
done
