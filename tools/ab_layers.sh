#!/bin/bash
# microbench of the main implicit-GEMM layer shapes of the gim_loftr forward (current library)
for L in "--cin 196 --cout 196 --k 3 --act leaky" "--cin 196 --cout 128 --k 3 --act none" "--cin 256 --cout 256 --k 3 --H 60 --W 80 --act relu" \
         "--cin 256 --cout 1024 --k 1 --H 60 --W 80 --act none" "--cin 128 --cout 512 --k 1 --H 120 --W 160 --res 1 --act relu" \
         "--cin 1024 --cout 256 --k 1 --H 60 --W 80 --act relu" "--cin 128 --cout 128 --k 3 --H 120 --W 160 --act relu" "--cin 256 --cout 256 --k 1 --H 60 --W 40 --act none"; do
  python tools/microbench_conv.py $L --iters 30 2>&1 | tail -1
done
