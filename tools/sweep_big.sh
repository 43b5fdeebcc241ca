for cfg in "768 8" "512 8" "256 8" "256 4" "512 4" "1024 2" "512 2"; do
  set -- $cfg
  echo "=== min_tiles=$1 min_nkt=$2"
  GIM_IGEMM_BIG_MIN_TILES=$1 GIM_IGEMM_BIG_MIN_NKT=$2 timeout 120 python tools/layer_profile.py 2>&1 | sed -n 2,4p
done
