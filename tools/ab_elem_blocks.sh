for b in 1536 3072 4096 4608 6144 9216; do
  for shp in "40 256 64" "8 256 64" "40 64 256" "8 128 128"; do
    echo "blocks=$b shape=$shp: $(SHM_ELEM_APPLY_BLOCKS=$b python tools/bench_elem.py $shp in_bwd_apply 2>/dev/null | grep float32 | awk '{print $5, $6, $7, $9}' | tr '\n' ' ')"
  done
done
