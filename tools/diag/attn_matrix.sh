for shp in "32 32 32 256 128 128" "32 32 32 256 256 128" "32 32 8 256 128 128" "32 32 8 256 256 128" "32 32 32 384 256 128" "32 32 32 384 384 128" "32 12 12 192 64 64" "32 12 12 256 256 64"; do
  echo "== $shp resident"; python tools/bench_attn2.py $shp 20 | tail -n +2
  echo "== $shp chunked WIDE_MIN=128"; ATTN_CHUNKED=1 MTL_ATTN_WIDE_MIN=128 python tools/bench_attn2.py $shp 20 | tail -n +2
done
