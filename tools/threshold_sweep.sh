cd $GRAFT_REPO_ROOT
for shape in "2048 2048" "3072 3072" "3600 3600"; do
  echo -n "default        "; timeout -k 10 120 python tools/shape_bench.py $shape 1000 fused add 2>/dev/null
  echo -n "codes+fill100  "; WDPM_FILL_PERCENT=100 WDPM_DEM32=2 timeout -k 10 120 python tools/shape_bench.py $shape 1000 fused add 2>/dev/null
  echo -n "fp64+fill100   "; WDPM_FILL_PERCENT=100 WDPM_DEM32=0 timeout -k 10 120 python tools/shape_bench.py $shape 1000 fused add 2>/dev/null
done
for shape in "512 8190" "1024 1024" "2048 2048" "3072 3072"; do for f in 50 100; do echo -n "fill=$f  "; WDPM_FILL_PERCENT=$f timeout -k 10 120 python tools/shape_bench.py $shape 1000 fused drain 2>/dev/null; done; done
