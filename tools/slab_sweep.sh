cd $GRAFT_REPO_ROOT
for shape in "2079 16384" "4126 16384" "8190 16384"; do for f in 50 100; do for d in 1 0; do echo -n "fill=$f dem32env=$d  "; WDPM_FILL_PERCENT=$f WDPM_DEM32=$d timeout -k 10 120 python tools/shape_bench.py $shape 300 fused add 2>/dev/null; done; done; done
for shape in "1055 8190" "2079 8190" "4126 8190"; do for f in 50 100; do echo -n "fill=$f  "; WDPM_FILL_PERCENT=$f timeout -k 10 120 python tools/shape_bench.py $shape 300 fused drain 2>/dev/null; done; done
