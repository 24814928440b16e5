cd $GRAFT_REPO_ROOT
for n in 16318 16350 16366 16374 16382 16384 16390 16398 16414 16446 16510; do
  timeout -k 10 200 python bench.py --size $n --steps 150 --warmup 10 --no-cpu-baseline 2>/dev/null | python -c "
import sys,json; d=json.loads(sys.stdin.read()); n=$n; print('n=%5d pitch=%6d B (mod 4096 = %4d)  %.4g cell-updates/s  kernel %.4f ms  %.3f ps/cell' % (n, (n+2)*8, ((n+2)*8)%4096, d['value'], d['roofline']['kernel_ms_per_iteration'], d['roofline']['kernel_ms_per_iteration']*1e9/(n*n)))"
done
