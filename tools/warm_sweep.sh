cd $GRAFT_REPO_ROOT
line() { python -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print('  ms/step %.4f  kernel_ms %.4f all %.4f job_frac %.3f' % (d['ms_per_step'], r['kernel_ms_per_iteration'], r['kernel_ms_per_iteration_all_launches'], r['job_frac']))"; }
for w in 5 50 400; do for rep in 1 2; do echo -n "warmup $w steps 20:"; timeout -k 10 200 python bench.py --steps 20 --warmup $w --no-cpu-baseline 2>/dev/null | line; done; done
echo -n "warmup 5 steps 1000:"; timeout -k 10 200 python bench.py --steps 1000 --warmup 5 --no-cpu-baseline 2>/dev/null | line
