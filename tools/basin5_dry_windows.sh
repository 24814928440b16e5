cd $GRAFT_REPO_ROOT; T=$(mktemp -d); zcat tests/golden/basin5.asc.gz > $T/basin5.asc
( cd $T && timeout -k 10 300 $GRAFT_REPO_ROOT/wdpm_amd/bin/WDPMCL add basin5.asc NULL out.asc NULL 100 1.0 1.0 1 1 0.005 0 > report.txt 2>err.txt; tail -n 12 report.txt )
python - $T/out.asc <<'PY'
import sys, numpy as np
f=open(sys.argv[1]); hdr=[f.readline() for _ in range(6)]; w=np.array(f.read().split(),dtype=float).reshape(int(hdr[1].split()[1]),int(hdr[0].split()[1]))
nd=float(hdr[5].split()[1]); valid = w != nd
nz=(w>0)&valid
print("valid %.1f %%  wet of valid %.1f %%" % (valid.mean()*100, nz.sum()/valid.sum()*100))
for rr,cc in ((3,3),(3,24),(7,64),(7,192),(9,192),(3,192),(24,192)):
    R,C=nz.shape[0]//rr*rr, nz.shape[1]//cc*cc
    b=nz[:R,:C].reshape(R//rr,rr,C//cc,cc).any(axis=(1,3))
    print("dry %dx%d: %.0f %%" % (rr,cc,(~b).mean()*100))
PY
