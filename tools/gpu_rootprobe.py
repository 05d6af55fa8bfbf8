"""Development probe: the device's bracketed root finder on quartics dumped from real line searches
(/tmp-style dump produced by a patched oracle; file given as argv[1], format: n lo hi (x f g)*n step)."""
import sys, math
import numpy as np
sys.path.insert(0, ".")
from nav2_social_mpc_controller_amd.params import OptimizerParams
from nav2_social_mpc_controller_amd.solver import BatchSolver
def hermite(f0,g0,x1,f1,g1,x2,f2,g2):
    r1=1/x1; r2=1/x2; r12=1/(x2-x1)
    d01=(f1-f0)*r1; d12=(f2-f1)*r12
    e0=(d01-g0)*r1; e1=(g1-d01)*r1; e2=(d12-g1)*r12; e3=(g2-d12)*r12
    h0=(e1-e0)*r1; h1=(e2-e1)*r2; h2=(e3-e2)*r12
    k0=(h1-h0)*r2; k1=(h2-h1)*r2
    m0=(k1-k0)*r2
    return [m0,k0-m0*(2*x1+x2),h0-2*k0*x1+m0*(x1*x1+2*x1*x2),e0-h0*x1+k0*x1*x1-m0*x1*x1*x2,g0,f0]
rows=[]
for l in open(sys.argv[1]):
    v=l.split(); n=int(v[0])
    if n!=3: continue
    lo=float(v[1]); hi=float(v[2]); s=[float(x) for x in v[3:12]]
    c=hermite(s[1],s[2],s[3],s[4],s[5],s[6],s[7],s[8])
    q=[5*c[0],4*c[1],3*c[2],2*c[3],c[4]]
    rows.append(q+[lo,hi,0.0])
a=np.array(rows)
s=BatchSolver(OptimizerParams.readme())
root,trips=s.math_probe(7,a.ravel())
ok=~np.isnan(root)
np.savez("gpurun_out/rootprobe.npz", a=a, root=root, trips=trips)
print("problems",len(a),"with a root in [lo,hi]",ok.sum(),"trips mean",trips[ok].mean(),"p90",np.percentile(trips[ok],90),"max",trips[ok].max())
