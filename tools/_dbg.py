import sys, numpy as np
sys.path.insert(0,'.')
from hunter_bipedal_control_amd import ingest, abi
from hunter_bipedal_control_amd.solver import HunterSolver
from oracle import workloads
params=ingest.load_packaged()
B,N=8,20
refs,x0,rbd,t_now=workloads.trot_batch(params,B,n_intervals=N)
s=HunterSolver(params,batch=B,max_nodes=N)
s.set_references(refs); s.reset(x0); s.mpc_solve(x0)
xg,ug=s.get_solution()
bad=x0.copy(); bad[3,7]=np.nan; bad[5,14]=np.inf
s.mpc_solve(bad)
print('status',s.mpc_status())
xa,ua=s.get_solution()
print('perf',s.get_performance()[:,3])
mask=np.zeros(B,dtype=np.uint8); mask[[3,5]]=1
s.reset_masked(mask,x0)
xr,ur=s.get_solution()
for i in (2,3,5):
    print(i, 'rows differing from x0:', [k for k in range(N+1) if not np.array_equal(xr[i,k],x0[i])], 'equal to x_after rows:', [k for k in range(N+1) if np.array_equal(xr[i,k],xa[i,k])], 'xa==xg rows', [k for k in range(N+1) if np.array_equal(xa[i,k],xg[i,k])])
print(np.isnan(xr[3]).sum(), np.isnan(xa[3]).sum(), xr[3,1,:4], x0[3,:4])
