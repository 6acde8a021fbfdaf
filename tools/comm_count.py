import sys, numpy as np
sys.path.insert(0,'/root/repo'); sys.path.insert(0,'/root/repo/tests')
import __graft_entry__ as ge
prod = ge.load_product()
import golden_cases as gc
n=32; S=2; nz=n*S; dx=1.0/n
case = prod.make_case(prod.FY_SOLVER_PIMPLE, n,n,nz, dx, 1e-4, 1e-6, g=(0,0,-9.81), u_bc=[0]*6, u_val=[(0,0,0)]*6, p_bc=[2]*6, n_outer_correctors=1, n_correctors=2, p_solver=1)
vs = prod.VirtualSlabs(case, S)
rs=np.random.RandomState(1); npart=200000
rec=np.zeros((npart,10)); rec[:,0:3]=rs.random_sample((npart,3)); rec[:,2]*=0.6*S; rec[:,9]=0.2*dx
vs.set_particles(rec)
vs.step(); a=vs.comm_stats(0)
for _ in range(3): vs.step()
b=vs.comm_stats(0)
st=vs.stats()[0]
print("per step: exchanges %.1f allreduces %.1f allgathers %.1f MB sent %.2f | p_iters %d" % ((b[0]-a[0])/3,(b[1]-a[1])/3,(b[2]-a[2])/3,(b[3]-a[3])/3/1e6, st["p_iters_total"]))
vs.close()
