import sys; sys.path.insert(0,'.')
import numpy as np, torch, time
from gridcodegenerator_amd import host
from gridcodegenerator_amd.robots import get_robot
from oracle import rbd_oracle as O
from tests.conftest import make_inputs, relerr
robot='iiwa7'; n=7; K=130
vi = int(sys.argv[1])
q,qd,u = make_inputs(n,K,1)
x = np.ascontiguousarray(np.concatenate([q,qd,u],axis=1))
T = O.RobotTables(get_robot(robot))
ref = O.fd_grad(T,q.astype(np.float64),qd.astype(np.float64),u.astype(np.float64))
ref = np.concatenate([O.flat_colmajor(ref[:,:,:n]),O.flat_colmajor(ref[:,:,n:])],axis=1)
variants = [('fp64', ()), ('fp64', ('-mllvm','-amdgpu-spill-sgpr-to-vgpr=0')), ('fp64', ('-O1',)), ('fp64',('-mllvm','-amdgpu-use-aa-in-codegen=0'))]
prec, flags = variants[vi]
t=time.time()
host.build_library(robot, prec, force=True, extra_flags=flags)
print('built', flags, '%.1fs'%(time.time()-t), flush=True)
L = host.GridLibrary(robot, prec)
h = host.GridHandle(robot, precision=prec, library=L)
res=[]
for (b,t) in [(0,0),(1,64),(1,256),(5,32)]:
    d_in = torch.from_numpy(x).cuda(); d_out = torch.full((K,2*n*n), 7.5, dtype=torch.float32, device='cuda')
    h.forward_dynamics_gradient_device(d_out.data_ptr(), d_in.data_ptr(), 3*n, K, blocks=b, threads=t); h.synchronize()
    out = d_out.cpu().numpy()
    res.append(((b,t), '%.2e'%relerr(out,ref)[0], int((out==7.5).sum())))
print(prec, flags, L.kernel_attributes(4), res, flush=True)
h.close()
