import sys; sys.path.insert(0,'.')
import numpy as np, torch
from gridcodegenerator_amd import host
from gridcodegenerator_amd.robots import get_robot
from oracle import rbd_oracle as O
from tests.conftest import make_inputs, relerr
for robot, prec in [('iiwa7','fp64')]:
    h = host.GridHandle(robot, precision=prec); n=h.n; K=130
    q,qd,u = make_inputs(n,K,1)
    x = np.ascontiguousarray(np.concatenate([q,qd,u],axis=1))
    T = O.RobotTables(get_robot(robot))
    ref = O.fd_grad(T,q.astype(np.float64),qd.astype(np.float64),u.astype(np.float64))
    ref = np.concatenate([O.flat_colmajor(ref[:,:,:n]),O.flat_colmajor(ref[:,:,n:])],axis=1)
    for (b,t) in [(0,0),(3,64),(3,128),(5,32),(2,96),(1,256)]:
        d_in = torch.from_numpy(x).cuda(); d_out = torch.full((K,2*n*n), 7.5, dtype=torch.float32, device='cuda')
        h.forward_dynamics_gradient_device(d_out.data_ptr(), d_in.data_ptr(), 3*n, K, blocks=b, threads=t); h.synchronize()
        out = d_out.cpu().numpy()
        print(robot, prec, (b,t), 'relerr', relerr(out, ref), 'n7.5', (out==7.5).sum(), 'nnan', np.isnan(out).sum(), 'attrs', h.L.kernel_attributes(4))
        bad = np.abs(out-ref) > 1e-3*np.abs(ref).max()
        print('   bad count', bad.sum(), 'bad rows', np.unique(np.where(bad)[0])[:12], 'ncols', len(np.unique(np.where(bad)[1])))
    # other algs
    dq = torch.full((K,n), 7.5, dtype=torch.float32, device='cuda')
    h.forward_dynamics_device(dq.data_ptr(), d_in.data_ptr(), 3*n, K); h.synchronize()
    print('fd', relerr(dq.cpu().numpy(), O.forward_dynamics(T,q.astype(np.float64),qd.astype(np.float64),u.astype(np.float64))))
    h.close()
