"""Register-pressure estimate of a trace's emission order (max simultaneously-live values)."""
import sys; sys.path.insert(0, '.')


def emission_order(tr):
    emitted = [False] * len(tr.nodes); order = []
    for (dst, r) in tr.outputs:
        if isinstance(r, float):
            continue
        stack = [(abs(r), False)]
        while stack:
            k, exp = stack.pop()
            if emitted[k]:
                continue
            if exp:
                emitted[k] = True; order.append(k); continue
            stack.append((k, True))
            for d in reversed(tr._deps(k)):
                if not emitted[d]:
                    stack.append((d, False))
        order.append(('out', abs(r)))
    return order


def pressure(tr, order=None):
    order = order or emission_order(tr)
    pos = {}; last = {}
    for i, k in enumerate(order):
        if isinstance(k, tuple):
            last[k[1]] = i
        else:
            pos[k] = i
            for d in tr._deps(k):
                last[d] = i
    events = [0] * (len(order) + 2)
    for k, i in pos.items():
        events[i] += 1; events[last.get(k, i) + 1] -= 1
    cur = mx = 0; where = 0; prof = []
    for i, e in enumerate(events):
        cur += e; prof.append(cur)
        if cur > mx:
            mx, where = cur, i
    return mx, where, len(order), prof


if __name__ == '__main__':
    from gridcodegenerator_amd.robots import get_robot
    from gridcodegenerator_amd.emit.model import RobotSpec
    from gridcodegenerator_amd.emit import cores
    spec = RobotSpec(get_robot(sys.argv[1] if len(sys.argv) > 1 else 'atlas30'))
    for name, tr in [('ID', cores.core_inverse_dynamics(spec, False)), ('MINV', cores.core_direct_minv(spec)), ('FD', cores.core_forward_dynamics(spec)),
                     ('ID_DU', cores.core_inverse_dynamics_gradient(spec, False)), ('FD_DU', cores.core_forward_dynamics_gradient(spec, False))]:
        mx, where, n, prof = pressure(tr)
        step = max(1, n // 20)
        print(name, 'max live', mx, 'at', where, 'of', n, ' profile:', prof[::step])
